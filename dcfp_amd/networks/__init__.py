"""Mirror of the reference's `networks` package (networks/__init__.py:1-4) for the two model
files on the hot path."""
from . import deeplabv3, simple  # noqa: F401
from . import backbone  # noqa: F401
