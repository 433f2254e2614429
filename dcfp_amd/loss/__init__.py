from . import criterion  # noqa: F401
