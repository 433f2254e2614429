"""Mirror of the reference's `pruners` package exports (pruners/__init__.py:1-2)."""
from .channel_pruner import init_pruned_model  # noqa: F401
from .dcfp_pruner import dcfp_pruning, DCFPPruner  # noqa: F401
