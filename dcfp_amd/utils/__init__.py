"""Host-side helpers of the hot path: checkpoint loading (pyt_utils) and the FLOPs counter prune.py is judged by."""
