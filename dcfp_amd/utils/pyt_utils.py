"""The two helpers of utils/pyt_utils.py that sit on the hot path:
all_reduce_tensor (:34-40) and load_model (:43-96)."""
import logging
import time
from collections import OrderedDict

import torch
import torch.distributed as dist

logger = logging.getLogger("dcfp_amd")


def all_reduce_tensor(tensor, op=None, world_size=1, norm=True):
    """Clone, all-reduce (RCCL under backend 'nccl'), optionally divide by world_size."""
    op = dist.ReduceOp.SUM if op is None else op
    tensor = tensor.clone()
    dist.all_reduce(tensor, op)
    if norm:
        tensor.div_(world_size)
    return tensor


def load_model(model, model_file, is_restore=False, ignore_prefix=None, extra_prefix=None):
    """strict=False load that only logs key mismatches; accepts a path or a state_dict, and
    checkpoints wrapped as {'model': ...} / {'state_dict': ...}."""
    t0 = time.time()
    if isinstance(model_file, str):
        state_dict = torch.load(model_file, map_location=torch.device("cpu"))
        if "model" in state_dict.keys():
            state_dict = state_dict["model"]
        elif "state_dict" in state_dict.keys():
            state_dict = state_dict["state_dict"]
    else:
        state_dict = model_file
    t1 = time.time()
    if ignore_prefix is not None:
        state_dict = OrderedDict((k[len(ignore_prefix):] if k.startswith(ignore_prefix) else k, v)
                                 for k, v in state_dict.items())
    if extra_prefix is not None:
        state_dict = OrderedDict((extra_prefix + k, v) for k, v in state_dict.items())
    model.load_state_dict(state_dict, strict=False)
    ckpt, own = set(state_dict.keys()), set(model.state_dict().keys())
    missing = [k for k in own - ckpt if not k.endswith(".num_batches_tracked")]
    unexpected = [k for k in ckpt - own if not k.endswith(".num_batches_tracked")]
    if missing:
        logger.warning("Missing key(s) in state_dict: %s", ", ".join(sorted(missing)))
    if unexpected:
        logger.warning("Unexpected key(s) in state_dict: %s", ", ".join(sorted(unexpected)))
    logger.info("Load model, Time usage:\n\tIO: %s, initialize parameters: %s", t1 - t0, time.time() - t1)
