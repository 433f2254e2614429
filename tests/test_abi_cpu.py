"""CPU: the C-ABI shared library loads and exports every entry point include/dcfp_hip.h
declares (no compute calls without a GPU), and the ctypes table mirrors the header."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "dcfp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dcfp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_header_symbols():
    from dcfp_amd import _lib
    path = _lib.LIB_PATH
    if not os.path.exists(path):
        _lib.build()
    h = ctypes.CDLL(path)
    names = header_functions()
    assert len(names) >= 24
    for n in names:
        assert hasattr(h, n), n
    assert sorted(_lib.SIGNATURES.keys()) == names
    assert h.dcfp_abi_version() == 2


def test_descriptor_errors_do_not_need_a_gpu():
    """Bad descriptors are rejected on the host side with the documented negative codes."""
    from dcfp_amd import _lib
    L = _lib.lib()
    d = _lib.ConvDesc(1, 8, 8, 8, 8, 5, 5, 1, 2, 1, 8, 8)          # 5x5 kernel: unsupported
    assert L.dcfp_conv2d_fwd_f32_nchw(ctypes.byref(d), None, None, None, None, 0, None, 0, 0, None) == -2
    d = _lib.ConvDesc(1, 8, 8, 8, 8, 3, 3, 1, 1, 1, 7, 8)          # wrong Hout
    assert L.dcfp_conv2d_fwd_f32_nchw(ctypes.byref(d), None, None, None, None, 0, None, 0, 0, None) == -1
    assert L.dcfp_bn_stats_f32(None, 0, 1, 1, 1, None, None, None, None, 0, None) == -1
    assert L.dcfp_conv2d_workspace_bytes(ctypes.byref(d), 2) == 0


def test_product_has_no_cpu_path():
    import pytest
    import torch
    from dcfp_amd import networks, ops
    m = networks.simple.Seg_Model(backbone="resnet50", backbone_para={"pretrained": False}, num_classes=19)
    with pytest.raises(RuntimeError):
        m(torch.zeros(2, 3, 32, 32))
    with pytest.raises(RuntimeError):
        ops.conv2d(torch.zeros(1, 3, 8, 8), torch.zeros(4, 3, 3, 3), None, 1, 1, 1)


def test_oracle_not_imported_by_product():
    """Nothing under dcfp_amd/ may import, call or link anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dcfp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dirpath, f)


def test_pitched_buffer_lease_follows_the_graph_lifetime():
    """ops.owner_pitched: a module's persistent row-pitched buffer is busy exactly while an autograd node holds a lease on
    it - released by that node's backward or by the node's death (a grad-enabled forward that never reaches backward)."""
    import gc
    import types
    import torch
    from dcfp_amd import ops
    owner = types.SimpleNamespace()
    shape, pitch, dev = (1, 2, 3, 8), 12, torch.device("cpu")
    v1, lease1 = ops.owner_pitched(owner, shape, pitch, dev)
    assert lease1 is not None and v1.stride(2) == pitch and tuple(v1.shape) == shape
    v2, lease2 = ops.owner_pitched(owner, shape, pitch, dev)           # first graph still alive: a fresh buffer, no lease
    assert lease2 is None and v2.data_ptr() != v1.data_ptr()
    del lease1                                                         # the graph died without backward
    gc.collect()
    v3, lease3 = ops.owner_pitched(owner, shape, pitch, dev)
    assert v3.data_ptr() == v1.data_ptr() and lease3 is not None
    lease3.release()                                                   # the normal end: backward ran
    v4, lease4 = ops.owner_pitched(owner, shape, pitch, dev)
    assert v4.data_ptr() == v1.data_ptr() and lease4 is not None
    v5, lease5 = ops.owner_pitched(owner, shape, pitch, dev, track=False)    # no_grad forward: nothing to hold
    assert lease5 is None
