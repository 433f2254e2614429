import sys, torch; sys.path.insert(0,'.')
from dcfp_amd import ops, _lib
import ctypes as C
dev=torch.device('cuda:0')
for Cc,S in ((256,1024),(1024,1024),(64,16384),(512,1024)):
    part=torch.randn(S*Cc*2,device=dev); mv=torch.empty(2,Cc,device=dev)
    L=_lib.lib()
    def run():
        ops.check(L.dcfp_bn_stats_from_partials_f32(ops._p(part), S, 128, Cc, ops._p(mv[0]), ops._p(mv[1]), None, ops._stream()), "x")
    for _ in range(5): run()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(50): run()
    e.record(); torch.cuda.synchronize(); print(Cc,S, s.elapsed_time(e)/50*1000, "us")
