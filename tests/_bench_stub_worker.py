"""Stand-in for a bench.py rank in tests/test_bench_launcher_cpu.py: prints what the launcher
handed it (no torch, no GPU).  STUB_FAIL_RANK makes that rank exit non-zero; STUB_HANG_RANK makes
that rank sleep so the launcher has something to terminate."""
import json
import os
import sys
import time

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if os.environ.get("STUB_FAIL_RANK") == str(rank):
    sys.exit(7)
if os.environ.get("STUB_HANG_RANK") == str(rank):
    print(f"stub rank {rank}: entering a collective that never completes", file=sys.stderr, flush=True)
    time.sleep(60)
print("noise line from rank", rank)
rec = {"n_gpus": world, "rccl_ranks": int(os.environ.get("STUB_RCCL_RANKS", world)), "rank": rank,
       "local_rank": int(os.environ["LOCAL_RANK"]), "master": os.environ["MASTER_ADDR"],
       "port": int(os.environ["MASTER_PORT"]), "argv": sys.argv[1:],
       "config": {"global_batch": 4 * world, "parallelism": f"dp{world}"}}
if rank == 0:
    print(json.dumps(rec))
else:
    print(json.dumps(rec))   # goes to the launcher's stderr, never to its stdout
