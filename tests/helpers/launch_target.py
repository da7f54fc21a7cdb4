"""A rank process for tests/test_launch_cpu.py: started N times by mofreak_amd.launch.spawn_ranks (the code
`python bench.py --gpus N` runs through), joins a gloo group and lets rank 0 print one JSON line."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if mode == "fail" and rank == 1:
    sys.exit(3)  # before the rendezvous: the other ranks would wait for ever
dist.init_process_group("gloo")
t = torch.tensor([rank + 1], dtype=torch.int64)
dist.all_reduce(t)
if mode == "hang" and rank == 0:
    time.sleep(600)
if rank == 0:
    print(json.dumps({"ranks_seen": dist.get_world_size(), "sum": int(t.item()), "launcher": os.environ.get("MOFREAK_LAUNCHER")}), flush=True)
else:
    print(f"rank {rank} says hello on stdout")  # must not end up in the job's stdout
dist.barrier()
dist.destroy_process_group()
