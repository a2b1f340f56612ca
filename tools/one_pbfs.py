"""A few searches through the in-library partitioned loop at world 1 (RCCL): python tools/one_pbfs.py <scale> <source idx|-1> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from gunrockinst_amd import multi_gpu as mg, devgraph
scale, k = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ro, ci = mg.partition_rmat_device(scale, 8, 0x6772, 0, 1)
src = devgraph.largest_degree_source(ro)[0] if k < 0 else devgraph.seeded_sources(ro, 64)[k]
eng = mg.HipEngine(1 << scale, 1, 0, ro, ci, 0)
bfs = mg.LibraryBfs(eng, mg.Comm(), "rccl")
import time
for r in range(reps):
    t0 = time.perf_counter(); lv, ms = bfs.search(src); print("levels", lv, "device ms", ms, "wall ms", (time.perf_counter() - t0) * 1e3)
eng.close()
dist.destroy_process_group()
