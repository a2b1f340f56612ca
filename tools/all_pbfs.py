"""Per-source times of the in-library partitioned loop at world 1 (RCCL) next to the single-GPU enactor: python tools/all_pbfs.py <scale>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import gunrockinst_amd as ga
from gunrockinst_amd import multi_gpu as mg, devgraph
scale = int(sys.argv[1])
ro, ci = mg.partition_rmat_device(scale, 8, 0x6772, 0, 1)
n, m = ro.shape[0] - 1, ci.shape[0]
sources = [devgraph.largest_degree_source(ro)[0]] + devgraph.seeded_sources(ro, 64)
eng = mg.HipEngine(1 << scale, 1, 0, ro, ci, 0)
bfs = mg.LibraryBfs(eng, mg.Comm(), "rccl", alpha=float(os.environ.get("PBFS_ALPHA", "0")))
p = ga.BfsProblem(False, True).init_device(n, m, ro.data_ptr(), ci.data_ptr())
p.set_inverse_graph()
tot_p = tot_s = 0.0
for i, s in enumerate(sources):
    best_p = min(bfs.search(s)[1] for _ in range(3))
    lv = bfs.search(s)[0]
    best_s = 1e9
    for _ in range(3):
        p.reset(s); best_s = min(best_s, p.enact(s, traversal_mode=2))
    tot_p += best_p; tot_s += best_s
    if not os.environ.get("PBFS_QUIET"): print("src %2d %9d  partitioned %.3f ms (%d levels)  single %.3f ms  ratio %.2f" % (i, s, best_p, lv, best_s, best_p / best_s))
print("mean partitioned %.3f  single %.3f" % (tot_p / len(sources), tot_s / len(sources)))
p.close(); eng.close()
dist.destroy_process_group()
