"""Per-level census of a top-down BFS on the bench graph: edges out of the frontier, edges into vertices unvisited at the
level's start (what a level-start screen lets through), discoveries, and how many of those edges a per-XCD hash bin sees."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gunrockinst_amd as ga
from gunrockinst_amd import devgraph
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ro, ci = devgraph.rmat_csr_device(scale, 8)
n, m = ro.shape[0] - 1, ci.shape[0]
src, md = devgraph.largest_degree_source(ro)
if len(sys.argv) > 2:
    src = devgraph.seeded_sources(ro, 64)[int(sys.argv[2])]
p = ga.BfsProblem(False, True).init_device(n, m, ro.data_ptr(), ci.data_ptr())
p.reset(src); p.enact(src, traversal_mode=0)
labels = devgraph.as_tensor(p.device_results()[0], n).clone()
deg = (ro[1:] - ro[:-1]).long()
rows = torch.repeat_interleave(torch.arange(n, device=ro.device, dtype=torch.int32), deg)
sl = labels[rows.long()]
dl = labels[ci.long()]
x = (ci >> 8)
binid = (x ^ (x >> 3) ^ (x >> 6) ^ (x >> 9) ^ (x >> 12) ^ (x >> 15)) & 7
for L in range(int(labels.max()) + 1):
    f = sl == L
    e = int(f.sum())
    s_mask = f & (dl == L + 1)
    s = int(s_mask.sum())
    new = int((labels == L + 1).sum())
    bins = torch.bincount(binid[s_mask].long(), minlength=8).tolist() if s else []
    print("level", L, "frontier", int((labels == L).sum()), "edges", e, "to-unvisited", s, "new", new, "bins", bins)
