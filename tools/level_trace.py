import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import gunrockinst_amd as ga
from gunrockinst_amd import devgraph
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ro, ci = devgraph.rmat_csr_device(scale, 8)
n, m = ro.shape[0]-1, ci.shape[0]
src, md = devgraph.largest_degree_source(ro)
if len(sys.argv) > 3:
    src = devgraph.seeded_sources(ro, 64)[int(sys.argv[3])]
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
p = ga.BfsProblem(False, True, instrument=True).init_device(n, m, ro.data_ptr(), ci.data_ptr())
if mode == 2:
    p.set_inverse_graph()
for rep in range(3):
    p.reset(src); ms = p.enact(src, traversal_mode=mode)
print("scale", scale, "n", n, "m", m, "src", src, "enact ms (instrumented)", ms)
q = ga.BfsProblem(False, True, instrument=False).init_device(n, m, ro.data_ptr(), ci.data_ptr())
if mode == 2:
    q.set_inverse_graph()
best = 1e9
for rep in range(5):
    q.reset(src); best = min(best, q.enact(src, traversal_mode=mode))
print("enact ms (plain, best of 5)", best, "depth", q.stats()["search_depth"])
for i, r in enumerate(p.level_trace()):
    print(i, r, "GB/s(col only)=%.1f" % (r["edges"]*4/ (r["ms"]*1e-3) /1e9))
