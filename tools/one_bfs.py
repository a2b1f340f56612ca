"""Run a few BFS enacts for one source (for rocprofv3 --kernel-trace): python tools/one_bfs.py <scale> <mode> <source index|-1>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gunrockinst_amd as ga
from gunrockinst_amd import devgraph
scale, mode, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
ro, ci = devgraph.rmat_csr_device(scale, 8)
n, m = ro.shape[0] - 1, ci.shape[0]
src = devgraph.largest_degree_source(ro)[0] if k < 0 else devgraph.seeded_sources(ro, 64)[k]
p = ga.BfsProblem(False, True).init_device(n, m, ro.data_ptr(), ci.data_ptr())
if mode == 2:
    p.set_inverse_graph()
for rep in range(reps):
    p.reset(src)
    print("enact ms", p.enact(src, traversal_mode=mode))
p.close()
