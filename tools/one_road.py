"""One grid-graph BFS in traversal_mode 1 (for rocprofv3 --kernel-trace): python tools/one_road.py <side> <twc_limit>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gunrockinst_amd as ga
from gunrockinst_amd import devgraph
side = int(sys.argv[1]); twc = int(sys.argv[2])
ro, ci = devgraph.grid_csr_device(side, 0.0)
n, m = ro.shape[0] - 1, ci.shape[0]
src = n // 2 + side // 2
p = ga.BfsProblem(False, True, instrument=False).init_device(n, m, ro.data_ptr(), ci.data_ptr())
p.set_twc_limit(twc)
for rep in range(2):
    p.reset(src); print("enact ms", p.enact(src, traversal_mode=1))
p.close()
