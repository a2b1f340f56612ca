"""PageRank timing: python tools/one_pr.py <scale> [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gunrockinst_amd as ga
from gunrockinst_amd import devgraph
scale = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ro, ci = devgraph.rmat_csr_device(scale, 8)
n, m = ro.shape[0] - 1, ci.shape[0]
p = ga.PrProblem().init_device(n, m, ro.data_ptr(), ci.data_ptr())
p.set_inverse_graph()
for rep in range(3):
    p.reset(-1, 0.85, 0.0)
    ms = p.enact(iters)
print("scale", scale, "m", m, "enact ms", ms, p.stats(), "ms/iter ~", ms / iters, "G edges/s per iter", m / (ms / iters) / 1e6)
p.close()
