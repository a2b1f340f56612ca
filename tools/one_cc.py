"""A few CC enacts (for rocprofv3 --kernel-trace): python tools/one_cc.py <scale> [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gunrockinst_amd as ga
from gunrockinst_amd import devgraph
scale = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ro, ci = devgraph.rmat_csr_device(scale, 8)
n, m = ro.shape[0] - 1, ci.shape[0]
p = ga.CcProblem(False).init_device(n, m, ro.data_ptr(), ci.data_ptr())
for rep in range(reps):
    p.reset(); ms = p.enact()
print("enact ms", ms, p.stats())
p.close()
