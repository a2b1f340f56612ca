"""A few SSSP enacts for one source (for rocprofv3 --kernel-trace): python tools/one_sssp.py <scale> [delta_factor] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gunrockinst_amd as ga
from gunrockinst_amd import devgraph
scale = int(sys.argv[1]); df = float(sys.argv[2]) if len(sys.argv) > 2 else 16.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ro, ci = devgraph.rmat_csr_device(scale, 8)
n, m = ro.shape[0] - 1, ci.shape[0]
gen = torch.Generator(device="cuda"); gen.manual_seed(0x6772)
w = torch.randint(1, 65, (m,), generator=gen, device="cuda", dtype=torch.int32)
deg = (ro[1:] - ro[:-1]).long()
delta = (int(float(w.double().mean())) * 32.0 / max(float(int(deg.double().mean())), 1.0)) * df
src = devgraph.largest_degree_source(ro)[0]
for inst in (False, True):
    p = ga.SsspProblem(False, inst).init_device(n, m, ro.data_ptr(), ci.data_ptr(), w.data_ptr(), delta)
    if os.environ.get("GUNROCK_SSSP_PULL") == "1":
        p.set_inverse_graph(pull_min_edges=int(os.environ.get("GUNROCK_SSSP_PULL_MIN", "-1")))
    for rep in range(reps):
        p.reset(src)
        ms = p.enact(src)
    print("instrument", inst, "delta", delta, "enact ms", ms, p.stats(), "pull levels", p.pull_levels())
    p.close()
