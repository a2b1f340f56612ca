import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gunrockinst_amd as ga
from gunrockinst_amd import devgraph
side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
plimit = int(sys.argv[3]) if len(sys.argv) > 3 else -1
ro, ci = devgraph.grid_csr_device(side, frac)
n, m = ro.shape[0] - 1, ci.shape[0]
src = n // 2 + side // 2
for mode in (0, 2):
    p = ga.BfsProblem(False, True, instrument=True).init_device(n, m, ro.data_ptr(), ci.data_ptr())
    if mode == 2: p.set_inverse_graph()
    if plimit >= 0: p.set_persistent_limit(plimit)
    best = 1e9
    for rep in range(3):
        p.reset(src); best = min(best, p.enact(src, traversal_mode=mode))
    st = p.stats(); tr = p.level_trace()
    kinds = {}
    for r in tr: kinds.setdefault(r["kind"], [0, 0.0]); kinds[r["kind"]][0] += 1; kinds[r["kind"]][1] += r["ms"]
    lab = devgraph.as_tensor(p.device_results()[0], n)
    deg = (ro[1:] - ro[:-1]).long()
    ev = int(deg[lab > -1].sum())
    print("grid %dx%d frac %.3f plimit %d mode %d: n=%d m=%d depth=%d enact %.3f ms -> %.1f MTEPS; launches %d by kind %s" %
          (side, side, frac, plimit, mode, n, m, st["search_depth"], best, ev / best / 1e3, st["kernel_launches"], kinds))
    p.close()
