"""Grid-graph BFS (road-like: degree 4, long diameter), TWC workgroup against the persistent levels kernel: python tools/road_sweep.py <side> [shortcut fraction]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gunrockinst_amd as ga
from gunrockinst_amd import devgraph
side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
ro, ci = devgraph.grid_csr_device(side, frac)
n, m = ro.shape[0] - 1, ci.shape[0]
src = n // 2 + side // 2
deg = (ro[1:] - ro[:-1]).long()
for twc in (0, 4096, 8192, 16384):
    p = ga.BfsProblem(False, True, instrument=False).init_device(n, m, ro.data_ptr(), ci.data_ptr())
    p.set_twc_limit(twc)
    best = 1e9
    for rep in range(3):
        p.reset(src); best = min(best, p.enact(src, traversal_mode=1))
    st = p.stats()
    lab = devgraph.as_tensor(p.device_results()[0], n)
    ev = int(deg[lab > -1].sum())
    print("grid %d frac %.3f twc_limit %7d: depth %d enact %.3f ms (%.2f us/level) %.1f MTEPS" %
          (side, frac, twc, st["search_depth"], best, best * 1e3 / max(st["search_depth"], 1), ev / best / 1e3))
    p.close()
