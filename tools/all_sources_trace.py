"""Per-source level trace summary of the bench's 65 sources: python tools/all_sources_trace.py [scale]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gunrockinst_amd as ga
from gunrockinst_amd import devgraph
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ro, ci = devgraph.rmat_csr_device(scale, 8)
n, m = ro.shape[0] - 1, ci.shape[0]
sources = [devgraph.largest_degree_source(ro)[0]] + devgraph.seeded_sources(ro, 64)
hp = int(sys.argv[2]) if len(sys.argv) > 2 else -1
VERBOSE = os.environ.get("TRACE_VERBOSE") == "1"
p = ga.BfsProblem(False, True, instrument=True).init_device(n, m, ro.data_ptr(), ci.data_ptr())
p.set_inverse_graph()
if hp >= 0: p.set_head_pass(hp, 0)
q = ga.BfsProblem(False, True, instrument=False).init_device(n, m, ro.data_ptr(), ci.data_ptr())
q.set_inverse_graph()
if hp >= 0: q.set_head_pass(hp, 0)
tot = {}
for s in sources:
    p.reset(s); p.enact(s, traversal_mode=2)
    q.reset(s); q.enact(s, traversal_mode=2)
    q.reset(s); ms = q.enact(s, traversal_mode=2)
    tr = p.level_trace()
    line = " ".join("%d:%s%.0f" % (r["kind"], ("f%d/e%d=" % (r["frontier"], r["edges"])) if (r["kind"] in (4,) or VERBOSE) else "", r["ms"] * 1e3) for r in tr)
    ksum = sum(r["ms"] for r in tr)
    for r in tr:
        tot[r["kind"]] = tot.get(r["kind"], 0.0) + r["ms"]
    print("src %9d enact %.0f us kernels %.0f us | %s" % (s, ms * 1e3, ksum * 1e3, line))
print("totals by kind (us per search):", {k: round(v * 1e3 / len(sources), 1) for k, v in sorted(tot.items())})
