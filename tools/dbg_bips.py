import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gunrockinst_amd as ga
from oracle import gr_oracle as o
g = o.build_market("tests/golden/bips98_606.mtx", undirected=True)
for src in (0, 566, 566):
    for instr in (True, False):
        p = ga.BfsProblem(False, False, instr).init(g.nodes, g.row_offsets, g.col_indices)
        p.reset(src); ms = p.enact(src, traversal_mode=0)
        labels, _ = p.extract()
        ref, _, depth = o.bfs(g, src)
        print("src", src, "instr", instr, "ok", bool((labels == ref).all()), "label[src]", labels[src], "reached", int((labels >= 0).sum()), p.stats())
        if instr:
            print(p.level_trace())
        p.close()
