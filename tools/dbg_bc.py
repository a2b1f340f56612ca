import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gunrockinst_amd as ga
from oracle import gr_oracle as o
rng = np.random.default_rng(11)
kind = rng.integers(0, 3)
scale = int(rng.integers(5, 16)); ef = int(rng.integers(1, 25)); und = bool(rng.integers(0, 2))
print("kind", kind, "scale", scale, "ef", ef, "undirected", und)
g = o.rmat_seeded(scale, ef << scale, undirected=und)
deg = np.diff(g.row_offsets)
for src in [int(np.argmax(deg)), 0, 1, 5]:
    bc, _ = ga.gunrock_bc(g.nodes, g.row_offsets, g.col_indices, src=src)
    ref, _ = o.bc(g, src)
    bad = np.nonzero(np.abs(bc - ref) > 1e-3 * np.abs(ref) + 1e-3)[0]
    lab, _, _ = o.bfs(g, src)
    print("src", src, "deg", deg[src], "bad", bad[:10], [(int(v), float(bc[v]), float(ref[v]), int(lab[v]), int(deg[v])) for v in bad[:6]])
