"""Time the device COO -> CSR step: python tools/ingest_time.py [scale]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gunrockinst_amd import devgraph
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
pairs = 8 << scale
torch.cuda.synchronize(); t0 = time.perf_counter()
rows, cols = devgraph.rmat_tuples_device(scale, pairs)
torch.cuda.synchronize(); t1 = time.perf_counter()
for rep in range(3):
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ro, ci = devgraph.csr_from_tuples_device(1 << scale, rows, cols, True)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print("scale %d: generate %.1f ms; COO->CSR of %d directed tuples -> %d edges in %.1f ms (%.2f G tuples/s)" %
          (scale, (t1 - t0) * 1e3, 2 * pairs, ci.shape[0], (t3 - t2) * 1e3, 2 * pairs / (t3 - t2) / 1e9))
