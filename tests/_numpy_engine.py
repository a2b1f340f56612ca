"""Test double for the per-rank local steps of the partitioned BFS (numpy, CPU).  TEST INFRASTRUCTURE: it lets the
engine-agnostic level loop + collectives of gunrockinst_amd/multi_gpu.py run under gloo without a GPU.  The product's
local steps are the HIP kernels behind grx_pbfs_* (multi_gpu.HipEngine)."""
import numpy as np
import torch

from gunrockinst_amd import multi_gpu as mg


class NumpyEngine:
    def __init__(self, n_global, parts, rank, ro, ci):
        self.n_global, self.parts, self.rank = n_global, parts, rank
        self.ro, self.ci = ro.astype(np.int64), ci.astype(np.int64)
        self.n_local = ro.shape[0] - 1
        self.words = mg.mask_words((n_global + parts - 1) // parts)

    def _deg(self, ids):
        return self.ro[ids + 1] - self.ro[ids]

    def _set_frontier(self, ids):
        ids = np.asarray(ids, dtype=np.int64)
        ids = ids[self._deg(ids) > 0] if ids.size else ids
        self.frontier = ids
        return int(ids.size), int(self._deg(ids).sum()) if ids.size else 0

    def reset(self, src):
        self.labels_ = np.full(self.n_local, -1, np.int32)
        self.sent = np.zeros(self.n_global, bool)
        self.sent[src] = True
        self.level = 0
        self.bitmap = np.zeros(self.words * 32, bool)
        if src % self.parts == self.rank:
            self.labels_[src // self.parts] = 0
            return self._set_frontier([src // self.parts])
        return self._set_frontier([])

    def advance_local(self):
        if self.frontier.size:
            nb = np.concatenate([self.ci[self.ro[v]:self.ro[v + 1]] for v in self.frontier])
        else:
            nb = np.empty(0, np.int64)
        nb = np.unique(nb)
        nb = nb[~self.sent[nb]]
        self.sent[nb] = True
        owner = nb % self.parts
        order = np.argsort(owner, kind="stable")
        counts = np.bincount(owner, minlength=self.parts).tolist()
        return counts, torch.from_numpy((nb[order] // self.parts).astype(np.int32))

    def filter_received(self, recv):
        ids = np.unique(recv.numpy().astype(np.int64))
        ids = ids[self.labels_[ids] == -1]
        self.labels_[ids] = self.level + 1
        self.level += 1
        return self._set_frontier(ids)

    def queue_to_bitmap(self):
        self.bitmap[:] = False
        self.bitmap[self.frontier] = True

    def frontier_bitmap(self):
        return torch.from_numpy(np.packbits(self.bitmap, bitorder="little").view(np.int32).copy())

    def bottom_up(self, gathered, words_per_rank):
        bits = np.unpackbits(gathered.numpy().view(np.uint8), bitorder="little").astype(bool).reshape(self.parts, -1)
        found = []
        for v in np.nonzero(self.labels_ == -1)[0]:
            nb = self.ci[self.ro[v]:self.ro[v + 1]]
            if nb.size and bits[nb % self.parts, nb // self.parts].any():
                found.append(v)
        found = np.asarray(found, dtype=np.int64)
        self.labels_[found] = self.level + 1
        self.level += 1
        self.bitmap[:] = False
        self.bitmap[found] = True
        self.frontier = found
        return int(found.size), int(self._deg(found).sum()) if found.size else 0

    def bitmap_to_queue(self):
        return self._set_frontier(np.nonzero(self.bitmap[:self.n_local])[0])

    def labels(self):
        return self.labels_
