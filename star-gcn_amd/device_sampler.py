"""Per-iteration samplers of the training loop ON THE DEVICE (second half).

Reference loop (experiments/STAR-GCN.py:583-600): `next(rating_sampler)` and `next(recon_sampler)` draw on the host with
numpy's Mersenne Twister (mxgraph/iterators.py:264-370); the batch's pairs, noise arrays and reconstruction ids are then
uploaded with the freshly built plan.  With the resident plan (resident.py) those draws were the last per-iteration host
work.  Here they are native device kernels (csrc/plan_build.hip, section 12 of include/stargcn.h):

  rating batch   B distinct edge ids of the training graph, sorted (sg_sample_distinct_hip + sg_sort_i32_hip); the pairs
                 are the CSR coordinates of those edges, the pair plan of the rating head is built from them on the device
  recon nodes    ceil(P_mask n) distinct nodes per node type and the embedding-noise array (sg_recon_mask_hip)

The generators are counter-based (element i of a draw depends on (seed, iteration, i) only): the same distributions as
the reference samplers, deliberately NOT the reference's random stream (a sequential Mersenne Twister cannot be run in
parallel).  Nothing here touches the host after construction.
"""
import math

import torch

from . import _lib as L


class DeviceBatchSampler(object):
    def __init__(self, resident, batch_size, embed_P_mask=0.1, embed_p_zero=0.0, seed=0, recon_candidates=None):
        """recon_candidates: optional {node key: int array of the node ids that occur in the TRAINING graph} -- the
        reference's `_recon_train_candidates` (iterators.py:332-346).  For a listed key the sampler is inductive: nodes
        outside the list get noise -1 ("nodes unseen in the training graph are masked as -1"), the reconstruction nodes are
        ceil(P_mask |list|) of the list.  A key without a list is transductive (every node is a candidate)."""
        self.res, self.batch_size, self.seed = resident, int(min(batch_size, resident.nnz)), int(seed)
        self._cand = {}
        n_of = {resident.U: resident.n_user, resident.I: resident.n_item}
        for key, ids in (recon_candidates or {}).items():
            # the device kernel assumes DISTINCT ids in [0, n) (the reference permutes a distinct candidate list,
            # iterators.py:332-346): duplicates would yield duplicate reconstruction nodes, an id out of range a -1 entry
            import numpy as np
            arr = np.asarray(ids.cpu() if torch.is_tensor(ids) else ids).astype(np.int64).reshape(-1)
            if key not in n_of:
                raise L.StarGCNError("recon_candidates: unknown node key %r" % (key,))
            if arr.size and (arr.min() < 0 or arr.max() >= n_of[key]):
                raise L.StarGCNError("recon_candidates[%r]: ids must lie in [0, %d)" % (key, n_of[key]))
            uniq = np.unique(arr)
            if uniq.size != arr.size:
                raise L.StarGCNError("recon_candidates[%r]: %d duplicate ids (the list must be distinct)" % (key, arr.size - uniq.size))
            self._cand[key] = torch.from_numpy(arr.astype(np.int32)).to(resident.device).contiguous()
        self.P_mask, self.p_zero = float(embed_P_mask), float(embed_p_zero)
        self.iteration = 0
        dev = resident.device
        # draw counter in device memory: lets a captured hipGraph of the whole iteration draw a new batch at every replay
        self.dev_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self._values = torch.from_numpy(resident.csr.values.astype("float32")).to(dev)
        self._n = {resident.U: resident.n_user, resident.I: resident.n_item}

    def _i32(self, n):
        return torch.empty(max(int(n), 1), dtype=torch.int32, device=self.res.device)

    def next_batch(self, advance_on_device=False):
        """-> dict(edge_ids (B,) sorted, users, items, ratings (B,), noise {key: (n,)}, recon {key: (k,)}) -- all device
        tensors, nothing synchronises.  advance_on_device=True: the draw index lives in device memory and is advanced by a
        kernel at the end of this call (for a captured hipGraph: every replay draws the next batch)."""
        lib, st, res = L.lib(), L.stream_ptr(), self.res
        if advance_on_device:
            it, dc = 0, L.ptr(self.dev_counter)
        else:
            it, dc = self.iteration, None
            self.iteration += 1
        B = self.batch_size
        raw, ids = self._i32(B), self._i32(B)
        L.check(lib.sg_sample_distinct_dev_hip(L.ptr(raw), res.nnz, B, self.seed, 3 * it, dc, st), "sg_sample_distinct_hip")
        ws, wsn = L.workspace(lib.sg_sort_i32_workspace_bytes(B), res.device)
        L.check(lib.sg_sort_i32_hip(L.ptr(ids), None, L.ptr(raw), None, B, max(res.nnz - 1, 0), L.ptr(ws), wsn, st),
                "sg_sort_i32_hip")
        users, items = self._i32(B), self._i32(B)
        L.check(lib.sg_gather_i32_hip(L.ptr(users), L.ptr(res._edge_row), L.ptr(ids), B, st), "sg_gather_i32_hip")
        L.check(lib.sg_gather_i32_hip(L.ptr(items), L.ptr(res._edge_col), L.ptr(ids), B, st), "sg_gather_i32_hip")
        ratings = self._values[ids[:B].long()]
        noise, recon = dict(), dict()
        for j, (key, n) in enumerate(self._n.items()):
            cand = self._cand.get(key)
            if cand is not None:
                k = int(math.ceil(self.P_mask * cand.numel()))
                nz, rc = self._i32(n), self._i32(k)
                L.check(lib.sg_recon_mask_cand_dev_hip(L.ptr(nz), L.ptr(rc), n, L.ptr(cand), cand.numel(), k, self.p_zero,
                                                       self.seed, 3 * it + 1 + j, dc, st), "sg_recon_mask_cand_dev_hip")
            else:
                k = int(math.ceil(self.P_mask * n))
                nz, rc = self._i32(n), self._i32(k)
                L.check(lib.sg_recon_mask_dev_hip(L.ptr(nz), L.ptr(rc), n, k, self.p_zero, self.seed, 3 * it + 1 + j, dc, st),
                        "sg_recon_mask_hip")
            noise[key], recon[key] = nz[:n], rc[:k]
        if advance_on_device:
            L.check(lib.sg_counter_add_hip(L.ptr(self.dev_counter), 3, st), "sg_counter_add_hip")
        return dict(edge_ids=ids[:B], users=users[:B], items=items[:B], ratings=ratings, noise=noise, recon=recon)
