"""Mini-batch samplers that feed the hot path: reference mxgraph/iterators.py (`DataIterator` :120-236 in both the
transductive and the inductive setting, `rating_sampler` :264-307, `recon_nodes_sampler` :309-370).  Host-side numpy only.

The embedding-noise convention is the reference's (iterators.py:338-346): `embed_noise[key]` has one entry per node
of the whole graph; -1 masks the node's input embedding to zero, i keeps (or substitutes) embedding i.  Nodes picked
for reconstruction get -1 with probability `embed_p_zero` and their own id otherwise; nodes that are not training
candidates stay -1 -- in the inductive setting those are the held-out nodes, which the network therefore sees with a
zero input embedding and has to reconstruct from their neighbourhood.  The unused negative-edge generator
(iterators.py:8-111) is out of scope.
"""
import numpy as np


class DataIterator(object):
    def __init__(self, all_graph, name_user, name_item, test_node_pairs, valid_node_pairs, embed_P_mask=0.1,
                 embed_p_zero=0.0, embed_p_self=1.0, seed=None, is_inductive=False, inductive_key=None,
                 inductive_valid_ids=None, inductive_train_ids=None):
        self._rng = np.random.RandomState(seed=seed)
        self._all_graph, self._name_user, self._name_item = all_graph, name_user, name_item
        self._is_inductive = bool(is_inductive)
        # test graph: test ratings removed (reference :165)
        self._test_graph = all_graph.remove_edges_by_id(name_user, name_item, test_node_pairs)
        if not is_inductive:    # val/train graph: validation ratings removed too (reference :168-170)
            self._val_graph = self._test_graph.remove_edges_by_id(name_user, name_item, valid_node_pairs)
            self._train_graph = self._val_graph
        else:                   # reference :171-176: graphs of the train (+ validation) NODES of `inductive_key`
            assert inductive_key is not None and inductive_train_ids is not None and inductive_valid_ids is not None
            train_val_ids = np.concatenate((inductive_train_ids, inductive_valid_ids)).astype(np.int32)
            self._val_graph = all_graph.sel_subgraph_by_id(inductive_key, train_val_ids) \
                .remove_edges_by_id(name_user, name_item, valid_node_pairs)
            self._train_graph = all_graph.sel_subgraph_by_id(inductive_key, inductive_train_ids)
        self._test_node_pairs, self._valid_node_pairs = np.asarray(test_node_pairs), np.asarray(valid_node_pairs)
        csr = self._train_graph[name_user, name_item]
        self._train_node_pairs, self._train_ratings = csr.node_pair_ids, csr.values
        self._valid_ratings = all_graph.fetch_edges_by_id(name_user, name_item, self._valid_node_pairs)
        self._test_ratings = all_graph.fetch_edges_by_id(name_user, name_item, self._test_node_pairs)
        as_dict = lambda v: v if isinstance(v, dict) else {k: v for k in all_graph.meta_graph}
        self._embed_P_mask, self._embed_p_zero, self._embed_p_self = as_dict(embed_P_mask), as_dict(embed_p_zero), as_dict(embed_p_self)
        for key in self._embed_P_mask:
            assert self._embed_p_zero[key] + self._embed_p_self[key] == 1.0
        self._recon_train_candidates = {k: self._train_graph.node_ids_dict[k] for k in self._train_graph.meta_graph}
        self._evaluate_embed_noise_dict = dict()
        for key, ids in self._recon_train_candidates.items():
            noise = -np.ones(all_graph.node_ids_dict[key].shape, dtype=np.int32)
            noise[ids] = ids
            self._evaluate_embed_noise_dict[key] = noise

    possible_rating_values = property(lambda self: self._all_graph[self._name_user, self._name_item].multi_link)
    evaluate_embed_noise_dict = property(lambda self: self._evaluate_embed_noise_dict)
    is_inductive = property(lambda self: self._is_inductive)
    all_graph = property(lambda self: self._all_graph)
    test_graph = property(lambda self: self._test_graph)
    val_graph = property(lambda self: self._val_graph)
    train_graph = property(lambda self: self._train_graph)

    def rating_sampler(self, batch_size, segment='train', sequential=None, return_index=False):
        """Yields (node_pairs (2, B), ratings (B,)).  Train: random batches without replacement within a batch,
        forever; valid/test: sequential sweep once (reference :264-307).  return_index=True (train, random) also
        yields the positions of the batch inside the train pairs = the edge ids (CSR positions) of
        train_graph[user, item], which the resident plan masks on the device."""
        if segment == 'train':
            sequential = False if sequential is None else sequential
            pairs, ratings = self._train_node_pairs, self._train_ratings
        elif segment == 'valid':
            sequential = True if sequential is None else sequential
            pairs, ratings = self._valid_node_pairs, self._valid_ratings
        elif segment == 'test':
            sequential = True if sequential is None else sequential
            pairs, ratings = self._test_node_pairs, self._test_ratings
        else:
            raise NotImplementedError(segment)
        n = pairs.shape[1]
        batch_size = n if batch_size < 0 else min(batch_size, n)
        if sequential:
            for start in range(0, n, batch_size):
                yield pairs[:, start:start + batch_size], ratings[start:start + batch_size]
            return
        while True:
            if batch_size == n:
                yield (pairs, ratings, np.arange(n, dtype=np.int32)) if return_index else (pairs, ratings)
            else:
                sel = self._rng.choice(n, batch_size, replace=False)
                yield (pairs[:, sel], ratings[sel], sel.astype(np.int32)) if return_index else (pairs[:, sel], ratings[sel])

    def recon_nodes_sampler(self, batch_size):
        """Yields (embed_noise_dict, batch_recon_node_ids_dict, all_recon_node_ids_dict) forever (reference :309-370):
        each epoch re-draws the masked node set, then sweeps it in batches."""
        while True:
            noise_dict, recon_dict = dict(), dict()
            for key, ids in self._recon_train_candidates.items():
                k = int(np.ceil(self._embed_P_mask[key] * ids.size))
                perm = self._rng.permutation(ids)
                recon, remain = perm[:k], perm[k:]
                noise = -np.ones(self._all_graph.node_ids_dict[key].shape, dtype=np.int32)
                if recon.size > 0:
                    recon_dict[key] = recon
                    noise[remain] = remain
                    zero = self._rng.multinomial(1, [self._embed_p_zero[key], self._embed_p_self[key]], size=recon.size)[:, 0]
                    noise[recon] = np.where(zero == 1, -1, recon).astype(np.int32)
                else:
                    noise[ids] = ids
                noise_dict[key] = noise
            cur = {key: 0 for key in recon_dict}
            while True:
                batch = dict()
                for key, recon in recon_dict.items():
                    if cur[key] > recon.size:
                        continue
                    batch[key] = recon[cur[key]:cur[key] + batch_size]
                    cur[key] += batch_size
                if len(batch) != len(recon_dict) or any(v.size == 0 for v in batch.values()):
                    break
                yield noise_dict, batch, recon_dict
