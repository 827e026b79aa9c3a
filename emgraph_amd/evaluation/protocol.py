"""Host-side protocol functions — mirrors the hot-path part of emgraph/evaluation/protocol.py
(same names, argument meaning and error behaviour); the numeric work goes to libemgraph_hip.so.

Not mirrored (out of scope): train_test_split_no_unseen, select_best_model_ranking and the
param-grid helpers — pure callers of fit()/evaluate_performance() that run unchanged on top of this API.
"""
from __future__ import annotations

import logging
import warnings

import numpy as np
import torch

from .. import _lib as L
from .. import device as D

logger = logging.getLogger(__name__)

TOO_MANY_ENTITIES_TH = 50000  # protocol.py:21

_UNSEEN_MSG = (
    "Input triples include one or more {concept_type} not present in the training set. "
    "Please filter all concepts in X that do not occur in the training test "
    "(set filter_unseen=True in evaluate_performance) or retrain the model on a "
    "training set that includes all the desired concept types."
)


def create_mappings(X):
    """protocol.py:429-445: ids are ranks in np.unique (sorted) order.  Returns (rel_to_idx, ent_to_idx)."""
    unique_ent = np.unique(np.concatenate((X[:, 0], X[:, 2])))
    unique_rel = np.unique(X[:, 1])
    ent_to_idx = dict(zip(unique_ent, range(len(unique_ent))))
    rel_to_idx = dict(zip(unique_rel, range(len(unique_rel))))
    return rel_to_idx, ent_to_idx


def create_mappings_and_index(X):
    """create_mappings + to_idx of the SAME array in one vectorised pass (np.unique(return_inverse));
    replaces np.vectorize(dict.get) over the training set (protocol.py:682-684, minutes at 10M rows)."""
    n = X.shape[0]
    unique_ent, inv_e = np.unique(np.concatenate((X[:, 0], X[:, 2])), return_inverse=True)
    unique_rel, inv_r = np.unique(X[:, 1], return_inverse=True)
    X_idx = np.stack([inv_e[:n], inv_r, inv_e[n:]], axis=1).astype(np.int32)
    ent_to_idx = dict(zip(unique_ent, range(len(unique_ent))))
    rel_to_idx = dict(zip(unique_rel, range(len(unique_rel))))
    return rel_to_idx, ent_to_idx, X_idx


import weakref

# sorted key / id tables of the label dictionaries most recently looked up, keyed by the dictionary's identity.  Held
# WEAKLY where the dictionary allows it (so a deleted model's 1M-entity tables go with it) and validated by size plus a
# fingerprint of the first / last items (a dictionary changed in place between two calls is rebuilt, not trusted).
_LOOKUP_CACHE = []   # [(ref or dict, len, fingerprint, sorted keys, their ids, direct table or None)]


def _fingerprint(mapping):
    it = iter(mapping.items())
    first = next(it, None)
    last = next(reversed(mapping.items()), None) if first is not None else None
    return (first, last)


def _lookup_tables(mapping):
    """sorted key / id arrays of a label dictionary (+ a direct label -> id table for dense non-negative integer labels),
    remembered for the dictionaries most recently used: evaluate_performance maps the test set and the (large) filter
    set through the same two dictionaries every call."""
    fp = _fingerprint(mapping)
    for ent in list(_LOOKUP_CACHE):
        held = ent[0]() if isinstance(ent[0], weakref.ReferenceType) else ent[0]
        if held is None:
            _LOOKUP_CACHE.remove(ent)
        elif held is mapping and ent[1] == len(mapping) and ent[2] == fp:
            return ent[3], ent[4], ent[5]
    keys = np.array(list(mapping.keys()))
    vals = np.fromiter(mapping.values(), dtype=np.int64, count=len(mapping))
    order = np.argsort(keys, kind="stable")
    skeys, svals = keys[order], vals[order]
    table = None
    if skeys.dtype.kind in "iu" and len(skeys) and int(skeys[0]) >= 0 and int(skeys[-1]) < 8 * len(skeys) + 1024:
        table = np.full(int(skeys[-1]) + 1, -1, dtype=np.int64)
        table[skeys] = svals
    try:
        holder = weakref.ref(mapping)
    except TypeError:       # a plain dict cannot be weakly referenced: keep it (at most 4 entries live here)
        holder = mapping
    _LOOKUP_CACHE.insert(0, (holder, len(mapping), fp, skeys, svals, table))
    del _LOOKUP_CACHE[4:]
    return skeys, svals, table


def _lookup(col, mapping):
    """vectorised dict lookup; returns (ids, ok_mask)."""
    if len(mapping) == 0:
        return np.zeros(len(col), np.int64), np.zeros(len(col), bool)
    skeys, svals, table = _lookup_tables(mapping)
    col = np.asarray(col)
    if skeys.dtype.kind != col.dtype.kind and not (skeys.dtype.kind in "US" and col.dtype.kind in "US") \
            and not (skeys.dtype.kind in "iu" and col.dtype.kind in "iu"):
        got = [mapping.get(v) for v in col.tolist()]
        ok = np.array([g is not None for g in got], dtype=bool)
        return np.array([g if g is not None else 0 for g in got], dtype=np.int64), ok
    if table is not None and col.dtype.kind in "iu":   # dense integer labels: one gather
        inside = (col >= 0) & (col < len(table))
        ids = table[np.where(inside, col, 0)]
        ok = inside & (ids >= 0)
        return np.where(ok, ids, 0), ok
    if skeys.dtype.kind in "iu" and col.dtype.kind in "iu" and skeys.dtype != col.dtype:
        # mixed integer types (int64 labels looked up with uint32 ids, ...): compare as int64, never through float64
        if (col.dtype.kind == "u" and col.size and int(col.max()) > np.iinfo(np.int64).max) or \
                (skeys.dtype.kind == "u" and int(skeys[-1]) > np.iinfo(np.int64).max):
            got = [mapping.get(v) for v in col.tolist()]
            ok = np.array([g is not None for g in got], dtype=bool)
            return np.array([g if g is not None else 0 for g in got], dtype=np.int64), ok
        skeys, col = skeys.astype(np.int64), col.astype(np.int64)
    pos = np.searchsorted(skeys, col)
    pos_c = np.minimum(pos, len(skeys) - 1)
    ok = skeys[pos_c] == col
    return svals[pos_c], ok


def to_idx(X, ent_to_idx, rel_to_idx):
    """protocol.py:662-723.  Unseen entity/relation -> ValueError with the reference's message."""
    X = np.asarray(X)
    if X.ndim == 1:
        X = X[np.newaxis, :]
    s, ok_s = _lookup(X[:, 0], ent_to_idx)
    p, ok_p = _lookup(X[:, 1], rel_to_idx)
    o, ok_o = _lookup(X[:, 2], ent_to_idx)
    if not (ok_s.all() and ok_o.all()):
        msg = _UNSEEN_MSG.format(concept_type="entities")
        logger.error(msg)
        raise ValueError(msg)
    if not ok_p.all():
        msg = _UNSEEN_MSG.format(concept_type="relations")
        logger.error(msg)
        raise ValueError(msg)
    return np.dstack([s, p, o]).reshape((-1, 3))


def _as_int_seed(rnd):
    if rnd is None:
        return 0
    if isinstance(rnd, (int, np.integer)):
        return int(rnd)
    if isinstance(rnd, np.random.RandomState):
        return int(rnd.randint(0, 2 ** 31 - 1))
    raise ValueError("rnd must be None, an int seed or a numpy RandomState")


def generate_corruptions_for_fit(X, entities_list=None, eta=1, corrupt_side="s,o", entities_size=0, rnd=None,
                                 draw_counter=0):
    """protocol.py:531-659 on the GPU: eta-major corruptions of the positives ``X`` (int [n,3]).

    Same arguments as the reference.  The replacement/mask draws come from the on-device Philox4x32-10
    stream keyed by ``rnd`` (an int seed) and ``draw_counter`` — NOT TensorFlow's stream (which cannot be
    reproduced without TensorFlow); the distribution is the same: mask ~ U{0,1} (only for 's+o'/'s,o'),
    replacement ~ U{0..entities_size-1} or a uniform pick from ``entities_list`` / the batch entities.
    Returns an int32 ndarray [n*eta, 3]."""
    if corrupt_side == "s,o":
        corrupt_side = "s+o"
    if corrupt_side not in ["s+o", "s", "o"]:
        msg = "Invalid argument value {} for corruption side passed for evaluation.".format(corrupt_side)
        logger.error(msg)
        raise ValueError(msg)
    D.require_gpu()
    X = np.ascontiguousarray(np.asarray(X), dtype=np.int32).reshape(-1, 3)
    B = X.shape[0]
    dev = torch.device("cuda")
    elist = None
    if entities_size != 0:
        n_choices = int(entities_size)
    else:
        if entities_list is None:
            entities_list = batch_entities(X)
        elist = torch.from_numpy(np.ascontiguousarray(np.asarray(entities_list, dtype=np.int32).reshape(-1))).to(dev)
        n_choices = int(elist.numel())
    Xt = torch.from_numpy(X).to(dev)
    codes = D.corrupt_codes(B, int(eta), L.SIDE_IDS[corrupt_side], n_choices, dev, entities_list=elist,
                            seed=_as_int_seed(rnd), counter=int(draw_counter))
    return D.corrupt_expand(Xt, int(eta), codes).cpu().numpy()


def batch_entities(X):
    """protocol.py:621-633: tf.unique(concat(subjects, objects)) — first-appearance order."""
    cat = np.concatenate([X[:, 0], X[:, 2]])
    _, first = np.unique(cat, return_index=True)
    return cat[np.sort(first)].astype(np.int32)


def generate_corruptions_for_eval(X, entities_for_corruption, corrupt_side="s,o"):
    """protocol.py:448-528: the [|C|*sides, 3] corruption array of ONE triple (object block first).

    Index tiling only (no arithmetic).  Kept for API parity; the ranking path never materialises it —
    the 1-vs-all kernels enumerate the candidates implicitly."""
    X = np.asarray(X).reshape(1, 3)
    C = np.asarray(entities_for_corruption).reshape(-1)
    if corrupt_side == "s,o":
        corrupt_side = "s+o"
    if corrupt_side not in ["s+o", "s", "o"]:
        msg = "Invalid argument value for corruption side passed for evaluation"
        logger.error(msg)
        raise ValueError(msg)
    n = len(C)
    s, p, o = (np.full(n, X[0, i], dtype=C.dtype) for i in range(3))
    obj_block = np.stack([s, p, C], axis=1)
    subj_block = np.stack([C, p, o], axis=1)
    if corrupt_side == "s+o":
        return np.concatenate([obj_block, subj_block], axis=0)
    return obj_block if corrupt_side == "o" else subj_block


_POOL_WARNING = """You are attempting to use %d distinct entities to generate synthetic negatives in the evaluation
    protocol. This may be unnecessary and will lead to a 'harder' task. Besides, it will lead to a much slower
    evaluation procedure. We recommended to set the 'corruption_entities' argument to a reasonably sized set
    of entities. The size of corruption_entities depends on your domain-specific task."""


def check_filter_size(model, corruption_entities):
    """Warn when the corruption pool of an evaluation is very large (the reference's message, protocol.py:982-1011):
    the pool is every entity the model knows unless a subset is given."""
    pool = model.ent_to_idx if corruption_entities is None else corruption_entities
    if len(pool) < TOO_MANY_ENTITIES_TH:
        return
    warnings.warn(_POOL_WARNING % len(pool))
    logger.warning(_POOL_WARNING, len(pool))


def filter_unseen_entities(X, model, verbose=False):
    """protocol.py:1014-1041: drop triples whose subject or object the model has not seen."""
    X = np.asarray(X)
    # membership through the cached label tables of to_idx (the reference builds np.array(list(keys)) and runs np.isin
    # on every call: 60 ms per call at 1M entities)
    keep = _lookup(X[:, 0], model.ent_to_idx)[1] & _lookup(X[:, 2], model.ent_to_idx)[1]
    n_removed = int((~keep).sum())
    if n_removed > 0:
        msg = "Removing {} triples containing unseen entities. ".format(n_removed)
        if verbose:
            logger.info(msg)
        logger.debug(msg)
        return X[keep]
    return X


_SIDES = ("s", "o", "s+o", "s,o")
_STRATEGIES = ("worst", "best", "middle")


def _test_adapter(X, model, filter_unseen, verbose):
    """(adapter holding the mapped 'test' set, whether the caller supplied it).  An array of labels is wrapped in a
    NumpyDatasetAdapter carrying the model's mappings (protocol.py:879-895); an adapter is used as it is."""
    from ..datasets import EmgraphBaseDatasetAdaptor, NumpyDatasetAdapter
    if isinstance(X, EmgraphBaseDatasetAdaptor):
        return X, True
    if not isinstance(X, np.ndarray):
        msg = "X must be either a numpy array or an EmgraphBaseDatasetAdaptor."
        logger.error(msg)
        raise ValueError(msg)
    if filter_unseen:
        X = filter_unseen_entities(X, model, verbose=verbose)
    else:
        logger.warning("If your test set or filter triples contain unseen entities you may get a"
                       "runtime error. You can filter them by setting filter_unseen=True")
    adapter = NumpyDatasetAdapter()
    adapter.use_mappings(model.rel_to_idx, model.ent_to_idx)
    adapter.set_data(X, "test")
    return adapter, False


_FILTER_CACHE = []   # [(key, mapped int64 [n, 3], FilterIndex, ent_to_idx, rel_to_idx)] of the filter sets most recently installed (at most 2)


def _array_digest(a):
    """content hash of a numeric / fixed-width-string array (None for object arrays): 24 MB of filter triples in 2-3 ms"""
    a = np.asarray(a)
    if a.dtype.kind not in "iufUS" or a.size == 0:
        return None
    a = np.ascontiguousarray(a)
    try:
        import xxhash
        h = xxhash.xxh3_128(memoryview(a).cast("B")).hexdigest()
    except ImportError:
        import hashlib
        h = hashlib.blake2b(memoryview(a).cast("B"), digest_size=16).hexdigest()
    return (h, a.shape, a.dtype.str)


def _mapped_filter(adapter, filter_triples, model, filter_unseen, verbose):
    """filter_unseen_entities + to_idx + FilterIndex of a filter array — five label lookups over every filter triple and the
    index build, 45 of the 107 ms of an evaluate_performance call at 1M filter triples — remembered by CONTENT (digest of the
    array + identity and fingerprint of the two label dictionaries): early stopping and repeated evaluations pass the same
    filter every time.  A changed array, or changed mappings, miss."""
    dg = _array_digest(filter_triples)
    # the two dictionaries are HELD by the entry and compared with `is` (an id() alone can be reused by a later dictionary of the
    # same size and end items once the model is gone: round-5 advisor); at most two entries live here, clear_filter_cache() drops them
    key = None if dg is None else (dg, bool(filter_unseen), len(model.ent_to_idx), _fingerprint(model.ent_to_idx),
                                   len(model.rel_to_idx), _fingerprint(model.rel_to_idx))
    if key is not None:
        for ent in _FILTER_CACHE:
            if ent[0] == key and ent[3] is model.ent_to_idx and ent[4] is model.rel_to_idx:
                adapter.filter_adapter, adapter.filter_index = ent[1], ent[2]
                return
    if filter_unseen:
        filter_triples = filter_unseen_entities(filter_triples, model, verbose=verbose)
    adapter.set_filter(filter_triples)
    if key is not None and getattr(adapter, "filter_index", None) is not None:
        _FILTER_CACHE.insert(0, (key, adapter.filter_adapter, adapter.filter_index, model.ent_to_idx, model.rel_to_idx))
        del _FILTER_CACHE[2:]


def clear_filter_cache():
    """drop the remembered mapped filters, their indices and the label dictionaries they were mapped through"""
    del _FILTER_CACHE[:]
    del _LOOKUP_CACHE[:]


def _install_filter(adapter, own_adapter, filter_triples, model, filter_unseen, verbose):
    """filter_triples: an array of known positives (any test input), or — with a caller-supplied adapter — a bool saying
    whether the filter already set in that adapter is to be used (protocol.py:897-929)."""
    if filter_triples is None:
        return
    if isinstance(filter_triples, np.ndarray):
        if hasattr(adapter, "filter_index"):     # this package's NumpyDatasetAdapter: the mapped filter and its index are reusable
            _mapped_filter(adapter, filter_triples, model, filter_unseen, verbose)
        else:
            if filter_unseen:
                filter_triples = filter_unseen_entities(filter_triples, model, verbose=verbose)
            adapter.set_filter(filter_triples)
        model.set_filter_for_eval()
    elif not own_adapter:
        raise Exception("Invalid datatype for filter. Expected a numpy array or preset data in the adapter.")
    elif not isinstance(filter_triples, bool):
        raise Exception("Expected a boolean type")
    elif filter_triples:
        model.set_filter_for_eval()


def evaluate_performance(X, model, filter_triples=None, verbose=False, filter_unseen=True, entities_subset=None,
                         corrupt_side="s,o", ranking_strategy="worst", use_default_protocol=False):
    """Ranks of the positives in ``X`` against their corruptions (the reference's evaluation protocol,
    protocol.py:726-979; the ranking itself is ``model.get_ranks`` -> emgraph_amd.evaluation.ranking on the GPU).

    ``X``: ndarray [n,3] of labels, or an EmgraphBaseDatasetAdaptor holding a mapped 'test' set (then
    ``filter_triples`` is a bool: True = use the filter already set in the adapter).
    Returns an int ndarray [n] ('s', 'o', 's+o') or [n,2] = [subject_rank, object_rank] ('s,o').
    Whatever fails, the model leaves evaluation mode and an adapter created here is cleaned up before the error
    propagates (:975-979)."""
    if use_default_protocol:
        logger.warning("DeprecationWarning: use_default_protocol will be removed in future. "
                       "Please use corrupt_side argument instead.")
        corrupt_side = "s,o"
    adapter = None
    try:
        assert corrupt_side in _SIDES, "Invalid value for corrupt_side."
        adapter, own_adapter = _test_adapter(X, model, filter_unseen, verbose)
        _install_filter(adapter, own_adapter, filter_triples, model, filter_unseen, verbose)
        check_filter_size(model, entities_subset)
        assert ranking_strategy in _STRATEGIES, "Invalid ranking_strategy!"
        settings = {"corrupt_side": corrupt_side, "ranking_strategy": ranking_strategy}
        if entities_subset is not None:
            wanted = set(entities_subset)
            settings["corruption_entities"] = np.asarray([i for label, i in model.ent_to_idx.items() if label in wanted])
        model.configure_evaluation_protocol(settings)
        if getattr(model, "_ranks_as_array", False):   # this package's models hand the array over as it is
            ranks = np.asarray(model.get_ranks(adapter, as_array=True))
        else:                                          # any other model: the reference's lists
            ranks = np.array(model.get_ranks(adapter))
    except BaseException:
        model.end_evaluation()
        if adapter is not None:
            adapter.cleanup()
        raise
    model.end_evaluation()
    return ranks
