"""Ranking metrics with the reference's names and semantics (emgraph/evaluation/metrics.py:11-222).

All three rank statistics first flatten their input, so the [n, 2] array 's,o' evaluation returns is averaged
over its 2n entries (SURVEY A-10 notwithstanding: that is what the reference's numpy code does, and what its
golden values in tests/emgraph/evaluation/test_metrics.py pin)."""
import numpy as np


def _flat(ranks):
    """lists and [n, sides] arrays -> one 1-D array of ranks"""
    return np.asarray(ranks).reshape(-1)


def hits_at_n_score(ranks, n):
    """share of the ranks that are at most ``n`` (Hits@N)"""
    r = _flat(ranks)
    return np.count_nonzero(r <= n) / r.size


def mrr_score(ranks):
    """mean of the reciprocal ranks (MRR)"""
    r = _flat(ranks)
    return np.reciprocal(r.astype(np.float64)).sum() / r.size


def mr_score(ranks):
    """arithmetic mean of the ranks (MR)"""
    r = _flat(ranks)
    return r.sum() / r.size


def rank_score(y_true, y_pred, pos_lab=1):
    """1-based position of the first ``pos_lab`` label when the candidates are ordered by descending score"""
    by_score_desc = np.flip(np.argsort(np.asarray(y_pred)))
    hits = np.flatnonzero(np.asarray(y_true)[by_score_desc] == pos_lab)
    return int(hits[0]) + 1
