"""Ranking metrics — mirrors emgraph/evaluation/metrics.py (numpy one-liners consumed after ranking).
Ranks are flattened first (metrics.py:66,129,221), so [n,2] 's,o' ranks average over 2n entries."""
import numpy as np


def hits_at_n_score(ranks, n):
    """metrics.py:11-67: fraction of ranks <= n."""
    if isinstance(ranks, list):
        ranks = np.asarray(ranks)
    ranks = ranks.reshape(-1)
    return np.sum(ranks <= n) / len(ranks)


def mrr_score(ranks):
    """metrics.py:70-130: mean reciprocal rank."""
    if isinstance(ranks, list):
        ranks = np.asarray(ranks)
    ranks = ranks.reshape(-1)
    return np.sum(1 / ranks) / len(ranks)


def rank_score(y_true, y_pred, pos_lab=1):
    """metrics.py:133-164: rank of the positive element among the scores."""
    idx = np.argsort(y_pred)[::-1]
    y_ord = y_true[idx]
    rank = np.where(y_ord == pos_lab)[0][0] + 1
    return rank


def mr_score(ranks):
    """metrics.py:167-222: mean rank."""
    if isinstance(ranks, list):
        ranks = np.asarray(ranks)
    ranks = ranks.reshape(-1)
    return np.sum(ranks) / len(ranks)
