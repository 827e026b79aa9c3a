"""Mirrors emgraph/evaluation/__init__.py exports for the hot path."""
from .metrics import hits_at_n_score, mr_score, mrr_score, rank_score
from .protocol import (check_filter_size, create_mappings, evaluate_performance, filter_unseen_entities,
                       generate_corruptions_for_eval, generate_corruptions_for_fit, to_idx)
from .ranking import FilterIndex, L2Tables, PrefilterTables, SadTables, build_filter_csr, rank_triples_device, ranks_from_counts

__all__ = ["hits_at_n_score", "mr_score", "mrr_score", "rank_score", "check_filter_size", "create_mappings",
           "evaluate_performance", "filter_unseen_entities", "generate_corruptions_for_eval",
           "generate_corruptions_for_fit", "to_idx", "FilterIndex", "PrefilterTables", "SadTables", "L2Tables", "build_filter_csr", "rank_triples_device",
           "ranks_from_counts"]
