"""Mirrors emgraph/models/__init__.py for the hot-path models."""
from .embedding_model import (MODEL_REGISTRY, ComplEx, DistMult, EmbeddingModel, HolE, TransE,
                              reset_entity_threshold, set_entity_threshold)

__all__ = ["EmbeddingModel", "TransE", "DistMult", "ComplEx", "HolE", "MODEL_REGISTRY", "set_entity_threshold",
           "reset_entity_threshold"]
