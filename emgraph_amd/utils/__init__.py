"""Mirrors emgraph/utils/__init__.py: save_model / restore_model (utils/model_utils.py:22-164)."""
from .model_utils import restore_model, save_model

__all__ = ["save_model", "restore_model"]
