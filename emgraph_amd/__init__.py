"""emgraph_amd — MI355X-native replacement for bi-graph/Emgraph's per-batch hot path
(score functions, eta-way corruption generator, pairwise/NLL losses, filtered 1-vs-all ranking)
behind the reference's model.fit() / model.predict() / evaluate_performance() API.

Compute lives in libemgraph_hip.so (hand-written HIP for gfx950) reached through a ctypes C-ABI
(include/emgraph_hip.h); PyTorch-ROCm tensors only hold device memory.  No TensorFlow, no CPU fallback.
"""
__version__ = "0.1.0"

from .evaluation import (evaluate_performance, hits_at_n_score, mr_score, mrr_score, rank_score)  # noqa: E402,F401
from .models import ComplEx, DistMult, HolE, TransE  # noqa: E402,F401
from .utils import restore_model, save_model  # noqa: E402,F401
