"""EmbeddingModel and the four hot-path models — host-side mirror of emgraph/models/EmbeddingModel.py
(fit :1113-1492, predict :2101-2186, get_ranks :2046-2099, get_embeddings :455-488) and of
TransE.py / DistMult.py / ComplEx.py / HolE.py (which only contribute ``_fn`` and ``internal_k``).

Same constructor arguments, defaults (utils/constants.py) and error behaviour; the per-batch work is
done by libemgraph_hip.so through emgraph_amd.training.Trainer / emgraph_amd.evaluation.ranking.

Deliberate deviations from the reference's *literal* TF2-port behaviour, all in
favour of its intended (AmpliGraph-1.x) semantics: every batch is trained on exactly once (A-2);
optimizer state persists across batches (A-3); ranks are computed per test triple (A-1); both
'corrupt_side' and 'corrupt_sides' keys are honoured (A-5); large-graph host paging is not needed on a
288 GB GPU, so |E| > ENTITY_THRESHOLD neither pages nor forces SGD.
"""
from __future__ import annotations

import abc
import logging
import os

import numpy as np
import torch

from .. import _lib as L
from .. import device as D
from .. import parallel
from ..datasets import EmgraphBaseDatasetAdaptor
from ..evaluation.metrics import hits_at_n_score, mrr_score
from ..evaluation.protocol import _lookup, create_mappings_and_index, to_idx
from ..evaluation.ranking import FilterIndex, rank_triples_device
from ..training import Trainer, alloc_table

logger = logging.getLogger(__name__)

MODEL_REGISTRY = {}

# utils/constants.py
DEFAULT_EMBEDDING_SIZE = 100
DEFAULT_ETA = 2
DEFAULT_EPOCH = 100
DEFAULT_BATCH_COUNT = 100
DEFAULT_SEED = 0
DEFAULT_OPTIM = "adam"
DEFAULT_LR = 0.0005
DEFAULT_LOSS = "nll"
DEFAULT_REGULARIZER = None
DEFAULT_INITIALIZER = "glorot_uniform"
DEFAULT_VERBOSE = False
DEFAULT_NORM_TRANSE = 1
DEFAULT_CORRUPTION_ENTITIES = "all"
DEFAULT_CORRUPT_SIDE_TRAIN = ["s,o"]
DEFAULT_CORRUPT_SIDE_EVAL = "s,o"
DEFAULT_NORMALIZE_EMBEDDINGS = False
DEFAULT_BURN_IN_EARLY_STOPPING = 100
DEFAULT_CHECK_INTERVAL_EARLY_STOPPING = 10
DEFAULT_STOP_INTERVAL_EARLY_STOPPING = 3
DEFAULT_CRITERIA_EARLY_STOPPING = "mrr"
DEFAULT_RANK_COMPARE_STRATEGY = "worst"
# initializers/_initializer_constants.py
DEFAULT_UNIFORM_LOW, DEFAULT_UNIFORM_HIGH = -0.05, 0.05
DEFAULT_NORMAL_MEAN, DEFAULT_NORMAL_STD = 0, 0.05
DEFAULT_GLOROT_IS_UNIFORM = False

ENTITY_THRESHOLD = 5e5  # EmbeddingModel.py:37 (kept for API parity; no host paging here)

LOSSES = ("pairwise", "nll", "absolute_margin", "self_adversarial", "multiclass_nll")
OPTIMIZERS = ("adam", "adagrad", "sgd", "momentum", "adam_lazy")
INITIALIZERS = ("glorot_uniform", "normal", "uniform", "constant")


def set_entity_threshold(threshold):
    """EmbeddingModel.py:41-49 (API parity)."""
    global ENTITY_THRESHOLD
    ENTITY_THRESHOLD = threshold


def reset_entity_threshold():
    """EmbeddingModel.py:52-58 (API parity)."""
    global ENTITY_THRESHOLD
    ENTITY_THRESHOLD = 5e5


def register_model(name):
    def deco(cls):
        MODEL_REGISTRY[name] = cls
        cls.name = name
        return cls
    return deco


def _initial_table(kind, params, rnd, rows, cols, concept):
    """initializers/*.py numpy paths.  glorot_uniform follows the reference's TF path, which ALWAYS
    returns tf.initializers.GlorotUniform() whatever the 'uniform' flag says (glorot_uniform.py:59-74,
    SURVEY A-11): U(+-sqrt(6/(rows+cols))).  The draws themselves are parity-unpinned (TF RNG)."""
    if kind == "glorot_uniform":
        limit = np.sqrt(6 / (rows + cols))
        return rnd.uniform(-limit, limit, size=(rows, cols)).astype(np.float32)
    if kind == "normal":
        return rnd.normal(params.get("mean", DEFAULT_NORMAL_MEAN), params.get("std", DEFAULT_NORMAL_STD),
                          size=(rows, cols)).astype(np.float32)
    if kind == "uniform":
        return rnd.uniform(params.get("low", DEFAULT_UNIFORM_LOW), params.get("high", DEFAULT_UNIFORM_HIGH),
                           size=(rows, cols)).astype(np.float32)
    if kind == "constant":
        arr = np.asarray(params["entity" if concept == "e" else "relation"], dtype=np.float32)
        assert arr.shape == (rows, cols), "Invalid shape for {} initializer!".format(
            "entity" if concept == "e" else "relation")
        return arr
    raise ValueError("Unsupported initializer: {}".format(kind))


def _initial_table_device(kind, params, seed, rows, cols, concept, device):
    """the same initialisers drawn on the device (emg_init_table, Philox keyed by (seed, table)): no 1M x 400 float64
    array on the host.  Returns a float32 device tensor [rows, cols] (row stride padded to 16 bytes), or None for
    initialisers that have no device form ('constant')."""
    from ..training import alloc_table
    stream_id = 1 if concept == "e" else 2
    if kind == "glorot_uniform":
        limit = float(np.sqrt(6 / (rows + cols)))
        spec = ("uniform", -limit, limit)
    elif kind == "uniform":
        spec = ("uniform", params.get("low", DEFAULT_UNIFORM_LOW), params.get("high", DEFAULT_UNIFORM_HIGH))
    elif kind == "normal":
        spec = ("normal", params.get("mean", DEFAULT_NORMAL_MEAN), params.get("std", DEFAULT_NORMAL_STD))
    else:
        return None
    t = alloc_table(rows, cols, device)
    return D.init_table(t, cols, spec[0], spec[1], spec[2], seed, stream_id)


class EmbeddingModel(abc.ABC):  # noqa: B024
    """Abstract base of the embedding models (EmbeddingModel.py:96-336)."""

    name = "EmbeddingModel"

    def __init__(self, k=DEFAULT_EMBEDDING_SIZE, eta=DEFAULT_ETA, epochs=DEFAULT_EPOCH,
                 batches_count=DEFAULT_BATCH_COUNT, seed=DEFAULT_SEED, embedding_model_params={},
                 optimizer=DEFAULT_OPTIM, optimizer_params={"lr": DEFAULT_LR}, loss=DEFAULT_LOSS, loss_params={},
                 regularizer=DEFAULT_REGULARIZER, regularizer_params={}, initializer=DEFAULT_INITIALIZER,
                 initializer_params={"uniform": DEFAULT_GLOROT_IS_UNIFORM}, large_graphs=False,
                 verbose=DEFAULT_VERBOSE):
        if loss == "bce":  # EmbeddingModel.py:206-210: BCE is ConvE-only
            raise ValueError("Invalid Model - Loss combination. "
                             "ConvE model can be used with BCE loss only and vice versa.")
        self.all_params = {
            "k": k, "eta": eta, "epochs": epochs, "batches_count": batches_count, "seed": seed,
            "embedding_model_params": embedding_model_params, "optimizer": optimizer,
            "optimizer_params": optimizer_params, "loss": loss, "loss_params": loss_params,
            "regularizer": regularizer, "regularizer_params": regularizer_params, "initializer": initializer,
            "initializer_params": initializer_params, "verbose": verbose,
        }
        self.seed = seed
        self.loss_params = loss_params
        # a copy: the subclasses' default dictionaries are shared objects (EmbeddingModel.py has the same mutable defaults;
        # there a model that edits its params edits every later model's defaults)
        self.embedding_model_params = dict(embedding_model_params)
        self.k = k
        self.internal_k = k
        self.epochs = epochs
        self.eta = eta
        self.regularizer_params = regularizer_params
        self.batches_count = batches_count
        self.dealing_with_large_graphs = large_graphs
        if batches_count == 1:
            logger.warning("All triples will be processed in the same batch (batches_count=1). "
                           "When processing large graphs it is recommended to batch the input knowledge graph "
                           "instead.")
        if loss not in LOSSES:
            msg = "Unsupported loss function: {}".format(loss)
            logger.error(msg)
            raise ValueError(msg)
        self.loss = loss
        if regularizer is not None and regularizer != "LP":
            msg = "Unsupported regularizer: {}".format(regularizer)
            logger.error(msg)
            raise ValueError(msg)
        self.regularizer = regularizer
        if optimizer not in OPTIMIZERS:
            msg = "Unsupported optimizer: {}".format(optimizer)
            logger.error(msg)
            raise ValueError(msg)
        self.optimizer = optimizer
        self.optimizer_params = optimizer_params
        self.verbose = verbose
        if initializer not in INITIALIZERS:
            msg = "Unsupported initializer: {}".format(initializer)
            logger.error(msg)
            raise ValueError(msg)
        self.initializer = initializer
        self.initializer_params = initializer_params
        self.trained_model_params = []
        self.is_fitted = False
        self.is_filtered = False
        self.eval_config = {}
        self.is_calibrated = False
        self.calibration_parameters = []
        self.ent_to_idx = {}
        self.rel_to_idx = {}
        self._dev = None  # (ent, rel) device tables of the trained parameters

    # ---- model-specific hooks ----
    def _model_id(self):
        raise NotImplementedError

    def _scale(self):
        return 1.0

    def _fn(self, e_s, e_p, e_o):
        """Score of already-gathered embedding rows [n, internal_k] — the reference's extension contract
        (EmbeddingModel.py:316-336; TransE.py:208-216, DistMult.py:201, ComplEx.py:288-298, HolE.py:189).
        Runs through the same HIP gather+score kernel as everything else (rows are staged as two small
        tables); fit/predict/evaluation never call it because their gathers are fused into the kernels."""
        D.require_gpu()
        dev = torch.device("cuda")
        rows = [torch.as_tensor(np.asarray(a, dtype=np.float32) if not isinstance(a, torch.Tensor) else a,
                                dtype=torch.float32, device=dev) for a in (e_s, e_p, e_o)]
        n, kk = rows[0].shape
        ent = alloc_table(2 * n, kk, dev)
        rel = alloc_table(n, kk, dev)
        ent[:n].copy_(rows[0])
        ent[n:].copy_(rows[2])
        rel.copy_(rows[1])
        ar = torch.arange(n, dtype=torch.int32, device=dev)
        spo = torch.stack([ar, ar, ar + n], dim=1).contiguous()
        return D.score_triples(self._model_id(), ent, rel, kk, self._scale(), spo).cpu().numpy()

    # ---- bookkeeping mirrored from the reference ----
    def get_hyperparameter_dict(self):
        return self.all_params

    def get_embedding_model_params(self, output_dict):
        output_dict["model_params"] = self.trained_model_params
        output_dict["large_graph"] = self.dealing_with_large_graphs
        output_dict["calibration_parameters"] = self.calibration_parameters

    def restore_model_params(self, in_dict):
        self.trained_model_params = [np.asarray(p, dtype=np.float32) for p in in_dict["model_params"]]
        self.calibration_parameters = in_dict.get("calibration_parameters", [])
        self.dealing_with_large_graphs = in_dict.get("large_graph", False)
        self._dev = None

    def get_embeddings(self, entities, embedding_type="entity"):
        """EmbeddingModel.py:455-488."""
        if not self.is_fitted:
            msg = "Model has not been fitted."
            logger.error(msg)
            raise RuntimeError(msg)
        if embedding_type == "entity":
            emb_list, lookup_dict = self.trained_model_params[0], self.ent_to_idx
        elif embedding_type == "relation":
            emb_list, lookup_dict = self.trained_model_params[1], self.rel_to_idx
        else:
            msg = "Invalid entity type: {}".format(embedding_type)
            logger.error(msg)
            raise ValueError(msg)
        # one sorted-key search for the whole array (the reference maps label by label: np.vectorize(dict.get)); a label the
        # model was not fitted on fails like the reference's None index does
        labels = np.asarray(entities)
        idxs, known = _lookup(labels.reshape(-1), lookup_dict)
        if not known.all():
            raise IndexError("get_embeddings: label(s) not seen in training: %r" % (labels.reshape(-1)[~known][:5].tolist(),))
        return emb_list[idxs.reshape(labels.shape)]

    def is_fitted_on(self, X):
        """EmbeddingModel.py:2188-2210."""
        if not self.is_fitted:
            msg = "Model has not been fitted."
            logger.error(msg)
            raise RuntimeError(msg)
        unique_ent = np.unique(np.concatenate((X[:, 0], X[:, 2])))
        unique_rel = np.unique(X[:, 1])
        return len(unique_ent) == len(self.ent_to_idx) and len(unique_rel) == len(self.rel_to_idx)

    # ---- training ----
    def _corrupt_sides(self):
        p = self.embedding_model_params
        sides = p.get("corrupt_side", p.get("corrupt_sides", DEFAULT_CORRUPT_SIDE_TRAIN))  # SURVEY A-5
        if not isinstance(sides, list):
            sides = [sides]
        for s in sides:
            if s not in ("s", "o", "s+o", "s,o"):
                raise ValueError("Invalid argument value {} for corruption side passed for evaluation.".format(s))
        return sides

    def _negative_pool(self, X_idx, batch_size):
        """negative_corruption_entities (EmbeddingModel.py:731-783): 'all' | 'batch' | list of labels | int.
        Returns (n_choices or None, fixed entities_list tensor or None, per-batch lists or None)."""
        nce = self.embedding_model_params.get("negative_corruption_entities", DEFAULT_CORRUPTION_ENTITIES)
        if isinstance(nce, str) and nce == "all":
            return len(self.ent_to_idx), None, None
        if isinstance(nce, str) and nce == "batch":
            from ..evaluation.protocol import batch_entities
            lists = []
            for i in range(self.batches_count):
                xb = X_idx[i * batch_size:(i + 1) * batch_size]
                lists.append(torch.from_numpy(batch_entities(xb)).cuda() if len(xb) else None)
            return None, None, lists
        if isinstance(nce, list):
            wanted = set(nce)
            ids = np.asarray([idx for uri, idx in self.ent_to_idx.items() if uri in wanted], dtype=np.int32)
            return len(ids), torch.from_numpy(ids).cuda(), None
        if isinstance(nce, (int, np.integer)) and not isinstance(nce, bool):
            return int(nce), None, None
        raise ValueError("Invalid negative_corruption_entities: {}".format(nce))

    def fit(self, X, early_stopping=False, early_stopping_params={}, focusE_numeric_edge_values=None,
            tensorboard_logs_path=None):
        """Train the model (EmbeddingModel.py:1113-1492).  ``X``: ndarray [n,3] of labels."""
        D.require_gpu()
        if focusE_numeric_edge_values is not None:
            raise NotImplementedError("FocusE numeric edge values are outside the accelerated hot path")
        if self.embedding_model_params.get("non_linearity", "linear") != "linear":
            raise NotImplementedError("non_linearity other than 'linear' is outside the accelerated hot path")
        # EmbeddingModel.py:1218-1248: an ndarray is wrapped in a NumpyDatasetAdapter; an adapter is used as it is
        if isinstance(X, np.ndarray):
            if X.ndim != 2 or X.shape[1] != 3:
                msg = "Invalid size for input X. Expected (n,3):  got {}".format(X.shape)
                logger.error(msg)
                raise ValueError(msg)
            handle = None
        elif isinstance(X, EmgraphBaseDatasetAdaptor):
            handle = X
        else:
            msg = "Invalid type for input X. Expected ndarray/EmgraphDataset object, got {}".format(type(X))
            logger.error(msg)
            raise ValueError(msg)
        if handle is None:
            self.rel_to_idx, self.ent_to_idx, X_idx = create_mappings_and_index(X)
        else:
            # the adapter owns mapping and batching (abstract_dataset_adapter.py): ids come from ITS dictionaries and
            # the training set is the concatenation of its batches, which stay contiguous slices of the resident copy
            self.rel_to_idx, self.ent_to_idx = handle.generate_mappings()
            handle.map_data()
            parts = [np.asarray(b[0] if isinstance(b, (list, tuple)) else b, dtype=np.int32).reshape(-1, 3)
                     for b in handle.get_next_batch(self.batches_count, "train")]
            X_idx = np.concatenate(parts, axis=0) if parts else np.zeros((0, 3), np.int32)
            if X_idx.shape[0] != handle.get_size("train"):
                raise ValueError("the adapter's batches do not add up to get_size('train')")
        n = X_idx.shape[0]
        batch_size = int(np.ceil(n / self.batches_count))  # EmbeddingModel.py:1297-1301
        self.batch_size = batch_size
        rnd = np.random.RandomState(self.seed)  # refit -> same seed -> same run (EmbeddingModel.py:1286-1290)
        n_ent, n_rel = len(self.ent_to_idx), len(self.rel_to_idx)
        # drawn on the device where the initialiser has a device form (a pure function of (seed, table, shape): the same
        # table for any number of GPUs); 'constant' comes from the caller's arrays
        dev = torch.device("cuda")
        ent0 = _initial_table_device(self.initializer, self.initializer_params, self.seed, n_ent, self.internal_k, "e", dev)
        rel0 = _initial_table_device(self.initializer, self.initializer_params, self.seed, n_rel, self.internal_k, "r", dev)
        if ent0 is None:
            ent0 = _initial_table(self.initializer, self.initializer_params, rnd, n_ent, self.internal_k, "e")
            rel0 = _initial_table(self.initializer, self.initializer_params, rnd, n_rel, self.internal_k, "r")
        normalize = self.embedding_model_params.get("normalize_ent_emb", DEFAULT_NORMALIZE_EMBEDDINGS)
        rank, world = parallel.rank_world()
        # multi-GPU plan (one process per GPU): "k" = column slabs + all-reduce of partial scores (default);
        # "batch" = replicated tables, batch split over the ranks, sparse gradient-row exchange (parallel.py)
        sharding = self.embedding_model_params.get("sharding", os.environ.get("EMG_SHARDING", "k")) if world > 1 else None
        if sharding not in (None, "k", "batch"):
            raise ValueError("Invalid sharding {!r}: expected 'k' or 'batch'".format(sharding))
        self._sharded = sharding == "k"
        k_local = self.internal_k
        if self._sharded:
            # one process per GPU: this rank trains a COLUMN slab of both tables (parallel.py)
            if world > self.k:
                raise ValueError("k-sharded training needs k >= number of ranks (k={}, ranks={})".format(self.k, world))
            if normalize:
                raise NotImplementedError("normalize_ent_emb needs full rows; not available with k-sharded training")
            cplx = self.internal_k != self.k
            ent0 = parallel.shard_columns(ent0, rank, world, cplx)
            rel0 = parallel.shard_columns(rel0, rank, world, cplx)
            k_local = ent0.shape[1]
        tr = Trainer(self._model_id(), k_local, self._scale(), ent0, rel0, self.eta, loss=self.loss,
                     loss_params=self.loss_params, optimizer=self.optimizer, optimizer_params=self.optimizer_params,
                     corrupt_sides=self._corrupt_sides(), batches_count=self.batches_count, seed=self.seed,
                     regularizer=self.regularizer, regularizer_params=self.regularizer_params,
                     normalize_ent_emb=normalize, sharded=sharding or False,
                     shard_state=bool(self.embedding_model_params.get("shard_state", False)) and sharding == "batch")
        tr.set_training_set(X_idx, batch_size)
        n_choices, fixed_list, batch_lists = self._negative_pool(X_idx, batch_size)
        if normalize:  # EmbeddingModel.py:1371-1380: both tables clipped once before the loop
            D.clip_rows(tr.rel, self.internal_k, 1.0)
            D.clip_rows(tr.ent, self.internal_k, 1.0)
        self.early_stopping_params = early_stopping_params
        es = self._initialize_early_stopping() if early_stopping else None
        # reported average: sum(losses) / (batch_size * batches_count), batch_size *= eta when the loss
        # tiles the positives (EmbeddingModel.py:1343-1344,1456)
        denom = batch_size * self.batches_count
        if self.loss in ("pairwise", "nll", "absolute_margin"):
            denom *= self.eta
        try:
            from tqdm import tqdm
            epochs_iter = tqdm(range(1, self.epochs + 1), disable=(not self.verbose), unit="epoch")
        except ImportError:  # pragma: no cover
            epochs_iter = range(1, self.epochs + 1)
        self._trainer = tr
        self.epoch_losses = []   # sum of the batch losses of each epoch (what the progress message averages)

        def batch_args(epoch, batch):
            """(start, B, epoch, batch, n_choices, entities_list) of a batch, or None past the end"""
            epoch, batch = epoch + (batch - 1) // self.batches_count, (batch - 1) % self.batches_count + 1
            if epoch > self.epochs:
                return None
            start = (batch - 1) * batch_size
            B = max(0, min(batch_size, n - start))  # last batch may be short / empty (numpy_adapter.py:105-112)
            if batch_lists is not None:
                elist = batch_lists[batch - 1]
                return (start, B, epoch, batch, elist.numel() if elist is not None else 0, elist)
            return (start, B, epoch, batch, n_choices, fixed_list)

        for epoch in epochs_iter:
            tr.run_batches([batch_args(epoch, batch) for batch in range(1, self.batches_count + 1)])
            loss_epoch = tr.read_loss()
            self.epoch_losses.append(loss_epoch)
            # EmbeddingModel.py:1422-1427 (per epoch here).  The reference's loss is a float32 tensor: a loss beyond float32's range IS inf
            # there; the device accumulates the epoch in a double, which would let such a run pass (seed 96078 of the round-6 soak)
            with np.errstate(over="ignore"):
                loss32 = np.float32(loss_epoch)
            if np.isnan(loss32) or np.isinf(loss32):
                msg = "Loss is {}. Please change the hyperparameters.".format(loss32)
                logger.error(msg)
                raise ValueError(msg)
            if self.verbose:
                msg = "Average {} Loss: {:10f}".format(self.name, loss_epoch / denom)
                if es is not None and es["best"] is not None:
                    msg += " — Best validation ({}): {:5f}".format(es["criteria"], es["best"])
                logger.debug(msg)
                if hasattr(epochs_iter, "set_description"):
                    epochs_iter.set_description(msg)
            if es is not None and self._perform_early_stopping_test(epoch, es, tr):
                self.is_fitted = True
                return
        self._save_trained_params(tr, live=True)
        self.is_fitted = True

    def _save_trained_params(self, tr, live=False):
        """Snapshot the tables into ``trained_model_params`` (EmbeddingModel.py:385-401).  ``live=True`` only for
        the final save at the end of fit(): inference may then read the trainer's tables directly.  A snapshot
        taken by early stopping must NOT alias them — training continues in place for ``stop_interval`` more
        checks, and predict()/get_ranks() have to use the best snapshot (the reference always infers from
        trained_model_params, :403-453) — so the device copy is dropped and re-uploaded lazily."""
        if getattr(self, "_sharded", False):
            cplx = self.internal_k != self.k
            ent = parallel.unshard_columns(parallel.gather_slabs(tr.ent), self.k, cplx)
            rel = parallel.unshard_columns(parallel.gather_slabs(tr.rel), self.k, cplx)
            self.trained_model_params = [ent, rel]
            self._dev = None  # full tables are uploaded lazily for predict / evaluation
            return
        ent, rel = tr.tables_numpy()
        self.trained_model_params = [ent, rel]
        self._dev = (tr.ent, tr.rel) if live else None

    def _eval_tables(self, tr):
        """full-width device tables of the CURRENT training state (early stopping)"""
        if not getattr(self, "_sharded", False):
            tr.materialize()   # (deferred dense decay: every row up to date before it is read)
            return tr.ent, tr.rel
        cplx = self.internal_k != self.k
        ent = parallel.unshard_columns(parallel.gather_slabs(tr.ent), self.k, cplx)
        rel = parallel.unshard_columns(parallel.gather_slabs(tr.rel), self.k, cplx)
        dev = torch.device("cuda")
        return alloc_table(ent.shape[0], ent.shape[1], dev, init=ent), alloc_table(rel.shape[0], rel.shape[1], dev, init=rel)

    # ---- early stopping (EmbeddingModel.py:824-1020) ----
    def _initialize_early_stopping(self):
        p = self.early_stopping_params
        try:
            x_valid = p["x_valid"]
        except KeyError:
            msg = "x_valid must be passed for early fitting."
            logger.error(msg)
            raise KeyError(msg)
        if isinstance(x_valid, EmgraphBaseDatasetAdaptor):   # EmbeddingModel.py:868-880: validation data held by an adapter
            if not x_valid.data_exists("valid"):
                msg = "Dataset `valid` has not been set in the DatasetAdapter."
                logger.error(msg)
                raise ValueError(msg)
            x_valid.use_mappings(self.rel_to_idx, self.ent_to_idx)
            x_valid.map_data()
            x_valid = np.concatenate([np.asarray(b[0]) for b in x_valid.get_next_batch(1, "valid")], axis=0)
            p = dict(p, x_valid=None)
            mapped_valid = x_valid
        else:
            mapped_valid = None
        if mapped_valid is None and not isinstance(x_valid, np.ndarray):
            msg = "Invalid type for input X. Expected ndarray/EmgraphDataset object, got {}".format(type(x_valid))
            logger.error(msg)
            raise ValueError(msg)
        if x_valid.ndim <= 1 or np.shape(x_valid)[1] != 3:
            msg = "Invalid size for input x_valid. Expected (n,3):  got {}".format(np.shape(x_valid))
            logger.error(msg)
            raise ValueError(msg)
        criteria = p.get("criteria", DEFAULT_CRITERIA_EARLY_STOPPING)
        if criteria not in ["hits10", "hits1", "hits3", "mrr"]:
            msg = "Unsupported early stopping criteria."
            logger.error(msg)
            raise ValueError(msg)
        ce = p.get("corruption_entities", DEFAULT_CORRUPTION_ENTITIES)
        subset = None
        if isinstance(ce, list):
            wanted = set(ce)
            subset = np.asarray([idx for uri, idx in self.ent_to_idx.items() if uri in wanted])
        findex = None
        if "x_filter" in p:
            x_filter = p["x_filter"]
            if x_filter.ndim <= 1 or np.shape(x_filter)[1] != 3:
                msg = "Invalid size for input x_valid. Expected (n,3):  got {}".format(np.shape(x_filter))
                logger.error(msg)
                raise ValueError(msg)
            findex = FilterIndex(to_idx(x_filter, ent_to_idx=self.ent_to_idx, rel_to_idx=self.rel_to_idx))
        return {"x_valid": (mapped_valid if mapped_valid is not None else
                            to_idx(x_valid, ent_to_idx=self.ent_to_idx, rel_to_idx=self.rel_to_idx)),
                "criteria": criteria, "subset": subset, "filter": findex,
                "corrupt_side": p.get("corrupt_side", DEFAULT_CORRUPT_SIDE_EVAL),
                "best": None, "first": None, "counter": 0, "epoch": None}

    def _perform_early_stopping_test(self, epoch, es, tr):
        p = self.early_stopping_params
        if not (epoch >= p.get("burn_in", DEFAULT_BURN_IN_EARLY_STOPPING)
                and epoch % p.get("check_interval", DEFAULT_CHECK_INTERVAL_EARLY_STOPPING) == 0):
            return False
        ent_f, rel_f = self._eval_tables(tr)
        ranks = rank_triples_device(self._model_id(), ent_f, rel_f, self.internal_k, self._scale(), es["x_valid"],
                                    es["corrupt_side"], DEFAULT_RANK_COMPARE_STRATEGY, filter_triples=es["filter"],
                                    entities_subset=es["subset"],
                                    shard=parallel.rank_world() if parallel.is_active() else None,
                                    precision=self._eval_precision())
        crit = es["criteria"]
        cur = mrr_score(ranks) if crit == "mrr" else hits_at_n_score(ranks, int(crit[4:]))
        if es["best"] is None:
            es["best"] = es["first"] = cur
        elif es["best"] >= cur:
            es["counter"] += 1
            if es["counter"] == p.get("stop_interval", DEFAULT_STOP_INTERVAL_EARLY_STOPPING):
                if es["best"] == es["first"]:
                    self._save_trained_params(tr)
                if self.verbose:
                    logger.info("Early stopping at epoch:{}".format(epoch))
                    logger.info("Best {}: {:10f}".format(crit, es["best"]))
                self.early_stopping_epoch = epoch
                return True
        else:
            es["best"] = cur
            es["counter"] = 0
            self._save_trained_params(tr)
        return False

    # ---- inference ----
    def _device_tables(self):
        if self._dev is None:
            D.require_gpu()
            ent, rel = self.trained_model_params
            self._dev = (alloc_table(ent.shape[0], ent.shape[1], torch.device("cuda"), init=ent),
                         alloc_table(rel.shape[0], rel.shape[1], torch.device("cuda"), init=rel))
        return self._dev

    def predict(self, X, from_idx=False, chunk=1 << 22, sharded=False):
        """Scores of the triples X (EmbeddingModel.py:2101-2186).

        ``sharded=True`` (multi-GPU, one process per GPU): a COLLECTIVE — every rank must call it with the same X; rank r
        scores the r-th contiguous range of the list and the ranges are summed into place.  The default is rank-local
        (every rank scores everything it is given), so `if rank == 0: model.predict(...)` and models restored in one
        process of a distributed job work as on a single GPU."""
        if not self.is_fitted:
            msg = "Model has not been fitted."
            logger.error(msg)
            raise RuntimeError(msg)
        if self.embedding_model_params.get("non_linearity", "linear") != "linear":
            raise NotImplementedError("non_linearity other than 'linear' is outside the accelerated hot path")
        if type(X) is not np.ndarray:
            X = np.array(X)
        if X.ndim == 1:
            X = X[np.newaxis, :]
        if not from_idx:
            X = to_idx(X, ent_to_idx=self.ent_to_idx, rel_to_idx=self.rel_to_idx)
        X = np.ascontiguousarray(X, dtype=np.int32)
        ent, rel = self._device_tables()
        out = np.zeros(X.shape[0], dtype=np.float32)
        # multi-GPU (one process per GPU, tables replicated): rank r scores the r-th contiguous range of the triple
        # list and the ranges are summed into place (predict shards trivially, no data-path exchange)
        rank, world = parallel.rank_world() if sharded else (0, 1)
        r0, r1 = parallel.entity_range(X.shape[0], rank, world)
        for c0 in range(r0, r1, chunk):  # SURVEY A-17: chunk instead of one giant gather
            c1 = min(c0 + chunk, r1)
            xt = torch.from_numpy(X[c0:c1]).cuda()
            out[c0:c1] = D.score_triples(self._model_id(), ent, rel, self.internal_k, self._scale(), xt).cpu().numpy()
        if world > 1:   # disjoint ranges, zeros elsewhere: the sum puts every range in place exactly
            out = parallel.allreduce_sum_(torch.from_numpy(out).cuda()).cpu().numpy()
        return out

    # ---- evaluation protocol plumbing (EmbeddingModel.py:1494-1518,2035-2099) ----
    def set_filter_for_eval(self):
        """Configures to use filter (:1494-1496)."""
        self.is_filtered = True

    def configure_evaluation_protocol(self, config=None):
        """:1498-1518: keys corruption_entities ('all' | ids), corrupt_side, ranking_strategy."""
        if config is None:
            config = {"corruption_entities": DEFAULT_CORRUPTION_ENTITIES, "corrupt_side": DEFAULT_CORRUPT_SIDE_EVAL}
        self.eval_config = config

    def end_evaluation(self):
        """:2035-2044."""
        handle = getattr(self, "eval_dataset_handle", None)
        if self.is_filtered and handle is not None:
            handle.cleanup()
        self.eval_dataset_handle = None
        self.is_filtered = False
        self.eval_config = {}

    _ranks_as_array = True   # (evaluate_performance: get_ranks(..., as_array=True) is understood)

    def get_ranks(self, dataset_handle, as_array=False):
        """Ranks of the adapter's 'test' triples under the configured protocol (EmbeddingModel.py:2046-2099, with the
        intended one-rank-per-test-triple semantics, SURVEY A-1).  The filter comes from the adapter: its FilterIndex
        when it is this package's NumpyDatasetAdapter, otherwise the per-triple lists its
        get_next_batch(-1, 'test', use_filter=True) yields (the protocol of numpy_adapter.py:79-131).
        Returns the reference's Python lists; ``as_array=True`` (this package's evaluate_performance, which turns the lists
        into an array at once: building 4096 two-element lists and parsing them back was 1.6 of a 10.8 ms call) the int64 array."""
        if not self.is_fitted:
            msg = "Model has not been fitted."
            logger.error(msg)
            raise RuntimeError(msg)
        self.eval_dataset_handle = dataset_handle
        cfg = self.eval_config
        corrupt_side = cfg.get("corrupt_side", DEFAULT_CORRUPT_SIDE_EVAL)
        strategy = cfg.get("ranking_strategy", DEFAULT_RANK_COMPARE_STRATEGY)
        subset = cfg.get("corruption_entities", DEFAULT_CORRUPTION_ENTITIES)
        subset = None if isinstance(subset, str) else np.asarray(subset)
        findex = None
        if self.is_filtered:
            findex = getattr(dataset_handle, "filter_index", None)
        if self.is_filtered and findex is None:
            # a foreign adapter: collect its per-triple filter lists into the filter triples they stand for
            X_parts, rows = [], []
            for out in dataset_handle.get_next_batch(-1, "test", use_filter=True):
                x, objs, subs = np.asarray(out[0]).reshape(-1, 3), np.asarray(out[-2]).reshape(-1), np.asarray(out[-1]).reshape(-1)
                X_parts.append(x)
                rows.append(np.stack([np.full(len(objs), x[0, 0]), np.full(len(objs), x[0, 1]), objs], 1))
                rows.append(np.stack([subs, np.full(len(subs), x[0, 1]), np.full(len(subs), x[0, 2])], 1))
            X_idx = np.concatenate(X_parts, 0) if X_parts else np.zeros((0, 3), np.int64)
            findex = FilterIndex(np.concatenate(rows, 0)) if rows else None
        else:
            parts = [np.asarray(b[0] if isinstance(b, (list, tuple)) else b).reshape(-1, 3)
                     for b in dataset_handle.get_next_batch(1, "test")]
            X_idx = np.concatenate(parts, 0) if parts else np.zeros((0, 3), np.int64)
        ranks = self.get_ranks_idx(X_idx, filter_idx=findex, corrupt_side=corrupt_side, ranking_strategy=strategy,
                                   corruption_entities=subset)
        if as_array:
            return ranks
        return [list(r) for r in ranks] if corrupt_side == "s,o" else list(ranks)

    def get_ranks_idx(self, X_idx, filter_idx=None, corrupt_side=DEFAULT_CORRUPT_SIDE_EVAL,
                      ranking_strategy=DEFAULT_RANK_COMPARE_STRATEGY, corruption_entities=None, verbose=False):
        """get_ranks (EmbeddingModel.py:2046-2099) on integer ids, intended per-triple semantics."""
        if not self.is_fitted:
            msg = "Model has not been fitted."
            logger.error(msg)
            raise RuntimeError(msg)
        ent, rel = self._device_tables()
        precision, tables = self._eval_precision(), None
        if precision == "auto":
            from ..evaluation.ranking import resolve_auto_precision
            precision = resolve_auto_precision(self._model_id(), self.internal_k, int(np.asarray(X_idx).reshape(-1, 3).shape[0]),
                                               int(ent.shape[0]), corruption_entities)
        if precision == 2 and corruption_entities is None:
            # what the exact-fast mode derives from the tables (half-precision copy, norm bounds, ...) is kept with the
            # device copy of the fitted parameters: repeated evaluations of a fitted model do not rebuild it (0.63 M -> 0.75 M
            # ranks/s at |E| = 1M, bench `eval.exact_fast.*.uncached_tables`); a new fit / restore replaces both
            from ..evaluation.ranking import derived_tables
            # (after fit() the device tables may be the trainer's LIVE ones: keyed on its step count too, so that further steps —
            # early stopping's evaluations between epochs, a caller driving the trainer — never meet stale half-precision copies)
            tr = getattr(self, "_trainer", None)
            stamp = (self._dev, tr.step_count if tr is not None else -1)
            cached = getattr(self, "_derived_cache", None)
            if cached is None or cached[0][0] is not stamp[0] or cached[0][1] != stamp[1]:
                cached = self._derived_cache = (stamp, derived_tables(self._model_id(), ent, rel, self.internal_k))
            tables = cached[1]
        return rank_triples_device(self._model_id(), ent, rel, self.internal_k, self._scale(), X_idx, corrupt_side,
                                   ranking_strategy, filter_triples=filter_idx, entities_subset=corruption_entities,
                                   shard=parallel.rank_world() if parallel.is_active() else None,
                                   precision=precision, ent_f16=tables)

    def _eval_precision(self):
        """embedding_model_params['eval_precision'] / EMG_EVAL_PRECISION: 0 exact f32 kernel, 2 exact ranks through the
        half-precision MFMA prefilter (bit-equal to 0), 1 bf16 throughput mode (statistical agreement only: never picked
        implicitly), 'auto' (default) = 2 where it applies and pays, else 0."""
        v = self.embedding_model_params.get("eval_precision", os.environ.get("EMG_EVAL_PRECISION", "auto"))
        if v in ("auto", 0, 1, 2):
            return v
        if str(v) in ("0", "1", "2"):
            return int(v)
        raise ValueError("eval_precision must be 0, 1, 2 or 'auto', got {!r}".format(v))


@register_model("TransE")
class TransE(EmbeddingModel):
    """TransE (TransE.py): f = -||e_s + r_p - e_o||_n, n = embedding_model_params['norm'] in {1, 2}."""

    def __init__(self, k=DEFAULT_EMBEDDING_SIZE, eta=DEFAULT_ETA, epochs=DEFAULT_EPOCH,
                 batches_count=DEFAULT_BATCH_COUNT, seed=DEFAULT_SEED,
                 embedding_model_params={"norm": DEFAULT_NORM_TRANSE, "normalize_ent_emb": DEFAULT_NORMALIZE_EMBEDDINGS,
                                         "negative_corruption_entities": DEFAULT_CORRUPTION_ENTITIES,
                                         "corrupt_sides": DEFAULT_CORRUPT_SIDE_TRAIN},
                 optimizer=DEFAULT_OPTIM, optimizer_params={"lr": DEFAULT_LR}, loss=DEFAULT_LOSS, loss_params={},
                 regularizer=DEFAULT_REGULARIZER, regularizer_params={}, initializer=DEFAULT_INITIALIZER,
                 initializer_params={"uniform": DEFAULT_GLOROT_IS_UNIFORM}, verbose=DEFAULT_VERBOSE,
                 large_graphs=False):
        super().__init__(k=k, eta=eta, epochs=epochs, batches_count=batches_count, seed=seed,
                         embedding_model_params=embedding_model_params, optimizer=optimizer,
                         optimizer_params=optimizer_params, loss=loss, loss_params=loss_params,
                         regularizer=regularizer, regularizer_params=regularizer_params, initializer=initializer,
                         initializer_params=initializer_params, verbose=verbose, large_graphs=large_graphs)

    def _norm(self):
        norm = self.embedding_model_params.get("norm", DEFAULT_NORM_TRANSE)
        if norm == "euclidean":
            norm = 2
        try:
            ok = float(norm) > 0
        except (TypeError, ValueError):
            ok = False
        if not ok:
            raise ValueError("TransE norm {!r}: expected a positive order of the vector norm (1, 2, ..., np.inf)".format(norm))
        return norm

    def _model_id(self):
        """norm 1 / 2: the tuned models; any other positive order (TransE.py:208-216 passes `norm` to tf.norm as ord): EMG_TRANSE_P —
        generic kernels for fit (unfused forward / loss / backward), predict, get_ranks and evaluate_performance (exact path)"""
        norm = self._norm()
        return L.TRANSE_L1 if norm == 1 else (L.TRANSE_L2 if norm == 2 else L.TRANSE_P)

    def _scale(self):
        return float(self._norm()) if self._model_id() == L.TRANSE_P else 1.0



@register_model("DistMult")
class DistMult(EmbeddingModel):
    """DistMult (DistMult.py): f = <r_p, e_s, e_o>."""

    def __init__(self, k=DEFAULT_EMBEDDING_SIZE, eta=DEFAULT_ETA, epochs=DEFAULT_EPOCH,
                 batches_count=DEFAULT_BATCH_COUNT, seed=DEFAULT_SEED,
                 embedding_model_params={"normalize_ent_emb": DEFAULT_NORMALIZE_EMBEDDINGS,
                                         "negative_corruption_entities": DEFAULT_CORRUPTION_ENTITIES,
                                         "corrupt_sides": DEFAULT_CORRUPT_SIDE_TRAIN},
                 optimizer=DEFAULT_OPTIM, optimizer_params={"lr": DEFAULT_LR}, loss=DEFAULT_LOSS, loss_params={},
                 regularizer=DEFAULT_REGULARIZER, regularizer_params={}, initializer=DEFAULT_INITIALIZER,
                 initializer_params={"uniform": DEFAULT_GLOROT_IS_UNIFORM}, verbose=DEFAULT_VERBOSE):
        super().__init__(k=k, eta=eta, epochs=epochs, batches_count=batches_count, seed=seed,
                         embedding_model_params=embedding_model_params, optimizer=optimizer,
                         optimizer_params=optimizer_params, loss=loss, loss_params=loss_params,
                         regularizer=regularizer, regularizer_params=regularizer_params, initializer=initializer,
                         initializer_params=initializer_params, verbose=verbose)

    def _model_id(self):
        return L.DISTMULT



@register_model("ComplEx")
class ComplEx(EmbeddingModel):
    """ComplEx (ComplEx.py): f = Re(<r_p, e_s, conj(e_o)>); rows are [re | im], internal_k = 2k (:224)."""

    def __init__(self, k=DEFAULT_EMBEDDING_SIZE, eta=DEFAULT_ETA, epochs=DEFAULT_EPOCH,
                 batches_count=DEFAULT_BATCH_COUNT, seed=DEFAULT_SEED,
                 embedding_model_params={"negative_corruption_entities": DEFAULT_CORRUPTION_ENTITIES,
                                         "corrupt_sides": DEFAULT_CORRUPT_SIDE_TRAIN},
                 optimizer=DEFAULT_OPTIM, optimizer_params={"lr": DEFAULT_LR}, loss=DEFAULT_LOSS, loss_params={},
                 regularizer=DEFAULT_REGULARIZER, regularizer_params={}, initializer=DEFAULT_INITIALIZER,
                 initializer_params={"uniform": DEFAULT_GLOROT_IS_UNIFORM}, verbose=DEFAULT_VERBOSE):
        super().__init__(k=k, eta=eta, epochs=epochs, batches_count=batches_count, seed=seed,
                         embedding_model_params=embedding_model_params, optimizer=optimizer,
                         optimizer_params=optimizer_params, loss=loss, loss_params=loss_params,
                         regularizer=regularizer, regularizer_params=regularizer_params, initializer=initializer,
                         initializer_params=initializer_params, verbose=verbose)
        self.internal_k = self.k * 2

    def _model_id(self):
        return L.COMPLEX



@register_model("HolE")
class HolE(ComplEx):
    """HolE (HolE.py:189): f = (2/k) * f_ComplEx (Hayashi & Shimbo equivalence; NOT an FFT, SURVEY A-12)."""

    def _model_id(self):
        return L.HOLE

    def _scale(self):
        return float(np.float32(2 / self.k))

