"""Device-resident training engine: what EmbeddingModel.fit's inner loop (EmbeddingModel.py:1388-1440)
needs per batch, restructured for one MI355X:

    corruption codes (Philox, on device)  ->  fused gather+score of positives and negatives
    ->  loss + dL/dscore  ->  fused backward (gradient rows, no atomics)
    ->  deterministic sort + row-sparse optimizer apply (entities, relations)

All buffers (tables, optimizer state, the whole mapped training set, per-batch scratch) are allocated
once as torch tensors and stay in HBM; a batch is a pointer offset into the resident training set
(the reference re-feeds numpy slices through tf.data each batch, EmbeddingModel.py:1044-1111,1329-1337).
No host synchronisation happens inside an epoch: the loss accumulates in a device double that the
caller reads once per epoch (the reference's per-batch `.numpy()` NaN check, :1421-1427, becomes a
per-epoch check).
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import _lib as L
from . import device as D
from . import parallel

DEFAULT_LR = 0.0005            # training/_optimizer_constants.py:9
DEFAULT_MOMENTUM = 0.9         # :11
DEFAULT_DECAY_CYCLE = 0
DEFAULT_DECAY_CYCLE_MULTIPLE = 1
DEFAULT_LR_DECAY_FACTOR = 2
DEFAULT_END_LR = 1e-8
DEFAULT_SINE = False
ADAM_BETA1, ADAM_BETA2, KERAS_EPS = 0.9, 0.999, 1e-7   # tf.optimizers defaults (TF 2.2)
ADAGRAD_INIT_ACC = 0.1


def sgd_learning_rate(params, batches_count, epoch, batch):
    """Learning rate of plain SGD for (epoch, batch), both 1-based: a pure function of the optimizer parameters.

    Semantics of the reference's schedule (training/sgd.py:127-185; its TF2 port never calls it, SURVEY A-4),
    values pinned by tests/emgraph/models/test_optimizers.py:21,41-43,72-79:
      * neither 'decay_cycle' nor 'cosine_decay': constant 'lr';
      * 'cosine_decay': warm restarts.  Cycle i (i = 0, 1, ...) lasts decay_cycle * expand_factor**i epochs and
        anneals from lr / decay_lr_rate**i down to 'end_lr' along half a cosine, batch by batch;
      * fixed decay ('decay_cycle' > 0): the rate is divided by 'decay_lr_rate' at the first batch of epoch
        b_1 = decay_cycle + 1, b_(j+1) = decay_cycle + (b_j - 1) * expand_factor + 1, never below 'end_lr'
        (once 'end_lr' is reached the schedule stops)."""
    lr0 = params.get("lr", DEFAULT_LR)
    cycle = params.get("decay_cycle", DEFAULT_DECAY_CYCLE)
    end_lr = params.get("end_lr", DEFAULT_END_LR)
    expand = params.get("expand_factor", DEFAULT_DECAY_CYCLE_MULTIPLE)
    shrink = params.get("decay_lr_rate", DEFAULT_LR_DECAY_FACTOR)
    if cycle <= 0:
        return lr0
    if params.get("cosine_decay", DEFAULT_SINE):
        first, length, top = 0, cycle, lr0          # the cycle holding `epoch` covers epochs (first, first + length]
        while epoch > first + length:
            first, length, top = first + length, length * expand, top / shrink
        done = ((epoch - 1 - first) * batches_count + (batch - 1)) / (length * batches_count)
        return max(end_lr, end_lr + (top - end_lr) * 0.5 * (1 + math.cos(math.pi * done)))
    lr, boundary = lr0, cycle + 1
    while boundary <= epoch and lr > end_lr:
        lr, boundary = max(end_lr, lr / shrink), cycle + (boundary - 1) * expand + 1
    return lr


def _padded_ld(k_int):
    """row stride: multiple of 4 floats so every row is 16-byte aligned for dwordx4 loads"""
    return ((k_int + 3) // 4) * 4


def alloc_table(rows, k_int, device, init=None, fill=None):
    ld = _padded_ld(k_int)
    buf = torch.zeros((rows, ld), dtype=torch.float32, device=device)
    view = buf[:, :k_int]
    if init is not None:   # a host array, or a device tensor (tables initialised on the device)
        view.copy_(init if torch.is_tensor(init) else torch.from_numpy(np.ascontiguousarray(init, dtype=np.float32)))
    elif fill is not None:
        view.fill_(fill)
    return view  # 2-D view with stride(0) = ld


# batches prepared ahead of the one computing (= side streams = extra batch slots); see Trainer.step.
# Two, not more: main + apply_rel stream + 2 side streams = 4 = the HIP runtime's hardware queues per device;
# a fifth stream is multiplexed onto them and the step got SLOWER (measured 0.47 -> 0.67 ms at 3, 0.73 at 4).
LOSS_SLOTS = int(os.environ.get("EMG_LOSS_SLOTS", "64"))   # doubles the fused kernel spreads the batch loss over (1: one, the old form; A/B aid)
LOOKAHEAD = int(os.environ.get("EMG_LOOKAHEAD", "2"))
GRAPH_MAX_ROWS = int(os.environ.get("EMG_GRAPH_MAX_ROWS", "200000"))   # entity contribution rows per batch up to which steps run as graph replays
AUX_MIN_ROWS = int(os.environ.get("EMG_AUX_MIN_ROWS", "100000"))  # (env: A/B aid) entity contribution rows per batch above which apply_rel gets its own stream


class Trainer:
    """One model's device state + the per-batch step."""

    def __init__(self, model_id, k_int, scale, ent_init, rel_init, eta, loss="nll", loss_params=None,
                 optimizer="adam", optimizer_params=None, corrupt_sides=("s,o",), batches_count=1, seed=0,
                 regularizer=None, regularizer_params=None, normalize_ent_emb=False, device="cuda", fused=True,
                 inplace=True, pipeline=True, sharded=False, deferred_dense=None, shard_state=False):
        """``sharded=True`` / ``"k"``: ent_init / rel_init are this rank's COLUMN slabs (emgraph_amd.parallel.shard_columns)
        and k_int is the local width; every step all-reduces the partial scores.
        ``sharded="batch"``: full tables on every rank; each rank scores its rows of the global batch, gradient rows
        travel to the owner of their destination, which applies them and all-gathers the updated rows (parallel.py).
        ``shard_state=True`` (with ``sharded="batch"``; SGD / momentum / Adagrad without regulariser): the optimizer state is
        sharded by OWNER — each rank holds the state rows of its own id range only, applies the optimizer to the summed gradients of
        ITS rows and all-gathers the UPDATED rows (the same bytes as the summed gradients), which every replica copies into its
        table: 1/N of the state and of the update arithmetic, same bits.  Keras Adam cannot take this form (every row moves every
        step: the replicas would need the whole table back), nor can a folded LP regulariser.
        ``deferred_dense`` (one GPU, Adam and / or an LP regulariser; default: tables of >= 256 MB, or env EMG_ADAM_DEFERRED=0/1):
        the dense pass (Keras Adam's decay, the regulariser's gradient) is replayed only for the rows a batch reads and
        updates (emg_deferred_catchup) instead of passing over the whole table every step — same bits; ``materialize()``
        brings every row up to date before the tables are read."""
        D.require_gpu()
        self.device = torch.device(device)
        self.model_id, self.k_int, self.scale, self.eta = model_id, int(k_int), float(scale), int(eta)
        self.loss_id = L.LOSS_IDS[loss]
        lp = loss_params or {}
        default_margin = 3.0 if loss == "self_adversarial" else 1.0   # losses/_loss_constants.py:8-12
        self.margin = float(lp.get("margin", default_margin))
        self.alpha = float(lp.get("alpha", 0.5))
        self.sides = [L.SIDE_IDS[s] for s in corrupt_sides]
        self.n_sides = len(self.sides)
        self.eta_total = self.eta * self.n_sides
        self.seed = int(seed)
        self.batches_count = int(batches_count)
        self.normalize = bool(normalize_ent_emb)

        self.ent = alloc_table(ent_init.shape[0], k_int, self.device, init=ent_init)
        self.rel = alloc_table(rel_init.shape[0], k_int, self.device, init=rel_init)
        self.n_ent, self.n_rel = ent_init.shape[0], rel_init.shape[0]

        op = optimizer_params or {}
        self.opt_name = optimizer
        self.opt_id = L.OPT_IDS[optimizer]
        self.lr = float(op.get("lr", DEFAULT_LR))
        self.momentum = float(op.get("momentum", DEFAULT_MOMENTUM))
        self.sgd_params = dict(op) if optimizer == "sgd" else None
        self.state_ent = [None, None]
        self.state_rel = [None, None]
        if optimizer == "momentum":
            self.state_ent[0] = alloc_table(self.n_ent, k_int, self.device)
            self.state_rel[0] = alloc_table(self.n_rel, k_int, self.device)
        elif optimizer == "adagrad":
            self.state_ent[0] = alloc_table(self.n_ent, k_int, self.device, fill=ADAGRAD_INIT_ACC)
            self.state_rel[0] = alloc_table(self.n_rel, k_int, self.device, fill=ADAGRAD_INIT_ACC)
        elif optimizer in ("adam", "adam_lazy"):
            for st, n in ((self.state_ent, self.n_ent), (self.state_rel, self.n_rel)):
                st[0] = alloc_table(n, k_int, self.device)
                st[1] = alloc_table(n, k_int, self.device)
        self.tag_ent = torch.zeros(self.n_ent, dtype=torch.int32, device=self.device)
        self.tag_rel = torch.zeros(self.n_rel, dtype=torch.int32, device=self.device)
        self.step_count = 0

        self.reg = None
        if regularizer is not None:
            rp = regularizer_params or {}
            lam = rp.get("lambda", 1e-5)      # regularizers/_regularizer_constants.py:8-10
            p = rp.get("p", 2)
            if not isinstance(p, (int, np.integer)):
                raise Exception("Invalid value for regularizer parameter p:{}. Supported type int, np.int32 or "
                                "np.int64".format(p))
            if np.isscalar(lam):
                lam = [lam, lam]
            elif not (isinstance(lam, list) and len(lam) == 2):
                raise ValueError("Regularizer weight must be a scalar or a list with length equal to number of "
                                 "params passes")
            self.reg = (float(lam[0]), float(lam[1]), int(p))
        # The LP penalty covers the full tables (lp.py:107-113), so its gradient is dense.  It is FOLDED into the
        # optimizer step: rows with contributions (or updated in place) add lambda*p*|w|^(p-1)*sign(w) to their summed
        # gradient, all other rows get it in one dense pass (emg_apply_grouped / untouched_rows_kernel) — no extra
        # contribution rows, one update per row per step with the gradient of the whole loss.
        self.reg_rows = False

        self.plan = None
        self.graph = False
        # the data loss: LOSS_SLOTS doubles, workgroup b of the fused kernel adds to slot b mod LOSS_SLOTS (same-address atomics of a
        # short launch queue up behind each other: include/emgraph_hip.h, emg_backward_args.loss_slots); every other producer adds to
        # slot 0; read_loss sums
        self.loss_accum = torch.zeros(LOSS_SLOTS, dtype=torch.float64, device=self.device)
        self.reg_accum = torch.zeros(1, dtype=torch.float64, device=self.device)  # LP term (column-local when sharded)
        # sum |w|^p per table, accumulated by the kernels that fold the regulariser (scaled by lambda in read_loss)
        self.lp_sum = torch.zeros(2, dtype=torch.float64, device=self.device)
        self.X = None
        self._cap = 0
        self.stage_events = None  # filled by enable_stage_timing()
        # execution plan
        #  fused   : score -> pair-local loss -> gradient in ONE kernel (pairwise / nll / absolute_margin)
        #  inplace : rows whose destination is hit once in the batch are updated from registers
        #            (plain SGD only: _choose_inplace)
        #  pipeline: codes + destination grouping of batches t+1, t+2 run on a side stream while batch t computes
        self.batch_sharded = sharded == "batch"
        self.sharded = bool(sharded) and not self.batch_sharded          # k (column) sharding
        # rows wider than 512 columns (per half for ComplEx / HolE) do not fit the register-tiled kernels: the separate
        # forward / loss / backward path handles them in column blocks (emg_score.hip::run_group_pass)
        self.wide = (self.k_int // 2 if model_id in (L.COMPLEX, L.HOLE) else self.k_int) > 512
        # TransE with an order of the norm other than 1 / 2 (EMG_TRANSE_P; `scale` carries the order): generic kernels — the separate
        # forward / loss / backward step, every gradient row through the apply
        self.generic = model_id == L.TRANSE_P
        if self.generic and self.sharded:
            raise ValueError("TransE with a norm other than 1 / 2 does not train on column slabs (sharding 'k'): use sharding 'batch'")
        self.fused = fused and loss in ("pairwise", "nll", "absolute_margin") and not self.sharded and not self.wide and not self.generic
        # (an LP regulariser is folded into every update: by the apply kernel, and by the in-place form of plain SGD — its
        # own instantiation (IP 3), so that the pow / sign code stays out of the forms that have no regulariser)
        n_cols = self.k_int // 2 if model_id in (L.COMPLEX, L.HOLE) else self.k_int
        self.inplace_mode = 0   # 0 off | 1 singletons in place | 2 the same through a stateful optimizer's window form (_choose_inplace)
        self.inplace = self._inplace_wanted = inplace and not self.generic and (self.reg is None or (self.opt_id == L.OPT_SGD and self.reg[2] <= 3
                                                                                and n_cols % 4 == 0))
        # default: on where the dense pass is what a step costs — an entity table of 256 MB or more (C3 with Adam: 2.6 -> 1.65
        # ms/step); small tables keep the dense pass (6-14 us at C1 / C2 / C5, inside the step graph)
        env_def = os.environ.get("EMG_ADAM_DEFERRED")
        want_deferred = deferred_dense if deferred_dense is not None else \
            (env_def == "1" if env_def in ("0", "1") else self.n_ent * k_int * 4 >= (256 << 20))
        # (what has a dense pass at all: Keras Adam's decay, the LP regulariser's gradient)
        self.deferred = bool(want_deferred) and (self.opt_id == L.OPT_ADAM or self.reg is not None) and not self.sharded \
            and not self.batch_sharded and not normalize_ent_emb
        self._lr_t_hist = None          # device float32 [steps]: learning rate (Adam: lr_t) of every optimizer step (deferred pass)
        self._lr_t_filled = 0
        self._lr_host = [0.0]           # host copy (index = step)
        #  factored: bilinear models write a negative's gradient row as (one float) x (one of the group's two query
        #            rows) instead of eta full rows per group (emg_backward_args.fac_ws_ent); EMG_FACTORED=0 = A/B switch
        self.factored = (model_id not in (L.TRANSE_L1, L.TRANSE_L2, L.TRANSE_P) and not self.batch_sharded
                         and os.environ.get("EMG_FACTORED", "1") != "0")
        self.pipeline = pipeline
        if self.batch_sharded:
            # a destination's contributions come from several ranks: no rank may update a row in place, and the
            # step has collectives in the middle (single stream; the exchange is what bounds it, not the enqueue).
            # Optimizer state is REPLICATED: the owner of a destination only SUMS its rows (in global slot order), the sums
            # are all-gathered and every replica applies the optimizer to them — touched rows from the sums, all others
            # through the dense pass — so Keras' dense-equivalent Adam and the folded LP regulariser work as on one GPU.
            self.inplace = self.pipeline = pipeline = False
            self.xgmi_bytes = 0          # bytes this rank sent + received over the interconnect (gradient rows + summed rows)
            self._owner_ws = {}
            # device-resident exchange (parallel.RowExchange; round 4): sizes settled in a table-independent metadata phase, the owner's
            # order by a keyed counting grouping — which needs a range of at most 2^20 + 16 n rows per owner; EMG_XCHG=host: round 3's
            # host-driven exchange (argsort / unique, a count exchange and fresh buffers every step)
            _, world = parallel.rank_world()
            self._device_exchange = (os.environ.get("EMG_XCHG", "device") != "host" and os.environ.get("EMG_GROUPING") != "sort"
                                     and -(-max(self.n_ent, self.n_rel) // max(1, world)) <= (1 << 20))
            self._xchg_objs, self._gslot_cache = {}, {}
            # rows this rank's optimizer updated (upper bounds from the count matrix: no host read), and what the other form would have
            self.opt_rows = self.opt_rows_owner_form = self.opt_rows_replicated_form = 0
            self._bs_ready = {}          # batches whose metadata phase has been issued ahead: key -> prepared state
            self._bs_parity = 0
            self._bs_done = [torch.cuda.Event(), torch.cuda.Event()]   # main-stream work that last used slot / workspaces of a parity
            self._bs_ahead = self._device_exchange and os.environ.get("EMG_XCHG_AHEAD", "1") != "0"
            self._bs_stream = torch.cuda.Stream(device=self.device) if self._bs_ahead else None
        self.shard_state = bool(shard_state)
        if self.shard_state:
            if not self.batch_sharded or self.opt_id not in (L.OPT_SGD, L.OPT_MOMENTUM, L.OPT_ADAGRAD) or self.reg is not None:
                raise ValueError("shard_state needs sharded='batch' and SGD / momentum / Adagrad without a regulariser (Keras Adam and the "
                                 "LP regulariser move every row every step: their state cannot live at the owner alone)")
            if not self._device_exchange:
                raise ValueError("shard_state needs the device-resident exchange (EMG_XCHG=device, owner ranges of <= 2^20 rows + 16 per slot)")
            rank_, world_ = parallel.rank_world()
            for st_, n_ in ((self.state_ent, self.n_ent), (self.state_rel, self.n_rel)):
                e0_, e1_ = parallel.entity_range(n_, rank_, world_)
                if st_[0] is not None:   # this rank's rows only
                    st_[0] = alloc_table(max(1, e1_ - e0_), k_int, self.device, fill=ADAGRAD_INIT_ACC if optimizer == "adagrad" else None)
        # high priority: the many small kernels must not queue behind the big ones.  TWO side streams used
        # alternately: a preparation chain is latency-bound (each small launch waits for a CU slot), so two
        # chains in flight double the rate at which prepared batches arrive
        self.lookahead = LOOKAHEAD if pipeline else 0
        if self.lookahead <= 0:  # EMG_LOOKAHEAD=0: nothing to prepare ahead on, run the plain single-stream plan
            self.pipeline, self.lookahead = False, 0
        self.sides_st = [torch.cuda.Stream(device=self.device, priority=-1) for _ in range(self.lookahead)]
        self._side_rr = 0
        self.aux = torch.cuda.Stream(device=self.device) if self.pipeline else None  # apply_rel under apply_ent
        self.aux_fork, self.aux_join = torch.cuda.Event(), torch.cuda.Event()
        self.slots = []

    # ---- data ----
    def set_training_set(self, X_idx, batch_size):
        """Upload the whole mapped training set once; allocate per-batch scratch for ``batch_size``."""
        X_idx = np.ascontiguousarray(X_idx, dtype=np.int32)
        self.X = torch.from_numpy(X_idx).to(self.device)
        if int(batch_size) <= self._cap:
            self._make_plan()            # the plan points at the resident training set
        self._alloc_scratch(int(batch_size))

    def _choose_inplace(self, B):
        """In-place singleton updates: the fused kernel reads a singleton's row anyway and writes it back updated — no contribution
        row, and the apply never sees 70 % of C3's slots.  Plain SGD always takes it.  A STATEFUL optimizer takes it where the
        kernel can fetch the optimizer state rows together with the table row (round 4: fused kernels, 16-byte rows of at most 64
        chunks — emg_score_kernels.hpp::ip_traits 4 / 5 / 6): round 3's form read the state at the update, a dependent round trip
        per row at one wave per SIMD, and lost to the apply everywhere (C1 Adam 0.081 / 0.078 ms per step in place / through the
        apply, C3 Adagrad 0.842 / 0.755).  Under the deferred dense pass (Adam on a large table) the singletons among the negatives
        are replayed inside the kernel as they are gathered: (w, m, v) read once and written once where catch-up + scoring + apply
        moved the row twelve times.  Results are the same bits either way (one optimizer rule, one summation order:
        tests/test_config_widths.py::test_inplace_choice_does_not_change_bits)."""
        if not self._inplace_wanted or self.batch_sharded:
            return 0
        stateful = self.opt_id != L.OPT_SGD
        n_cols = self.k_int // 2 if self.model_id in (L.COMPLEX, L.HOLE) else self.k_int
        window = self.fused and n_cols % 4 == 0 and n_cols // 4 <= 64 and os.environ.get("EMG_INPLACE_STATE", "1") != "0"
        # what CAN run in place: SGD anything; a stateful optimizer anything (window form or chunk-wise) unless its dense pass is
        # deferred — then only Adam through the window form's replay, finished by the descriptor-driven apply (rows of > 16 chunks)
        can = not (stateful and self.deferred) or (window and self.opt_id == L.OPT_ADAM and self.reg is None and self.k_int > 64)
        # modes: 1 = singletons in place (SGD; a stateful optimizer's chunk-wise form), 2 = a stateful optimizer's window form
        mode = 2 if (stateful and window) else 1
        if os.environ.get("EMG_INPLACE") in ("0", "1"):      # A/B aid
            return mode if (os.environ["EMG_INPLACE"] == "1" and can) else 0
        if not stateful or os.environ.get("EMG_INPLACE_ALWAYS"):
            return mode if can else 0
        # small batches (the graph-replay range) stay with the apply: its launch is latency-bound there and does not shrink with its
        # item count, so the in-place work only lengthens the scoring kernel (measured: C1 0.0778 / 0.0747, C2 0.0637 / 0.0585, C5
        # 0.1527 / 0.1319 ms per step in place / through the apply; C3 Adagrad 0.612 / 0.695, C3 Adam 0.90 / 1.15)
        return 2 if (can and window and (2 + self.eta_total) * B > GRAPH_MAX_ROWS) else 0

    def _settle_deferred(self, B):
        """decide, BEFORE the in-place form is chosen from it, whether the deferred dense pass can go on with scratch for B positives
        (it lives in the plan's step and walks the counting / bucket grouping's descriptors); if it ends after steps have run, the
        lagging rows and their state are brought up to date first"""
        if not self.deferred:
            return
        ok = not (self.sharded or self.batch_sharded or self.X is None or os.environ.get("EMG_PY_PLAN"))
        ok = ok and bool(L.load().emg_plan_deferred_ok(B, self.eta_total, self.n_ent, self.n_rel))
        if not ok:
            self.materialize()
            self.deferred = False

    def _alloc_scratch(self, B):
        if B <= self._cap:
            return
        self._settle_deferred(B)
        self.inplace_mode = self._choose_inplace(B)
        if self.inplace_mode == 2 and (self.sharded or os.environ.get("EMG_PY_PLAN")):
            self.inplace_mode = 1      # (the window form is the plan's; host-driven steps keep the chunk-wise form)
        self.inplace = self.inplace_mode != 0
        torch.cuda.synchronize()
        dev, k, et = self.device, self.k_int, self.eta_total
        ldc = _padded_ld(k)
        xe = self.n_ent if self.reg_rows else 0
        xr = self.n_rel if self.reg_rows else 0
        n_ce, n_cr = (2 + et) * B + xe, B + xr
        self.scores_all = torch.empty(B * (1 + et), dtype=torch.float32, device=dev)  # [pos | neg], one all-reduce
        self.g_all = torch.empty(B * (1 + et), dtype=torch.float32, device=dev)       # dL/dscore, same layout
        self.g_pos, self.g_neg = self.g_all[:B], self.g_all[B:]
        # factored: subject rows | object rows | query rows (object side) | query rows (subject side)
        self.contrib_ent = torch.empty((4 * B if self.factored else n_ce, ldc), dtype=torch.float32, device=dev)[:, :k]
        self.contrib_rel = torch.empty((n_cr, ldc), dtype=torch.float32, device=dev)[:, :k]
        self.slots = []
        n_slots = 2 if (self.batch_sharded and getattr(self, "_bs_ahead", False)) else 1 + self.lookahead
        for _ in range(n_slots):  # current batch + the ones being prepared ahead
            sl = {
                "codes": torch.empty(B * et, dtype=torch.int32, device=dev),
                "dest_ent": torch.empty(n_ce, dtype=torch.int32, device=dev),
                "dest_rel": torch.empty(n_cr, dtype=torch.int32, device=dev),
                "single": torch.empty(n_ce, dtype=torch.uint8, device=dev),
                "ws_ent": torch.empty(D.apply_workspace_bytes(n_ce, self.n_ent, k), dtype=torch.uint8, device=dev),
                "ws_rel": torch.empty(D.apply_workspace_bytes(n_cr, self.n_rel, k), dtype=torch.uint8, device=dev),
                "ready": torch.cuda.Event(), "done": torch.cuda.Event(), "key": None,
            }
            if self.reg_rows:  # LP gradient rows come FIRST ([0, n_rows)); their destinations never change
                sl["dest_ent"][:xe] = torch.arange(self.n_ent, dtype=torch.int32, device=dev)
                sl["dest_rel"][:xr] = torch.arange(self.n_rel, dtype=torch.int32, device=dev)
            self.slots.append(sl)
        self._cap = B
        self._xe, self._xr = xe, xr
        torch.cuda.synchronize()
        self._make_plan()

    # ---- the step plan in the library (emg_plan.hip): one call per batch instead of a dozen ----
    def _make_plan(self):
        import ctypes as C
        if self.plan is not None:
            L.check(L.load().emg_plan_destroy(self.plan), "emg_plan_destroy")
            self.plan = None
        if self.sharded or self.batch_sharded or self.X is None or os.environ.get("EMG_PY_PLAN"):
            self.materialize()      # (no-op unless deferred steps have run)
            self.deferred = False   # (the deferred dense decay lives in the plan's step)
            return   # multi-GPU steps have a collective in the middle: driven from the host (see step / _compute)
        c = L.PlanConfig()
        c.model, c.k_int, c.scale, c.eta, c.n_sides = self.model_id, self.k_int, self.scale, self.eta, self.n_sides
        for i, sd in enumerate(self.sides):
            c.sides[i] = sd
        c.ent, c.n_ent, c.ld_ent = self.ent.data_ptr(), self.n_ent, self.ent.stride(0)
        c.rel, c.n_rel, c.ld_rel = self.rel.data_ptr(), self.n_rel, self.rel.stride(0)
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        c.ent_state0, c.ent_state1 = ptr(self.state_ent[0]), ptr(self.state_ent[1])
        c.rel_state0, c.rel_state1 = ptr(self.state_rel[0]), ptr(self.state_rel[1])
        c.tag_ent, c.tag_rel = self.tag_ent.data_ptr(), self.tag_rel.data_ptr()
        c.opt, c.loss, c.margin, c.alpha = self.opt_id, self.loss_id, self.margin, self.alpha
        c.seed, c.batches_count = self.seed & 0xFFFFFFFFFFFFFFFF, self.batches_count
        c.X, c.n_triples, c.cap_B = self.X.data_ptr(), self.X.shape[0], self._cap
        c.scores, c.g = self.scores_all.data_ptr(), self.g_all.data_ptr()
        c.contrib_ent, c.contrib_rel, c.ldc = self.contrib_ent.data_ptr(), self.contrib_rel.data_ptr(), self.contrib_ent.stride(0)
        c.loss_accum, c.lp_sum = self.loss_accum.data_ptr(), self.lp_sum.data_ptr()
        c.loss_slots = LOSS_SLOTS
        c.factored = int(self.factored)
        if self.reg is not None:
            c.lp_lambda_ent, c.lp_lambda_rel, c.lp_p = self.reg[0], self.reg[1], self.reg[2]
        c.fused, c.inplace, c.normalize = int(self.fused), int(self.inplace_mode), int(self.normalize)
        c.n_slots = len(self.slots)
        for i, sl in enumerate(self.slots):
            ps = c.slots[i]
            ps.codes, ps.dest_ent, ps.dest_rel = sl["codes"].data_ptr(), sl["dest_ent"].data_ptr(), sl["dest_rel"].data_ptr()
            ps.single = sl["single"].data_ptr()
            ps.ws_ent, ps.ws_ent_bytes = sl["ws_ent"].data_ptr(), sl["ws_ent"].numel()
            ps.ws_rel, ps.ws_rel_bytes = sl["ws_rel"].data_ptr(), sl["ws_rel"].numel()
        c.aux_min_rows = AUX_MIN_ROWS
        self._ctl_buf = torch.zeros(4096, dtype=torch.uint8, device=self.device)   # emg_step_ctl records of a graph replay
        c.ctl_buf, c.ctl_bytes = self._ctl_buf.data_ptr(), self._ctl_buf.numel()
        if self.deferred and not L.load().emg_plan_deferred_ok(self._cap, self.eta_total, self.n_ent, self.n_rel):
            # the catch-up walks the counting grouping's segment descriptors; a table far longer than a batch has gradient
            # rows (or EMG_GROUPING=sort) is grouped by the radix-sort backend: keep the dense pass there
            self.materialize()
            self.deferred = False
        if self.deferred:
            if self._lr_t_hist is None:
                self._lr_t_hist = torch.zeros(self.LR_TABLE_STEPS, dtype=torch.float32, device=self.device)
            c.lr_t_hist = self._lr_t_hist.data_ptr()
        h = C.c_void_p()
        L.check(L.load().emg_plan_create(C.byref(c), C.byref(h)), "emg_plan_create")
        self.plan = h
        self._plan_cfg = c   # keeps nothing alive the tensors do not, but documents what the plan points at
        # Steps as graph replays (emg_plan_run) where a step is shorter than its launches take to issue: small batches.
        # EMG_GRAPH=1 / 0 forces it on (where the plan can) / off.
        n_ce = (2 + self.eta_total) * self._cap
        env = os.environ.get("EMG_GRAPH")
        self.graph = bool(L.load().emg_plan_graph_ok(self.plan)) and (env == "1" or (env != "0" and n_ce <= GRAPH_MAX_ROWS))

    def __del__(self):
        try:
            if getattr(self, "plan", None) is not None:
                L.load().emg_plan_destroy(self.plan)
                self.plan = None
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def _plan_batch(self, out, spec):
        start, B, epoch, batch = spec[:4]
        out.start, out.B, out.epoch, out.batch = int(start), int(B), int(epoch), int(batch)
        nc = spec[4] if len(spec) > 4 else None
        el = spec[5] if len(spec) > 5 else None
        out.n_choices = int(nc) if nc is not None else 0
        out.entities_list = el.data_ptr() if el is not None else None
        out.inj_mask = out.inj_repl = None
        return el

    def _plan_step(self, start, B, epoch, batch, n_choices, entities_list, inj_mask, inj_repl, prefetch):
        import ctypes as C
        self._deferred_table_check(1)
        self.step_count += 1
        lr = (sgd_learning_rate(self.sgd_params, self.batches_count, epoch, batch) if self.sgd_params is not None
              else self.lr)
        if self.deferred:
            self._fill_lr_t(self.step_count, lr)
        cur = L.PlanBatch()
        keep = [self._plan_batch(cur, (start, B, epoch, batch, n_choices, entities_list)), inj_mask, inj_repl]
        if inj_repl is not None:
            cur.inj_repl = inj_repl.data_ptr()
            cur.inj_mask = inj_mask.data_ptr() if inj_mask is not None else None
        nxt = (L.PlanBatch * 3)()
        n_next = 0
        if prefetch is not None:
            for pf in ([prefetch] if isinstance(prefetch, tuple) else list(prefetch))[:3]:
                if pf is None or pf[1] <= 0:
                    continue
                spec = tuple(pf) if len(pf) > 4 else tuple(pf[:4]) + (n_choices, entities_list)
                keep.append(self._plan_batch(nxt[n_next], spec))
                n_next += 1
        h6 = (C.c_float * 6)(*self._hyper(lr))
        L.check(L.load().emg_plan_step(self.plan, C.byref(cur), self.step_count, h6, nxt, n_next,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream)), "emg_plan_step")

    # ---- optional per-stage HIP-event timing (bench.py) ----
    def enable_stage_timing(self, max_samples=16):
        """per-stage HIP events for the next ``max_samples`` launches of each stage (creating and recording ~10
        timing events per step costs ~0.1 ms of host time: sampling keeps long runs from turning host-bound)"""
        self.stage_events = {}
        self._stage_max = int(max_samples)
        if self.plan is not None:
            L.check(L.load().emg_plan_timing(self.plan, int(max_samples)), "emg_plan_timing")

    def _timed(self, name, fn):
        if self.stage_events is None or len(self.stage_events.get(name, ())) >= self._stage_max:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st = D.active_stream()
        e0.record(st)
        r = fn()
        e1.record(st)
        self.stage_events.setdefault(name, []).append((e0, e1))
        return r

    def stage_times_ms(self):
        torch.cuda.synchronize()
        if self.plan is not None:
            import ctypes as C
            ms, cnt = (C.c_float * 9)(), (C.c_int32 * 9)()
            L.check(L.load().emg_plan_stage_ms(self.plan, ms, cnt), "emg_plan_stage_ms")
            names = ("prepare", "fused", "forward", "loss", "backward", "apply_ent", "apply_rel", "clip", "catchup")
            return {n: [float(ms[i])] for i, n in enumerate(names) if cnt[i] > 0}
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in (self.stage_events or {}).items()}

    # ---- one batch ----
    def _hyper(self, lr, table=None):
        """(lr, momentum, beta1, beta2, eps, lr_t) + (lambda, p) of the folded LP regulariser for table 0 (entities)
        / 1 (relations); table=None: no regulariser folded in"""
        t = self.step_count
        lr_t = lr * math.sqrt(1.0 - ADAM_BETA2 ** t) / (1.0 - ADAM_BETA1 ** t)
        h = (lr, self.momentum, ADAM_BETA1, ADAM_BETA2, KERAS_EPS, lr_t)
        if table is None or self.reg is None:
            return h
        return h + (self.reg[table], float(self.reg[2]))

    def _prepare(self, sl, start, B, epoch, batch, n_choices, entities_list, inj_mask, inj_repl):
        """Everything about a batch that does not depend on the tables: corruption codes (Philox, draw counter
        a pure function of (epoch, batch, side) so a refit reproduces the same negatives), destination ids of
        the gradient rows, their stable grouping and the singleton flags."""
        et, eta = self.eta_total, self.eta
        pos = self.X[start:start + B]
        n_choices = self.n_ent if n_choices is None else int(n_choices)
        codes = sl["codes"][:B * et]

        def run():
            # draw counter of side sd = ((epoch-1)*batches_count + (batch-1))*n_sides + sd
            counter0 = ((epoch - 1) * self.batches_count + (batch - 1)) * self.n_sides
            xe, xr = self._xe, self._xr
            n_ce = (2 + et) * B
            D.prepare_batch(pos, eta, self.sides, n_choices, codes, sl["dest_ent"][:xe + n_ce], sl["dest_rel"][:xr + B],
                            self.n_ent, self.n_rel, sl["ws_ent"], sl["ws_rel"], entities_list=entities_list,
                            seed=self.seed, counter0=counter0, inj_mask=inj_mask, inj_repl=inj_repl, n_extra_ent=xe,
                            n_extra_rel=xr, single_flags=sl["single"][:n_ce] if self.inplace else None,
                            factored=self.factored)

        if self.pipeline:
            side = self.sides_st[self._side_rr]
            self._side_rr = (self._side_rr + 1) % len(self.sides_st)
            side.wait_event(sl["done"])  # the compute that last used this slot has finished
            if sl["key"] is not None:    # evicting a prepared-but-never-consumed slot (caller's prefetch list did
                side.wait_event(sl["ready"])  # not match its steps): its chain may still run on the other side stream
            with D.on_stream(side):
                self._timed("prepare", run)
                sl["ready"].record(side)
        else:
            self._timed("prepare", run)
        sl["key"] = (start, B, epoch, batch)

    def step(self, start, B, epoch=1, batch=1, n_choices=None, entities_list=None, inj_mask=None, inj_repl=None,
             prefetch=None):
        """Train on resident triples [start, start+B).  ``prefetch`` = (start, B, epoch, batch[, n_choices,
        entities_list]) of the NEXT batch — or a list of the next two — lets their preparation overlap this
        batch's compute.  Two ahead matters: the preparation chain (~14 small dependent launches) takes about one
        step of wall time when it shares the GPU with the big kernels, so one batch of lookahead leaves it on the
        critical path."""
        if B <= 0:
            return
        if self.batch_sharded:
            return self._step_batch_sharded(start, B, epoch, batch, n_choices, entities_list, prefetch)
        self._alloc_scratch(B)
        if self.plan is not None:
            return self._plan_step(start, B, epoch, batch, n_choices, entities_list, inj_mask, inj_repl, prefetch)
        self.step_count += 1
        key = (start, B, epoch, batch)
        sl = next((s for s in self.slots if s["key"] == key), None)
        if sl is None:
            sl = self.slots[0] if not self.pipeline else min(self.slots, key=lambda s: s["key"] is not None)
            self._prepare(sl, start, B, epoch, batch, n_choices, entities_list, inj_mask, inj_repl)
        if self.pipeline and prefetch is not None:
            # one batch or a list of the next ones (nearest first): each goes to a free slot unless already held
            upcoming = [prefetch] if isinstance(prefetch, tuple) else list(prefetch)
            for pf in upcoming:
                if pf is None or pf[1] <= 0 or any(s["key"] == tuple(pf[:4]) for s in self.slots):
                    continue
                free = next((s for s in self.slots if s is not sl and s["key"] is None), None)
                if free is None:
                    break
                self._prepare(free, pf[0], pf[1], pf[2], pf[3], pf[4] if len(pf) > 4 else n_choices,
                              pf[5] if len(pf) > 5 else entities_list, None, None)
        main = torch.cuda.current_stream()  # looked up once per step; every call below is routed explicitly
        if self.pipeline:
            main.wait_event(sl["ready"])
        with D.on_stream(main):
            self._compute(sl, start, B, epoch, batch, main)
        if self.pipeline:
            sl["done"].record(main)
        sl["key"] = None

    def run_batches(self, specs):
        """Train on a sequence of batches: ``specs`` = [(start, B, epoch, batch[, n_choices, entities_list]), ...] (empty
        batches allowed).  Small batches run as graph replays — one library call for the whole sequence
        (emg_plan_run: two launches per 32 steps from the host); otherwise one ``step`` per batch with the next batches
        prepared ahead."""
        specs = [s for s in specs if s is not None and s[1] > 0]
        if not specs:
            return
        if not (self.plan is not None and self.graph and self.stage_events is None):
            for i, s in enumerate(specs):
                self.step(s[0], s[1], epoch=s[2], batch=s[3], n_choices=s[4] if len(s) > 4 else None,
                          entities_list=s[5] if len(s) > 5 else None, prefetch=specs[i + 1:i + 4])
            return
        import ctypes as C
        self._alloc_scratch(max(s[1] for s in specs))
        n = len(specs)
        arr = (L.PlanBatch * n)()
        hyp = (C.c_float * (6 * n))()
        keep = []
        first = self.step_count + 1
        for i, s in enumerate(specs):
            keep.append(self._plan_batch(arr[i], s))
            self.step_count += 1
            lr = (sgd_learning_rate(self.sgd_params, self.batches_count, s[2], s[3]) if self.sgd_params is not None else self.lr)
            hyp[6 * i:6 * i + 6] = self._hyper(lr)
        L.check(L.load().emg_plan_run(self.plan, arr, n, first, hyp, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                "emg_plan_run")

    def _compute(self, sl, start, B, epoch, batch, main):
        pos = self.X[start:start + B]
        et, eta = self.eta_total, self.eta
        codes = sl["codes"][:B * et]
        n_ce, n_cr, xe, xr = (2 + et) * B, B, self._xe, self._xr
        ce, cr = self.contrib_ent[xe:xe + n_ce], self.contrib_rel[xr:xr + B]   # batch rows follow the LP rows
        if self.factored:
            ce = self.contrib_ent[:4 * B]
        single = sl["single"][:n_ce] if self.inplace else None
        lr = (sgd_learning_rate(self.sgd_params, self.batches_count, epoch, batch) if self.sgd_params is not None
              else self.lr)
        hyper_e, hyper_r = self._hyper(lr, 0), self._hyper(lr, 1)
        lp_e, lp_r = (self.lp_sum[0:1], self.lp_sum[1:2]) if len(hyper_e) == 8 else (None, None)
        fold = single is not None and len(hyper_e) == 8     # in-place updates fold the regulariser themselves (plain SGD)
        inplace_kw = dict(single_ent=single, opt_id=self.opt_id, step=self.step_count, hyper=hyper_e if fold else hyper_e[:6],
                          ent_state0=self.state_ent[0], ent_state1=self.state_ent[1], tag_ent=self.tag_ent,
                          fac_ws_ent=sl["ws_ent"] if self.factored else None, lp_accum=lp_e if fold else None)
        if self.fused:
            self._timed("fused", lambda: D.train_backward_ex(
                self.model_id, self.ent, self.rel, self.k_int, self.scale, pos, et, codes, ce, cr,
                fused_loss=self.loss_id, margin=self.margin, loss_accum=self.loss_accum, loss_slots=LOSS_SLOTS, **inplace_kw))
        else:
            sall = self.scores_all[:B * (1 + et)]
            sp, sn = sall[:B], sall[B:]
            flags = L.SCORE_PARTIAL if self.sharded else L.SCORE_FINAL
            self._timed("forward", lambda: D.train_forward(self.model_id, self.ent, self.rel, self.k_int, self.scale,
                                                           pos, et, codes, flags=flags, scores_pos=sp, scores_neg=sn))
            bw = {}
            if self.sharded:
                # the ONE collective of a training step: sum the k-slab partial scores over the ranks
                self._timed("allreduce", lambda: (parallel.allreduce_sum_(sall),
                                                  D.finalize_scores(self.model_id, self.scale, sall)))
                if self.model_id == L.TRANSE_L2:  # its gradient needs the full norm, not the slab's
                    bw = dict(bw_scores_pos=sp, bw_scores_neg=sn)
            elif self.wide and self.model_id == L.TRANSE_L2:   # column blocks: the same need
                bw = dict(bw_scores_pos=sp, bw_scores_neg=sn)
            gp, gn = self.g_pos[:B], self.g_neg[:B * et]
            self._timed("loss", lambda: D.loss(self.loss_id, sp, sn, B, eta, self.n_sides, self.margin, self.alpha,
                                               self.loss_accum[0:1], gp, gn))
            self._timed("backward", lambda: D.train_backward_ex(
                self.model_id, self.ent, self.rel, self.k_int, self.scale, pos, et, codes, ce, cr, fused_loss=-1,
                g_pos=gp, g_neg=gn, **bw, **inplace_kw))
        apply_rel = lambda: D.apply_grouped(self.opt_id, self.rel, self.k_int, self.state_rel[0],  # noqa: E731
                                            self.state_rel[1], self.tag_rel, self.step_count, cr, n_cr,
                                            False, hyper_r, sl["ws_rel"], lp_accum=lp_r)
        # (small batches stay on one stream: the fork/join costs ~60 us of host time, more than the overlap buys)
        use_aux = self.aux is not None and n_ce >= AUX_MIN_ROWS
        if use_aux:
            # the relation table's apply (few, long segments: latency-bound, ~0.06 ms at 0.5 TB/s) is independent
            # of the entity table's: it runs on a second stream underneath it
            self.aux_fork.record(main)
            self.aux.wait_event(self.aux_fork)
            with D.on_stream(self.aux):
                self._timed("apply_rel", apply_rel)
                self.aux_join.record(self.aux)
        self._timed("apply_ent", lambda: D.apply_grouped(self.opt_id, self.ent, self.k_int, self.state_ent[0],
                                                         self.state_ent[1], self.tag_ent, self.step_count, ce, n_ce,
                                                         self.inplace, hyper_e, sl["ws_ent"], lp_accum=lp_e,
                                                         factored=self.factored))
        if use_aux:
            main.wait_event(self.aux_join)
        else:
            self._timed("apply_rel", apply_rel)
        if self.normalize:
            # EmbeddingModel.py:1434-1440: tf.clip_by_norm(ent_emb, clip_norm=1, axes=1) after each batch
            D.clip_rows(self.ent, self.k_int, 1.0)

    # ---- batch-sharded step (parallel.py: BATCH sharding) ----
    def _bs_prepare(self, key, n_choices, entities_list, parity):
        """Everything of a batch-sharded step that does not read the tables: this rank's rows of the global batch, their Philox
        negatives and local grouping (emg_prepare_batch) and — device exchange — the exchange's metadata phase for both tables.
        Issued on the side stream a step ahead when the caller names the next batch (``prefetch``)."""
        start, B, epoch, batch = key
        rank, world = parallel.rank_world()
        r0, r1 = parallel.batch_rows(B, rank, world)
        Bl = r1 - r0
        et, eta = self.eta_total, self.eta
        n_ce = (2 + et) * Bl
        st = {"key": key, "parity": parity, "r0": r0, "Bl": Bl, "n_ce": n_ce, "pe": None, "pr": None}
        dev = self.device
        if Bl > 0:
            self._alloc_scratch(Bl)
        sl = self.slots[parity % len(self.slots)]
        if Bl > 0:
            pos = self.X[start + r0:start + r1]
            counter0 = ((epoch - 1) * self.batches_count + (batch - 1)) * self.n_sides
            D.prepare_batch(pos, eta, self.sides, self.n_ent if n_choices is None else int(n_choices), sl["codes"][:Bl * et],
                            sl["dest_ent"][:n_ce], sl["dest_rel"][:Bl], self.n_ent, self.n_rel, sl["ws_ent"], sl["ws_rel"],
                            entities_list=entities_list, seed=self.seed, counter0=counter0, B_global=B, row_offset=r0)
        if self._device_exchange:
            if Bl > 0:
                gslot_e, gslot_r = self._gslots(Bl, B, r0, n_ce)      # slot of the same row in the whole batch's layout
                sorted_e = D.apply_workspace_views(sl["ws_ent"], n_ce)   # the local grouping: destinations ascending, slots in order
                sorted_r = D.apply_workspace_views(sl["ws_rel"], Bl)
            else:
                z32 = torch.zeros(0, dtype=torch.int32, device=dev)
                gslot_e = gslot_r = z32
                sorted_e = sorted_r = (z32, z32)
            st["pe"] = self._plan_exchange("ent", sorted_e[0], sorted_e[1], gslot_e, parity)
            st["pr"] = self._plan_exchange("rel", sorted_r[0], sorted_r[1], gslot_r, parity)
        return st

    def _step_batch_sharded(self, start, B, epoch, batch, n_choices, entities_list, prefetch=None):
        """[start, start + B) is the GLOBAL batch.  This rank scores rows [r0, r1) of it with the negatives the whole batch draws
        for them, then: gradient rows -> owners (all_to_all) -> one summed row per destination, added in global slot order ->
        all-gather -> every replica applies the optimizer (parallel.py: BATCH sharding).  Device exchange (default): the
        table-independent half — preparation, row counts, (destination, slot) pairs, groupings — is issued for the NEXT batch on a
        side stream as soon as this step's kernels are enqueued; between the scoring kernel and the apply there is no host read."""
        key = (start, B, epoch, batch)
        self.step_count += 1
        et, eta, k = self.eta_total, self.eta, self.k_int
        lr = (sgd_learning_rate(self.sgd_params, self.batches_count, epoch, batch) if self.sgd_params is not None
              else self.lr)
        dev = self.device
        main = torch.cuda.current_stream()
        st = self._bs_ready.pop(key, None)
        if st is None:
            self._bs_ready.clear()                 # (a prefetch list that did not match the steps: nothing prepared is kept)
            st = self._bs_prepare(key, n_choices, entities_list, self._bs_parity)
        else:
            main.wait_event(st["event"])           # the side stream's preparation and metadata phase of this batch
        self._bs_parity = st["parity"] ^ 1
        Bl, r0, n_ce = st["Bl"], st["r0"], st["n_ce"]
        sl = self.slots[st["parity"] % len(self.slots)]
        if Bl > 0:
            pos = self.X[start + r0:start + r0 + Bl]
            codes = sl["codes"][:Bl * et]
            ce, cr = self.contrib_ent[:n_ce], self.contrib_rel[:Bl]
            if self.fused:
                D.train_backward_ex(self.model_id, self.ent, self.rel, k, self.scale, pos, et, codes, ce, cr,
                                    fused_loss=self.loss_id, margin=self.margin, loss_accum=self.loss_accum, loss_slots=LOSS_SLOTS)
            else:
                sall = self.scores_all[:Bl * (1 + et)]
                sp, sn = sall[:Bl], sall[Bl:]
                D.train_forward(self.model_id, self.ent, self.rel, k, self.scale, pos, et, codes, scores_pos=sp, scores_neg=sn)
                gp, gn = self.g_pos[:Bl], self.g_neg[:Bl * et]
                D.loss(self.loss_id, sp, sn, Bl, eta, self.n_sides, self.margin, self.alpha, self.loss_accum[0:1], gp, gn)
                D.train_backward_ex(self.model_id, self.ent, self.rel, k, self.scale, pos, et, codes, ce, cr, fused_loss=-1,
                                    g_pos=gp, g_neg=gn)
            dest_e, dest_r = sl["dest_ent"][:n_ce], sl["dest_rel"][:Bl]
            rows_e, rows_r = ce, cr
        else:
            dest_e = dest_r = torch.zeros(0, dtype=torch.int32, device=dev)
            rows_e = rows_r = torch.zeros((0, k), dtype=torch.float32, device=dev)
        lp = (self.lp_sum[0:1], self.lp_sum[1:2]) if self.reg is not None else (None, None)
        he, hr = self._hyper(lr, 0), self._hyper(lr, 1)
        if self._device_exchange:
            # data phase: rows -> owners, summed in global slot order, sums -> every replica, optimizer
            self._exchange_apply_planned("ent", st["pe"], rows_e, self.ent, self.state_ent, self.tag_ent, he, lp[0], st["parity"])
            self._exchange_apply_planned("rel", st["pr"], rows_r, self.rel, self.state_rel, self.tag_rel, hr, lp[1], st["parity"])
        else:
            if Bl > 0:
                gslot_e, gslot_r = self._gslots(Bl, B, r0, n_ce)
            else:
                gslot_e = gslot_r = dest_e
            self._exchange_apply(self.ent, self.n_ent, self.state_ent, self.tag_ent, dest_e, gslot_e.to(torch.int64), rows_e, he, "ent", lp[0])
            self._exchange_apply(self.rel, self.n_rel, self.state_rel, self.tag_rel, dest_r, gslot_r.to(torch.int64), rows_r, hr, "rel", lp[1])
        if self.normalize:
            D.clip_rows(self.ent, k, 1.0)
        self._bs_done[st["parity"]].record(main)
        # the next batch's table-independent half, on the side stream, now: it runs beside this step's exchange and apply
        if self._bs_ahead and prefetch:
            nxt = prefetch if isinstance(prefetch, tuple) else next((p for p in prefetch if p is not None and p[1] > 0), None)
            if nxt is not None and tuple(nxt[:4]) not in self._bs_ready:
                par = st["parity"] ^ 1
                side = self._bs_stream
                side.wait_event(self._bs_done[par])          # the step that last used that slot and those workspaces (two steps back)
                with torch.cuda.stream(side):
                    nst = self._bs_prepare(tuple(nxt[:4]), nxt[4] if len(nxt) > 4 else n_choices, nxt[5] if len(nxt) > 5 else entities_list, par)
                    nst["event"] = torch.cuda.Event()
                    nst["event"].record(side)
                for pl in (nst["pe"], nst["pr"]):            # (allocated on the side stream, read by the main stream's data phase)
                    if pl is not None:
                        pl.order.record_stream(main)
                        pl.uniq_local.record_stream(main)
                self._bs_ready[tuple(nxt[:4])] = nst

    def _gslots(self, Bl, B, r0, n_ce):
        """global contribution slot of every local slot (int32): local slot t = role block t // Bl, row t % Bl -> block * B + r0 + row"""
        key = (Bl, B, r0, n_ce)
        if self._gslot_cache.get("key") != key:
            t = torch.arange(n_ce, dtype=torch.int64, device=self.device)
            self._gslot_cache = {"key": key, "e": ((t // Bl) * B + r0 + (t % Bl)).to(torch.int32),
                                 "r": (r0 + torch.arange(Bl, dtype=torch.int64, device=self.device)).to(torch.int32)}
        return self._gslot_cache["e"], self._gslot_cache["r"]

    def _xchg(self, which):
        x = self._xchg_objs.get(which)
        if x is None:
            x = parallel.RowExchange(self.n_ent if which == "ent" else self.n_rel, self.k_int, self.device)
            x.sums = alloc_table(max(1, x.e1 - x.e0), self.k_int, self.device)   # this rank's range: one summed gradient row per owned row
            x.compact = None
            x.ws_owner, x.ws_rep = [None, None], [None, None]
            self._xchg_objs[which] = x
        return x

    @staticmethod
    def _grown(buf, need, device):
        if buf is None or buf.numel() < need:
            return torch.empty(int(need * 1.25) + 1024, dtype=torch.uint8, device=device)
        return buf

    def _plan_exchange(self, which, dest_sorted, order, gslot, parity=0):
        """METADATA phase of one table's exchange (parallel.RowExchange): nothing here reads the tables.  The two grouping
        workspaces exist twice (``parity``): the next batch's are filled while this batch's are being applied from."""
        x, k = self._xchg(which), self.k_int
        pl = x.plan_counts(dest_sorted, order, gslot)
        owned = max(1, x.e1 - x.e0)
        if pl.m:
            # the owner's order: by destination, a destination's rows in GLOBAL slot order (the order one GPU adds them in)
            x.ws_owner[parity] = self._grown(x.ws_owner[parity], D.apply_workspace_bytes(pl.m, owned, k), self.device)
            D.group_dest_keyed(pl.dest_o, pl.gslot_o, pl.m, owned, x.ws_owner[parity])
            keys_o = D.apply_workspace_views(x.ws_owner[parity], pl.m)[0]
        else:
            keys_o = pl.dest_o
        x.plan_unique(pl, keys_o)
        n_all = x.world * pl.cap_u
        x.ws_rep[parity] = self._grown(x.ws_rep[parity], D.apply_workspace_bytes(n_all, x.n_rows, k), self.device)
        D.group_dest(pl.ids_all, n_all, x.n_rows, x.ws_rep[parity])        # (the sentinel ids of the padding are dropped: no row to update)
        return pl

    def _exchange_apply_planned(self, which, pl, rows, table, state, tag, hyper, lp_accum, parity=0):
        """DATA phase: gradient rows -> owners (all_to_all) -> one summed row per destination, added in global slot order ("SGD with
        lr = -1 on a zero row" is 0 - (-1 * g) = g exactly: emg_apply_grouped as a segmented sum) -> all-gather -> every replica
        applies the optimizer to the same sums with its own (replicated, hence identical) state"""
        x, k = self._xchg(which), self.k_int
        recv = x.send_rows(pl, rows)
        x.sums.index_fill_(0, pl.uniq_local, 0.0)
        if pl.m:
            D.apply_grouped(L.OPT_SGD, x.sums, k, None, None, None, 1, recv, pl.m, False, (-1.0, 0, 0, 0, 0, 0), x.ws_owner[parity], factored=True)
        if x.compact is None or x.compact.shape[0] < pl.cap_u:
            x.compact = torch.empty((int(pl.cap_u * 1.25) + 64, k), dtype=torch.float32, device=self.device)
        comp = x.compact[:pl.cap_u]
        torch.index_select(x.sums, 0, pl.uniq_local, out=comp)
        if self.shard_state:
            # the OWNER applies the optimizer to its rows (its shard of the state), the UPDATED rows are all-gathered and every
            # replica copies them into its table: one update per touched row in the whole job instead of one per replica
            own = table[x.e0:x.e1]
            dest_local = torch.where(pl.mine < x.n_rows, pl.mine - x.e0, torch.full_like(pl.mine, x.e1 - x.e0))   # padding: outside the range
            x.ws_own = self._grown(getattr(x, "ws_own", None), D.apply_workspace_bytes(pl.cap_u, max(1, x.e1 - x.e0), k), self.device)
            D.apply_rows(self.opt_id, own, k, state[0], state[1], None, self.step_count, comp, dest_local, pl.cap_u, hyper, x.ws_own)
            torch.index_select(own[:, :k], 0, pl.uniq_local, out=comp)          # (padding entries: row 0 of the range; their ids are skipped)
            g = x.gather_sums(pl, comp)
            D.scatter_rows(table, k, g, pl.ids_all)
            self.opt_rows += pl.rows_mine
        else:
            g = x.gather_sums(pl, comp)
            D.apply_grouped(self.opt_id, table, k, state[0], state[1], tag, self.step_count, g, x.world * pl.cap_u, False, hyper, x.ws_rep[parity],
                            lp_accum=lp_accum)
            self.opt_rows += pl.rows_all
        self.opt_rows_owner_form += pl.rows_mine
        self.opt_rows_replicated_form += pl.rows_all
        self.xgmi_bytes += pl.sent_bytes + pl.recv_bytes

    def _exchange_apply(self, table, n_rows, state, tag, dest, gslot, rows, hyper, which, lp_accum=None):
        """gradient rows -> owners (all_to_all) -> per destination ONE summed row, added in global slot order (the order a
        single GPU adds them in) -> all-gather of (destination, summed row) -> every replica applies the optimizer to the
        same sums with its own (replicated, hence identical) state: tables, state and loss are those of one GPU, bit for
        bit, for every optimizer — Keras Adam's dense decay and a folded LP regulariser included (they need every row)."""
        k = self.k_int
        dest_o, gslot_o, rows_o, sent = parallel.exchange_rows(dest, gslot, rows, n_rows)
        m = int(dest_o.numel())
        if m:
            # the sum order: by slot of the (global) batch layout; the stable grouping by destination below keeps it.
            # Destinations are renumbered 0..u-1 so that the sums land in a compact [u, k] buffer: "SGD with lr = -1 on a
            # zero table" is 0 - (-1 * g) = g exactly, i.e. emg_apply_grouped used as a segmented sum
            perm = torch.argsort(gslot_o, stable=True)
            dest_p = dest_o.index_select(0, perm)
            upd, compact = torch.unique(dest_p, sorted=True, return_inverse=True)
            u = int(upd.numel())
            ws = self._owner_ws.get(which)
            need = D.apply_workspace_bytes(m, u, k)
            if ws is None or ws.numel() < need:
                ws = self._owner_ws[which] = torch.empty(int(need * 1.5) + 1024, dtype=torch.uint8, device=self.device)
            cid = compact.to(torch.int32).contiguous()
            D.group_dest(cid, m, u, ws)
            keys, vals = D.apply_workspace_views(ws, m)
            vals.copy_(perm.index_select(0, vals.to(torch.int64)).to(torch.int32))   # positions in RECEIVE order
            sums = alloc_table(u, k, self.device)        # (zero-filled)
            D.apply_grouped(L.OPT_SGD, sums, k, None, None, None, 1, rows_o, m, False, (-1.0, 0, 0, 0, 0, 0), ws)
            upd = upd.to(torch.int32)
        else:
            upd = torch.zeros(0, dtype=torch.int32, device=self.device)
            sums = alloc_table(1, k, self.device)[:0]
        oid, orow, recvd = parallel.allgather_rows(upd, sums)
        ids = torch.cat([upd, oid]) if oid.numel() else upd
        g = torch.cat([sums, orow.to(sums.dtype)]) if oid.numel() else sums
        n = int(ids.numel())
        # every replica: one optimizer step per touched row from its summed gradient (each destination occurs once: a
        # segment of one row), the dense pass for the rest where the optimizer / regulariser needs one
        ws2 = self._owner_ws.get(which + "2")
        need2 = D.apply_workspace_bytes(max(n, 1), n_rows, k)
        if ws2 is None or ws2.numel() < need2:
            ws2 = self._owner_ws[which + "2"] = torch.empty(int(need2 * 1.5) + 1024, dtype=torch.uint8, device=self.device)
        gg = alloc_table(max(n, 1), k, self.device)
        if n:
            gg[:n].copy_(g)
            D.group_dest(ids.contiguous(), n, n_rows, ws2)
        D.apply_grouped(self.opt_id, table, k, state[0], state[1], tag, self.step_count, gg, n, False, hyper, ws2, lp_accum=lp_accum)
        self.xgmi_bytes += sent + recvd

    def read_loss(self, reset=True):
        """data loss (identical on every rank) + LP term (summed over the column slabs when sharded)"""
        if self.deferred and self.reg is not None:
            self.materialize()   # the regulariser's value over the rows whose dense updates are still pending
        reg, data = self.reg_accum, self.loss_accum
        if self.sharded:  # sum a COPY over the ranks: the accumulator itself stays rank-local (reset=False calls)
            reg = parallel.allreduce_sum_(self.reg_accum.clone())
        if self.batch_sharded:  # every rank saw its rows of each batch only; the LP term was computed by every replica
            data = parallel.allreduce_sum_(self.loss_accum.clone())
        v = float(data.sum().item()) + float(reg.item())
        if self.reg is not None:   # (batch-sharded: every replica folded the regulariser over the full tables itself)
            lp = self.lp_sum if not self.sharded else parallel.allreduce_sum_(self.lp_sum.clone())
            lp = lp.cpu()
            v += self.reg[0] * float(lp[0]) + self.reg[1] * float(lp[1])
        if reset:
            self.loss_accum.zero_()
            self.reg_accum.zero_()
            self.lp_sum.zero_()
        return v

    def _step_lr(self, t):
        """the learning rate optimizer step t uses (Adam: lr_t), assuming fit()'s order of batches (step t = batch (t - 1) mod
        batches_count of epoch (t - 1) // batches_count); a step issued out of that order corrects its own entry"""
        if self.opt_id == L.OPT_ADAM:
            return self.lr * math.sqrt(1.0 - ADAM_BETA2 ** t) / (1.0 - ADAM_BETA1 ** t)
        if self.sgd_params is not None:
            return sgd_learning_rate(self.sgd_params, self.batches_count, (t - 1) // self.batches_count + 1, (t - 1) % self.batches_count + 1)
        return self.lr

    LR_TABLE_STEPS = 1 << 22    # entries of the deferred pass's learning-rate table (16 MB)

    def _deferred_table_check(self, n_more):
        """the deferred pass reads each replayed step's learning rate from a table of LR_TABLE_STEPS entries the plan points at; a
        fit that long (e.g. batches_count = 1000 for 4200 epochs) brings every row up to date once and goes on with the dense pass"""
        if self.deferred and self._lr_t_hist is not None and self.step_count + n_more >= self._lr_t_hist.numel():
            self.materialize()
            self.deferred = False
            self._make_plan()

    def _fill_lr_t(self, upto, lr_now=None):
        """learning rates of the steps (filled, upto] into the device table the replay reads: the values _hyper hands the
        kernels, float32-rounded the same way; ``lr_now``: what step ``upto`` really uses"""
        if upto >= self._lr_t_hist.numel():
            raise RuntimeError("deferred dense pass: more than %d optimizer steps" % self._lr_t_hist.numel())
        if upto > self._lr_t_filled:
            hi = min(max(upto, self._lr_t_filled + 8192), self._lr_t_hist.numel() - 1)   # ahead in blocks: one upload per 8192 steps
            vals = [self._step_lr(t) for t in range(self._lr_t_filled + 1, hi + 1)]
            self._lr_host.extend(np.asarray(vals, dtype=np.float64).astype(np.float32).tolist())
            self._lr_t_hist[self._lr_t_filled + 1:hi + 1] = torch.tensor(vals, dtype=torch.float64).to(torch.float32).to(self.device)
            self._lr_t_filled = hi
        if lr_now is not None and self.opt_id != L.OPT_ADAM:
            v = float(np.float32(lr_now))
            if self._lr_host[upto] != v:      # a step outside fit()'s order (tests drive epochs / batches freely)
                self._lr_host[upto] = v
                self._lr_t_hist[upto:upto + 1] = torch.tensor([lr_now], dtype=torch.float64).to(torch.float32).to(self.device)

    def materialize(self):
        """deferred dense pass: bring every row of both tables (and its optimizer state) up to the current step"""
        if not self.deferred or self.step_count == 0 or self._lr_t_hist is None:
            return
        import ctypes as C
        lib = L.load()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        for i, (tab, n, state, tag) in enumerate(((self.ent, self.n_ent, self.state_ent, self.tag_ent),
                                                  (self.rel, self.n_rel, self.state_rel, self.tag_rel))):
            hy = self._hyper(self.lr, i)
            h = (C.c_float * 8)(*(hy if len(hy) == 8 else hy + (0.0, 0.0)))
            lp = self.lp_sum[i:i + 1].data_ptr() if self.reg is not None else None
            L.check(lib.emg_deferred_materialize(self.opt_id, tab.data_ptr(), n, tab.stride(0), self.k_int, ptr(state[0]), ptr(state[1]),
                                                 tag.data_ptr(), h, self._lr_t_hist.data_ptr(), self.step_count, lp, st),
                    "emg_deferred_materialize")

    def tables_numpy(self):
        self.materialize()
        # (.cpu() of a device tensor is a fresh host copy already; only padded rows need compacting)
        return np.ascontiguousarray(self.ent.cpu().numpy()), np.ascontiguousarray(self.rel.cpu().numpy())
