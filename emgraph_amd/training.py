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

import numpy as np
import torch

from . import _lib as L
from . import device as D

DEFAULT_LR = 0.0005            # training/_optimizer_constants.py:9
DEFAULT_MOMENTUM = 0.9         # :11
DEFAULT_DECAY_CYCLE = 0
DEFAULT_DECAY_CYCLE_MULTIPLE = 1
DEFAULT_LR_DECAY_FACTOR = 2
DEFAULT_END_LR = 1e-8
DEFAULT_SINE = False
ADAM_BETA1, ADAM_BETA2, KERAS_EPS = 0.9, 0.999, 1e-7   # tf.optimizers defaults (TF 2.2)
ADAGRAD_INIT_ACC = 0.1


class SGDSchedule:
    """Learning-rate schedule of training/sgd.py:127-185 (`update_feed_dict`), restated as host logic.
    The reference's TF2 port never calls it (SURVEY A-4); values pinned by
    tests/emgraph/models/test_optimizers.py:21,41-43,72-79."""

    def __init__(self, params, batches_count):
        self.batches_count = batches_count
        self.start_lr = params.get("lr", DEFAULT_LR)
        self.current_lr = self.start_lr
        self.decay_cycle_rate = params.get("decay_cycle", DEFAULT_DECAY_CYCLE)
        self.end_lr = params.get("end_lr", DEFAULT_END_LR)
        self.is_cosine_decay = params.get("cosine_decay", DEFAULT_SINE)
        self.next_cycle_epoch = self.decay_cycle_rate + 1
        self.decay_cycle_expand_factor = params.get("expand_factor", DEFAULT_DECAY_CYCLE_MULTIPLE)
        self.decay_lr_rate = params.get("decay_lr_rate", DEFAULT_LR_DECAY_FACTOR)
        self.curr_cycle_length = self.decay_cycle_rate
        self.curr_start = 0

    def lr(self, batch_num, epoch_num):
        if self.is_cosine_decay:
            current_cycle_num = ((epoch_num - 1 - self.curr_start) * self.batches_count + (batch_num - 1)) / (
                self.curr_cycle_length * self.batches_count)
            self.current_lr = self.end_lr + (self.start_lr - self.end_lr) * 0.5 * (
                1 + math.cos(math.pi * current_cycle_num))
            if epoch_num % (self.next_cycle_epoch - 1) == 0 and batch_num == self.batches_count:
                self.curr_cycle_length = self.curr_cycle_length * self.decay_cycle_expand_factor
                self.next_cycle_epoch = self.next_cycle_epoch + self.curr_cycle_length
                self.curr_start = epoch_num
                self.start_lr = self.start_lr / self.decay_lr_rate
            if self.current_lr < self.end_lr:
                self.current_lr = self.end_lr
        elif self.decay_cycle_rate > 0:
            if epoch_num % self.next_cycle_epoch == 0 and batch_num == 1:
                if self.current_lr > self.end_lr:
                    self.next_cycle_epoch = (self.decay_cycle_rate
                                             + ((self.next_cycle_epoch - 1) * self.decay_cycle_expand_factor) + 1)
                    self.current_lr = self.current_lr / self.decay_lr_rate
                    if self.current_lr < self.end_lr:
                        self.current_lr = self.end_lr
        return self.current_lr


def _padded_ld(k_int):
    """row stride: multiple of 4 floats so every row is 16-byte aligned for dwordx4 loads"""
    return ((k_int + 3) // 4) * 4


def alloc_table(rows, k_int, device, init=None, fill=None):
    ld = _padded_ld(k_int)
    buf = torch.zeros((rows, ld), dtype=torch.float32, device=device)
    view = buf[:, :k_int]
    if init is not None:
        view.copy_(torch.from_numpy(np.ascontiguousarray(init, dtype=np.float32)))
    elif fill is not None:
        view.fill_(fill)
    return view  # 2-D view with stride(0) = ld


class Trainer:
    """One model's device state + the per-batch step."""

    def __init__(self, model_id, k_int, scale, ent_init, rel_init, eta, loss="nll", loss_params=None,
                 optimizer="adam", optimizer_params=None, corrupt_sides=("s,o",), batches_count=1, seed=0,
                 regularizer=None, regularizer_params=None, normalize_ent_emb=False, device="cuda"):
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
        self.schedule = SGDSchedule(op, self.batches_count) if optimizer == "sgd" else None
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
        # SGD folds the dense LP gradient into an in-place pass; every other optimizer needs the true
        # per-element gradient sum, so the LP gradient is appended as one contribution row per table row
        self.reg_rows = self.reg is not None and optimizer != "sgd"

        self.loss_accum = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.X = None
        self._cap = 0
        self.stage_events = None  # filled by enable_stage_timing()

    # ---- data ----
    def set_training_set(self, X_idx, batch_size):
        """Upload the whole mapped training set once; allocate per-batch scratch for ``batch_size``."""
        X_idx = np.ascontiguousarray(X_idx, dtype=np.int32)
        self.X = torch.from_numpy(X_idx).to(self.device)
        self._alloc_scratch(int(batch_size))

    def _alloc_scratch(self, B):
        if B <= self._cap:
            return
        dev, k, et = self.device, self.k_int, self.eta_total
        ldc = _padded_ld(k)
        self.codes = torch.empty(B * et, dtype=torch.int32, device=dev)
        self.scores_pos = torch.empty(B, dtype=torch.float32, device=dev)
        self.scores_neg = torch.empty(B * et, dtype=torch.float32, device=dev)
        self.g_pos = torch.empty(B, dtype=torch.float32, device=dev)
        self.g_neg = torch.empty(B * et, dtype=torch.float32, device=dev)
        xe = self.n_ent if self.reg_rows else 0
        xr = self.n_rel if self.reg_rows else 0
        self.contrib_ent = torch.empty(((2 + et) * B + xe, ldc), dtype=torch.float32, device=dev)[:, :k]
        self.contrib_rel = torch.empty((B + xr, ldc), dtype=torch.float32, device=dev)[:, :k]
        self.dest_ent = torch.empty((2 + et) * B + xe, dtype=torch.int32, device=dev)
        self.dest_rel = torch.empty(B + xr, dtype=torch.int32, device=dev)
        nb = max(D.apply_workspace_bytes((2 + et) * B + xe, self.n_ent), D.apply_workspace_bytes(B + xr, self.n_rel))
        self.workspace = torch.empty(nb, dtype=torch.uint8, device=dev)
        self._cap = B

    # ---- optional per-stage HIP-event timing (bench.py) ----
    def enable_stage_timing(self):
        self.stage_events = {}

    def _timed(self, name, fn):
        if self.stage_events is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.stage_events.setdefault(name, []).append((e0, e1))
        return r

    def stage_times_ms(self):
        torch.cuda.synchronize()
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in (self.stage_events or {}).items()}

    # ---- one batch ----
    def _hyper(self, lr):
        t = self.step_count
        lr_t = lr * math.sqrt(1.0 - ADAM_BETA2 ** t) / (1.0 - ADAM_BETA1 ** t)
        return (lr, self.momentum, ADAM_BETA1, ADAM_BETA2, KERAS_EPS, lr_t)

    def step(self, start, B, epoch=1, batch=1, n_choices=None, entities_list=None, inj_mask=None, inj_repl=None):
        """Train on resident triples [start, start+B).  Draw counter is a pure function of
        (epoch, batch, side) so a refit with the same seed reproduces the same negatives."""
        if B <= 0:
            return
        self._alloc_scratch(B)
        self.step_count += 1
        pos = self.X[start:start + B]
        et, eta = self.eta_total, self.eta
        n_choices = self.n_ent if n_choices is None else int(n_choices)
        codes = self.codes[:B * et]

        def gen():
            for sd, side in enumerate(self.sides):
                counter = ((epoch - 1) * self.batches_count + (batch - 1)) * self.n_sides + sd
                sl = slice(sd * eta * B, (sd + 1) * eta * B)
                D.corrupt_codes(B, eta, side, n_choices, self.device, entities_list=entities_list, seed=self.seed,
                                counter=counter, inj_mask=None if inj_mask is None else inj_mask[sl],
                                inj_repl=None if inj_repl is None else inj_repl[sl], out=codes[sl])
        self._timed("corrupt", gen)
        sp, sn = self.scores_pos[:B], self.scores_neg[:B * et]
        self._timed("forward", lambda: D.train_forward(self.model_id, self.ent, self.rel, self.k_int, self.scale, pos,
                                                       et, codes, scores_pos=sp, scores_neg=sn))
        gp, gn = self.g_pos[:B], self.g_neg[:B * et]
        self._timed("loss", lambda: D.loss(self.loss_id, sp, sn, B, eta, self.n_sides, self.margin, self.alpha,
                                           self.loss_accum, gp, gn))
        n_ce = (2 + et) * B
        ce, cr = self.contrib_ent[:n_ce], self.contrib_rel[:B]
        de, dr = self.dest_ent[:n_ce], self.dest_rel[:B]
        self._timed("backward", lambda: D.train_backward(self.model_id, self.ent, self.rel, self.k_int, self.scale, pos,
                                                         et, codes, gp, gn, ce, cr, de, dr))
        lr = self.schedule.lr(batch, epoch) if self.schedule is not None else self.lr
        hyper = self._hyper(lr)
        n_cr = B
        if self.reg is not None and not self.reg_rows:
            # dense LP term: value + SGD-style in-place step, both evaluated at the pre-update tables
            # (the sparse contributions above were also computed from the pre-update tables)
            self._timed("regularizer", lambda: (
                D.lp_regularizer(self.ent, self.k_int, self.reg[0], self.reg[2], lr, self.loss_accum),
                D.lp_regularizer(self.rel, self.k_int, self.reg[1], self.reg[2], lr, self.loss_accum)))
        elif self.reg_rows:
            # LP gradient as one extra contribution row per table row (dense by definition, lp.py:107-113)
            self._timed("regularizer", lambda: (
                D.lp_grad_rows(self.ent, self.k_int, self.reg[0], self.reg[2], self.contrib_ent[n_ce:n_ce + self.n_ent],
                               self.dest_ent[n_ce:n_ce + self.n_ent], self.loss_accum),
                D.lp_grad_rows(self.rel, self.k_int, self.reg[1], self.reg[2], self.contrib_rel[B:B + self.n_rel],
                               self.dest_rel[B:B + self.n_rel], self.loss_accum)))
            ce, de = self.contrib_ent[:n_ce + self.n_ent], self.dest_ent[:n_ce + self.n_ent]
            cr, dr = self.contrib_rel[:B + self.n_rel], self.dest_rel[:B + self.n_rel]
            n_ce, n_cr = n_ce + self.n_ent, B + self.n_rel
        self._timed("apply_ent", lambda: D.apply_rows(self.opt_id, self.ent, self.k_int, self.state_ent[0],
                                                      self.state_ent[1], self.tag_ent, self.step_count, ce, de, n_ce,
                                                      hyper, self.workspace))
        self._timed("apply_rel", lambda: D.apply_rows(self.opt_id, self.rel, self.k_int, self.state_rel[0],
                                                      self.state_rel[1], self.tag_rel, self.step_count, cr, dr, n_cr,
                                                      hyper, self.workspace))
        if self.normalize:
            # EmbeddingModel.py:1434-1440: tf.clip_by_norm(ent_emb, clip_norm=1, axes=1) after each batch
            D.clip_rows(self.ent, self.k_int, 1.0)

    def read_loss(self, reset=True):
        v = float(self.loss_accum.item())
        if reset:
            self.loss_accum.zero_()
        return v

    def tables_numpy(self):
        return self.ent.cpu().numpy().copy(), self.rel.cpu().numpy().copy()
