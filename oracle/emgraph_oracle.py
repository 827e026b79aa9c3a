"""CPU ORACLE — numpy restatement of bi-graph/Emgraph's per-batch hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product (``emgraph_amd``) never imports, links or executes anything
under ``oracle/`` and fails loudly when its HIP library is missing.

Every function cites the reference ``file:line`` (relative to the reference
tree) whose op graph it restates, op by op, in float32 numpy — deliberately
*unfused* (materialised gathers, separate elementwise passes) so that it is
also the closest stand-in for the reference's TF-eager CPU path.

Pinning status:
  * corruption generators, ``to_idx``/``create_mappings``, LP regulariser and
    metrics are pinned on the reference's OWN golden vectors
    (tests/emgraph/evaluation/test_protocol.py:418-455,490-496,530-605,
    tests/emgraph/models/test_regularizers.py:7-38,
    tests/emgraph/evaluation/test_metrics.py:6-39) — tests/test_oracle_golden.py;
  * score functions, losses and eval-corruption layouts are pinned on fixtures
    produced by executing the reference's own functions in the build container
    (tests/golden/make_golden.py, numpy stand-in for the TF leaf ops);
  * PARITY UNPINNED (lives in TensorFlow, which is absent): the
    ``tf.random.uniform`` stream, the summation order inside ``tf.reduce_sum``
    / ``tf.norm``, Keras optimizer update arithmetic, initializer draws.  The
    oracle fixes its own choices for these (Philox4x32-10 below, numpy
    pairwise sums, published Keras-2.2 update rules) and says so.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
I32 = np.int32

# utils/constants.py:87
SCORE_COMPARISON_PRECISION = 1e5
# losses/_loss_constants.py:8-16
DEFAULT_MARGIN = 1
DEFAULT_ALPHA_ADVERSARIAL = 0.5
DEFAULT_MARGIN_ADVERSARIAL = 3
DEFAULT_CLIP_EXP_LOWER = -75.0
DEFAULT_CLIP_EXP_UPPER = 75.0

MODEL_IDS = {"TransE_L1": 0, "TransE_L2": 1, "DistMult": 2, "ComplEx": 3, "HolE": 4}


# --------------------------------------------------------------------------
# a1  EmbeddingModel._lookup_embeddings / _entity_lookup  (EmbeddingModel.py:490-533)
# --------------------------------------------------------------------------
def lookup_embeddings(ent_emb, rel_emb, x):
    """Three materialised row gathers: e_s=E[x[:,0]], e_p=R[x[:,1]], e_o=E[x[:,2]]."""
    x = np.asarray(x)
    e_s = ent_emb[x[:, 0]]
    e_p = rel_emb[x[:, 1]]
    e_o = ent_emb[x[:, 2]]
    return e_s, e_p, e_o


# --------------------------------------------------------------------------
# a2..a5  score functions
# --------------------------------------------------------------------------
def fn_transe(e_s, e_p, e_o, norm=1):
    """TransE.py:208-216: -||e_s + e_p - e_o||_ord over axis 1 (ord default 1, constants.py:30)."""
    d = (e_s + e_p) - e_o
    if norm == 1:
        return -np.sum(np.abs(d), axis=1, dtype=F32)
    if norm == 2:
        return -np.sqrt(np.sum(d * d, axis=1, dtype=F32)).astype(F32)
    return -np.linalg.norm(d, ord=norm, axis=1).astype(F32)


def fn_distmult(e_s, e_p, e_o):
    """DistMult.py:201: reduce_sum(e_s * e_p * e_o, axis=1)."""
    return np.sum(e_s * e_p * e_o, axis=1, dtype=F32)


def fn_complex(e_s, e_p, e_o):
    """ComplEx.py:288-298: halves layout [re | im]; four separate reduce_sums added in this order."""
    e_s_real, e_s_img = np.split(e_s, 2, axis=1)
    e_p_real, e_p_img = np.split(e_p, 2, axis=1)
    e_o_real, e_o_img = np.split(e_o, 2, axis=1)
    return (
        np.sum(e_p_real * e_s_real * e_o_real, axis=1, dtype=F32)
        + np.sum(e_p_real * e_s_img * e_o_img, axis=1, dtype=F32)
        + np.sum(e_p_img * e_s_real * e_o_img, axis=1, dtype=F32)
        - np.sum(e_p_img * e_s_img * e_o_real, axis=1, dtype=F32)
    )


def fn_hole(e_s, e_p, e_o, k):
    """HolE.py:189: (2 / k) * ComplEx._fn — Python-float scalar times the f32 tensor."""
    return (F32(2 / k) * fn_complex(e_s, e_p, e_o)).astype(F32)


def score_fn(model, e_s, e_p, e_o, k=None):
    """Dispatch by model name ('TransE' uses norm 1; 'TransE_L2' norm 2; 'TransE_P:<ord>' any order tf.norm takes)."""
    if model in ("TransE", "TransE_L1"):
        return fn_transe(e_s, e_p, e_o, 1)
    if model.startswith("TransE_P:"):
        return fn_transe(e_s, e_p, e_o, float(model.split(":")[1]))
    if model == "TransE_L2":
        return fn_transe(e_s, e_p, e_o, 2)
    if model == "DistMult":
        return fn_distmult(e_s, e_p, e_o)
    if model == "ComplEx":
        return fn_complex(e_s, e_p, e_o)
    if model == "HolE":
        return fn_hole(e_s, e_p, e_o, k if k is not None else e_s.shape[1] // 2)
    raise ValueError(model)


def score_triples(model, ent_emb, rel_emb, x, k=None):
    """predict(): EmbeddingModel.py:2132-2133 — lookup then _fn."""
    return score_fn(model, *lookup_embeddings(ent_emb, rel_emb, x), k=k)


# --------------------------------------------------------------------------
# K14  counter-based PRNG (our choice; the TF stream is parity-unpinned)
# --------------------------------------------------------------------------
_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al. 2011).  All args uint32 arrays/scalars."""
    c0 = np.asarray(c0, dtype=np.uint32).copy()
    c1 = np.broadcast_to(np.asarray(c1, dtype=np.uint32), c0.shape).copy()
    c2 = np.broadcast_to(np.asarray(c2, dtype=np.uint32), c0.shape).copy()
    c3 = np.broadcast_to(np.asarray(c3, dtype=np.uint32), c0.shape).copy()
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PHILOX_M0 * c0.astype(np.uint64)
            p1 = _PHILOX_M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = p0.astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + _PHILOX_W0)
            k1 = np.uint32(k1 + _PHILOX_W1)
    return c0, c1, c2, c3


def philox_corruption_draws(seed, counter, n_rows, n_choices):
    """Draws for ``n_rows`` corruption rows: (keep_subject_mask in {0,1}, replacement index
    in [0, n_choices)).  Row j uses Philox counter (j_lo, j_hi, counter_lo, counter_hi), key
    (seed_lo, seed_hi); mask = out0 & 1; index = mulhi64((out2<<32)|out1, n_choices).
    Mirrored bit-for-bit by emgraph_amd/csrc/emg_common.hpp::philox4x32_10 / corruption_draw and by oracle/emg_oracle.c."""
    j = np.arange(n_rows, dtype=np.uint64)
    seed = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    counter = np.uint64(counter & 0xFFFFFFFFFFFFFFFF)
    o0, o1, o2, _ = philox4x32_10(
        (j & np.uint64(0xFFFFFFFF)).astype(np.uint32),
        (j >> np.uint64(32)).astype(np.uint32),
        np.uint32(counter & np.uint64(0xFFFFFFFF)),
        np.uint32(counter >> np.uint64(32)),
        np.uint32(seed & np.uint64(0xFFFFFFFF)),
        np.uint32(seed >> np.uint64(32)),
    )
    mask = (o0 & np.uint32(1)).astype(I32)
    r64 = (o2.astype(np.uint64) << np.uint64(32)) | o1.astype(np.uint64)
    # mulhi64(r64, n) with python ints (exact)
    n = int(n_choices)
    idx = np.array([(int(v) * n) >> 64 for v in r64.tolist()], dtype=np.int64) if n_rows else np.zeros(0, np.int64)
    return mask, idx.astype(I32)


# --------------------------------------------------------------------------
# a6  generate_corruptions_for_fit  (protocol.py:531-659)
# --------------------------------------------------------------------------
def batch_unique_entities(X):
    """protocol.py:621-633: tf.unique(concat(s_col, o_col)) — FIRST-APPEARANCE order."""
    cat = np.concatenate([X[:, 0], X[:, 2]])
    _, first = np.unique(cat, return_index=True)
    return cat[np.sort(first)].astype(I32)


def generate_corruptions_for_fit(X, entities_list=None, eta=1, corrupt_side="s,o", entities_size=0,
                                 mask_draw=None, repl_draw=None):
    """protocol.py:531-659 with the two ``tf.random.uniform`` draws INJECTED.

    ``mask_draw``  int[B*eta] in {0,1}  (only consumed for 's+o' / 's,o', :600-604; drawn FIRST)
    ``repl_draw``  int[B*eta]: replacement id in [0,entities_size) (:616-619) or an index into
                   ``entities_list`` / the batch-unique list (:620-641).
    Output row j corrupts positive j mod B (eta-major tiling, :598)."""
    X = np.asarray(X, dtype=I32)
    if corrupt_side == "s,o":
        corrupt_side = "s+o"
    if corrupt_side not in ("s+o", "s", "o"):
        raise ValueError("Invalid argument value {} for corruption side passed for evaluation.".format(corrupt_side))
    B = X.shape[0]
    dataset = np.tile(X.reshape(-1), eta).reshape(B * eta, 3)  # :598
    if corrupt_side == "s+o":
        keep_subj_mask = np.asarray(mask_draw).astype(bool)  # :600-604
        assert keep_subj_mask.shape == (B * eta,)
    else:
        keep_subj_mask = np.ones(B * eta, dtype=bool)  # :606
        if corrupt_side == "s":
            keep_subj_mask = ~keep_subj_mask  # :607-608
    keep_obj_mask = (~keep_subj_mask).astype(I32)
    keep_subj_mask = keep_subj_mask.astype(I32)
    repl_draw = np.asarray(repl_draw, dtype=I32)
    assert repl_draw.shape == (B * eta,)
    if entities_size != 0:
        replacements = repl_draw  # :616-619
    else:
        if entities_list is None:
            entities_list = batch_unique_entities(X)  # :621-633
        entities_list = np.asarray(entities_list, dtype=I32)
        replacements = entities_list[repl_draw]  # :635-641
    subjects = keep_subj_mask * dataset[:, 0] + keep_obj_mask * replacements  # :643-646
    relationships = dataset[:, 1]
    objects = keep_obj_mask * dataset[:, 2] + keep_subj_mask * replacements  # :650-653
    return np.stack([subjects, relationships, objects]).T.astype(I32)  # :656


def n_corruption_choices(X, entities_list, entities_size):
    if entities_size != 0:
        return int(entities_size)
    if entities_list is None:
        return len(batch_unique_entities(np.asarray(X)))
    return len(entities_list)


def generate_corruptions_for_fit_philox(X, entities_list=None, eta=1, corrupt_side="s,o", entities_size=0,
                                        seed=0, counter=0):
    """Same generator with the draws taken from our Philox stream (what the HIP kernel does)."""
    X = np.asarray(X, dtype=I32)
    n = X.shape[0] * eta
    mask, idx = philox_corruption_draws(seed, counter, n, n_corruption_choices(X, entities_list, entities_size))
    return generate_corruptions_for_fit(X, entities_list, eta, corrupt_side, entities_size, mask, idx)


# --------------------------------------------------------------------------
# a10  generate_corruptions_for_eval  (protocol.py:448-528)
# --------------------------------------------------------------------------
def generate_corruptions_for_eval(X, entities_for_corruption, corrupt_side="s,o"):
    """For one triple x and candidate list C: 'o' -> (s,p,c_i); 's' -> (c_i,p,o);
    's+o'/'s,o' -> object block FIRST then subject block (:511-518)."""
    X = np.asarray(X).reshape(1, 3)
    C = np.asarray(entities_for_corruption).reshape(-1)
    if corrupt_side == "s,o":
        corrupt_side = "s+o"
    if corrupt_side not in ("s+o", "s", "o"):
        raise ValueError("Invalid argument value for corruption side passed for evaluation")
    n = len(C)
    s = np.full(n, X[0, 0], dtype=C.dtype)
    p = np.full(n, X[0, 1], dtype=C.dtype)
    o = np.full(n, X[0, 2], dtype=C.dtype)
    obj_block = np.stack([s, p, C], axis=1)
    subj_block = np.stack([C, p, o], axis=1)
    if corrupt_side == "s+o":
        return np.concatenate([obj_block, subj_block], axis=0)
    if corrupt_side == "o":
        return obj_block
    return subj_block


# --------------------------------------------------------------------------
# a8, a9, K5c-e  losses  (losses/*.py).  pos is ALREADY tiled to [B*eta] by the caller
# when the loss requires same sizes (EmbeddingModel.py:724-729, losses/utils.py:24-32).
# --------------------------------------------------------------------------
REQUIRE_SAME_SIZE = {"pairwise": True, "nll": True, "absolute_margin": True,
                     "self_adversarial": False, "multiclass_nll": False}


def clip_before_exp(v):
    """losses/utils.py:44-53."""
    return np.clip(v, F32(DEFAULT_CLIP_EXP_LOWER), F32(DEFAULT_CLIP_EXP_UPPER))


def loss_pairwise(scores_pos, scores_neg, margin=DEFAULT_MARGIN):
    """pairwise.py:66-70."""
    return np.sum(np.maximum(F32(margin) - scores_pos + scores_neg, F32(0)), dtype=F32)


def loss_nll(scores_pos, scores_neg):
    """nll.py:55-59 — naive log(1+exp(x)) after +-75 clip; positives counted eta times."""
    scores_neg = clip_before_exp(scores_neg)
    scores_pos = clip_before_exp(scores_pos)
    scores = np.concatenate([-scores_pos, scores_neg], 0)
    return np.sum(np.log(F32(1) + np.exp(scores)), dtype=F32)


def loss_absolute_margin(scores_pos, scores_neg, margin=DEFAULT_MARGIN):
    """absolute_margin.py:66-70: reduce_sum(max(margin + neg, 0) - pos)."""
    return np.sum(np.maximum(F32(margin) + scores_neg, F32(0)) - scores_pos, dtype=F32)


def _log_sigmoid(x):
    return (-np.logaddexp(F32(0), -x)).astype(F32)


def _softmax0(x):
    m = np.max(x, axis=0, keepdims=True)
    e = np.exp(x - m)
    return (e / np.sum(e, axis=0, keepdims=True)).astype(F32)


def loss_self_adversarial(scores_pos, scores_neg, eta, margin=DEFAULT_MARGIN_ADVERSARIAL,
                          alpha=DEFAULT_ALPHA_ADVERSARIAL):
    """self_adversarial.py:90-112.  pos [B]; neg [eta*B] reshaped [eta, B] (eta-major)."""
    B = scores_pos.shape[0]
    scores_neg_reshaped = scores_neg.reshape(eta, B)
    p_neg = _softmax0(F32(alpha) * scores_neg_reshaped)
    loss = np.sum(-_log_sigmoid(F32(margin) + scores_pos), dtype=F32) - np.sum(
        p_neg * _log_sigmoid(-scores_neg_reshaped - F32(margin)), dtype=F32)
    return F32(loss)


def loss_multiclass_nll(scores_pos, scores_neg, eta):
    """nll_multiclass.py:70-81: -sum log( e^pos / (sum_eta e^neg + e^pos) ) after +-75 clip."""
    scores_neg = clip_before_exp(scores_neg)
    scores_pos = clip_before_exp(scores_pos)
    B = scores_pos.shape[0]
    neg_exp = np.exp(scores_neg.reshape(eta, B))
    pos_exp = np.exp(scores_pos)
    softmax_score = pos_exp / (np.sum(neg_exp, axis=0, dtype=F32) + pos_exp)
    return F32(-np.sum(np.log(softmax_score), dtype=F32))


def loss_apply(name, scores_pos, scores_neg, eta, params=None):
    """Loss.apply (loss.py:123-138) incl. the same-size assert (:81-105)."""
    params = params or {}
    if REQUIRE_SAME_SIZE[name] and eta != 1:
        assert scores_pos.shape[0] == scores_neg.shape[0]
    if name == "pairwise":
        return loss_pairwise(scores_pos, scores_neg, params.get("margin", DEFAULT_MARGIN))
    if name == "nll":
        return loss_nll(scores_pos, scores_neg)
    if name == "absolute_margin":
        return loss_absolute_margin(scores_pos, scores_neg, params.get("margin", DEFAULT_MARGIN))
    if name == "self_adversarial":
        return loss_self_adversarial(scores_pos, scores_neg, eta,
                                     params.get("margin", DEFAULT_MARGIN_ADVERSARIAL),
                                     params.get("alpha", DEFAULT_ALPHA_ADVERSARIAL))
    if name == "multiclass_nll":
        return loss_multiclass_nll(scores_pos, scores_neg, eta)
    raise ValueError(name)


def loss_grads(name, pos, neg, eta, params=None):
    """Analytic dL/dpos [B], dL/dneg [B*eta] of ``loss_apply(name, tile(pos), neg)`` (TF autodiff
    equivalent; checked against float64 finite differences in tests/test_oracle_golden.py).
    ``pos`` is the UN-tiled [B] vector; the eta-fold tiling's adjoint (sum over tiles) is included."""
    params = params or {}
    pos = pos.astype(np.float64)
    neg = neg.astype(np.float64)
    B = pos.shape[0]
    negr = neg.reshape(eta, B)
    if name == "pairwise":
        # tf.maximum's gradient goes to x where x >= y (MaximumGrad uses greater_equal)
        act = ((params.get("margin", DEFAULT_MARGIN) - pos[None, :] + negr) >= 0).astype(np.float64)
        return (-act.sum(0)).astype(F32), act.reshape(-1).astype(F32)
    if name == "nll":
        lo, hi = DEFAULT_CLIP_EXP_LOWER, DEFAULT_CLIP_EXP_UPPER
        inp = ((pos >= lo) & (pos <= hi)).astype(np.float64)  # clip_by_value passes grad inside [lo,hi]
        inn = ((negr >= lo) & (negr <= hi)).astype(np.float64)
        gp = -eta * inp / (1.0 + np.exp(np.clip(pos, lo, hi)))
        gn = inn / (1.0 + np.exp(-np.clip(negr, lo, hi)))
        return gp.astype(F32), gn.reshape(-1).astype(F32)
    if name == "absolute_margin":
        act = ((params.get("margin", DEFAULT_MARGIN) + negr) >= 0).astype(np.float64)
        return np.full(B, -float(eta), dtype=F32), act.reshape(-1).astype(F32)
    if name == "self_adversarial":
        m = params.get("margin", DEFAULT_MARGIN_ADVERSARIAL)
        a = params.get("alpha", DEFAULT_ALPHA_ADVERSARIAL)
        gp = -1.0 / (1.0 + np.exp(m + pos))
        z = a * negr
        w = np.exp(z - z.max(0, keepdims=True))
        w = w / w.sum(0, keepdims=True)
        ell = -np.logaddexp(0.0, negr + m)  # log_sigmoid(-neg - m)
        dell = -1.0 / (1.0 + np.exp(-(negr + m)))
        # L_neg = -sum_j w_j ell_j ; softmax NOT stop-gradiented (self_adversarial.py:98-110)
        s = (w * ell).sum(0, keepdims=True)
        gn = -(w * dell + a * w * (ell - s))
        return gp.astype(F32), gn.reshape(-1).astype(F32)
    if name == "multiclass_nll":
        lo, hi = DEFAULT_CLIP_EXP_LOWER, DEFAULT_CLIP_EXP_UPPER
        inp = ((pos >= lo) & (pos <= hi)).astype(np.float64)
        inn = ((negr >= lo) & (negr <= hi)).astype(np.float64)
        ep = np.exp(np.clip(pos, lo, hi))
        en = np.exp(np.clip(negr, lo, hi))
        den = en.sum(0) + ep
        gp = -(1.0 - ep / den) * inp
        gn = (en / den[None, :]) * inn
        return gp.astype(F32), gn.reshape(-1).astype(F32)
    raise ValueError(name)


# --------------------------------------------------------------------------
# a15  LPRegularizer._apply  (regularizers/lp.py:81-113)
# --------------------------------------------------------------------------
def lp_regularizer(trainable_params, lam=1e-5, p=2):
    """sum_i lambda_i * sum(|W_i|^p); scalar lambda is broadcast to every param (lp.py:95-113)."""
    if np.isscalar(lam):
        lam = [lam] * len(trainable_params)
    loss_reg = F32(0)
    for i, w in enumerate(trainable_params):
        loss_reg = loss_reg + F32(lam[i]) * np.sum(np.power(np.abs(w), p), dtype=F32)
    return F32(loss_reg)


# --------------------------------------------------------------------------
# a7  _get_model_loss assembly  (EmbeddingModel.py:614-822)
# --------------------------------------------------------------------------
def model_loss(model, ent_emb, rel_emb, x_pos, eta, loss="nll", loss_params=None, corrupt_sides=("s,o",),
               x_negs=None, regularizer=None, k=None):
    """pos score -> tile pos x eta if the loss requires same size -> per side: score the
    (given) corruptions, loss += Loss.apply -> + LP over the FULL tables.
    ``x_negs``: list (one per side) of int[B*eta,3] corruption arrays (produced by
    generate_corruptions_for_fit*)."""
    scores_pos = score_triples(model, ent_emb, rel_emb, x_pos, k=k)
    if REQUIRE_SAME_SIZE[loss]:
        scores_pos_in = np.tile(scores_pos, eta)  # :724-729
    else:
        scores_pos_in = scores_pos
    total = F32(0)
    per_side_neg = []
    for side, x_neg in zip(corrupt_sides, x_negs):
        scores_neg = score_triples(model, ent_emb, rel_emb, x_neg, k=k)
        per_side_neg.append(scores_neg)
        total = F32(total + loss_apply(loss, scores_pos_in, scores_neg, eta, loss_params))  # :816
    if regularizer is not None:
        total = F32(total + lp_regularizer([ent_emb, rel_emb], **regularizer))  # :818-820
    return total, scores_pos, per_side_neg


def score_grad_rows(model, ent_emb, rel_emb, x, g, k=None):
    """Adjoint of _fn per triple: given g = dL/dscore [n] return the float64 gradient rows (d/de_s, d/de_p, d/de_o), each
    [n, k_int] — the IndexedSlices TF autodiff hands the optimizer, before densification.  Analytic per model."""
    x = np.asarray(x)
    e_s, e_p, e_o = [a.astype(np.float64) for a in lookup_embeddings(ent_emb, rel_emb, x)]
    g = np.asarray(g, dtype=np.float64)[:, None]
    if model in ("TransE", "TransE_L1"):
        sg = np.sign((e_s + e_p) - e_o)
        gs, gp, go = -g * sg, -g * sg, g * sg
    elif model == "TransE_L2":
        d = (e_s + e_p) - e_o
        nrm = np.sqrt((d * d).sum(1, keepdims=True))
        u = np.divide(d, nrm, out=np.zeros_like(d), where=nrm > 0)
        gs, gp, go = -g * u, -g * u, g * u
    elif model.startswith("TransE_P:"):
        # tf.norm(ord) = reduce_sum(|d|^ord)^(1/ord) (ord = inf: reduce_max |d|, whose gradient tied maxima share): autodiff gives
        # d||d||/dd = sgn(d) |d|^(ord-1) / ||d||^(ord-1); a zero vector gets gradient zero here
        order = float(model.split(":")[1])
        d = (e_s + e_p) - e_o
        ad = np.abs(d)
        if np.isinf(order):
            top = ad.max(1, keepdims=True)
            hit = (ad == top) & (top > 0)
            u = np.sign(d) * hit / np.maximum(hit.sum(1, keepdims=True), 1)
        else:
            nrm = (ad ** order).sum(1, keepdims=True) ** (1.0 / order)
            with np.errstate(divide="ignore", invalid="ignore"):
                u = np.where((nrm > 0) & (ad > 0), np.sign(d) * ad ** (order - 1) / nrm ** (order - 1), 0.0)
        gs, gp, go = -g * u, -g * u, g * u
    elif model == "DistMult":
        gs, gp, go = g * e_p * e_o, g * e_s * e_o, g * e_s * e_p
    elif model in ("ComplEx", "HolE"):
        if model == "HolE":
            kk = k if k is not None else e_s.shape[1] // 2
            g = g * float(F32(2 / kk))
        sr, si = np.split(e_s, 2, axis=1)
        pr, pi = np.split(e_p, 2, axis=1)
        orr, oi = np.split(e_o, 2, axis=1)
        gs = g * np.concatenate([pr * orr + pi * oi, pr * oi - pi * orr], 1)
        gp = g * np.concatenate([sr * orr + si * oi, sr * oi - si * orr], 1)
        go = g * np.concatenate([pr * sr - pi * si, pr * si + pi * sr], 1)
    else:
        raise ValueError(model)
    return gs, gp, go


def score_grads(model, ent_emb, rel_emb, x, g, k=None):
    """Adjoint of lookup+_fn: given g = dL/dscore [n] return dense (dE, dR) float64 accumulations
    (what TF autodiff + IndexedSlices densification would give)."""
    x = np.asarray(x)
    gs, gp, go = score_grad_rows(model, ent_emb, rel_emb, x, g, k=k)
    dE = np.zeros(ent_emb.shape, dtype=np.float64)
    dR = np.zeros(rel_emb.shape, dtype=np.float64)
    np.add.at(dE, x[:, 0], gs)
    np.add.at(dE, x[:, 2], go)
    np.add.at(dR, x[:, 1], gp)
    return dE, dR


def train_grads(model, ent_emb, rel_emb, x_pos, eta, loss, loss_params, x_negs, k=None):
    """Dense gradients of model_loss (without regulariser) wrt the two tables."""
    scores_pos = score_triples(model, ent_emb, rel_emb, x_pos, k=k)
    dE = np.zeros(ent_emb.shape, dtype=np.float64)
    dR = np.zeros(rel_emb.shape, dtype=np.float64)
    for x_neg in x_negs:
        scores_neg = score_triples(model, ent_emb, rel_emb, x_neg, k=k)
        gp, gn = loss_grads(loss, scores_pos, scores_neg, eta, loss_params)
        for xx, gg in ((x_pos, gp), (x_neg, gn)):
            a, b = score_grads(model, ent_emb, rel_emb, xx, gg, k=k)
            dE += a
            dR += b
    return dE, dR


def train_grads_sparse(model, ent_emb, rel_emb, x_pos, eta, loss, loss_params, x_negs, k=None, chunk=65536):
    """train_grads for tables too large to densify a float64 gradient for: returns (entity ids, their summed float64 gradient
    rows, relation ids, their rows, data loss) over the rows the batch touches — the same per-triple rows, grouped by destination
    (triples taken ``chunk`` at a time: float64 rows of 344 k triples x 400 columns are 1 GB apiece)."""
    x_pos = np.asarray(x_pos)
    x_negs = [np.asarray(x) for x in x_negs]
    scores_pos = score_triples(model, ent_emb, rel_emb, x_pos, k=k)
    ue = np.unique(np.concatenate([x_pos[:, 0], x_pos[:, 2]] + [x[:, 0] for x in x_negs] + [x[:, 2] for x in x_negs]))
    ur = np.unique(x_pos[:, 1])
    ge = np.zeros((len(ue), ent_emb.shape[1]), dtype=np.float64)
    gr = np.zeros((len(ur), rel_emb.shape[1]), dtype=np.float64)
    total = F32(0.0)
    for x_neg in x_negs:
        scores_neg = np.concatenate([score_triples(model, ent_emb, rel_emb, x_neg[c0:c0 + chunk], k=k)
                                     for c0 in range(0, len(x_neg), chunk)])
        pos_in = np.tile(scores_pos, eta) if REQUIRE_SAME_SIZE[loss] else scores_pos
        total = F32(total + loss_apply(loss, pos_in, scores_neg, eta, loss_params))
        gp, gn = loss_grads(loss, scores_pos, scores_neg, eta, loss_params)
        for xx, gg in ((x_pos, gp), (x_neg, gn)):
            for c0 in range(0, len(xx), chunk):
                xc = xx[c0:c0 + chunk]
                gs, gpp, go = score_grad_rows(model, ent_emb, rel_emb, xc, gg[c0:c0 + chunk], k=k)
                np.add.at(ge, np.searchsorted(ue, xc[:, 0]), gs)
                np.add.at(ge, np.searchsorted(ue, xc[:, 2]), go)
                np.add.at(gr, np.searchsorted(ur, xc[:, 1]), gpp)
    return ue, ge, ur, gr, float(total)


# --------------------------------------------------------------------------
# a16  optimizers — published Keras (TF 2.2) update rules.  PARITY UNPINNED (no reference
# test pins a post-update value; the rules live in tensorflow~=2.2.3, requirements/default.txt:11).
# --------------------------------------------------------------------------
def opt_init(name, shape):
    if name == "sgd":
        return {}
    if name == "momentum":
        return {"m": np.zeros(shape, F32)}
    if name == "adagrad":
        return {"acc": np.full(shape, 0.1, F32)}  # Keras initial_accumulator_value=0.1
    if name == "adam":
        return {"m": np.zeros(shape, F32), "v": np.zeros(shape, F32), "t": 0}
    raise ValueError(name)


def opt_apply(name, w, g, state, lr=0.0005, momentum=0.9, beta1=0.9, beta2=0.999, eps=1e-7, touched=None):
    """One dense-equivalent step.  ``touched`` (bool rows) restricts SGD/momentum/Adagrad to the
    IndexedSlices rows (Keras sparse apply touches only those); Adam's Keras sparse apply is
    dense-equivalent (decays every row's m, v and updates every row)."""
    g = g.astype(F32)
    w = w.astype(F32).copy()
    rows = slice(None) if touched is None else touched
    if name == "sgd":
        w[rows] = w[rows] - F32(lr) * g[rows]
    elif name == "momentum":
        m = state["m"]
        m[rows] = F32(momentum) * m[rows] - F32(lr) * g[rows]  # keras SGD(momentum): v = mu*v - lr*g ; w += v
        w[rows] = w[rows] + m[rows]
    elif name == "adagrad":
        acc = state["acc"]
        acc[rows] = acc[rows] + g[rows] * g[rows]
        w[rows] = w[rows] - F32(lr) * g[rows] / (np.sqrt(acc[rows]) + F32(eps))
    elif name == "adam":
        state["t"] += 1
        t = state["t"]
        lr_t = F32(lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t))
        state["m"][:] = F32(beta1) * state["m"] + F32(1 - beta1) * g
        state["v"][:] = F32(beta2) * state["v"] + F32(1 - beta2) * g * g
        w = w - lr_t * state["m"] / (np.sqrt(state["v"]) + F32(eps))
    else:
        raise ValueError(name)
    return w.astype(F32)


# --------------------------------------------------------------------------
# a12  perform_comparision  (EmbeddingModel.py:1989-2033)
# --------------------------------------------------------------------------
def to_cmp_int(score):
    """tf.cast(score * 1e5, tf.int32): f32 multiply then truncation toward zero (:2010-2014)."""
    return (np.asarray(score, dtype=F32) * F32(SCORE_COMPARISON_PRECISION)).astype(I32)


def perform_comparison(score_corr, score_pos, strategy="worst"):
    c = to_cmp_int(score_corr)
    p = to_cmp_int(score_pos)
    if strategy == "best":
        return int(np.sum(c > p))
    if strategy == "middle":
        return int(np.sum(c > p)) + int(np.ceil(np.sum(c == p) / 2))
    if strategy == "worst":
        return int(np.sum(c >= p))
    raise AssertionError("Invalid score comparision type!")


# --------------------------------------------------------------------------
# a14  SQLiteAdapter.get_participating_entities  (sqlite_adapter.py:449-508)
# --------------------------------------------------------------------------
def participating_entities(filter_triples, x):
    """({o} U {o' : (s,p,o') in F},  {s} U {s' : (s',p,o) in F}) — 'select <id> union select
    distinct ...' => the test entity is always included, result de-duplicated (SQL UNION)."""
    F = np.asarray(filter_triples)
    s, p, o = int(x[0]), int(x[1]), int(x[2])
    objs = set(F[(F[:, 0] == s) & (F[:, 1] == p), 2].tolist()) | {o}
    subs = set(F[(F[:, 1] == p) & (F[:, 2] == o), 0].tolist()) | {s}
    return np.array(sorted(objs), dtype=np.int64), np.array(sorted(subs), dtype=np.int64)


# --------------------------------------------------------------------------
# a11, a13  per-test-triple ranking  (EmbeddingModel.py:1845-1986)
# --------------------------------------------------------------------------
def rank_triple(model, ent_emb, rel_emb, x, corrupt_side="s,o", strategy="worst",
                corruption_entities=None, filter_triples=None, k=None, score_override=None):
    """Intended per-triple semantics of _initialize_eval_graph + get_ranks (see SURVEY A-1).
    Returns int rank ('s','o','s+o') or [rank_s, rank_o] ('s,o').
    ``corruption_entities`` None -> all entity ids 0..|E|-1 (:1847-1850).
    With a subset, filter ids not in the subset are dropped and the rest mapped to subset
    positions (:1898-1940, intended semantics; SURVEY A-7).
    ``score_override(x_rows)->scores`` lets tests plug a different-but-equivalent scorer."""
    x = np.asarray(x).reshape(1, 3)
    n_ent = ent_emb.shape[0]
    if corruption_entities is None:
        C = np.arange(n_ent, dtype=np.int64)
    else:
        C = np.asarray(corruption_entities, dtype=np.int64).reshape(-1)
    scorer = score_override or (lambda rows: score_triples(model, ent_emb, rel_emb, rows, k=k))
    out_corr = generate_corruptions_for_eval(x, C, corrupt_side)  # :1856-1858
    scores_predict = scorer(out_corr)  # :1861-1862
    score_positive = scorer(x)[0]  # :1865-1866
    nC = len(C)
    if corrupt_side == "s,o":
        obj_scores = scores_predict[:nC]  # :1883-1892
        subj_scores = scores_predict[nC:]
    pos_obj_higher = 0
    pos_sub_higher = 0
    if filter_triples is not None:
        indices_obj, indices_sub = participating_entities(filter_triples, x[0])
        if corruption_entities is not None:  # remap ids -> positions in C; drop absentees
            pos_of = {int(e): i for i, e in enumerate(C.tolist())}
            indices_obj = np.array([pos_of[i] for i in indices_obj.tolist() if i in pos_of], dtype=np.int64)
            indices_sub = np.array([pos_of[i] for i in indices_sub.tolist() if i in pos_of], dtype=np.int64)
        if corrupt_side == "s,o":
            sp_obj = obj_scores[indices_obj]
            sp_sub = subj_scores[indices_sub]
        else:
            sp_obj = scores_predict[indices_obj]
            if corrupt_side == "s+o":
                sp_sub = scores_predict[indices_sub + nC]  # :1950-1953
            else:
                sp_sub = scores_predict[indices_sub]
        if "o" in corrupt_side:
            pos_obj_higher = perform_comparison(sp_obj, score_positive, strategy)
        if "s" in corrupt_side:
            pos_sub_higher = perform_comparison(sp_sub, score_positive, strategy)
    if corrupt_side == "s,o":
        return [perform_comparison(subj_scores, score_positive, strategy) + 1 - pos_sub_higher,
                perform_comparison(obj_scores, score_positive, strategy) + 1 - pos_obj_higher]
    return perform_comparison(scores_predict, score_positive, strategy) + 1 - pos_sub_higher - pos_obj_higher


def get_ranks(model, ent_emb, rel_emb, X_test, **kw):
    """get_ranks loop (EmbeddingModel.py:2084-2097), intended per-triple re-evaluation."""
    return np.array([rank_triple(model, ent_emb, rel_emb, x, **kw) for x in np.asarray(X_test)])


# --------------------------------------------------------------------------
# a17  metrics  (evaluation/metrics.py:62-67,125-130,217-222): ranks are flattened first
# (``ranks.reshape(-1)``), so [n,2] 's,o' ranks average over 2n entries (SURVEY A-10 is wrong
# about this; the generated fixtures caught it).
# --------------------------------------------------------------------------
def hits_at_n_score(ranks, n):
    ranks = np.asarray(ranks).reshape(-1)
    return np.sum(ranks <= n) / len(ranks)


def mrr_score(ranks):
    ranks = np.asarray(ranks).reshape(-1)
    return np.sum(1 / ranks) / len(ranks)


def mr_score(ranks):
    ranks = np.asarray(ranks).reshape(-1)
    return np.sum(ranks) / len(ranks)


def rank_score(y_true, y_pred, pos_lab=1):
    """metrics.py:167-219."""
    idx = np.argsort(y_pred)[::-1]
    y_ord = np.asarray(y_true)[idx]
    return int(np.where(y_ord == pos_lab)[0][0] + 1)


# --------------------------------------------------------------------------
# caller side: mappings and batching  (protocol.py:429-445,662-723; numpy_adapter.py:105-131)
# --------------------------------------------------------------------------
def create_mappings(X):
    """np.unique order = id (protocol.py:443-445, _create_unique_mappings)."""
    unique_ent = np.unique(np.concatenate((X[:, 0], X[:, 2])))
    unique_rel = np.unique(X[:, 1])
    ent_to_idx = dict(zip(unique_ent, range(len(unique_ent))))
    rel_to_idx = dict(zip(unique_rel, range(len(unique_rel))))
    return rel_to_idx, ent_to_idx


def to_idx(X, ent_to_idx, rel_to_idx):
    """protocol.py:662-723; unseen -> ValueError (:684-701)."""
    try:
        x_idx_s = np.vectorize(ent_to_idx.__getitem__)(X[:, 0])
        x_idx_p = np.vectorize(rel_to_idx.__getitem__)(X[:, 1])
        x_idx_o = np.vectorize(ent_to_idx.__getitem__)(X[:, 2])
    except KeyError as e:
        raise ValueError("Input triples include one or more concepts not present in the training set: %s" % e)
    return np.dstack([x_idx_s, x_idx_p, x_idx_o]).reshape((-1, 3))


def batches(X_idx, batches_count):
    """numpy_adapter.py:105-112: B=ceil(n/batches_count); contiguous unshuffled slices; the last
    may be short (or empty)."""
    n = X_idx.shape[0]
    bs = int(np.ceil(n / batches_count))
    for i in range(batches_count):
        yield np.int32(X_idx[i * bs:(i + 1) * bs, :])
