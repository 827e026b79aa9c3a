"""GPU parity of the DEFAULT product path at the widths and optimizers of BASELINE.json's configurations.

The training step `fit()` runs is: emg_prepare_batch (Philox codes, destination grouping, singleton flags) ->
emg_train_backward_ex(fused_loss >= 0, single_ent != NULL)  [scores + pair-local loss + gradients in one pass,
singleton destinations updated IN PLACE with the optimizer rule] -> emg_apply_grouped(skip_single = 1) for the
entity table and emg_apply_grouped for the relation table.  Round 1 compared that path with the oracle only at
k <= 12; here it is driven through the C-ABI at

    C1  TransE-L1 k=100 eta=20 pairwise, Adam   (train_backward_kernel<TransE, 4, 1, 32, true, 2>)
    C2  DistMult  k=200 eta=10 NLL,      Adam   (<DistMult, 4, 1, 64, true, 2>)
    C5  HolE      k=200 eta=20 NLL,      Adam   (<HolE, 4, 1, 64, true, 2>)
    C3  ComplEx   k=200 eta=20 NLL,      SGD    (<ComplEx, 4, 1, 64, true, 1>)
    (+ Adagrad / momentum once each: the other two state layouts of the in-place path)

against oracle/emgraph_oracle.py::train_grads + opt_apply (EmbeddingModel.py:614-822, losses/pairwise.py:66-70,
nll.py:55-59, training/*.py) over TWO consecutive steps (the second one reads optimizer state written by the first),
on a UNIFORM batch over 20 000 entities (most destinations are singletons -> in-place path) and on a ZIPF(1.0)
batch over 2 000 entities (hot rows, long segments, hardly any singleton -> segmented apply path).

Tolerances (fp32 path vs a float64-accumulating oracle): loss rtol 2e-5; gradients — read back through the
optimizer state where one exists (Adam m = 0.1 g after step 1, momentum buffer, Adagrad accumulator) and through the
SGD update otherwise — rtol 1e-4 with an absolute floor of 1e-5 x the largest gradient entry; tables after Adam /
Adagrad within 1e-2 x lr, a secondary check (a gradient entry inside fp32 noise of zero may take either sign, and
the first Adam steps move every entry by ~lr x g / (|g| + 3e-6): measured worst case 2.05e-3 x lr on 1 of 800 000
entries); rows no kernel may touch stay BIT-identical.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import c_oracle as co  # noqa: E402
from oracle import emgraph_oracle as orc  # noqa: E402

F32 = np.float32
MID = orc.MODEL_IDS
B1, B2, EPS = 0.9, 0.999, 1e-7

CONFIGS = {
    # name: model, k, eta, loss, optimizer, n_rel
    "C1": ("TransE_L1", 100, 20, "pairwise", "adam", 11),
    "C2": ("DistMult", 200, 10, "nll", "adam", 237),
    "C5": ("HolE", 200, 20, "nll", "adam", 1345),
    "C3": ("ComplEx", 200, 20, "nll", "sgd", 1000),
    "C2-adagrad": ("DistMult", 200, 10, "nll", "adagrad", 237),
    "C1-momentum": ("TransE_L1", 100, 20, "pairwise", "momentum", 11),
}


def _batch(kind, n_ent, n_rel, n, seed):
    rs = np.random.RandomState(seed)
    if kind == "zipf":   # Zipf(1.0) over the entity ids: a few hub entities take most of the slots
        w = 1.0 / np.arange(1, n_ent + 1)
        s, o = (rs.choice(n_ent, n, p=w / w.sum()) for _ in range(2))
    else:
        s, o = rs.randint(0, n_ent, n), rs.randint(0, n_ent, n)
    return np.stack([s, rs.randint(0, n_rel, n), o], 1).astype(np.int32)


def _hyper(lr, step):
    return (lr, 0.9, B1, B2, EPS, lr * np.sqrt(1.0 - B2 ** step) / (1.0 - B1 ** step))


@pytest.mark.parametrize("kind", ["uniform", "zipf"])
@pytest.mark.parametrize("cfg", list(CONFIGS))
def test_fused_inplace_step_vs_oracle_at_config_width(cfg, kind):
    _check_step_vs_oracle(cfg, kind, 20000 if kind == "uniform" else 2000, 512)


# BASELINE.json's configurations at their EXACT shapes (C1 WN11 |E| = 38 600 (literature) / 11 relations,
# B = ceil(110361 / 64) = 1725; C2 FB15k-237 14 541 / 237, B = ceil(272115 / 100) = 2722; C5 FB15k 14 951 / 1 345,
# B = ceil(483142 / 100) = 4832), eta and optimizer as configured: the default product step against the oracle
@pytest.mark.parametrize("cfg,n_ent,B", [("C1", 38600, 1725), ("C2", 14541, 2722), ("C5", 14951, 4832)])
def test_fused_inplace_step_vs_oracle_at_baseline_shape(cfg, n_ent, B):
    _check_step_vs_oracle(cfg, "uniform", n_ent, B, expect_singletons=None)


def _check_step_vs_oracle(cfg, kind, n_ent, B, expect_singletons="by kind"):
    from emgraph_amd import _lib as L
    from emgraph_amd import device as d
    from emgraph_amd.training import Trainer, alloc_table
    d.require_gpu()
    model, k, eta, loss, opt, n_rel = CONFIGS[cfg]
    seed = 3
    lr = 0.1 if opt == "sgd" else 0.01     # SGD: a large step keeps the table's own fp32 rounding below the 1e-4 bar
    ki = 2 * k if model in ("ComplEx", "HolE") else k
    sc = float(F32(2 / k)) if model == "HolE" else 1.0
    rs = np.random.RandomState(sum(map(ord, cfg + kind)))
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = _batch(kind, n_ent, n_rel, 2 * B, seed=17)
    dev = torch.device("cuda")

    # ---------------- device: the C-ABI sequence fit() issues ----------------
    Et, Rt = alloc_table(n_ent, ki, dev, init=E0), alloc_table(n_rel, ki, dev, init=R0)
    ns = {"sgd": 0, "momentum": 1, "adagrad": 1, "adam": 2}[opt]
    fill = 0.1 if opt == "adagrad" else 0.0
    se = [alloc_table(n_ent, ki, dev, fill=fill) for _ in range(ns)] + [None] * (2 - ns)
    sr = [alloc_table(n_rel, ki, dev, fill=fill) for _ in range(ns)] + [None] * (2 - ns)
    tag_e = torch.zeros(n_ent, dtype=torch.int32, device=dev)
    tag_r = torch.zeros(n_rel, dtype=torch.int32, device=dev)
    n_ce = (2 + eta) * B
    we = torch.empty(d.apply_workspace_bytes(n_ce, n_ent, ki), dtype=torch.uint8, device=dev)   # incl. long-segment scratch
    wr = torch.empty(d.apply_workspace_bytes(B, n_rel, ki), dtype=torch.uint8, device=dev)
    codes = torch.empty(B * eta, dtype=torch.int32, device=dev)
    de = torch.empty(n_ce, dtype=torch.int32, device=dev)
    dr = torch.empty(B, dtype=torch.int32, device=dev)
    single = torch.zeros(n_ce, dtype=torch.uint8, device=dev)
    Xt = torch.from_numpy(X).to(dev)
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    n_single, dev_states, dev_tables, dev_loss = [], [], [], []
    for step in (1, 2):
        pos = Xt[(step - 1) * B:step * B]
        d.prepare_batch(pos, eta, [L.SIDE_SO], n_ent, codes, de, dr, n_ent, n_rel, we, wr, seed=seed, counter0=step - 1,
                        single_flags=single)
        # poison: a slot the backward kernel applied in place must never be read by the apply kernel
        ce = torch.full((n_ce, Et.stride(0)), float("nan"), dtype=torch.float32, device=dev)[:, :ki]
        cr = torch.full((B, Et.stride(0)), float("nan"), dtype=torch.float32, device=dev)[:, :ki]
        hyper = _hyper(lr, step)
        d.train_backward_ex(MID[model], Et, Rt, ki, sc, pos, eta, codes, ce, cr, fused_loss=L.LOSS_IDS[loss], margin=1.0,
                            loss_accum=acc, single_ent=single, opt_id=L.OPT_IDS[opt], step=step, hyper=hyper,
                            ent_state0=se[0], ent_state1=se[1], tag_ent=tag_e)
        d.apply_grouped(L.OPT_IDS[opt], Et, ki, se[0], se[1], tag_e, step, ce, n_ce, True, hyper, we)
        d.apply_grouped(L.OPT_IDS[opt], Rt, ki, sr[0], sr[1], tag_r, step, cr, B, False, hyper, wr)
        n_single.append(int(single.sum().item()))
        dev_loss.append(float(acc.item()))
        dev_tables.append((Et.cpu().numpy().copy(), Rt.cpu().numpy().copy()))
        dev_states.append(([t.cpu().numpy().copy() for t in se if t is not None],
                           [t.cpu().numpy().copy() for t in sr if t is not None]))
        np.testing.assert_array_equal(codes.cpu().numpy(), co.corrupt_codes(B, eta, 2, n_ent, seed, step - 1))
    # both data-movement paths are exercised where the test says so
    if expect_singletons is not None:
        if kind == "uniform":
            assert min(n_single) > 0.3 * n_ce, n_single
        else:
            assert max(n_single) < 0.3 * n_ce, n_single

    # ---------------- oracle ----------------
    E, R = E0.copy(), R0.copy()
    stE, stR = orc.opt_init(opt, E.shape), orc.opt_init(opt, R.shape)
    loss_sum = 0.0
    never_e = np.ones(n_ent, bool)
    for step in (1, 2):
        xb = X[(step - 1) * B:step * B]
        xneg = orc.generate_corruptions_for_fit_philox(xb, eta=eta, corrupt_side="s,o", entities_size=n_ent, seed=seed,
                                                       counter=step - 1)
        val, _, _ = orc.model_loss(model, E, R, xb, eta, loss, None, ("s,o",), [xneg], k=k)
        loss_sum += float(val)
        dE, dR = orc.train_grads(model, E, R, xb, eta, loss, None, [xneg], k=k)
        tE, tR = np.zeros(n_ent, bool), np.zeros(n_rel, bool)
        for xx in (xb, xneg):
            tE[xx[:, 0]] = True
            tE[xx[:, 2]] = True
        tR[xb[:, 1]] = True
        never_e &= ~tE
        E_prev, R_prev = E, R
        E = orc.opt_apply(opt, E, dE, stE, lr=lr, touched=None if opt == "adam" else tE)
        R = orc.opt_apply(opt, R, dR, stR, lr=lr, touched=None if opt == "adam" else tR)
        gE, gR = np.abs(dE).max(), np.abs(dR).max()
        dE_t, dR_t = dev_tables[step - 1]
        (se_np, sr_np) = dev_states[step - 1]
        np.testing.assert_allclose(dev_loss[step - 1], loss_sum, rtol=2e-5, err_msg="loss, step %d" % step)
        if opt == "sgd":      # the update IS the gradient: W_prev - W_new = lr * g (+ one fp32 rounding of the table entry)
            prevE, prevR = (E0, R0) if step == 1 else dev_tables[0]
            np.testing.assert_allclose(prevE.astype(np.float64) - dE_t, lr * dE, rtol=1e-4, atol=2e-7 + 1e-5 * lr * gE)
            np.testing.assert_allclose(prevR.astype(np.float64) - dR_t, lr * dR, rtol=1e-4, atol=2e-7 + 1e-5 * lr * gR)
        elif opt == "adam":   # first moment: 0.1 g (+ 0.9 m): the gradient, read back through the state
            np.testing.assert_allclose(se_np[0], stE["m"], rtol=1e-4, atol=1e-6 * gE)
            np.testing.assert_allclose(sr_np[0], stR["m"], rtol=1e-4, atol=1e-6 * gR)
            np.testing.assert_allclose(se_np[1], stE["v"], rtol=2e-4, atol=1e-8 * gE * gE)
            np.testing.assert_allclose(sr_np[1], stR["v"], rtol=2e-4, atol=1e-8 * gR * gR)
            # the table itself: Adam's normalised step m / (sqrt(v) + eps) amplifies the last bit of a gradient that is
            # (nearly) zero by cancellation — there its sign is rounding noise, in the oracle's arithmetic as much as here,
            # and the step may differ by up to Adam's largest step (~3.2 lr at step 1); everywhere else 1e-2 lr
            for got_t, want_t, g_t, g_max in ((dE_t, E, dE, gE), (dR_t, R, dR, gR)):
                solid = np.abs(g_t) > 1e-4 * g_max
                np.testing.assert_allclose(got_t[solid], want_t[solid], rtol=0, atol=1e-2 * lr)
                np.testing.assert_allclose(got_t[~solid], want_t[~solid], rtol=0, atol=7.0 * lr)
        elif opt == "momentum":
            np.testing.assert_allclose(se_np[0], stE["m"], rtol=1e-4, atol=1e-5 * gE * lr)
            np.testing.assert_allclose(sr_np[0], stR["m"], rtol=1e-4, atol=1e-5 * gR * lr)
            np.testing.assert_allclose(dE_t, E, rtol=1e-5, atol=1e-5 * gE * lr)
        else:                 # adagrad
            np.testing.assert_allclose(se_np[0], stE["acc"], rtol=2e-4, atol=1e-8 * gE * gE)
            np.testing.assert_allclose(sr_np[0], stR["acc"], rtol=2e-4, atol=1e-8 * gR * gR)
            np.testing.assert_allclose(dE_t, E, rtol=0, atol=1e-2 * lr)
    # rows no triple of either batch touched: bit-identical (Adam's dense-equivalent decay of an all-zero state
    # subtracts lr_t * 0 / (0 + eps) = 0)
    assert never_e.any() or kind == "zipf" or expect_singletons is None
    np.testing.assert_array_equal(dev_tables[1][0][never_e], E0[never_e])

    # ---------------- the product's Trainer (fused + in-place + pipelined plan) == the sequence above, bitwise ----
    tr = Trainer(MID[model], ki, sc, E0, R0, eta, loss=loss, optimizer=opt, optimizer_params={"lr": lr}, batches_count=2,
                 seed=seed)
    assert tr.fused and tr.pipeline   # (in-place singleton updates: chosen by the expected singleton share, same bits either way)
    tr.set_training_set(X, B)
    tr.step(0, B, epoch=1, batch=1, prefetch=[(B, B, 1, 2)])
    tr.step(B, B, epoch=1, batch=2)
    Et2, Rt2 = tr.tables_numpy()
    np.testing.assert_array_equal(Et2, dev_tables[1][0])
    np.testing.assert_array_equal(Rt2, dev_tables[1][1])
    assert tr.read_loss() == dev_loss[1]


# ------------------------------------------------------------------------------------------------
# factored entity contributions: a negative's gradient row is (one float) x (a query row of its triple group)
# ------------------------------------------------------------------------------------------------
def _run_steps(factored, inplace, model, k, eta, loss, opt, n_ent, n_rel, B, kind, steps=3, seed=5, pair=False):
    """the C-ABI sequence of fit(): emg_prepare_batch -> emg_train_backward_ex -> emg_apply_grouped[_factored] (entities)
    -> emg_apply_grouped (relations); returns tables, optimizer state, tags and the loss after `steps` batches"""
    from emgraph_amd import _lib as L
    from emgraph_amd import device as d
    from emgraph_amd.training import alloc_table
    ki = 2 * k if model in ("ComplEx", "HolE") else k
    sc = float(F32(2 / k)) if model == "HolE" else 1.0
    rs = np.random.RandomState(101)
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = _batch(kind, n_ent, n_rel, steps * B, seed=23)
    dev = torch.device("cuda")
    Et, Rt = alloc_table(n_ent, ki, dev, init=E0), alloc_table(n_rel, ki, dev, init=R0)
    ns = {"sgd": 0, "momentum": 1, "adagrad": 1, "adam": 2, "adam_lazy": 2}[opt]
    fill = 0.1 if opt == "adagrad" else 0.0
    se = [alloc_table(n_ent, ki, dev, fill=fill) for _ in range(ns)] + [None] * (2 - ns)
    sr = [alloc_table(n_rel, ki, dev, fill=fill) for _ in range(ns)] + [None] * (2 - ns)
    tag_e = torch.zeros(n_ent, dtype=torch.int32, device=dev)
    tag_r = torch.zeros(n_rel, dtype=torch.int32, device=dev)
    n_ce = (2 + eta) * B
    we = torch.empty(d.apply_workspace_bytes(n_ce, n_ent, ki), dtype=torch.uint8, device=dev)
    wr = torch.empty(d.apply_workspace_bytes(B, n_rel, ki), dtype=torch.uint8, device=dev)
    codes = torch.empty(B * eta, dtype=torch.int32, device=dev)
    de = torch.empty(n_ce, dtype=torch.int32, device=dev)
    dr = torch.empty(B, dtype=torch.int32, device=dev)
    single = torch.zeros(n_ce, dtype=torch.uint8, device=dev)
    Xt = torch.from_numpy(X).to(dev)
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    n_single = 0
    for step in range(1, steps + 1):
        pos = Xt[(step - 1) * B:step * B]
        d.prepare_batch(pos, eta, [L.SIDE_SO], n_ent, codes, de, dr, n_ent, n_rel, we, wr, seed=seed, counter0=step - 1,
                        single_flags=single if inplace else None, factored=factored)
        # poison: nothing may be read that this step did not write
        ce = torch.full((4 * B if factored else n_ce, Et.stride(0)), float("nan"), dtype=torch.float32, device=dev)[:, :ki]
        cr = torch.full((B, Et.stride(0)), float("nan"), dtype=torch.float32, device=dev)[:, :ki]
        hyper = _hyper(0.002, step)
        d.train_backward_ex(MID[model], Et, Rt, ki, sc, pos, eta, codes, ce, cr, fused_loss=L.LOSS_IDS[loss], margin=1.0,
                            loss_accum=acc, single_ent=single if inplace else None, opt_id=L.OPT_IDS[opt], step=step,
                            hyper=hyper, ent_state0=se[0], ent_state1=se[1], tag_ent=tag_e, fac_ws_ent=we if factored else None)
        if pair:   # both tables through shared launches (what emg_plan_step does for large batches)
            d.apply_grouped_pair(
                dict(opt_id=L.OPT_IDS[opt], table=Et, k_int=ki, state0=se[0], state1=se[1], tag=tag_e, step=step, contrib=ce,
                     n_contrib=n_ce, skip_single=inplace, hyper=hyper, workspace=we, factored=factored),
                dict(opt_id=L.OPT_IDS[opt], table=Rt, k_int=ki, state0=sr[0], state1=sr[1], tag=tag_r, step=step, contrib=cr,
                     n_contrib=B, skip_single=0, hyper=hyper, workspace=wr))
        else:
            d.apply_grouped(L.OPT_IDS[opt], Et, ki, se[0], se[1], tag_e, step, ce, n_ce, inplace, hyper, we,
                            factored=factored)
            d.apply_grouped(L.OPT_IDS[opt], Rt, ki, sr[0], sr[1], tag_r, step, cr, B, 0, hyper, wr)
        if inplace:
            n_single += int(single.sum().item())
    torch.cuda.synchronize()
    return ([Et.cpu().numpy(), Rt.cpu().numpy()] + [t.cpu().numpy() for t in se + sr if t is not None]
            + [tag_e.cpu().numpy(), np.array([acc.item()])]), n_single


FACTORED_CASES = {
    # name: model, k, eta, loss, optimizer, n_ent, n_rel, B, batch kind
    "C3-sgd": ("ComplEx", 200, 20, "nll", "sgd", 60000, 1000, 4096, "uniform"),
    "C3-zipf": ("ComplEx", 200, 20, "nll", "sgd", 3000, 1000, 4096, "zipf"),        # hub rows: deferred / block-tree sums
    "C2-adam": ("DistMult", 200, 10, "nll", "adam", 15000, 237, 4096, "uniform"),
    "C2-adagrad-zipf": ("DistMult", 200, 10, "nll", "adagrad", 1500, 237, 4096, "zipf"),
    "C5-adam_lazy": ("HolE", 200, 20, "nll", "adam_lazy", 8000, 1345, 2048, "zipf"),
    "k32-momentum": ("DistMult", 32, 5, "pairwise", "momentum", 3000, 7, 2048, "uniform"),       # skinny rows: sub-wave segments
    "k50-sgd": ("DistMult", 50, 8, "pairwise", "sgd", 4000, 7, 2048, "uniform"),                 # k % 4 != 0: scalar path
    "k512-adam": ("ComplEx", 256, 4, "nll", "adam", 5000, 20, 1024, "uniform"),                  # two 16-byte chunks per lane
}


@pytest.mark.parametrize("inplace", [1, 0])
@pytest.mark.parametrize("case", list(FACTORED_CASES))
def test_factored_contributions_are_bit_identical_to_full_rows(case, inplace):
    """eta full gradient rows per triple group (emg_apply_grouped) == two query rows per group + one float per
    negative (emg_prepare_args.factored + emg_backward_args.fac_ws_ent + emg_apply_grouped_factored), BIT for BIT over three steps: tables, optimizer
    state, tags, loss.  The factored apply adds the separately rounded product coef * q exactly where the other path
    adds the stored row coef * q, in the same order."""
    from emgraph_amd import device as d
    d.require_gpu()
    full, n1 = _run_steps(False, inplace, *FACTORED_CASES[case])
    fac, n2 = _run_steps(True, inplace, *FACTORED_CASES[case])
    assert n1 == n2
    for a, b in zip(full, fac):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)


def test_factored_contributions_refused_for_transe():
    from emgraph_amd import _lib as L
    from emgraph_amd import device as d
    from emgraph_amd.training import alloc_table
    d.require_gpu()
    dev = torch.device("cuda")
    Et, Rt = alloc_table(50, 8, dev, fill=0.1), alloc_table(3, 8, dev, fill=0.1)
    pos = torch.zeros((4, 3), dtype=torch.int32, device=dev)
    codes = torch.zeros(8, dtype=torch.int32, device=dev)
    ce, cr = alloc_table(16, 8, dev), alloc_table(4, 8, dev)
    ws = torch.empty(d.apply_workspace_bytes(16, 50, 8), dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError, match="bilinear"):
        d.train_backward_ex(MID["TransE_L1"], Et, Rt, 8, 1.0, pos, 2, codes, ce, cr, fused_loss=L.LOSS_IDS["pairwise"],
                            loss_accum=torch.zeros(1, dtype=torch.float64, device=dev), fac_ws_ent=ws)


@pytest.mark.parametrize("case", list(FACTORED_CASES))
def test_pair_apply_is_bit_identical(case):
    """emg_apply_grouped_pair (entity and relation table through one window launch and one task launch) == the two
    separate calls, BIT for BIT over three steps — incl. Keras Adam's dense pass, scalar (k % 4 != 0) and <= 16-chunk rows,
    where the pair call falls back to two launches"""
    from emgraph_amd import device as d
    d.require_gpu()
    one, _ = _run_steps(True, 1, *FACTORED_CASES[case])
    two, _ = _run_steps(True, 1, *FACTORED_CASES[case], pair=True)
    for a, b in zip(one, two):
        np.testing.assert_array_equal(a, b)


# ------------------------------------------------------------------------------------------------
# the in-place choice is a speed choice: same bits either way
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("opt", ["adam", "adagrad", "momentum"])
@pytest.mark.parametrize("model,k", [("HolE", 200), ("ComplEx", 50), ("ComplEx", 150), ("DistMult", 200), ("TransE", 100)])
def test_inplace_choice_does_not_change_bits(monkeypatch, model, k, opt):
    """Trainer._choose_inplace picks, for a stateful optimizer, between updating singleton destinations inside the fused
    kernel and sending every row through the contribution buffer.  Both forms are different instantiations of the same
    kernel template: tables, optimizer state and loss must agree bit for bit on a popularity-skewed batch (rows with
    hundreds of contributions next to singletons)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    mid = {"HolE": L.HOLE, "ComplEx": L.COMPLEX, "DistMult": L.DISTMULT, "TransE": L.TRANSE_L1}[model]
    n_ent, n_rel, B, eta = 2000, 45, 512, 20
    ki = 2 * k if model in ("HolE", "ComplEx") else k
    rs = np.random.RandomState(7)
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    w = 1.0 / np.arange(1, n_ent + 1)
    s, o = (rs.choice(n_ent, 2 * B, p=w / w.sum()) for _ in range(2))
    X = np.stack([s, rs.randint(0, n_rel, 2 * B), o], 1).astype(np.int32)
    sc = float(F32(2 / k)) if model == "HolE" else 1.0

    def run(inplace, window=True):
        monkeypatch.setenv("EMG_INPLACE", "1" if inplace else "0")
        monkeypatch.setenv("EMG_INPLACE_STATE", "1" if window else "0")
        tr = Trainer(mid, ki, sc, E0, R0, eta, loss="nll", optimizer=opt, optimizer_params={"lr": 0.01}, batches_count=2,
                     seed=3)
        tr.set_training_set(X, B)
        tr.step(0, B, epoch=1, batch=1, prefetch=[(B, B, 1, 2)])
        tr.step(B, B, epoch=1, batch=2)
        Et, Rt = tr.tables_numpy()
        states = [t.cpu().numpy().copy() for t in tr.state_ent + tr.state_rel if t is not None]
        return tr.inplace, Et, Rt, states, tr.read_loss()

    # singletons in place with the negatives' state rows travelling in the rolling window (round 4: ip 4 / 5) | every row through the
    # contribution buffer | singletons in place with the state read chunk by chunk at the update (round 3: ip 2)
    a, b, c = run(True), run(False), run(True, window=False)
    assert a[0] and not b[0] and c[0]
    for other in (b, c):
        np.testing.assert_array_equal(a[1], other[1])
        np.testing.assert_array_equal(a[2], other[2])
        for x, y in zip(a[3], other[3]):
            np.testing.assert_array_equal(x, y)
        assert a[4] == other[4]


@pytest.mark.parametrize("model,k,case", [("ComplEx", 52, "fifth"), ("DistMult", 68, "fifth"), ("TransE", 100, "fifth"), ("HolE", 128, "fifth"),
                                          ("ComplEx", 100, "long_gaps"), ("DistMult", 200, "long_gaps"), ("TransE", 256, "hubs")])
def test_adam_replayed_in_the_scoring_kernel_gives_the_dense_pass_bits(monkeypatch, model, k, case):
    """Keras Adam under the deferred dense pass with in-place singletons (ip 6, round 4): the scoring kernel fetches (w, m, v) of a
    singleton negative's row together, replays the steps the row missed (the dense pass's own update, g = 0, each step's lr_t) in
    registers, scores it, updates it and writes the three rows once; emg_deferred_catchup and the apply (skip_single = 2) handle
    the subject / object slots and the rows hit more than once.  Tables, both state arrays and the loss must equal the DENSE form
    with every gradient row through the apply, bit for bit: on a table of which a batch touches a fifth, on one where a fifth of
    the gaps is longer than the 64 learning rates a lane register holds (the replay then reads the table), and with hub rows
    (block tasks finished by several waves)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    mid = {"ComplEx": L.COMPLEX, "HolE": L.HOLE, "DistMult": L.DISTMULT, "TransE": L.TRANSE_L1}[model]
    if case == "fifth":
        n_ent, n_rel, B, eta, nb, epochs = 40000, 30, 1024, 6, 4, 3
    else:
        n_ent, n_rel, B, eta, nb, epochs = 20000, 12, 128, 2, 24, 3
    ki = 2 * k if model in ("ComplEx", "HolE") else k
    sc = float(F32(2 / k)) if model == "HolE" else 1.0
    rs = np.random.RandomState(5)
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = np.stack([rs.randint(0, n_ent, nb * B), rs.randint(0, n_rel, nb * B), rs.randint(0, n_ent, nb * B)], 1).astype(np.int32)
    if case == "hubs":
        hub = (np.arange(nb * B) // B) % 2 == 0
        X[hub & (rs.rand(nb * B) < 0.6), 0] = 7
        X[hub & (rs.rand(nb * B) < 0.4), 2] = 11

    def run(deferred, inplace):
        monkeypatch.setenv("EMG_INPLACE", "1" if inplace else "0")
        # (TransE at k = 256: |score| > 75, where the NLL's clip has no gradient — the pairwise loss has)
        tr = Trainer(mid, ki, sc, E0, R0, eta, loss="pairwise" if model == "TransE" else "nll", optimizer="adam", optimizer_params={"lr": 0.01},
                     batches_count=nb, seed=3, deferred_dense=deferred)
        tr.set_training_set(X, B)
        assert tr.deferred == deferred and tr.inplace == inplace
        for ep in range(1, epochs + 1):
            for b in range(nb):
                tr.step(b * B, B, epoch=ep, batch=b + 1, prefetch=[(((b + 1) % nb) * B, B, ep + (b + 1) // nb, (b + 1) % nb + 1)])
        Et, Rt = tr.tables_numpy()
        return Et, Rt, [t.cpu().numpy().copy() for t in tr.state_ent + tr.state_rel], tr.read_loss()

    a, b = run(True, True), run(False, False)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)
    assert a[3] == b[3]
    assert not np.array_equal(a[0], E0)


@pytest.mark.parametrize("form", ["adagrad", "momentum", "adam_dense", "adam_deferred", "sgd"])
@pytest.mark.parametrize("model,k,eta", [("ComplEx", 100, 70), ("DistMult", 200, 130), ("TransE", 100, 65)])
def test_window_forms_with_more_than_64_negatives_keep_the_bits(monkeypatch, model, k, eta, form):
    """more negatives per positive than the wave has lanes (eta > 64): the group's codes / flags / positions are gathered 64 at a
    time and the loop over the rolling window runs once per gather.  The refills issued past a chunk's end are never taken; they
    must have LANDED before the next gather reuses registers (round-4 advisor finding: the drain sat behind the whole loop).
    Window forms (ip 4: Adagrad / momentum, ip 5: Adam with its dense pass, ip 6: Adam under the deferred pass) and the plain
    in-place form (ip 1) against every row through the contribution buffer: tables, optimizer state and loss bit for bit."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    mid = {"ComplEx": L.COMPLEX, "DistMult": L.DISTMULT, "TransE": L.TRANSE_L1}[model]
    n_ent, n_rel, B, nb = 60000, 9, 256, 3             # most slots are singletons
    ki = 2 * k if model == "ComplEx" else k
    rs = np.random.RandomState(eta)
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = np.stack([rs.randint(0, n_ent, nb * B), rs.randint(0, n_rel, nb * B), rs.randint(0, n_ent, nb * B)], 1).astype(np.int32)
    opt = {"adam_dense": "adam", "adam_deferred": "adam"}.get(form, form)
    deferred = {"adam_dense": False, "adam_deferred": True}.get(form)

    def run(inplace):
        monkeypatch.setenv("EMG_INPLACE", "1" if inplace else "0")
        monkeypatch.setenv("EMG_INPLACE_STATE", "1")
        tr = Trainer(mid, ki, 1.0, E0, R0, eta, loss="pairwise" if model == "TransE" else "nll", optimizer=opt, optimizer_params={"lr": 0.01},
                     batches_count=nb, seed=3, deferred_dense=deferred if inplace or deferred is None else False)
        tr.set_training_set(X, B)
        assert bool(tr.inplace) == inplace
        for ep in (1, 2):
            for b in range(nb):
                tr.step(b * B, B, epoch=ep, batch=b + 1)
        Et, Rt = tr.tables_numpy()
        return Et, Rt, [t.cpu().numpy().copy() for t in tr.state_ent + tr.state_rel if t is not None], tr.read_loss()

    a, b = run(True), run(False)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)
    assert a[3] == b[3]
    assert not np.array_equal(a[0], E0)


@pytest.mark.parametrize("p", [1, 2, 3, 4])
@pytest.mark.parametrize("model,k", [("ComplEx", 100), ("DistMult", 200), ("TransE", 100)])
def test_inplace_sgd_folds_the_lp_regulariser(monkeypatch, model, k, p):
    """SGD + LP: singleton destinations are updated inside the fused kernel with the regulariser's gradient folded in (its own
    instantiation, p <= 3), everything else by the apply kernel and the dense pass — the same tables, bit for bit, as sending
    every row through the apply; the regulariser's value (per-lane float partial sums, then double atomics, from three kernels instead of two) to 1e-9.
    p = 4 keeps every row in the apply (the in-place form carries no powf)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    mid = {"ComplEx": L.COMPLEX, "DistMult": L.DISTMULT, "TransE": L.TRANSE_L1}[model]
    n_ent, n_rel, B, eta = 30000, 45, 512, 10          # most of a batch's slots are singletons
    ki = 2 * k if model == "ComplEx" else k
    rs = np.random.RandomState(11)
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = np.stack([rs.randint(0, n_ent, 3 * B), rs.randint(0, n_rel, 3 * B), rs.randint(0, n_ent, 3 * B)], 1).astype(np.int32)

    def run(force_off):
        if force_off:
            monkeypatch.setenv("EMG_INPLACE", "0")
        else:
            monkeypatch.delenv("EMG_INPLACE", raising=False)
        tr = Trainer(mid, ki, 1.0, E0, R0, eta, loss="nll", optimizer="sgd", optimizer_params={"lr": 0.05}, batches_count=3,
                     seed=3, regularizer="LP", regularizer_params={"lambda": 1e-3, "p": p})
        tr.set_training_set(X, B)
        for b in range(3):
            tr.step(b * B, B, epoch=1, batch=b + 1, prefetch=[((b + 1) * B, B, 1, b + 2)] if b < 2 else None)
        Et, Rt = tr.tables_numpy()
        return tr.inplace, Et, Rt, tr.read_loss()

    a, b = run(False), run(True)
    assert a[0] == (p <= 3) and not b[0]
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[2], b[2])
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)     # (per-lane float partial sums group differently in the two forms)


@pytest.mark.parametrize("opt,reg", [("adam", None), ("sgd", 2), ("adagrad", 3), ("adam", 1), ("momentum", 2)])
@pytest.mark.parametrize("model,k", [("ComplEx", 50), ("DistMult", 64), ("TransE", 100)])
def test_deferred_adam_decay_gives_the_dense_pass_bits(model, k, opt, reg):
    """Keras Adam decays m, v and moves w of EVERY row every step, an LP regulariser gives every row a gradient every step.  With
    deferred_dense a row nothing touches is left alone and
    the missed steps are replayed — the dense pass's own update with g = 0 and each step's lr_t — when a batch is about to
    read it (emg_deferred_catchup) or when the tables are read (materialize): tables, both state arrays and the loss must equal
    the dense form bit for bit, on a table of which a batch touches a fifth (rows stay untouched for several steps, some
    for all of them)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    mid = {"ComplEx": L.COMPLEX, "DistMult": L.DISTMULT, "TransE": L.TRANSE_L1}[model]
    n_ent, n_rel, B, eta, nb = 40000, 30, 1022, 6, 4     # (1022: the last workgroup of a wave-per-group kernel holds two clamped copies of the last group)
    ki = 2 * k if model == "ComplEx" else k
    rs = np.random.RandomState(5)
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = np.stack([rs.randint(0, n_ent, nb * B), rs.randint(0, n_rel, nb * B), rs.randint(0, n_ent, nb * B)], 1).astype(np.int32)

    def run(deferred):
        kw = dict(regularizer="LP", regularizer_params={"lambda": 1e-3, "p": reg}) if reg else {}
        op = {"lr": 0.01}
        if opt == "sgd":      # (a learning rate that changes per step: the replay takes each step's own from the device table)
            op = {"lr": 0.05, "decay_cycle": 1, "decay_lr_rate": 2, "end_lr": 1e-4}
        tr = Trainer(mid, ki, 1.0, E0, R0, eta, loss="nll", optimizer=opt, optimizer_params=op, batches_count=nb,
                     seed=3, deferred_dense=deferred, **kw)
        tr.set_training_set(X, B)
        assert tr.deferred == deferred
        for ep in (1, 2, 3):
            for b in range(nb):
                tr.step(b * B, B, epoch=ep, batch=b + 1, prefetch=[(((b + 1) % nb) * B, B, ep + (b + 1) // nb, (b + 1) % nb + 1)])
        Et, Rt = tr.tables_numpy()
        states = [t.cpu().numpy().copy() for t in tr.state_ent + tr.state_rel if t is not None]
        return Et, Rt, states, tr.read_loss()

    a, b = run(True), run(False)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)
    if reg:   # (the regulariser's value is summed by other kernels in another order: double atomics over float partial sums)
        np.testing.assert_allclose(a[3], b[3], rtol=1e-9)
    else:
        assert a[3] == b[3]
    assert not np.array_equal(a[0], E0)


def test_deferred_pass_falls_back_to_the_dense_pass_where_it_cannot_run():
    """(advisor, round 3) emg_deferred_catchup walks the counting grouping's segment descriptors.  A table far longer than a batch
    has gradient rows is grouped by the radix-sort backend (n_rows > 16 n + 2^20): a Trainer asked to defer there must keep the
    dense pass instead of failing in its first step — and a fit longer than the learning-rate table the replay reads must bring
    every row up to date once and go on with the dense pass.  Both equal the dense form bit for bit."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    lib = L.load()
    assert lib.emg_plan_deferred_ok(16384, 20, 1000000, 1000) == 1          # C3
    assert lib.emg_plan_deferred_ok(16, 1, 1200000, 5) == 0                  # 48 gradient rows per batch against 1.2 M entities
    rs = np.random.RandomState(2)

    def run(n_ent, B, eta, nb, ki, deferred, table_steps=None, epochs=2):
        E0 = (rs.__class__(7).randn(n_ent, ki) * 0.3).astype(F32)
        R0 = (rs.__class__(8).randn(5, ki) * 0.3).astype(F32)
        r = rs.__class__(9)
        X = np.stack([r.randint(0, n_ent, nb * B), r.randint(0, 5, nb * B), r.randint(0, n_ent, nb * B)], 1).astype(np.int32)
        tr = Trainer(L.DISTMULT, ki, 1.0, E0, R0, eta, loss="nll", optimizer="adam", optimizer_params={"lr": 0.01}, batches_count=nb,
                     seed=3, deferred_dense=deferred)
        if table_steps:
            tr.LR_TABLE_STEPS = table_steps
        tr.set_training_set(X, B)
        was = tr.deferred
        for ep in range(1, epochs + 1):
            for b in range(nb):
                tr.step(b * B, B, epoch=ep, batch=b + 1)
        Et, Rt = tr.tables_numpy()
        return was, tr.deferred, Et, Rt, [t.cpu().numpy().copy() for t in tr.state_ent + tr.state_rel], tr.read_loss()

    # the sort backend's shape: asked to defer, runs dense
    a, b = run(1200000, 16, 1, 3, 8, True), run(1200000, 16, 1, 3, 8, False)
    assert a[0] is False and a[1] is False
    # a learning-rate table of 6 steps under a fit of 12
    c, d = run(5000, 64, 2, 4, 72, True, table_steps=6, epochs=3), run(5000, 64, 2, 4, 72, False, epochs=3)
    assert c[0] is True and c[1] is False
    for x, y in ((a, b), (c, d)):
        np.testing.assert_array_equal(x[2], y[2])
        np.testing.assert_array_equal(x[3], y[3])
        for u, v in zip(x[4], y[4]):
            np.testing.assert_array_equal(u, v)
        assert x[5] == y[5]


@pytest.mark.parametrize("model,k,opt,reg", [("ComplEx", 200, "adam", None), ("DistMult", 600, "sgd", 2), ("DistMult", 101, "adagrad", 3),
                                               ("TransE", 256, "momentum", 4), ("ComplEx", 100, "adam", 2),
                                               ("ComplEx", 64, "adam", "hubs"), ("DistMult", 300, "adam", "hubs")])
def test_deferred_pass_row_widths_and_long_gaps(model, k, opt, reg):
    """The catch-up's pipelined form holds a row in 1, 2 or 4 sixteen-byte chunks per lane (k_int <= 256 / 512 / 1024) and the
    next 64 learning rates in one register; rows that missed more than 64 steps, rows whose width is no multiple of 4 and
    regularisers with p > 3 take the generic replay; Adam without a regulariser writes back w alone and lets the apply redo
    the decay of m, v (the "hubs" cases add rows the apply finishes with several waves).  72 steps of small batches (2.5 % of the rows touched per step: a
    fifth of the gaps is longer than 64 steps) over these widths == the dense pass, bit for bit."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    mid = {"ComplEx": L.COMPLEX, "DistMult": L.DISTMULT, "TransE": L.TRANSE_L1}[model]
    n_ent, n_rel, B, eta, nb = 20000, 12, 128, 2, 24
    ki = 2 * k if model == "ComplEx" else k
    rs = np.random.RandomState(11)
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = np.stack([rs.randint(0, n_ent, nb * B), rs.randint(0, n_rel, nb * B), rs.randint(0, n_ent, nb * B)], 1).astype(np.int32)
    if reg == "hubs":   # two rows with ~80 and ~50 contributions per batch (block tasks: several waves finish such a row, and with
        reg = None      # Adam's w-only catch-up they must find m, v complete), in every second batch only (gaps in between)
        hub = (np.arange(nb * B) // B) % 2 == 0
        X[hub & (rs.rand(nb * B) < 0.6), 0] = 7
        X[hub & (rs.rand(nb * B) < 0.4), 2] = 11

    def run(deferred):
        kw = dict(regularizer="LP", regularizer_params={"lambda": 1e-3, "p": reg}) if reg else {}
        tr = Trainer(mid, ki, 1.0, E0, R0, eta, loss="nll", optimizer=opt, optimizer_params={"lr": 0.01}, batches_count=nb,
                     seed=3, deferred_dense=deferred, **kw)
        tr.set_training_set(X, B)
        assert tr.deferred == deferred
        for ep in (1, 2, 3):
            for b in range(nb):
                tr.step(b * B, B, epoch=ep, batch=b + 1, prefetch=[(((b + 1) % nb) * B, B, ep + (b + 1) // nb, (b + 1) % nb + 1)])
        Et, Rt = tr.tables_numpy()
        states = [t.cpu().numpy().copy() for t in tr.state_ent + tr.state_rel if t is not None]
        return Et, Rt, states, tr.read_loss()

    a, b = run(True), run(False)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)
    if reg:   # (per-lane float partial sums of the penalty, grouped by row here and by step there; the reference's loss is f32)
        np.testing.assert_allclose(a[3], b[3], rtol=1e-7)
    else:
        assert a[3] == b[3]


@pytest.mark.parametrize("opt,reg", [("sgd", None), ("adam", None), ("adagrad", 2), ("momentum", None)])
@pytest.mark.parametrize("model,k", [("TransE", 100), ("DistMult", 100), ("DistMult", 128), ("ComplEx", 36), ("TransE", 68)])
def test_two_segments_per_wave_give_the_one_segment_bits(monkeypatch, model, k, opt, reg):
    """Rows of 17..32 sixteen-byte chunks (k = 100 of the real-valued models, the reference's default width): the apply kernel
    gives each half of a wave its own destination row (segment_update_half) instead of one row to 25 of the wave's 64 lanes.
    Same additions in the same order: tables, optimizer state and loss equal the one-segment form (EMG_APPLY_HALF=0) bit for
    bit — hub rows (block tasks) and an odd number of items per wave included."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    mid = {"ComplEx": L.COMPLEX, "DistMult": L.DISTMULT, "TransE": L.TRANSE_L1}[model]
    n_ent, n_rel, B, eta, nb = 5000, 9, 1111, 5, 3
    ki = 2 * k if model == "ComplEx" else k
    rs = np.random.RandomState(23)
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = np.stack([rs.randint(0, n_ent, nb * B), rs.randint(0, n_rel, nb * B), rs.randint(0, n_ent, nb * B)], 1).astype(np.int32)
    X[rs.rand(nb * B) < 0.1, 0] = 3     # a hub row: more than 32 contributions per batch
    kw = dict(regularizer="LP", regularizer_params={"lambda": 1e-3, "p": reg}) if reg else {}

    def run(half):
        monkeypatch.setenv("EMG_APPLY_HALF", half)
        tr = Trainer(mid, ki, 1.0, E0, R0, eta, loss="nll", optimizer=opt, optimizer_params={"lr": 0.02}, batches_count=nb, seed=3, **kw)
        tr.set_training_set(X, B)
        for ep in (1, 2):
            for b in range(nb):
                tr.step(b * B, B, epoch=ep, batch=b + 1, prefetch=[(((b + 1) % nb) * B, B, ep + (b + 1) // nb, (b + 1) % nb + 1)])
        Et, Rt = tr.tables_numpy()
        states = [t.cpu().numpy().copy() for t in tr.state_ent + tr.state_rel if t is not None]
        return Et, Rt, states, tr.read_loss()

    a, b = run("1"), run("0")
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    for x, y in zip(a[2], b[2]):
        np.testing.assert_array_equal(x, y)
    assert a[3] == b[3] if not reg else abs(a[3] - b[3]) <= 1e-9 * abs(b[3])
    assert not np.array_equal(a[0], E0)


@pytest.mark.parametrize("opt", ["sgd", "adam"])
@pytest.mark.parametrize("model,k,eta", [("DistMult", 32, 20), ("DistMult", 64, 40), ("TransE", 100, 40), ("ComplEx", 32, 70)])
def test_more_negatives_than_lanes_per_group(monkeypatch, model, k, eta, opt):
    """The fused kernel gathers the corruption codes / in-place flags / factor positions of LPG negatives at a time (LPG = lanes
    per group: 16, 32 or 64 by row width).  More negatives than that run as several chunks — here 20 and 40 negatives on
    16- / 32-lane groups and 70 on 64 lanes — and must give the bits of the one-chunk form (a wave per group, LPG = 64,
    which EMG_WIDE_GROUPS=1 selects for the narrow rows; the 70-negative case is checked against the unfused kernels)."""
    from emgraph_amd import _lib as L
    from emgraph_amd.training import Trainer
    if eta > 62 and opt == "adam":
        pytest.skip("the unfused comparison is a tolerance check: Adam's normalised step amplifies last-bit gradient noise "
                    "(the chunk logic is the optimizer's business nowhere; the narrow-row cases cover Adam bit for bit)")
    mid = {"ComplEx": L.COMPLEX, "DistMult": L.DISTMULT, "TransE": L.TRANSE_L1}[model]
    n_ent, n_rel, B = 3000, 7, 300
    ki = 2 * k if model == "ComplEx" else k
    rs = np.random.RandomState(k + eta)
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    X = np.stack([rs.randint(0, n_ent, 2 * B), rs.randint(0, n_rel, 2 * B), rs.randint(0, n_ent, 2 * B)], 1).astype(np.int32)

    def run(wide, fused=True):
        monkeypatch.setenv("EMG_WIDE_GROUPS", "1" if wide else "0")
        monkeypatch.setenv("EMG_GRAPH", "0")
        tr = Trainer(mid, ki, 1.0, E0, R0, eta, loss="nll", optimizer=opt, optimizer_params={"lr": 0.02}, batches_count=2, seed=3,
                     fused=fused)
        tr.set_training_set(X, B)
        tr.step(0, B, epoch=1, batch=1, prefetch=[(B, B, 1, 2)])
        tr.step(B, B, epoch=1, batch=2)
        Et, Rt = tr.tables_numpy()
        return Et, Rt, tr.read_loss()

    a = run(False)
    b = run(True) if eta <= 62 else run(False, fused=False)
    if eta <= 62:
        np.testing.assert_array_equal(a[0], b[0])
        np.testing.assert_array_equal(a[1], b[1])
        assert a[2] == b[2]
    else:      # (the unfused kernels sum the loss in another order and form dL/dscore in their own kernel)
        np.testing.assert_allclose(a[0], b[0], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(a[1], b[1], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(a[2], b[2], rtol=1e-6)
    assert not np.array_equal(a[0], E0)


@pytest.mark.parametrize("seed", range(int(os.environ.get("EMG_SOAK_OFFSET", "0")), int(os.environ.get("EMG_SOAK_OFFSET", "0")) + int(os.environ.get("EMG_SOAK_SEEDS", "4"))))
def test_inplace_forms_random_configurations_do_not_change_bits(monkeypatch, seed):
    """soak of the scoring kernel's in-place forms (csrc/emg_score_kernels.hpp::ip_traits: plain SGD; the stateful optimizers with
    their state rows in the rolling window or read at the update; Keras Adam replayed under the deferred pass; SGD + LP with its
    replay) against the form that sends every row through the contribution buffer and the apply: a random model, width (3 ... 256),
    eta (1 ... 70: more negatives than a wave's lanes), loss, optimizer, regulariser, table size (tiny ... 200 000 rows: both
    groupings, dense pass inside the apply or deferred), batch split and graph shape per seed — tables, optimizer state and loss
    must agree bit for bit (what `test_inplace_choice_does_not_change_bits` checks for fifteen hand-picked cases)."""
    from emgraph_amd import _lib as L
    from emgraph_amd import device as d
    from emgraph_amd.training import Trainer
    d.require_gpu()
    rs = np.random.RandomState(13000 + seed)
    model = ("ComplEx", "DistMult", "TransE", "HolE")[rs.randint(0, 4)]
    k = int(rs.choice([3, 8, 16, 25, 50, 64, 100, 128, 150, 200, 256]))
    if model in ("ComplEx", "HolE"):
        k = min(k, 200)
    eta = int(rs.choice([1, 2, 5, 10, 20, 33, 65, 70]))
    opt = ("sgd", "adam", "adagrad", "momentum")[rs.randint(0, 4)]
    loss = ("nll", "pairwise", "absolute_margin")[rs.randint(0, 3)]
    reg = {"lambda": float(rs.choice([1e-3, 1e-2])), "p": int(rs.choice([1, 2, 3]))} if rs.randint(0, 4) == 0 else None
    n_ent = int(rs.choice([60, 500, 2000, 20000, 140000, 200000]))
    n_rel = int(rs.choice([1, 5, 45, 400]))
    B = int(rs.choice([3, 64, 300, 512, 1500]))
    deferred = None
    if n_ent >= 20000 and (opt == "adam" or reg is not None):
        deferred = [None, True, False][rs.randint(0, 3)]
    ki = 2 * k if model in ("HolE", "ComplEx") else k
    mid = {"HolE": L.HOLE, "ComplEx": L.COMPLEX, "DistMult": L.DISTMULT, "TransE": [L.TRANSE_L1, L.TRANSE_L2][rs.randint(0, 2)]}[model]
    E0 = (rs.randn(n_ent, ki) * 0.3).astype(F32)
    R0 = (rs.randn(n_rel, ki) * 0.3).astype(F32)
    if rs.randint(0, 2):
        w = 1.0 / np.arange(1, n_ent + 1)
        s, o = (rs.choice(n_ent, 2 * B, p=w / w.sum()) for _ in range(2))
    else:
        s, o = rs.randint(0, n_ent, 2 * B), rs.randint(0, n_ent, 2 * B)
    X = np.stack([s, rs.randint(0, n_rel, 2 * B), o], 1).astype(np.int32)
    sc = float(F32(2 / k)) if model == "HolE" else 1.0
    kw = dict(regularizer="LP", regularizer_params=reg) if reg else {}
    what = str((model, mid, k, eta, opt, loss, reg, n_ent, n_rel, B, deferred))

    def run(inplace, window=True):
        monkeypatch.setenv("EMG_INPLACE", "1" if inplace else "0")
        monkeypatch.setenv("EMG_INPLACE_STATE", "1" if window else "0")
        tr = Trainer(mid, ki, sc, E0, R0, eta, loss=loss, optimizer=opt, optimizer_params={"lr": 0.01}, batches_count=2, seed=3,
                     deferred_dense=deferred, **kw)
        tr.set_training_set(X, B)
        for ep in (1, 2):
            tr.step(0, B, epoch=ep, batch=1, prefetch=[(B, B, ep, 2)])
            tr.step(B, B, epoch=ep, batch=2, prefetch=[(0, B, ep + 1, 1)] if ep == 1 else None)
        Et, Rt = tr.tables_numpy()
        states = [t.cpu().numpy().copy() for t in tr.state_ent + tr.state_rel if t is not None]
        return tr.inplace, Et, Rt, states, tr.read_loss()

    a, b, c = run(True), run(False), run(True, window=False)
    for other in (b, c):
        np.testing.assert_array_equal(a[1], other[1], err_msg=what)
        np.testing.assert_array_equal(a[2], other[2], err_msg=what)
        for x, y in zip(a[3], other[3]):
            np.testing.assert_array_equal(x, y, err_msg=what)
        if reg:     # (the regulariser's value: float partials per wave, of the fused kernel's fold in one form and of the apply's in the other: 1.07e-8 apart in seed 28534)
            np.testing.assert_allclose(a[4], other[4], rtol=1e-6, equal_nan=True, err_msg=what)
        elif not abs(a[4]) <= 1e9:    # a diverged run (seeds 2248: -8.8e24, 4193 / 10036: 1e13, 29361: nan): float partials that far apart no longer add exactly in a double
            np.testing.assert_allclose(a[4], other[4], rtol=1e-12, equal_nan=True, err_msg=what)
        else:
            assert a[4] == other[4], what
