"""GPU tests: the library's A/B switches that claim "same bits" give the same bits.  The switches are read once per process, so
every leg is a subprocess (tests/_switch_worker.py: a small fit — untouched rows, singletons and shared rows in every step — whose
trained tables, optimizer state and epoch losses are compared byte for byte).
  EMG_DENSE_FUSED  Keras Adam's dense-equivalent pass over the untouched rows inside the descriptor-driven apply launch / as a
                   launch of its own (emg_apply.hip: ApplyParams.dense_here)
  EMG_APPLY_FIX    the apply's optimizer rule fixed at compile time / the run-time switch (apply_segments_kernel<..., FIX>)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, tag, env_over, name, k, loss, opt, *extra):
    out = str(tmp_path / ("%s.npz" % tag))
    env = dict(os.environ)
    env.update(env_over)
    env["EMG_GRAPH"] = env_over.get("EMG_GRAPH", "1")
    subprocess.run([sys.executable, "-m", "tests._switch_worker", out, name, str(k), loss, opt] + list(extra), cwd=ROOT, env=env, check=True, timeout=600)
    return dict(np.load(out))


def _same(a, b):
    assert sorted(a) == sorted(b)
    for key in a:
        assert a[key].tobytes() == b[key].tobytes(), "%s differs" % key


@pytest.mark.parametrize("name,k,loss", [("TransE", 100, "pairwise"),      # rows of 25 chunks: two segments per wave
                                         ("DistMult", 200, "nll"),         # 50 chunks: a wave per segment, factored contributions
                                         ("ComplEx", 100, "nll")])
def test_adam_dense_pass_inside_the_apply_launch_gives_the_same_bits(tmp_path, name, k, loss):
    a = _run(tmp_path, "fused", {"EMG_DENSE_FUSED": "1"}, name, k, loss, "adam")
    b = _run(tmp_path, "alone", {"EMG_DENSE_FUSED": "0"}, name, k, loss, "adam")
    assert "state_ent1" in a   # (Adam's second moments were dumped: the comparison covers m and v)
    _same(a, b)


@pytest.mark.parametrize("name,k,loss,opt", [("TransE", 100, "pairwise", "adam"), ("DistMult", 200, "nll", "adam"),
                                             ("ComplEx", 100, "nll", "adagrad"), ("TransE", 100, "nll", "adagrad")])
def test_compile_time_optimizer_forms_of_the_apply_give_the_same_bits(tmp_path, name, k, loss, opt):
    a = _run(tmp_path, "fix", {"EMG_APPLY_FIX": "1"}, name, k, loss, opt)
    b = _run(tmp_path, "switch", {"EMG_APPLY_FIX": "0"}, name, k, loss, opt)
    _same(a, b)
    c = _run(tmp_path, "nograph", {"EMG_APPLY_FIX": "1", "EMG_GRAPH": "0"}, name, k, loss, opt)   # (and as single steps)
    _same(a, c)


@pytest.mark.parametrize("name,k,loss,p", [("ComplEx", 100, "nll", 2), ("DistMult", 200, "pairwise", 2), ("TransE", 100, "nll", 3)])
def test_sgd_with_the_lp_regulariser_folds_the_same_bits_in_every_form(tmp_path, name, k, loss, p):
    """plain SGD + LP: the apply's compile-time form for p = 2 (apply_segments_kernel<..., kFixSgdLp2>) against the run-time switch,
    and the two-multiplication fold lp_fold_p2 that every p = 2 path takes since round 5 (lambda 2 |w| sgn w = fl(2 lambda w)) — the
    tables byte for byte; the regulariser's value is a sum of float partials added as double atomics from several kernels: equal to 1e-9.  p = 3 keeps the
    generic fold in both legs (the switch must not touch it)."""
    a = _run(tmp_path, "fix", {"EMG_APPLY_FIX": "1"}, name, k, loss, "sgd", "lp%d" % p)
    b = _run(tmp_path, "switch", {"EMG_APPLY_FIX": "0"}, name, k, loss, "sgd", "lp%d" % p)
    c = _run(tmp_path, "noinplace", {"EMG_APPLY_FIX": "1", "EMG_INPLACE": "0", "EMG_GRAPH": "0"}, name, k, loss, "sgd", "lp%d" % p)   # every row through the apply
    for other in (b, c):
        for key in ("E", "R"):
            assert a[key].tobytes() == other[key].tobytes(), "%s differs" % key
        np.testing.assert_allclose(a["losses"], other["losses"], rtol=1e-9)   # (float partials per wave, added as doubles: the partition of the rows over the waves differs between the forms)
