"""subprocess body of tests/test_library_switches.py: fit a small model with the library's A/B switches as the environment sets them
(they are read once per process) and dump the trained tables, the optimizer state and the epoch losses.
usage: python -m tests._switch_worker OUT.npz MODEL K LOSS OPTIMIZER [lp<p>]"""
import sys

import numpy as np


def main():
    out, name, k, loss, opt = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    reg = {} if len(sys.argv) < 7 else dict(regularizer="LP", regularizer_params={"lambda": 1e-3, "p": int(sys.argv[6][2:])})
    from tests.test_api import _models, synth_graph
    n_ent, n_rel, n = 900, 7, 2003   # 6 batches of 334 / 333 triples, eta 5: ~4000 slots over 900 rows — touched and untouched rows in every step
    X = synth_graph(n_ent, n_rel, n, seed=3)
    rs = np.random.RandomState(5)
    ki = 2 * k if name in ("ComplEx", "HolE") else k
    ent0 = (rs.randn(n_ent, ki) * 0.3).astype(np.float32)
    rel0 = (rs.randn(n_rel, ki) * 0.3).astype(np.float32)
    m = _models()[name](k=k, initializer="constant", initializer_params={"entity": ent0, "relation": rel0}, eta=5, epochs=2,
                        batches_count=6, seed=11, loss=loss, optimizer=opt, optimizer_params={"lr": 0.02}, **reg)
    m.fit(X)
    E, R = m.trained_model_params
    tr = m._trainer
    state = {}
    for nm in ("state_ent", "state_rel"):
        st = getattr(tr, nm, None)
        if st:
            for i, t in enumerate(st):
                if t is not None:
                    state["%s%d" % (nm, i)] = t.detach().cpu().numpy()
    np.savez(out, E=np.array(E), R=np.array(R), losses=np.array(m.epoch_losses, dtype=np.float64), **state)


if __name__ == "__main__":
    main()
