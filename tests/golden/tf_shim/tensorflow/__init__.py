"""Numpy-backed stand-in for the handful of TensorFlow symbols that the
reference's *pure* hot-path functions call (score ``_fn``s, corruption
generators, losses, LP regulariser).

TEST-ONLY TOOL.  TensorFlow is not installable in the build container, so the
golden-vector generator (``tests/golden/make_golden.py``) puts this directory
ahead of ``/root/reference`` on ``sys.path`` in order to *execute the
reference's own Python source* on numpy arrays.  The control flow / op graph
that runs is the reference's; only the leaf arithmetic is numpy's.

Nothing here is imported by the product (``emgraph_amd``), by ``bench.py`` or
by the ``-m gpu`` tests; it is only usable where ``/root/reference`` exists.

``tf.random.uniform`` does NOT generate numbers: it pops the next array from
``random._INJECTED`` so that the caller decides the draws (the TF Philox stream
cannot be reproduced without TF; see DESIGN.md "parity unpinned").
"""
import contextlib
import sys
import types

import numpy as np

float32 = np.float32
float64 = np.float64
int32 = np.int32
int64 = np.int64
bool = np.bool_  # noqa: A001  (tf.bool)


class Tensor(np.ndarray):
    pass


class Op:
    pass


class _Inert:
    """Permissive object for import-time / constructor-time TF symbols that the
    pure functions never touch (tf.device, tf.config..., initializer classes,
    @tf.custom_gradient, tf.compat.v1.logging...)."""

    def __getattr__(self, name):
        return self

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and callable(a[0]) and not isinstance(a[0], _Inert):
            return a[0]  # used as a decorator
        return self

    def __iter__(self):
        return iter(())

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_inert = _Inert()


def _np(x):
    return np.asarray(x)


def cast(x, dtype):
    return _np(x).astype(dtype)


def constant(v, dtype=None, name=None, shape=None):
    return np.array(v, dtype=dtype)


def shape(x):
    return np.array(_np(x).shape, dtype=np.int64)


def reshape(x, shp):
    shp = tuple(int(np.asarray(s).reshape(()) if np.ndim(s) else s) for s in (shp if np.ndim(shp) else [shp]))
    return _np(x).reshape(shp)


def tile(x, multiples):
    return np.tile(_np(x), [int(m) for m in multiples])


def squeeze(x, axis=None):
    return np.squeeze(_np(x), axis=axis)


def expand_dims(x, axis):
    return np.expand_dims(_np(x), axis)


def slice(x, begin, size):  # noqa: A001
    x = _np(x)
    idx = tuple(np.s_[int(b): (None if int(s) == -1 else int(b) + int(s))] for b, s in zip(begin, size))
    return x[idx]


def concat(values, axis):
    return np.concatenate([_np(v) for v in values], axis=axis)


def stack(values, axis=0):
    return np.stack([_np(v) for v in values], axis=axis)


def transpose(x, perm=None):
    return np.transpose(_np(x), perm)


def split(x, n, axis=0):
    return np.split(_np(x), n, axis=axis)


def reduce_sum(x, axis=None):
    x = _np(x)
    return np.sum(x, axis=axis, dtype=x.dtype if x.dtype.kind == "f" else None)


def reduce_mean(x, axis=None):
    return np.mean(_np(x), axis=axis)


def negative(x):
    return np.negative(_np(x))


def norm(x, ord=2, axis=None):  # noqa: A002
    return np.linalg.norm(_np(x), ord=ord, axis=axis)


def maximum(a, b):
    return np.maximum(a, b)


def exp(x):
    return np.exp(_np(x))


def abs(x):  # noqa: A001
    return np.abs(_np(x))


def pow(x, p):  # noqa: A001
    return np.power(_np(x), p)


def multiply(a, b):
    return np.multiply(a, b)


def add(a, b):
    return np.add(a, b)


def sigmoid(x):
    x = _np(x)
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def tanh(x):
    return np.tanh(_np(x))


def clip_by_value(x, clip_value_min, clip_value_max):
    x = _np(x)
    return np.clip(x, x.dtype.type(clip_value_min), x.dtype.type(clip_value_max))


def gather(params, indices, axis=0):
    return np.take(_np(params), _np(indices), axis=axis)


def logical_not(x):
    return np.logical_not(x)


def ones(shp, dtype=np.float32):
    return np.ones(tuple(int(s) for s in np.atleast_1d(shp)), dtype=dtype)


def zeros(shp, dtype=np.float32):
    return np.zeros(tuple(int(s) for s in np.atleast_1d(shp)), dtype=dtype)


def equal(a, b):
    return np.equal(a, b)


def Assert(cond, data=None, **k):  # noqa: N802
    assert np.all(cond), data
    return None


def unique(x):
    """tf.unique: values in FIRST-APPEARANCE order (+ inverse index)."""
    x = _np(x)
    _, first = np.unique(x, return_index=True)
    order = np.sort(first)
    vals = x[order]
    lut = {v: i for i, v in enumerate(vals.tolist())}
    idx = np.array([lut[v] for v in x.tolist()], dtype=np.int32)
    return vals, idx


def control_dependencies(deps):
    return contextlib.nullcontext()


def convert_to_tensor(x, dtype=None):
    return np.asarray(x, dtype=dtype)


class _Math(types.ModuleType):
    @staticmethod
    def log(x):
        return np.log(_np(x))

    @staticmethod
    def add(a, b):
        return np.add(a, b)

    @staticmethod
    def multiply(a, b):
        return np.multiply(a, b)

    @staticmethod
    def log_sigmoid(x):
        x = _np(x)
        return (-np.logaddexp(x.dtype.type(0), -x)).astype(x.dtype)

    @staticmethod
    def ceil(x):
        return np.ceil(x)


math = _Math("tensorflow.math")


class _NN(types.ModuleType):
    @staticmethod
    def embedding_lookup(params, ids):
        return _np(params)[_np(ids)]

    @staticmethod
    def softmax(x, axis=-1):
        x = _np(x)
        m = np.max(x, axis=axis, keepdims=True)
        e = np.exp(x - m)
        return (e / np.sum(e, axis=axis, keepdims=True)).astype(x.dtype)

    def __getattr__(self, name):
        return _inert


nn = _NN("tensorflow.nn")


class _Random(types.ModuleType):
    """tf.random.uniform pops pre-injected draws (FIFO)."""

    _INJECTED = []

    @classmethod
    def inject(cls, *arrays):
        cls._INJECTED = [np.asarray(a) for a in arrays]

    @classmethod
    def uniform(cls, shape, minval=0, maxval=None, dtype=np.float32, seed=None):  # noqa: A002
        if not cls._INJECTED:
            raise RuntimeError("tf_shim: tf.random.uniform called with no injected draws left")
        a = cls._INJECTED.pop(0)
        n = int(np.prod([int(s) for s in np.atleast_1d(shape)]))
        assert a.size == n, (a.size, n)
        if maxval is not None:
            assert a.min() >= minval and a.max() < int(maxval), (a.min(), a.max(), minval, maxval)
        return a.astype(dtype).reshape([int(s) for s in np.atleast_1d(shape)])

    @staticmethod
    def set_seed(seed):
        return None

    def __getattr__(self, name):
        return _inert


random = _Random("tensorflow.random")


class _Backend:
    @staticmethod
    def repeat(x, n):
        """keras.backend.repeat: [batch, dim] -> [batch, n, dim]."""
        x = _np(x)
        assert x.ndim == 2
        return np.repeat(x[:, None, :], int(n), axis=1)


class _Keras(_Inert):
    backend = _Backend()


keras = _Keras()

# import-time-only symbols -> inert
device = _inert
compat = _inert
config = _inert
initializers = _inert
optimizers = _inert
data = _inert
lookup = _inert
summary = _inert
Variable = _inert
custom_gradient = _inert
zeros_initializer = _inert
random_normal_initializer = _inert
random_uniform_initializer = _inert
constant_initializer = _inert
GradientTape = _inert
function = _inert
name_scope = _inert
TensorShape = _inert
__version__ = "0.0-numpy-shim"


def __getattr__(name):  # anything else touched at import time
    return _inert


for _n, _m in (("math", math), ("nn", nn), ("random", random)):
    sys.modules["tensorflow." + _n] = _m
