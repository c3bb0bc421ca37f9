"""A stand-in for the handful of TensorFlow calls c3_amd/tf_bridge.py makes -- NOT TensorFlow, and no autodiff: tensors are
numpy arrays with `.numpy()`, `py_function` calls the host function at once, and `custom_gradient` records the gradient
function of every op it wraps so that a test can feed it a cotangent by hand (what a GradientTape would do)."""
import numpy as np


class T(np.ndarray):
    def numpy(self):
        return np.asarray(self)


def _t(x):
    return np.asarray(x).view(T)


class _Math:
    @staticmethod
    def real(x):
        return _t(np.real(np.asarray(x)))

    @staticmethod
    def reduce_mean(x, axis=None):
        return _t(np.mean(np.asarray(x), axis=axis))

    @staticmethod
    def reduce_variance(x, axis=None):
        return _t(np.var(np.asarray(x), axis=axis))

    @staticmethod
    def reduce_max(x, axis=None):
        return _t(np.max(np.asarray(x), axis=axis))

    @staticmethod
    def imag(x):
        return _t(np.imag(np.asarray(x)))

    @staticmethod
    def abs(x):
        return _t(np.abs(np.asarray(x)))


class StandIn:
    complex128 = np.complex128
    float64 = np.float64
    math = _Math()

    def __init__(self):
        self.grad_fns = []  # one per custom_gradient op executed, in order
        self.py_function_calls = 0

    @staticmethod
    def executing_eagerly():
        return True

    def cast(self, x, dtype):
        a = np.asarray(x)
        if np.iscomplexobj(a) and not np.issubdtype(dtype, np.complexfloating):
            raise TypeError("stand-in: casting complex to real discards the imaginary part (tf.cast does too): use tf.math.real first")
        return _t(a.astype(dtype))

    def stack(self, xs, axis=0):
        return _t(np.stack([np.asarray(x) for x in xs], axis=axis))

    def transpose(self, x):
        return _t(np.asarray(x).T)

    def stop_gradient(self, x):
        return _t(x)

    def ensure_shape(self, x, shape):
        a = np.asarray(x)
        shape = tuple(shape)
        assert a.ndim == len(shape) and all(s is None or int(s) == d for s, d in zip(shape, a.shape)), (a.shape, shape)
        return _t(a)

    def py_function(self, func, inp, Tout):
        self.py_function_calls += 1
        outs = func(*[_t(i) for i in inp])
        if not isinstance(outs, (tuple, list)):
            outs = (outs,)
        assert len(outs) == len(Tout)
        return [_t(np.asarray(o).astype(dt)) for o, dt in zip(outs, Tout)]

    def custom_gradient(self, f):
        def wrapped(*args):
            y, g = f(*[_t(a) for a in args])
            self.grad_fns.append(g)
            return y

        return wrapped

    def vectorized_map(self, fn, xs):
        return _t(np.stack([np.asarray(fn(x)) for x in np.asarray(xs)]))
