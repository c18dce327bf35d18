"""Keras 2.2.4 / TF 1.14 library semantics the hot path relies on (ORACLE, test infra only).

None of these are present in /root/reference (the libraries are not vendored); they restate
the published behaviour of the pinned versions (reference README.md:8-10) at the reference's
call sites.  SURVEY.md Appendix B lists them; each sits behind one function so it can be
corrected if a real Keras run ever disagrees.
"""
import numpy as np

KERAS_EPSILON = 1e-7   # K.epsilon(); used by Adam when epsilon=None (BS_brain.py:212)
HUBER_DELTA = 1.0      # tf.losses.huber_loss default delta (BS_brain.py:86-87)


def glorot_uniform(rng, shape, dtype=np.float64):
    """keras.initializers.glorot_uniform: U(-l, l), l = sqrt(6/(fan_in+fan_out)).
    Call sites: GNNLayer.build (BS_brain.py:26-37), Dense default kernel init (:176-179)."""
    fan_in, fan_out = shape[-2], shape[-1]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(dtype)


def huber_mean(y_true, y_pred):
    """tf.losses.huber_loss(labels, predictions), delta=1, reduction SUM_BY_NONZERO_WEIGHTS
    == mean over all elements (BS_brain.py:86-87; Appendix B.4)."""
    err = y_pred - y_true
    a = np.abs(err)
    quad = np.minimum(a, HUBER_DELTA)
    lin = a - quad
    return np.mean(0.5 * quad * quad + HUBER_DELTA * lin, dtype=y_pred.dtype)


def huber_grad(y_true, y_pred):
    """d huber_mean / d y_pred  = clip(pred - true, -delta, delta) / numel."""
    err = y_pred - y_true
    return (np.clip(err, -HUBER_DELTA, HUBER_DELTA) / err.size).astype(y_pred.dtype)


class KerasAdam:
    """keras.optimizers.Adam(lr=0.001, beta_1=0.5, beta_2=0.999, epsilon=None)
    (BS_brain.py:212).  Keras 2.2.4 update (Appendix B.6):
        t    = iterations + 1
        lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t)
        m    = b1*m + (1-b1)*g ;  v = b2*v + (1-b2)*g^2
        p    = p - lr_t * m / (sqrt(v) + eps),   eps = K.epsilon() = 1e-7
    Operates on a flat list of arrays, in place."""

    def __init__(self, lr=0.001, beta_1=0.5, beta_2=0.999, epsilon=KERAS_EPSILON):
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.iterations = 0
        self.m = None
        self.v = None

    def step(self, params, grads):
        if self.m is None:
            self.m = [np.zeros_like(p) for p in params]
            self.v = [np.zeros_like(p) for p in params]
        self.iterations += 1
        t = self.iterations
        dt = params[0].dtype.type
        lr_t = dt(self.lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t))
        b1, b2, eps = dt(self.b1), dt(self.b2), dt(self.eps)
        one = dt(1.0)
        for p, g, m, v in zip(params, grads, self.m, self.v):
            m *= b1
            m += (one - b1) * g
            v *= b2
            v += (one - b2) * (g * g)
            p -= lr_t * m / (np.sqrt(v) + eps)
