"""Fused Keras-Nadam step over the model's flat parameter / gradient buffers.

Reference: train.py:197,224 -- `tf.keras.optimizers.Nadam(learning_rate=1e-4)` applied to all trainable variables after
the cross-replica gradient SUM.  Keras defaults beta_1 .9, beta_2 .999, epsilon 1e-7, momentum-cache schedule
mu_t = beta_1 (1 - 0.5 * 0.96^(0.004 t)).  One HIP launch (stj_nadam_step) updates every tensor: the
parameters are slices of ONE flat f32 buffer and their gradients of another (the all-reduce bucket)."""
import torch

from .ops import _p, _st, call


class Nadam:
    def __init__(self, flat_weights, flat_grads, lr=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        if flat_weights.dtype != torch.float32 or flat_grads.dtype != torch.float32 or flat_weights.shape != flat_grads.shape:
            raise ValueError('Nadam needs matching flat float32 weight and gradient buffers')
        if not flat_weights.is_cuda:
            raise RuntimeError('Nadam: CUDA (ROCm) buffers only: the HIP path has no CPU fallback')
        self.w, self.g = flat_weights, flat_grads
        self.lr, self.b1, self.b2, self.eps = float(lr), float(beta_1), float(beta_2), float(epsilon)
        self.m = torch.zeros_like(flat_weights)
        self.v = torch.zeros_like(flat_weights)
        self.t = 0
        self.m_schedule = 1.0

    @classmethod
    def for_model(cls, model, **kw):
        """Optimizer over every parameter of a strajnet_amd.STrajNet (its flat master / gradient buffers)."""
        return cls(model._flat, model._gflat, **kw)

    def _mu(self, t):
        return self.b1 * (1.0 - 0.5 * 0.96 ** (0.004 * t))

    def step(self, grad_scale=1.0):
        self.t += 1
        mu_t, mu_n = self._mu(self.t), self._mu(self.t + 1)
        sched_t = self.m_schedule * mu_t
        sched_n = sched_t * mu_n
        self.m_schedule = sched_t
        cg = (1.0 - mu_t) / (1.0 - sched_t)
        cm = mu_n / (1.0 - sched_n)
        vs = 1.0 / (1.0 - self.b2 ** self.t)
        call('stj_nadam_step', _p(self.w), _p(self.g), _p(self.m), _p(self.v), self.w.numel(), self.lr, self.b1, self.b2, self.eps,
             cg, cm, vs, float(grad_scale), _st())
