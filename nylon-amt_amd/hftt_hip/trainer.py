"""Fast training step over the HIP engine (no per-parameter autograd): forward plan -> fused loss kernel ->
backward plan -> fused Adam on the flat parameter buffer.  Mirrors the body of the reference's hot loop
(training/train.py:89-160: zero_grad, forward, 8 criteria, weighted sum, backward, optimizer.step) and
m_training.py:146 (Adam defaults).  Gradients are exposed as ``p.grad`` views into the flat gradient buffer.
"""
from __future__ import annotations

import torch

from . import ops
from ._capi import HfttError


class FusedAdam:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0) over the engine's flat buffers."""

    def __init__(self, model_or_engine, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        engine = model_or_engine.hftt_engine() if hasattr(model_or_engine, 'hftt_engine') else model_or_engine
        self.engine = engine
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.exp_avg = torch.zeros_like(engine.flat_params)
        self.exp_avg_sq = torch.zeros_like(engine.flat_params)
        self.param_groups = [{'lr': lr, 'betas': betas, 'eps': eps}]     # ReduceLROnPlateau-compatible surface

    def step(self, grad_scale=1.0):
        self.step_count += 1
        lr = self.param_groups[0]['lr']
        ops.adam_step(self.engine.flat_params, self.engine.flat_grads, self.exp_avg, self.exp_avg_sq, self.step_count,
                      lr=lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps, grad_scale=grad_scale)

    def state_dict(self):
        return {'step': self.step_count, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq, 'param_groups': self.param_groups}

    def load_state_dict(self, sd):
        self.step_count = int(sd['step'])
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.param_groups = sd['param_groups']


class TrainStep:
    """One object per (model, optimizer): ``loss = step(spec, onset, offset, mpe, velocity)``."""

    def __init__(self, model, lr=1e-4, weight_A=1.0, weight_B=1.0, grad_sync=None, optimizer=None):
        self.model = model
        self.engine = model.hftt_engine()
        self.opt = optimizer if optimizer is not None else FusedAdam(self.engine, lr=lr)
        if self.opt.engine is not self.engine:
            raise HfttError('FusedAdam was built for a different engine binding')
        self.weight_A, self.weight_B = weight_A, weight_B
        self.grad_sync = grad_sync          # callable(flat_grads) -> None (DDP all-reduce), or None

    def forward_backward(self, spec, label_onset, label_offset, label_mpe, label_velocity):
        eng = self.model.hftt_engine()
        if eng is not self.engine:
            raise HfttError('model was moved / re-bound after the TrainStep was created')
        B = spec.shape[0]
        eng.forward(spec, training=self.model.training, save=True)
        loss = eng.loss(B, (label_onset, label_offset, label_mpe, label_velocity), self.weight_A, self.weight_B, with_grad=True)
        eng.backward(B, on_ready=getattr(self.grad_sync, 'bucket_ready', None))
        return loss

    def __call__(self, spec, label_onset, label_offset, label_mpe, label_velocity):
        loss = self.forward_backward(spec, label_onset, label_offset, label_mpe, label_velocity)
        scale = 1.0
        if self.grad_sync is not None:
            scale = self.grad_sync(self.engine.flat_grads) or 1.0     # all-reduce(sum); the 1/world factor goes into Adam
        self.opt.step(grad_scale=scale)
        return loss            # [9] device tensor: total + 8 terms (no host sync here)

    def expose_grads(self):
        for (name, p, o, n), g in zip(self.engine._bound, self.engine.grad_views()):
            p.grad = g
