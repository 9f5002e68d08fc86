"""torch.autograd glue: the whole model forward/backward as ONE autograd.Function over the HIP engine.

This is the compatibility path (``loss.backward()`` of training/train.py:158 works unchanged with any torch
optimizer).  The fast path used by ``training.train`` / bench.py drives the same engine plans without autograd.
"""
import torch

_D_NAMES = ('onset_A', 'offset_A', 'mpe_A', 'velocity_A', None, 'onset_B', 'offset_B', 'mpe_B', 'velocity_B')


class HfttModelFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, engine, training, *params):
        outs = engine.forward(spec, training=training, save=True)
        ctx.engine = engine
        ctx.params = params
        ctx.B = spec.shape[0]
        ctx.generation = engine.generation
        ctx.mark_non_differentiable(outs[4])     # attention map: returned for inspection, never part of the loss
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        eng = ctx.engine
        ws = eng._ws[ctx.B]
        for name, g in zip(_D_NAMES, gouts):
            if name is None:
                continue
            buf = ws['bufs']['d.' + name]
            if g is None:
                buf.zero_()
            else:
                buf.copy_(g.reshape(buf.shape))
        # The reference loop calls optimizer.zero_grad() every step (train.py:89; set_to_none since torch 2.0), so p.grad is None here
        # and autograd simply adopts what it is handed: views of the flat gradient buffer, no 22 MB copy per step.
        # A caller that KEEPS gradients across backward passes (accumulation, zero_grad(set_to_none=False)) may be holding exactly those
        # views from an earlier pass: the engine's backward overwrites the flat buffer, i.e. the held gradients, before autograd adds the
        # new ones into them (the result would be 2 * g2 instead of g1 + g2).  Then the old contents are put back and the new gradients
        # are handed over as a copy.
        flat = eng.flat_grads
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        held = [p for p in ctx.params if p.grad is not None]
        aliased = any(lo <= p.grad.data_ptr() < hi for p in held)
        old = flat.clone() if aliased else None
        eng.backward(ctx.B, ctx.generation)
        if not held:
            grads = eng.grad_views(None)
        else:
            new = flat.clone()
            if aliased:
                flat.copy_(old)
            grads = eng.grad_views(new)
        return (None, None, None) + tuple(grads)
