"""Fast training step over the HIP engine (no per-parameter autograd): forward plan -> fused loss kernel ->
backward plan -> fused Adam on the flat parameter buffer.  Mirrors the body of the reference's hot loop
(training/train.py:89-160: zero_grad, forward, 8 criteria, weighted sum, backward, optimizer.step) and
m_training.py:146-147 (Adam defaults, ReduceLROnPlateau on top).  Gradients are exposed as ``p.grad`` views into the flat
gradient buffer.
"""
from __future__ import annotations

import weakref

import torch

from . import ops
from ._capi import HfttError

# parameter -> engine whose flat buffer it is a view of (filled by HfttEngine.bind): lets FusedAdam(model.parameters()) find the engine
# although m_training.py:146 builds the optimizer before the first forward has bound anything
# (keyed by id: tensors compare element-wise, so they cannot be keys of a WeakKeyDictionary; the weak reference to the parameter
# guards against a recycled id)
_ENGINE_OF = {}


def register_binding(engine, params):
    eref = weakref.ref(engine)
    for p in params:
        key = id(p)
        _ENGINE_OF[key] = (weakref.ref(p, lambda _r, key=key: _ENGINE_OF.pop(key, None)), eref)


def engine_of(p):
    ent = _ENGINE_OF.get(id(p))
    if ent is None or ent[0]() is not p:
        return None
    return ent[1]()


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) as ONE kernel over the engine's flat buffers.

    Drop-in at m_training.py:146: ``optimizer = FusedAdam(model.parameters(), lr=args.lr)`` (a model is accepted too).  It is a real
    ``torch.optim.Optimizer``: ``ReduceLROnPlateau(optimizer)`` (:147) and ``scheduler.step`` (:437) drive ``param_groups[0]['lr']``,
    ``state_dict()`` / ``load_state_dict()`` (:374-392, :268-299) have torch Adam's layout (per-parameter ``step`` / ``exp_avg`` /
    ``exp_avg_sq``, here views into two flat moment buffers), so a reference ``.dat`` checkpoint's ``optimizer_dict`` loads."""

    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        model = params if hasattr(params, 'hftt_engine') else None
        if model is not None:
            params = model.parameters()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                                      differentiable=False, fused=None, decoupled_weight_decay=False))
        if len(self.param_groups) != 1:
            raise HfttError('FusedAdam takes one parameter group (the whole model), as m_training.py:146 builds it')
        self._model = model
        self.engine = None
        self.step_count = 0
        self.exp_avg = self.exp_avg_sq = None

    # ---- binding to the engine's flat buffers (lazy: the engine binds at the model's first forward)
    def _params(self):
        return self.param_groups[0]['params']

    def attach(self, engine):
        if self.engine is engine and self.exp_avg is not None and self.exp_avg.numel() == engine.flat_params.numel():
            return
        ps = self._params()
        bound = [p for _, p, _, _ in engine._bound]
        if len(ps) != len(bound) or any(a is not b for a, b in zip(ps, bound)):
            raise HfttError('FusedAdam must hold exactly model.parameters() of the bound model, in order')
        old = {p: dict(self.state[p]) for p in ps if p in self.state and 'exp_avg' in self.state[p]}
        self.engine = engine
        self.exp_avg = torch.zeros_like(engine.flat_params)
        self.exp_avg_sq = torch.zeros_like(engine.flat_params)
        for (name, p, o, n) in engine._bound:
            st = self.state[p]
            m, v = self.exp_avg[o:o + n].view(p.shape), self.exp_avg_sq[o:o + n].view(p.shape)
            if p in old:                          # state loaded (load_state_dict) or carried over a re-bind: copy into the flat moments
                m.copy_(old[p]['exp_avg']); v.copy_(old[p]['exp_avg_sq'])
                self.step_count = int(old[p]['step'])
            st['step'] = torch.tensor(float(self.step_count))
            st['exp_avg'], st['exp_avg_sq'] = m, v
        ctr = getattr(self, '_pending_counter', None)
        if ctr is not None:                       # a checkpoint loaded before any engine existed: resume its dropout stream position
            engine.step_counter = int(ctr)
            self._pending_counter = None

    def _find_engine(self):
        if self._model is not None:
            return self._model.hftt_engine()
        eng = engine_of(self._params()[0])
        if eng is None or not eng.is_bound():
            raise HfttError('FusedAdam.step: the parameters are not bound to a HIP engine yet (run a forward of the model on the GPU first)')
        return eng

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.attach(self._find_engine())
        g = self.param_groups[0]
        self.step_count += 1
        ops.adam_step(self.engine.flat_params, self.engine.flat_grads, self.exp_avg, self.exp_avg_sq, self.step_count,
                      lr=float(g['lr']), beta1=g['betas'][0], beta2=g['betas'][1], eps=g['eps'], grad_scale=grad_scale)
        self.engine._prepared_frozen = False          # (a frozen-weights inference engine must prepare its operands again)
        for p in self._params():
            self.state[p]['step'].fill_(self.step_count)
        return loss

    def state_dict(self):
        sd = super().state_dict()
        sd['hftt_step_counter'] = None if self.engine is None else int(self.engine.step_counter)      # dropout stream position (resume)
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        ctr = state_dict.pop('hftt_step_counter', None)
        super().load_state_dict(state_dict)          # torch's layout: state tensors are fresh copies, param_groups restored
        steps = [int(s['step']) for s in self.state.values() if 'step' in s]
        self.step_count = max(steps) if steps else 0
        for g in self.param_groups:                  # what the fused kernel does not implement must not be dropped silently
            if g.get('weight_decay', 0) != 0 or g.get('amsgrad', False) or g.get('maximize', False):
                raise HfttError('FusedAdam: the loaded state asks for weight_decay / amsgrad / maximize, which the fused Adam kernel does not do '
                                '(reference: optim.Adam(model.parameters(), lr), m_training.py:146)')
        eng, self.engine, self.exp_avg = self.engine, None, None
        self._pending_counter = None
        if eng is not None:
            self.attach(eng)                         # copy the loaded moments into the flat buffers
            if ctr is not None:
                eng.step_counter = int(ctr)          # applied now: a TrainStep built later must not rewind the dropout stream again
        else:
            self._pending_counter = ctr              # no engine yet: applied when one is attached (attach / TrainStep)


class TrainStep:
    """One object per (model, optimizer): ``loss = step(spec, onset, offset, mpe, velocity)``."""

    def __init__(self, model, lr=1e-4, weight_A=1.0, weight_B=1.0, grad_sync=None, optimizer=None):
        self.model = model
        self.engine = model.hftt_engine()
        self.opt = optimizer if optimizer is not None else FusedAdam(model, lr=lr)
        self.opt.attach(self.engine)
        ctr = getattr(self.opt, '_pending_counter', None)
        if ctr is not None:
            self.engine.step_counter = int(ctr)
            self.opt._pending_counter = None
        self.weight_A, self.weight_B = weight_A, weight_B
        self.grad_sync = grad_sync          # callable(flat_grads) -> None (DDP all-reduce), or None

    def forward_backward(self, spec, label_onset, label_offset, label_mpe, label_velocity):
        eng = self.model.hftt_engine()
        if eng is not self.engine:
            raise HfttError('model was moved / re-bound after the TrainStep was created')
        B = spec.shape[0]
        with torch.cuda.device(eng.device):
            eng.forward(spec, training=self.model.training, save=True)
            loss = eng.loss(B, (label_onset, label_offset, label_mpe, label_velocity), self.weight_A, self.weight_B, with_grad=True)
            if self.grad_sync is not None:
                self.grad_sync.begin_step()
            eng.backward(B, on_ready=getattr(self.grad_sync, 'bucket_ready', None))
        return loss

    def __call__(self, spec, label_onset, label_offset, label_mpe, label_velocity):
        loss = self.forward_backward(spec, label_onset, label_offset, label_mpe, label_velocity)
        scale = 1.0
        if self.grad_sync is not None:
            scale = self.grad_sync(self.engine.flat_grads) or 1.0     # all-reduce(sum); the 1/world factor goes into Adam
        with torch.cuda.device(self.engine.device):
            self.opt.step(grad_scale=scale)
        return loss            # [9] device tensor: total + 8 terms (no host sync here)

    def expose_grads(self):
        for (name, p, o, n), g in zip(self.engine._bound, self.engine.grad_views()):
            p.grad = g
