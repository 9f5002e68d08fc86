"""Drop-in replacement of the reference's ``training/train.py`` hot loops (``train`` :63-162, ``valid`` :168-254) for the
MI355X HIP model.  Same call signatures and return values, so ``m_training.py`` can ``import train`` unchanged.

Two execution paths, same kernels underneath:
  * fast path -- when ``optimizer`` is ``hftt_hip.trainer.FusedAdam`` (Adam defaults of m_training.py:146) and the eight
    criteria are the reference's (6x nn.BCELoss, 2x nn.CrossEntropyLoss, mean reduction): forward plan -> fused loss kernel
    -> backward plan -> fused Adam on the flat parameter buffer, no autograd graph, no per-step host sync;
  * compatibility path -- any other optimizer / criterion: the model's autograd.Function + torch criteria + ``optimizer.step()``
    exactly as training/train.py:89-160 does.
``valid(metrics=True)`` -- what an unchanged ``m_training.py`` calls at its last step by default (m_training.py:64,466-470) --
scores every batch as train.py:193-200 does (``reshape_for_mir_eval`` :9-57 on onset_B / offset_B against the onset labels, a
degenerate metric restated as it is) through ``evaluation.metrics`` (mir_eval is absent: parity unpinned at that library boundary),
prints the reference's three lines and writes ``test_performance.json`` with the keys ``precision`` / ``recall`` / ``f1`` (:235-251).

Data parallel (the reference is single-device): when torch.distributed is initialised with world > 1, every process runs these same
functions on ITS shard of the clips (``hftt_hip.ddp.shard_indices`` / ``DeviceClipStore.loader(rank=, world=)``: r::world, equal counts);
``train`` all-reduces the flat gradient inside the step (FlatGradSync, overlapped with the backward) and both functions exchange ONE
pair of scalars per epoch, so every rank returns the loss of the whole job: ``train`` -> global mean, ``valid`` -> (global sum, global
number of batches).  Checkpoints are written by rank 0 only (``hftt_hip.ddp.is_main``).
"""
import json
import os
import sys

import torch
import torch.nn as nn

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from hftt_hip._capi import HfttError           # noqa: E402
from hftt_hip.trainer import FusedAdam, TrainStep   # noqa: E402
from hftt_hip import ddp                        # noqa: E402
from evaluation.metrics import reshape_for_mir_eval, transcription_evaluate   # noqa: E402

try:
    from tqdm import tqdm
except Exception:      # pragma: no cover
    def tqdm(x, **k):
        return x


def _reference_criteria(cs):
    a = cs[:4] + cs[4:]
    return (all(isinstance(c, nn.BCELoss) and c.reduction == 'mean' and c.weight is None for c in (a[0], a[1], a[2], a[4], a[5], a[6]))
            and all(isinstance(c, nn.CrossEntropyLoss) and c.reduction == 'mean' and c.weight is None and c.ignore_index == -100
                    and c.label_smoothing == 0.0 for c in (a[3], a[7])))


def train(model, iterator, optimizer,
          criterion_onset_A, criterion_offset_A, criterion_mpe_A, criterion_velocity_A,
          criterion_onset_B, criterion_offset_B, criterion_mpe_B, criterion_velocity_B,
          weight_A, weight_B,
          device, verbose_flag):
    model.train()
    crits = (criterion_onset_A, criterion_offset_A, criterion_mpe_A, criterion_velocity_A,
             criterion_onset_B, criterion_offset_B, criterion_mpe_B, criterion_velocity_B)
    fast = isinstance(optimizer, FusedAdam) and _reference_criteria(crits)
    rank, world = ddp.rank_world()
    sync = None
    if world > 1:
        eng = model.hftt_engine()
        sync = model.__dict__.get('_hftt_sync')
        if sync is None or sync.engine is not eng:
            ddp.broadcast_parameters(eng)            # every rank starts from rank 0's parameters
            sync = ddp.FlatGradSync(eng, world, rank=rank)
            model.__dict__['_hftt_sync'] = sync
    step = None
    if fast:
        step = getattr(optimizer, '_train_step', None)
        if step is None or step.model is not model or step.engine is not model.hftt_engine():
            step = TrainStep(model, weight_A=weight_A, weight_B=weight_B, optimizer=optimizer, grad_sync=sync)
            optimizer._train_step = step
        step.weight_A, step.weight_B, step.grad_sync = weight_A, weight_B, sync
    epoch_loss = torch.zeros((), device=device, dtype=torch.float64) if fast else 0
    n = 0
    for i, (input_spec, label_onset, label_offset, label_mpe, label_velocity) in tqdm(enumerate(iterator), total=len(iterator)):
        input_spec = input_spec.to(device, non_blocking=True)
        label_onset = label_onset.to(device, non_blocking=True)
        label_offset = label_offset.to(device, non_blocking=True)
        label_mpe = label_mpe.to(device, non_blocking=True)
        label_velocity = label_velocity.to(device, non_blocking=True)
        n += 1
        if fast:
            loss9 = step(input_spec, label_onset.float().contiguous(), label_offset.float().contiguous(), label_mpe.float().contiguous(),
                         label_velocity.long().contiguous())
            epoch_loss += loss9[0].double()          # stays on the device: one host sync per epoch instead of per step (train.py:160)
            continue
        optimizer.zero_grad()
        (output_onset_A, output_offset_A, output_mpe_A, output_velocity_A, attention,
         output_onset_B, output_offset_B, output_mpe_B, output_velocity_B) = model(input_spec)
        output_velocity_A = output_velocity_A.contiguous().view(-1, output_velocity_A.shape[-1])
        output_velocity_B = output_velocity_B.contiguous().view(-1, output_velocity_B.shape[-1])
        label_onset = label_onset.contiguous().view(-1)
        label_offset = label_offset.contiguous().view(-1)
        label_mpe = label_mpe.contiguous().view(-1)
        label_velocity = label_velocity.contiguous().view(-1)
        loss_A = (criterion_onset_A(output_onset_A.contiguous().view(-1), label_onset) + criterion_offset_A(output_offset_A.contiguous().view(-1), label_offset)
                  + criterion_mpe_A(output_mpe_A.contiguous().view(-1), label_mpe) + criterion_velocity_A(output_velocity_A, label_velocity))
        loss_B = (criterion_onset_B(output_onset_B.contiguous().view(-1), label_onset) + criterion_offset_B(output_offset_B.contiguous().view(-1), label_offset)
                  + criterion_mpe_B(output_mpe_B.contiguous().view(-1), label_mpe) + criterion_velocity_B(output_velocity_B, label_velocity))
        loss = weight_A * loss_A + weight_B * loss_B
        if verbose_flag is True:
            print('(5) loss:' + str(loss.size()))
            print(loss)
        loss.backward()
        if sync is not None:                       # one all-reduce of the flat gradient buffer, then the mean
            eng = model.hftt_engine()
            flat = eng.flat_grads
            # p.grad are normally views of the flat buffer (autograd adopts what HfttModelFunction.backward hands it).  Where that is not so
            # (gradients kept across steps, a hook that re-created p.grad) the reduced buffer would never reach the optimizer and the ranks
            # would drift apart silently: bring such gradients into the buffer and point p.grad at it.
            base = flat.data_ptr()
            for _, p, o, n in eng._bound:
                if p.grad is not None and p.grad.data_ptr() != base + 4 * o:
                    flat[o:o + n].copy_(p.grad.reshape(-1))
                    p.grad = flat[o:o + n].view(p.shape)
            sync.begin_step()
            flat.mul_(sync(flat))
        optimizer.step()
        epoch_loss += loss.item()
    if fast:
        epoch_loss = float(epoch_loss.item())
    if world > 1:
        epoch_loss, n_batches = ddp.allreduce_sums(epoch_loss, len(iterator), device=device)
        return epoch_loss / n_batches
    return epoch_loss / len(iterator)


def valid(model, iterator,
          criterion_onset_A, criterion_offset_A, criterion_mpe_A, criterion_velocity_A,
          criterion_onset_B, criterion_offset_B, criterion_mpe_B, criterion_velocity_B,
          weight_A, weight_B,
          device,
          metrics=False):
    model.eval()
    precision = recall = f1 = 0.0
    crits = (criterion_onset_A, criterion_offset_A, criterion_mpe_A, criterion_velocity_A,
             criterion_onset_B, criterion_offset_B, criterion_mpe_B, criterion_velocity_B)
    fast = _reference_criteria(crits)
    epoch_loss = torch.zeros((), device=device, dtype=torch.float64) if fast else 0
    with torch.no_grad():
        for i, (input_spec, label_onset, label_offset, label_mpe, label_velocity) in enumerate(tqdm(iterator)):
            input_spec = input_spec.to(device, non_blocking=True)
            label_onset = label_onset.to(device, non_blocking=True)
            label_offset = label_offset.to(device, non_blocking=True)
            label_mpe = label_mpe.to(device, non_blocking=True)
            label_velocity = label_velocity.to(device, non_blocking=True)
            out = model(input_spec)
            if metrics:                                  # train.py:193-200 (one host copy of two [B, 128, 88] posteriors per batch)
                est_int, est_pitch = reshape_for_mir_eval(onset_matrix=out[5].detach().cpu().numpy(), offset_matrix=out[6].detach().cpu().numpy())
                ref = label_onset.detach().cpu().numpy()
                ref_int, ref_pitch = reshape_for_mir_eval(onset_matrix=ref, offset_matrix=ref)
                scores = transcription_evaluate(ref_int, ref_pitch, est_int, est_pitch)
                precision += scores['Precision']
                recall += scores['Recall']
                f1 += scores['F-measure']
            if fast:
                eng = model.hftt_engine()
                loss9 = eng.loss(input_spec.shape[0], (label_onset.float().contiguous(), label_offset.float().contiguous(),
                                                       label_mpe.float().contiguous(), label_velocity.long().contiguous()),
                                 weight_A, weight_B, with_grad=False)
                epoch_loss += loss9[0].double()          # one host sync per epoch (train.py:252 syncs per batch)
                continue
            oa, fa, ma, va, _att, ob, fb, mb, vb = out
            lo, lf, lm, lv = (t.contiguous().view(-1) for t in (label_onset, label_offset, label_mpe, label_velocity))
            loss_A = (criterion_onset_A(oa.contiguous().view(-1), lo) + criterion_offset_A(fa.contiguous().view(-1), lf)
                      + criterion_mpe_A(ma.contiguous().view(-1), lm) + criterion_velocity_A(va.contiguous().view(-1, va.shape[-1]), lv))
            loss_B = (criterion_onset_B(ob.contiguous().view(-1), lo) + criterion_offset_B(fb.contiguous().view(-1), lf)
                      + criterion_mpe_B(mb.contiguous().view(-1), lm) + criterion_velocity_B(vb.contiguous().view(-1, vb.shape[-1]), lv))
            epoch_loss += (weight_A * loss_A + weight_B * loss_B).item()
    if fast:
        epoch_loss = float(epoch_loss.item())
    world = ddp.rank_world()[1]
    n_batches = len(iterator)
    if world > 1:
        epoch_loss, n_batches = ddp.allreduce_sums(epoch_loss, len(iterator), device=device)
        n_batches = int(n_batches)
    if metrics:                                          # train.py:235-251
        if world > 1:                                    # every rank scored its shard: the job's mean over all batches
            precision, recall, f1 = ddp.allreduce_sums(precision, recall, f1, device=device)
        precision /= n_batches
        recall /= n_batches
        f1 /= n_batches
        print("Precision:", precision)
        print("Recall:", recall)
        print("F1:", f1)
        if ddp.is_main():
            with open("test_performance.json", mode="w") as opened_json:
                json.dump({"precision": precision, "recall": recall, "f1": f1}, opened_json)
    return epoch_loss, n_batches
