"""N>1 path on CPU: world_size-2 gloo processes.  The HIP kernels cannot run here, so the per-rank gradient is produced by
the CPU oracle; what is under test is the product's data-parallel logic (hftt_hip/ddp.py): clip sharding r::world, flat
gradient all-reduce in buckets, the 1/world scale folded into the optimizer, parameter broadcast -- and the claim it rests on:
with mean-reduction losses the averaged per-rank gradient equals the single-process gradient of the global batch."""
import os
import sys
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from util import O, MINI


class _FakeEngine:
    def __init__(self, n):
        self.flat_grads = torch.zeros(n)
        self.device = torch.device('cpu')


def _worker(rank, world, port, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from hftt_hip.ddp import FlatGradSync
    torch.set_num_threads(2)
    cfg = MINI
    model = util.build_model(cfg, 100 + rank)          # different init per rank on purpose: broadcast must fix it
    names = [n for n, _ in model.named_parameters()]
    flat = torch.cat([p.data.reshape(-1) for p in model.parameters()])
    dist.broadcast(flat, 0)
    off = 0
    sd = {}
    for n_, p in model.named_parameters():
        sd[n_] = flat[off:off + p.numel()].view(p.shape).clone().requires_grad_(True)
        off += p.numel()
    gB = 4
    x = O.synth_spec(gB, cfg, salt=77) * 0.5
    labels = O.synth_labels(gB, cfg, salt=78)
    xs = x[rank::world]
    ls = tuple(t[rank::world] for t in labels)
    O.spec2midi_loss(O.model_forward(sd, xs, cfg), *ls).backward()
    eng = _FakeEngine(flat.numel())
    eng.flat_grads.copy_(torch.cat([sd[n_].grad.reshape(-1) for n_ in names]))
    sync = FlatGradSync(eng, world, buckets=3)
    scale = sync(eng.flat_grads)
    avg = eng.flat_grads * scale
    # overlapped mode: the engine hands over three ranges as they become final (here: after the fact, any order)
    eng2 = _FakeEngine(flat.numel())
    eng2.flat_grads.copy_(torch.cat([sd[n_].grad.reshape(-1) for n_ in names]))
    sync2 = FlatGradSync(eng2, world)
    n_all = flat.numel()
    for lo, hi in ((n_all // 2, n_all), (n_all // 5, n_all // 2), (0, n_all // 5)):
        sync2.bucket_ready(lo, hi)
    assert sync2(eng2.flat_grads) == scale and torch.equal(eng2.flat_grads * scale, avg)
    sync2.bucket_ready(0, 10)                            # a step that forgot a range must not pass silently
    try:
        sync2(eng2.flat_grads)
        raise AssertionError('missing range not detected')
    except RuntimeError:
        pass
    if rank == 0:
        sd_full = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        O.spec2midi_loss(O.model_forward(sd_full, x, cfg), *labels).backward()
        ref = torch.cat([sd_full[n_].grad.reshape(-1) for n_ in names])
        err = (avg - ref).abs().max().item() / ref.abs().max().item()
        np.save(os.path.join(tmp, 'result.npy'), np.array([err, scale, len(sync.slices)]))
    gathered = [torch.zeros_like(avg) for _ in range(world)]
    dist.all_gather(gathered, avg)
    assert torch.equal(gathered[0], gathered[1])        # every rank holds the same averaged gradient
    dist.destroy_process_group()


def test_world2_flat_gradient_allreduce_equals_global_batch_gradient(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    err, scale, n_buckets = np.load(tmp_path / 'result.npy')
    assert scale == 0.5 and n_buckets == 3
    assert err < 1e-5
