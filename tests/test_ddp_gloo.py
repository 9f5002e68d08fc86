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
        self.base_seed = 1234


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
    assert eng.base_seed == 1234 + 7919 * rank          # every rank draws its own dropout masks
    FlatGradSync(eng, world)
    assert eng.base_seed == 1234 + 7919 * rank          # ... folded in once per engine
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
    sync2.bucket_ready(0, 10)                            # ranges of an abandoned backward (accumulation, exception) are dropped ...
    sync2.begin_step()
    assert sync2.launched == []
    sync2.bucket_ready(0, 10)                            # ... and a step that forgot a range must not pass silently
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


# ------------------------------------------------------------------------------------------------------------------------------
# inference scatter (model/amt.py) and the epoch-loss exchange of training/train.py under world 2
def _amt_worker(rank, world, port, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import test_amt_host as H
    from model.amt import AMT
    from hftt_hip import ddp
    g = util.golden('amt')
    assert ddp.rank_world() == (rank, world) and ddp.is_main() == (rank == 0)
    for bs, gather in ((1, 'host'), (3, 'host'), (3, 'all'), (3, 'none')):
        amt = AMT(H.CFG, None, batch_size=bs, device='cpu', gather=gather)        # rank / world picked up from torch.distributed
        assert (amt.rank, amt.world) == (rank, world)
        T = H.CFG['input']['num_frame']
        calls = []
        echo = H.EchoModel(H.CFG)
        amt.model = type('Counting', (), {'eval': lambda s: s, '__call__': lambda s, x: (calls.append(x.shape[0]), echo(x))[1]})()
        for n in (8, 21, 30):
            feat = g[f'tr.{n}.feature']
            calls.clear()
            outs = amt.transcript(feat)
            n_clips = -(-n // 8)
            n_batches = -(-n_clips // bs)
            assert len(calls) == len(range(rank, n_batches, world))      # this rank ran only its share of the batches ...
            # ... rank 0 (gather='host': the default, a host-side gather by clip index; 'all': every rank) holds the whole file's result, equal
            # to the reference's; the other ranks hold their own clips (replicas only: no collective on the data path)
            whole = gather == 'all' or (gather == 'host' and rank == 0)
            mine = np.zeros(n_clips * T, bool)
            for bi in range(rank, n_batches, world):
                mine[bi * bs * T:min((bi + 1) * bs, n_clips) * T] = True
            for i, o in enumerate(outs):
                ref = g[f'tr.{n}.out{i}']
                assert o.dtype == ref.dtype and o.shape == ref.shape
                if whole:
                    np.testing.assert_array_equal(o, ref)
                else:
                    np.testing.assert_array_equal(o[mine[:len(o)]], ref[mine[:len(o)]])
                    assert not o[~mine[:len(o)]].any()
            for i, o in enumerate(amt.transcript_stride(feat, 2)):
                if whole:
                    np.testing.assert_array_equal(o, g[f'trs.{n}.2.out{i}'])
    # clip sharding: disjoint, equal-sized, covering all but the n % world tail
    ids = [ddp.shard_indices(11, r, world) for r in range(world)]
    assert ids == [[0, 2, 4, 6, 8], [1, 3, 5, 7, 9]] and ddp.shard_indices(11) == ids[rank]
    # epoch-loss exchange: every rank ends with the global sums
    tot, cnt = ddp.allreduce_sums(1.5 + rank, 3)
    assert (tot, cnt) == (4.0, 6.0)
    dist.destroy_process_group()


def test_world2_inference_scatter_and_loss_exchange(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_amt_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
