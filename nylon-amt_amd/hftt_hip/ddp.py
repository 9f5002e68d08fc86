"""Clip-sharded data parallelism: one process per GPU, ONE flat fp32 gradient buffer, RCCL all-reduce over xGMI.

The reference is single-device (no torch.distributed anywhere); this is the multi-GPU path BASELINE.json asks for.
Mean-reduction losses (training/train.py:141-153) make the average of per-rank gradients equal the single-GPU gradient of
the concatenated batch, so the only collective per step is all_reduce(sum) of the flat gradient followed by a 1/world
scale that is folded into the fused Adam kernel (grad_scale).
"""
import torch
import torch.distributed as dist


def _host_transport():
    """gloo moves host memory: device tensors are staged through the host (the RCCL backend, 'nccl', takes them as they are)"""
    return dist.get_backend() == 'gloo'


def _all_reduce_sum(t):
    if t.is_cuda and _host_transport():
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


class FlatGradSync:
    """all-reduce the engine's flat gradient buffer on a side stream, bucket by bucket, overlapped with the backward.

    The whole backward of this path is one plan of launches on torch's current stream.  The engine reports three flat
    ranges as they become final (``HfttEngine.backward(on_ready=...)``): time decoder + heads B first, then the frequency
    decoder + heads A, then the encoder.  ``bucket_ready`` records an event on the compute stream, makes the side stream
    wait for it and enqueues the all-reduce of that range there, so only the last (encoder) bucket is exposed; ``__call__``
    joins the side stream back into the compute stream before the fused Adam.  22 MB of gradients against a >= 10 ms step
    is latency- rather than bandwidth-bound on xGMI, hence few large buckets.

    Without ``bucket_ready`` calls (plain ``sync(flat_grads)``) the whole buffer is reduced after the backward in
    ``buckets`` equal slices -- the path the CPU/gloo tests and non-engine callers use."""

    def __init__(self, engine, world, buckets=1, rank=None):
        self.engine = engine
        self.world = world
        # every rank must draw its own dropout masks: fold the rank into the engine's seed (bench.py used to be the only caller that did)
        if rank is None:
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        if not getattr(engine, '_seed_has_rank', False):
            engine.base_seed = int(engine.base_seed) + 7919 * int(rank)
            engine._seed_has_rank = True
        self.cuda = engine.flat_grads.is_cuda
        self.stream = torch.cuda.Stream(device=engine.flat_grads.device) if self.cuda else None
        n = engine.flat_grads.numel()
        step = (n + buckets - 1) // buckets
        step = (step + 1023) // 1024 * 1024
        self.slices = [(i, min(n, i + step)) for i in range(0, n, step)]
        self.scale = 1.0 / world
        self.launched = []                     # flat ranges already reduced (or in flight) for the current step
        # timing=True: an event behind every bucket's all-reduce (side stream) and one at the end of the backward (compute stream), so a
        # caller can check that the early buckets really finish under the rest of the backward (overlap_report; bench.py, the rehearsal test)
        self.timing = False
        self._bucket_events, self._bwd_end_event, self._last_report = [], None, None

    def begin_step(self):
        """called before every backward: ranges reported by a backward that was never followed by __call__ (gradient accumulation,
        an exception) must not be mistaken for this step's"""
        self.launched = []
        self._bucket_events, self._bwd_end_event = [], None

    def bucket_ready(self, lo, hi):
        """flat_grads[lo:hi] is final on the current stream: start its all-reduce now."""
        g = self.engine.flat_grads
        if hi <= lo:
            return
        if not self.cuda:
            dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM)
        else:
            self.stream.wait_stream(torch.cuda.current_stream(g.device))
            with torch.cuda.stream(self.stream):
                _all_reduce_sum(g[lo:hi])
                if self.timing:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(self.stream)
                    self._bucket_events.append(((lo, hi), ev))
        self.launched.append((lo, hi))

    def overlap_report(self):
        """after a step with timing=True (and a device synchronize): per bucket, in the order the backward released them, the milliseconds
        by which its all-reduce finished BEFORE the end of the backward (negative: it was still running -- exposed)"""
        if self._bwd_end_event is None or not self._bucket_events:
            return self._last_report
        rep = [{'range': [int(lo), int(hi)], 'ms_before_backward_end': float(ev.elapsed_time(self._bwd_end_event))} for (lo, hi), ev in self._bucket_events]
        self._last_report = rep
        return rep

    def __call__(self, flat_grads):
        if self.launched:                      # overlapped mode: every range was handed over by bucket_ready
            done, self.launched = sorted(self.launched), []
            if self.cuda:                           # join the side stream first: also on the error paths below
                if self.timing:                      # the last backward kernel has been enqueued on the compute stream: mark its end
                    self._bwd_end_event = torch.cuda.Event(enable_timing=True)
                    self._bwd_end_event.record(torch.cuda.current_stream(flat_grads.device))
                torch.cuda.current_stream(flat_grads.device).wait_stream(self.stream)
            pos = 0
            for lo, hi in done:
                if lo != pos:
                    raise RuntimeError('FlatGradSync: gradient range [%d, %d) was never reported ready' % (pos, lo))
                pos = hi
            if pos != flat_grads.numel():
                raise RuntimeError('FlatGradSync: gradient range [%d, %d) was never reported ready' % (pos, flat_grads.numel()))
            return self.scale
        if not self.cuda:                      # gloo / CPU tensors (tests): same bucketing, no streams
            for a, b in self.slices:
                dist.all_reduce(flat_grads[a:b], op=dist.ReduceOp.SUM)
            return self.scale
        cur = torch.cuda.current_stream(flat_grads.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for a, b in self.slices:
                _all_reduce_sum(flat_grads[a:b])
        cur.wait_stream(self.stream)
        return self.scale


# ---------------------------------------------------------------------------------------------------------------------------------
# Host-side helpers of the data-parallel job (one process per GPU).  The reference has no multi-device code at all: these are what a
# launcher around training/m_training.py and the training.train / model.amt mirrors use when WORLD_SIZE > 1.
def rank_world():
    """(rank, world) of the running job; (0, 1) outside torch.distributed."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def is_main():
    """rank 0 writes checkpoints / logs (m_training.py:372-392 runs on every rank otherwise)"""
    return rank_world()[0] == 0


def shard_indices(n, rank=None, world=None):
    """clip ids of this rank: r, r+world, r+2*world, ... (files differ in length; striding keeps the ranks balanced), truncated so that every
    rank holds the SAME number of clips -- ranks must make the same number of steps or the gradient all-reduce deadlocks."""
    if rank is None or world is None:
        rank, world = rank_world()
    per = n // world
    return list(range(rank, per * world, world))


def broadcast_parameters(engine, src=0):
    """every rank starts from rank `src`'s parameters (m_training.py:141 initialises per process): one broadcast of the flat buffer"""
    if rank_world()[1] > 1:                              # (the engine re-packs its GEMM operands from flat_params at every forward)
        if engine.flat_params.is_cuda and _host_transport():
            h = engine.flat_params.cpu()
            dist.broadcast(h, src)
            engine.flat_params.copy_(h)
        else:
            dist.broadcast(engine.flat_params, src)


def allreduce_sums(*values, device=None):
    """sum python scalars / 0-d tensors over the ranks in ONE collective; returns floats (the epoch-loss exchange of train / valid)"""
    rank, world = rank_world()
    if world > 1 and _host_transport():
        device = 'cpu'
    t = torch.stack([torch.as_tensor(v, dtype=torch.float64, device=device).reshape(()) for v in values])
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]
