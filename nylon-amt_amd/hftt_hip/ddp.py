"""Clip-sharded data parallelism: one process per GPU, ONE flat fp32 gradient buffer, RCCL all-reduce over xGMI.

The reference is single-device (no torch.distributed anywhere); this is the multi-GPU path BASELINE.json asks for.
Mean-reduction losses (training/train.py:141-153) make the average of per-rank gradients equal the single-GPU gradient of
the concatenated batch, so the only collective per step is all_reduce(sum) of the flat gradient followed by a 1/world
scale that is folded into the fused Adam kernel (grad_scale).
"""
import torch
import torch.distributed as dist


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

    def __init__(self, engine, world, buckets=1):
        self.engine = engine
        self.world = world
        self.cuda = engine.flat_grads.is_cuda
        self.stream = torch.cuda.Stream(device=engine.flat_grads.device) if self.cuda else None
        n = engine.flat_grads.numel()
        step = (n + buckets - 1) // buckets
        step = (step + 1023) // 1024 * 1024
        self.slices = [(i, min(n, i + step)) for i in range(0, n, step)]
        self.scale = 1.0 / world
        self.launched = []                     # flat ranges already reduced (or in flight) for the current step

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
                dist.all_reduce(g[lo:hi], op=dist.ReduceOp.SUM)
        self.launched.append((lo, hi))

    def __call__(self, flat_grads):
        if self.launched:                      # overlapped mode: every range was handed over by bucket_ready
            done, self.launched = sorted(self.launched), []
            pos = 0
            for lo, hi in done:
                if lo != pos:
                    raise RuntimeError('FlatGradSync: gradient range [%d, %d) was never reported ready' % (pos, lo))
                pos = hi
            if pos != flat_grads.numel():
                raise RuntimeError('FlatGradSync: gradient range [%d, %d) was never reported ready' % (pos, flat_grads.numel()))
            if self.cuda:
                torch.cuda.current_stream(flat_grads.device).wait_stream(self.stream)
            return self.scale
        if not self.cuda:                      # gloo / CPU tensors (tests): same bucketing, no streams
            for a, b in self.slices:
                dist.all_reduce(flat_grads[a:b], op=dist.ReduceOp.SUM)
            return self.scale
        cur = torch.cuda.current_stream(flat_grads.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for a, b in self.slices:
                dist.all_reduce(flat_grads[a:b], op=dist.ReduceOp.SUM)
        cur.wait_stream(self.stream)
        return self.scale
