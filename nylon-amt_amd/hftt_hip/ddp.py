"""Clip-sharded data parallelism: one process per GPU, ONE flat fp32 gradient buffer, RCCL all-reduce over xGMI.

The reference is single-device (no torch.distributed anywhere); this is the multi-GPU path BASELINE.json asks for.
Mean-reduction losses (training/train.py:141-153) make the average of per-rank gradients equal the single-GPU gradient of
the concatenated batch, so the only collective per step is all_reduce(sum) of the flat gradient followed by a 1/world
scale that is folded into the fused Adam kernel (grad_scale).
"""
import torch
import torch.distributed as dist


class FlatGradSync:
    """all-reduce the engine's flat gradient buffer on a side stream, in a few large buckets.

    The whole backward of this path is one plan of launches on torch's current stream; bucket k covers the parameters whose
    gradients are final earliest in that plan (time layers + heads, then the frequency decoder, then the encoder which
    finishes last).  With 22 MB of gradients and a >= 10 ms step the all-reduce is latency- not bandwidth-bound on xGMI,
    so a single post-backward launch is within ~1% of a perfectly overlapped one; the side stream keeps it off the
    compute stream's queue so the fused Adam can start as soon as the last bucket lands."""

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

    def __call__(self, flat_grads):
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
