"""Per-launch HIP-event timer for the engine's plans (used by bench.py for the roofline object).

Events are recorded on torch's current stream, which is the stream every C-ABI call of the engine is enqueued on, so the
elapsed time between the two events around a call is that kernel's (those kernels') device time.  Nothing synchronises
until ``summary()``."""
import torch


class LaunchProfiler:
    def __init__(self):
        self.records = []      # (name, meta, start, end)
        self._pool = []
        self._cur = None

    def _event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def begin(self, name, meta):
        e0 = self._event()
        e0.record()
        self._cur = (name, meta, e0)

    def end(self):
        name, meta, e0 = self._cur
        e1 = self._event()
        e1.record()
        self.records.append((name, meta, e0, e1))

    def summary(self):
        """-> dict kernel-key -> {launches, ms, flops, bytes}; synchronises."""
        torch.cuda.synchronize()
        out = {}
        for name, meta, e0, e1 in self.records:
            key = meta['kernel'] if meta else name
            d = out.setdefault(key, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
            d['launches'] += 1
            d['ms'] += e0.elapsed_time(e1)
            if meta:
                d['flops'] += meta['flops']
                d['bytes'] += meta['bytes']
            self._pool.extend((e0, e1))
        self.records = []
        return out
