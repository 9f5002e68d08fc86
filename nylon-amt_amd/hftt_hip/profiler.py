"""Per-launch HIP-event timer for the engine's plans (used by bench.py for the roofline objects, in a pass of its own AFTER the timed
region).

Events are recorded on torch's current stream, which is the stream every C-ABI call of the engine is enqueued on, so the elapsed time
between the two events around a call is that kernel's (those kernels') device time -- provided the queue never runs dry: when the host
falls behind (interpreter pauses), the gap before the kernel starts lands inside the interval.  ``step_done()`` therefore harvests after
every step and ``summary()`` replaces intervals far above their kernel's median (> 3x and > 50 us over) by the median, reporting how
many; the per-kernel averages agree with rocprofv3's (profiles/)."""
import statistics

import torch


class LaunchProfiler:
    def __init__(self):
        self.records = []      # (key, meta, start, end) of the step in flight
        self._pool = []
        self._cur = None
        self.samples = {}      # key -> {'ms': [..], 'flops': per launch, 'bytes': per launch}

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

    def step_done(self):
        """synchronise and move the step's intervals into the per-kernel sample lists (events go back to the pool)"""
        torch.cuda.synchronize()
        for name, meta, e0, e1 in self.records:
            key = meta['kernel'] if meta else name
            d = self.samples.setdefault(key, {'ms': [], 'flops': 0.0, 'bytes': 0.0, 'meta': []})
            d['ms'].append(e0.elapsed_time(e1))
            if meta:
                d['flops'] += meta['flops']
                d['bytes'] += meta['bytes']
                d['meta'].append(meta)
            self._pool.extend((e0, e1))
        self.records = []

    def summary(self):
        """-> dict kernel-key -> {launches, ms (sum, host-stall outliers replaced by the median), flops, bytes, stalls}"""
        if self.records:
            self.step_done()
        out = {}
        for key, d in self.samples.items():
            # host-stall filter: an interval far above the median OF ITS OWN SHAPE is a stalled host, not a long kernel (one symbol serves
            # launches of very different size -- S_e and S_n tokens, 256 x 256 and 88 x 88 attention -- so the median is taken per shape)
            shapes = [tuple(m.get('shape', ())) for m in d['meta']] if len(d['meta']) == len(d['ms']) else [()] * len(d['ms'])
            by_shape = {}
            for sh, t in zip(shapes, d['ms']):
                by_shape.setdefault(sh, []).append(t)
            meds = {sh: statistics.median(v) for sh, v in by_shape.items()}
            fixed = [meds[sh] if (t > 3.0 * meds[sh] and t > meds[sh] + 0.05) else t for sh, t in zip(shapes, d['ms'])]
            out[key] = {'launches': len(fixed), 'ms': sum(fixed), 'flops': d['flops'], 'bytes': d['bytes'],
                        'stalls': sum(1 for a, b in zip(fixed, d['ms']) if a != b), 'meta': d['meta']}
        return out
