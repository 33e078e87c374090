"""HIP-event timer for ONE named kernel launch site (bench.py's roofline).

Events are recorded on the stream the kernel is launched on, around every
launch inside the timed region, and read after it."""
import torch


class Probe:
    def __init__(self):
        self.name, self.flops, self.events = None, 0.0, []

    def record(self, name, flops, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.name, self.flops = name, flops
        self.events.append((s, e))
        return out

    def summary(self):
        if not self.events:
            return None
        ms = [s.elapsed_time(e) for s, e in self.events]
        return {'name': self.name, 'flops': self.flops, 'avg_ms': sum(ms) / len(ms), 'launches': len(ms)}


_probe = None


def set_probe(p):
    global _probe
    _probe = p


def probed(name, flops, fn):
    return _probe.record(name, flops, fn) if _probe is not None else fn()
