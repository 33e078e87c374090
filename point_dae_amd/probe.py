"""HIP-event timers for bench.py's rooflines.

`probed(name, flops, fn)`: ONE named kernel launch site (the `roofline` object).
`probed_family(family, flops, fn)`: every launch of a kernel FAMILY (the row GEMMs, the grouped weight
gradients: `roofline_dominant`, the time-weighted figure of the kernels that own most of the step).

Events are recorded on the stream the kernel is launched on, around every launch, and read afterwards."""
import torch


class Probe:
    def __init__(self):
        self.name, self.flops, self.events = None, 0.0, []
        self.families = {}                      # family -> [(start, end, flops)]

    def _time(self, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        return out, s, e

    def record(self, name, flops, fn):
        out, s, e = self._time(fn)
        self.name, self.flops = name, flops
        self.events.append((s, e))
        return out

    def record_family(self, family, flops, fn):
        out, s, e = self._time(fn)
        self.families.setdefault(family, []).append((s, e, flops))
        return out

    def summary(self):
        if not self.events:
            return None
        ms = [s.elapsed_time(e) for s, e in self.events]
        return {'name': self.name, 'flops': self.flops, 'avg_ms': sum(ms) / len(ms), 'launches': len(ms)}

    def family_summary(self):
        """-> {family: {'launches', 'ms', 'flops'}} summed over every recorded launch."""
        out = {}
        for fam, recs in self.families.items():
            out[fam] = {'launches': len(recs), 'ms': sum(s.elapsed_time(e) for s, e, _ in recs),
                        'flops': float(sum(f for _, _, f in recs))}
        return out


_probe = None


def set_probe(p):
    global _probe
    _probe = p


def probed(name, flops, fn):
    return _probe.record(name, flops, fn) if _probe is not None else fn()


def probed_family(family, flops, fn):
    return _probe.record_family(family, flops, fn) if _probe is not None else fn()
