"""HIP-event timers for bench.py's rooflines.

`probed(name, flops, fn)`: ONE named kernel launch site (the `roofline` object).
`probed_family(family, flops, fn, nbytes)`: every launch of a kernel FAMILY (the row GEMMs, the grouped weight
gradients: `roofline_dominant`, the time-weighted figure of the kernels that own most of the step).

Events are recorded on the stream the kernel is launched on, around every launch, and read afterwards."""
import torch


class Probe:
    def __init__(self):
        self.name, self.flops, self.events = None, 0.0, []
        self.families = {}                      # family -> [(start, end, flops)]
        self.keep_calls, self.calls = False, {}  # keep the launch closures of the families (family_replay_ms)

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

    def record_family(self, family, flops, fn, nbytes=0.0):
        out, s, e = self._time(fn)
        self.families.setdefault(family, []).append((s, e, flops, nbytes))
        if self.keep_calls:
            self.calls.setdefault(family, []).append(fn)      # (the closure keeps its operand tensors alive)
        return out

    def family_replay_ms(self, replays=10):
        """The kept launches of each family captured into ONE hipGraph per family and replayed back to back:
        ms per replay = the family's device time without host launch gaps (operands as the step left them)."""
        out = {}
        for fam, fns in self.calls.items():
            g = torch.cuda.CUDAGraph()
            for fn in fns:                                         # warm (the weight-gradient closure allocates its workspace
                                                                   # per call: torch's graph-private pool serves it under capture)
                fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                for fn in fns:
                    fn()
            g.replay()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(replays):
                g.replay()
            e.record()
            torch.cuda.synchronize()
            out[fam] = s.elapsed_time(e) / replays
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
            out[fam] = {'launches': len(recs), 'ms': sum(r[0].elapsed_time(r[1]) for r in recs),
                        'flops': float(sum(r[2] for r in recs)), 'bytes': float(sum(r[3] for r in recs))}
        return out


_probe = None


def set_probe(p):
    global _probe
    _probe = p


def probed(name, flops, fn):
    return _probe.record(name, flops, fn) if _probe is not None else fn()


def probed_family(family, flops, fn, nbytes=0.0):
    """nbytes: the launch's ALGORITHMIC bytes -- every operand and the result once (bench.py roofline.algorithmic_bytes)."""
    return _probe.record_family(family, flops, fn, nbytes) if _probe is not None else fn()
