"""Flat-buffer data parallelism for the pretraining step.

Replaces nn.parallel.DistributedDataParallel(find_unused_parameters=True) of
tools/runner_pretrain.py:79-85 (collective C1 of SURVEY.md 2.2).  Design for
xGMI (point-to-point links, per-link-bound rings): few, large collectives.

  * all parameters live in ONE contiguous fp32 buffer, all gradients in another
    (parameters / .grad are views), ordered so that the AdamW weight-decay and
    no-decay groups are two contiguous ranges -> the optimiser is two fused
    launches and a reduction is a plain slice of the flat gradient;
  * the gradient buffer is cut into a handful of buckets (default 32 MB) in
    reverse parameter order; a post-accumulate hook counts arrivals and issues
    an ASYNC all-reduce for a bucket as soon as it is complete, so RCCL moves
    the tail of the network's gradients while backward still computes the head;
  * `finish()` waits for the outstanding buckets and averages; no unused-
    parameter search (every parameter is used on the benchmarked paths).

  * `late_prefixes` names the parameters whose gradients arrive LAST in backward (the patch
    embedder).  They are laid at the two ends of the flat buffers, everything else forms one
    contiguous `early_range` in the middle: the split graphed step (graph_step.py) all-reduces
    that range -- 98 % of the bytes -- on RCCL's stream while the embedder backward still runs.

Works on CPU tensors with gloo, which is how tests cover world_size 2.
"""
import weakref

import torch
import torch.distributed as dist
from torch import nn

from .arena import arena


def _no_weight_decay(name, p):
    return p.dim() == 1 or name.endswith('.bias') or 'token' in name


class FlatDataParallel(nn.Module):
    def __init__(self, module, bucket_mb=32, process_group=None, broadcast=True, late_prefixes=None):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self.world_size = dist.get_world_size(process_group) if dist.is_initialized() else 1
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        if late_prefixes is None:
            late_prefixes = getattr(module, 'late_grad_prefixes', ())

        def late(n):
            return any(n.startswith(pre) for pre in late_prefixes)
        # no-decay range first, decay range second (tools/builder.py:41-98 grouping); inside them the
        # late parameters sit at the outer ends: [late nd | early nd | pad | early wd | late wd]
        nd = [(n, p) for n, p in named if _no_weight_decay(n, p)]
        wd = [(n, p) for n, p in named if not _no_weight_decay(n, p)]
        nd = [t for t in nd if late(t[0])] + [t for t in nd if not late(t[0])]
        wd = [t for t in wd if not late(t[0])] + [t for t in wd if late(t[0])]
        late_nd = sum(p.numel() for n, p in nd if late(n))
        late_wd = sum(p.numel() for n, p in wd if late(n))
        self.names = [n for n, _ in nd + wd]
        params = [p for _, p in nd + wd]
        dev, dt = params[0].device, params[0].dtype
        sizes = [p.numel() for p in params]
        self.no_decay_numel = sum(p.numel() for _, p in nd)
        pad = (-self.no_decay_numel) % 64          # decay range starts 256-byte aligned
        total = sum(sizes) + pad
        self.flat_param = torch.zeros(total, device=dev, dtype=dt)
        self.flat_grad = torch.zeros(total, device=dev, dtype=dt)
        self.no_decay_range = (0, self.no_decay_numel)
        self.decay_range = (self.no_decay_numel + pad, total)
        self.early_range = (late_nd, total - late_wd)
        self.late_ranges = [r for r in ((0, late_nd), (total - late_wd, total)) if r[1] > r[0]]
        self.offsets = []
        off = 0
        for i, (p, n) in enumerate(zip(params, sizes)):
            if i == len(nd):
                off += pad
            self.flat_param[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_param[off:off + n].view_as(p)
            p.grad = self.flat_grad[off:off + n].view_as(p)
            self.offsets.append((off, n))
            off += n
        self.params = params
        self.grad_views = [p.grad for p in params]
        # gradient sink (nn_ops._sink_views): armed by a graphed step around its own forward + backward only
        self.sink_armed, self.sink_written = False, set()
        self.sink_strided = []          # (flat view, 2-D tile, cols): gradients the gather copies out of a wider tile
        self.wgrad_queue = []           # the armed step's queued weight gradients (nn_ops.flush_wgrad_queue)
        self.wgrad_stream, self.wgrad_inflight = None, []     # optional side stream for them (graph_step)
        me = weakref.ref(self)
        for i, p in enumerate(params):
            p._pdae_flat = (me, i)
        if self.world_size > 1 and broadcast:
            dist.broadcast(self.flat_param, src=0, group=process_group)
            for b in module.buffers():
                dist.broadcast(b, src=0, group=process_group)
        # buckets over the flat gradient, filled from the END (backward order)
        cap = max(1, int(bucket_mb * 1024 * 1024 // self.flat_grad.element_size()))
        self.buckets = []            # (start, end, [param indices])
        end, members, count = total, [], 0
        for i in range(len(params) - 1, -1, -1):
            members.append(i)
            count += sizes[i]
            if count >= cap or i == 0:
                start = self.offsets[i][0]
                self.buckets.append((start, end, list(members)))
                end, members, count = start, [], 0
        self._bucket_of = {}
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self._bucket_of[i] = b
        self._pending = [len(m) for _, _, m in self.buckets]
        self._works = []
        self.require_sync = True
        if self.world_size > 1:
            for i, p in enumerate(params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(param):
            if not self.require_sync:
                return
            if param.grad is None or param.grad.data_ptr() != self.flat_grad.data_ptr() + self.offsets[i][0] * self.flat_grad.element_size():
                # autograd replaced the view (first accumulation after set_to_none): copy back
                off, n = self.offsets[i]
                self.flat_grad[off:off + n].copy_(param.grad.reshape(-1))
                param.grad = self.flat_grad[off:off + n].view_as(param)
            b = self._bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                s, e, _ = self.buckets[b]
                self._works.append(dist.all_reduce(self.flat_grad[s:e], op=dist.ReduceOp.SUM,
                                                   group=self.process_group, async_op=True))
        return hook

    def forward(self, *args, **kwargs):
        # gradients land in (or are gathered into) the flat buffer before the next forward: the
        # backward may hand out slices of the shared pre-zeroed arena (arena.py's contract)
        arena.lease()
        return self.module(*args, **kwargs)

    def finish(self):
        """Wait for the gradient all-reduce and turn the sum into the mean.
        Call after backward(), before optimizer.step().  (Gradient accumulation: the micro-steps before
        the last run with require_sync = False and only add into the flat buffer; the hooks of the last
        backward pass then reduce buckets that hold the accumulated sums.)"""
        if self.world_size == 1:
            return
        if not self.require_sync:          # local step (no_sync): nothing was sent, nothing to wait for
            self._pending = [len(m) for _, _, m in self.buckets]
            return
        if any(p != 0 for p in self._pending):
            # a parameter received no gradient this step: reduce whatever was not sent
            for b, (s, e, _) in enumerate(self.buckets):
                if self._pending[b] != 0:
                    self._works.append(dist.all_reduce(self.flat_grad[s:e], op=dist.ReduceOp.SUM,
                                                       group=self.process_group, async_op=True))
        for w in self._works:
            w.wait()
        self._works = []
        self._pending = [len(m) for _, _, m in self.buckets]
        self.flat_grad.div_(self.world_size)

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for p, v in zip(self.params, self.grad_views):
            if p.grad is not v:
                p.grad = v

    def param_groups(self, weight_decay):
        """The two AdamW groups of the reference over the flat ranges."""
        k = sum(1 for n, p in zip(self.names, self.params) if _no_weight_decay(n, p))
        return [{'params': self.params[:k], 'weight_decay': 0.},
                {'params': self.params[k:], 'weight_decay': weight_decay}]
