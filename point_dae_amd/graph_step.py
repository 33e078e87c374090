"""The optimisation step as replayed hipGraphs.

An eager step of this model is ~1800 kernel launches; on MI355X the kernels of
one step take ~25 ms while the Python/launch path needs ~33 ms, so the eager
step is host-bound.  Here forward + loss + backward are captured once per
number of visible tokens (the mask ratio is drawn per batch, so the encoder's
token count T_vis takes ~20 values) and replayed; the host only draws the
random mask / affine maps with the reference's RNG calls (MaskTransformer
._mask_center_rand, corrupt_data) and copies them into static device buffers.
All graphs share one memory pool (activations of one step, ~6 GB of the 288 GB).

The gradient all-reduce (world > 1) and the optimiser run outside the graphs.  With more than
one rank the step is TWO graphs (GraphedTrainStep.split): the all-reduce of the Transformer's
gradients starts after the first and runs on RCCL's stream while the second -- the patch
embedder's backward -- replays; only the embedder's 2 MB are reduced in the open.
"""
import os
import torch

from . import _lib
from .corrupt_util_tensor import draw_corruption
from .data_parallel import FlatDataParallel
from .point_cae_transformer import draw_mask, mask_row_ids


def use_created_stream(device=None):
    """Make a created (non-NULL) stream the current stream of this thread and return it.

    Call once before building a graphed step.  Up to round 3 a device-to-host copy on the legacy NULL
    stream (a checkpoint) made later replays sporadically return garbage; the round-4 bisection
    shows a replay-ordering race that the step's current graph no longer triggers, not a corrupted pool --
    one created stream for everything remains the rule because it costs nothing."""
    s = torch.cuda.Stream(device)
    s.wait_stream(torch.cuda.current_stream(device))
    torch.cuda.set_stream(s)
    return s


def _warn_if_null_stream():
    if torch.cuda.current_stream() == torch.cuda.default_stream():
        import warnings
        warnings.warn('graphed step built on the legacy NULL stream: call '
                      'point_dae_amd.graph_step.use_created_stream() first (NULL-stream work between '
                      'replays corrupts hipGraph replays on this platform)', RuntimeWarning, stacklevel=3)


def _bn_buffers(module):
    """running_mean / running_var / num_batches_tracked of every BatchNorm under `module`: the state a training-mode
    forward changes besides the gradients."""
    out = []
    for m in module.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.running_mean is not None:
            out += [m.running_mean, m.running_var, m.num_batches_tracked]
    return out


class _KeepBNState:
    """The eager warm-up pass in front of a capture runs the forward once more than the reference does for that batch
    (warm-up, then the replay): the running estimates are put back behind it, so that the batch that triggers a
    capture updates them ONCE (the replay), as every other batch does."""

    def __init__(self, module):
        self.bufs = _bn_buffers(module)

    def __enter__(self):
        self.saved = [b.clone() for b in self.bufs]
        return self

    def __exit__(self, *exc):
        for b, s in zip(self.bufs, self.saved):
            b.copy_(s)
        return False


SINK = os.environ.get('PDAE_GRAD_SINK', '1') != '0'
WGRAD_SIDE = os.environ.get('PDAE_WGRAD_SIDE', '0') != '0'     # the stacks' weight gradients on a side stream (lab)
_AVG_OK = {}


def _start_average(model, a, b):
    """Start the all-reduce of flat_grad[a:b] -> (work, needs_division).  RCCL averages in the
    collective (ReduceOp.AVG) when the build supports it; otherwise (and with gloo in the CPU-side
    tests) the sum is divided once every slice has arrived.  The NCCL/RCCL process group runs the
    collective on its own stream behind an event recorded on the current one, so whatever is issued
    on the current stream afterwards overlaps with it."""
    dist = torch.distributed
    view = model.flat_grad[a:b]
    backend = dist.get_backend(model.process_group)
    if backend == 'nccl' and _AVG_OK.get(backend, True):
        try:
            w = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=model.process_group, async_op=True)
            _AVG_OK[backend] = True
            return w, False
        except (RuntimeError, ValueError, NotImplementedError):     # rejected before anything ran
            _AVG_OK[backend] = False
    return dist.all_reduce(view, group=model.process_group, async_op=True), True


def _average_gradients(model):
    """One all-reduce of the flat gradient buffer (116 MB for the Transformer DAE)."""
    w, div = _start_average(model, 0, model.flat_grad.numel())
    w.wait()
    if div:
        model.flat_grad.div_(model.world_size)


MULTI_COPY = os.environ.get('PDAE_MULTI_COPY', os.environ.get('PDAE_GLUE', '1')) != '0'      # (A/B: 0 = torch._foreach_copy_, two launches per 123 tensors)


def _copy_into_views(have, strided=()):
    """[(flat gradient view, the gradient autograd produced -- None: no gradient, the view is zeroed)] -> ONE launch per 128
    tensors (csrc/glue.hip multi_copy); strided: [(view, 2-D tensor, cols)] gradients that are the leading columns of a wider
    tile (nn_ops._PosEmbed)."""
    fast, slow = [], []
    for v, g in have:
        if g is None:
            if MULTI_COPY and v.dtype == torch.float32:
                fast.append((v, None))
            else:
                v.zero_()
            continue
        ok = MULTI_COPY and g.is_contiguous() and g.dtype == torch.float32 and v.dtype == torch.float32 and g.device == v.device
        (fast if ok else slow).append((v, g))
    if fast or strided:
        _lib.multi_copy(fast, strided)
    if slow:
        torch._foreach_copy_([v for v, _ in slow], [g for _, g in slow])


class GraphedTrainStep:
    MAX_STEPS = 3          # affine_r3 applies 1-3 maps; shorter draws are padded with identities
    RING = int(os.environ.get("PDAE_RING", "4"))   # staging slots = how many steps the host may run ahead

    def __init__(self, model, optimizer, config, batch_size, npoints, warmup_eager=2, split=None,
                 step_per_update=None):
        """split: two-phase step -- graph 1 = forward + loss + the Transformer's backward, graph 2 = the
        patch embedder's backward (2.4 of the 14 ms); the all-reduce of the Transformer's gradients
        (model.early_range, 98 % of the bytes) is started between the two replays and runs on RCCL's
        stream under graph 2.  Default: on when world_size > 1.

        Every branch of the reference's step body (runner_pretrain.py:161-197) replays: the loss mix reads its
        normal-loss weight from a DEVICE scalar (set_gradual_weight, once per epoch: `xyznormal_gradual` /
        `xyznormal_warm` change it between replays), the un-masked model (NormalTransformer, :473-541) has one
        static shape and draws no mask, and with step_per_update > 1 the micro-steps' gradients are summed in a
        second flat buffer; the collective and AdamW run on the step that closes the group."""
        assert isinstance(model, FlatDataParallel)
        self.model, self.optimizer, self.config = model, optimizer, config
        self.net = model.module
        self.split = (model.world_size > 1) if split is None else bool(split)
        if self.split and not model.late_ranges:
            self.split = False                     # nothing was declared late: one phase
        e0, e1 = model.early_range
        self.early_idx = [i for i, (off, _) in enumerate(model.offsets) if e0 <= off < e1]
        self.late_idx = [i for i, (off, _) in enumerate(model.offsets) if not e0 <= off < e1]
        dev = model.flat_param.device
        self.B, self.G = batch_size, self.net.num_group
        self.masked = bool(getattr(self.net, 'masked', True))
        self.pts = torch.zeros(batch_size, npoints, 3, device=dev)
        # Everything the host draws per step travels in ONE packed buffer (one H2D copy): the affine steps, the
        # visible / masked row lists as int64 (index_select) and as int32 (the embedder's group lists), and the
        # decoder's token order [visible..., masked...] per sample (otherwise a cat launch inside the graph).
        BG = batch_size * self.G
        sizes = [self.MAX_STEPS * batch_size * 10 * 4, BG * 8, BG * 8, BG * 8, BG * 4, BG * 4]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + (n + 15) // 16 * 16)
        self._offs = offs

        def views(buf):
            cut = [buf[offs[i]:offs[i] + sizes[i]] for i in range(6)]
            return dict(steps=cut[0].view(torch.float32).view(self.MAX_STEPS, batch_size, 10),
                        vis=cut[1].view(torch.int64), msk=cut[2].view(torch.int64), order=cut[3].view(torch.int64),
                        vis32=cut[4].view(torch.int32), msk32=cut[5].view(torch.int32))
        self.stage = torch.zeros(offs[-1], dtype=torch.uint8, device=dev)
        dv = views(self.stage)
        self.steps, self.vis, self.msk, self.order = dv['steps'], dv['vis'], dv['msk'], dv['order']
        self.vis32, self.msk32 = dv['vis32'], dv['msk32']
        # The host runs several steps ahead of the GPU, so the pinned staging buffers of
        # the draws form a ring; a slot is reused only after the event recorded behind its
        # H2D copy has completed.
        self.ring = []
        for _ in range(self.RING):
            raw = torch.zeros(offs[-1], dtype=torch.uint8).pin_memory()
            slot = views(raw)
            slot.update(raw=raw, done=None)
            self.ring.append(slot)
        if not self.masked:                                        # every token visible, one static shape
            ar = torch.arange(BG)
            for slot in self.ring:
                slot['vis'].copy_(ar), slot['order'].copy_(ar), slot['vis32'].copy_(ar.to(torch.int32))
            self.stage.copy_(self.ring[0]['raw'])
        self.slot = 0
        self.graphs, self.outputs = {}, {}
        _warn_if_null_stream()
        self.pool = None
        self.eager_left = warmup_eager
        self.normal_weight = float(config.normal_weight)
        self.loss_type = config.loss_type
        if self.loss_type not in ('xyz', 'normal', 'xyznormal', 'xyznormal_gradual', 'xyznormal_warm'):
            raise NotImplementedError('graphed step: loss_type %s' % self.loss_type)
        # weight of the normal loss as the captured graphs read it: normal_weight (x the epoch's gradual weight)
        self.w_dev = torch.full((), self.normal_weight, device=dev)
        self.seed = torch.ones((), device=dev)
        if self.loss_type in ('xyznormal_gradual', 'xyznormal_warm'):
            self.w_dev.zero_()                                     # epoch 0 of both ramps
        self.spu = int(config.get('step_per_update', 1) if step_per_update is None else step_per_update)
        self.micro = 0
        self.accum = torch.zeros_like(model.flat_grad) if self.spu > 1 else None

    def invalidate(self):
        """Drop the captured graphs: a value they hold as a kernel argument changed (the BatchNorm momentum of
        misc.BNMomentumScheduler); the next step of every T_vis captures again."""
        self.graphs.clear()
        self.outputs.clear()

    def reset_micro(self):
        """the reference restarts its micro-step counter at every epoch (runner_pretrain.py:112), gradients carry over"""
        self.micro = 0

    def set_gradual_weight(self, gw):
        """runner_pretrain.py:113-122: the epoch's ramp factor of `xyznormal_gradual` / `xyznormal_warm`."""
        if self.loss_type in ('xyznormal_gradual', 'xyznormal_warm'):
            self.w_dev.fill_(self.normal_weight * float(gw))

    def _mix(self, lx, ln):
        lt = self.loss_type
        if lt == 'xyz' or (lt != 'normal' and not ln.requires_grad):
            # (the plain model's second loss is the constant zeros(1): lx + w * 0 is lx, bit for bit, without the
            # fill / sum / mul / add launches and their backward twins)
            return lx
        if lt == 'normal':
            return self.w_dev * ln.sum()
        if ln.numel() == 1 and lx.numel() == 1:                   # lx + w * ln as ONE launch (was sum, mul, add)
            return torch.addcmul(lx.reshape(()), self.w_dev, ln.reshape(()))
        return lx + self.w_dev * ln.sum()

    def _draw(self, tvis=None):
        """Host RNG, in the order the eager forward consumes it: corruption
        (corrupt_util_tensor.py:706-727) first, then the mask (:395-422; none for the un-masked model).
        tvis: (measurement aid, bench.py's per-T_vis tables) mask exactly G - tvis groups instead of drawing the ratio."""
        enc = self.net.MAE_encoder
        steps = draw_corruption(self.net.corrupt_type, self.B)
        if self.masked:
            if tvis is None:
                mask, enc.mask_ratio = draw_mask(self.B, self.G, enc.mask_ratio, enc.rand_ratio)
            else:
                mask, _ = draw_mask(self.B, self.G, (self.G - tvis + 0.5) / self.G, 'False')
            enc.num_mask = int(enc.mask_ratio * self.G) if tvis is None else self.G - tvis
        n = steps.shape[0]
        slot = self.ring[self.slot]
        self.slot = (self.slot + 1) % self.RING
        if slot['done'] is not None:
            slot['done'].synchronize()                         # its previous copies have been consumed
        slot['steps'].zero_()
        slot['steps'][:, :, 1:4] = 1.0                         # identity 'multiply' steps
        if n:
            slot['steps'][:n].copy_(steps)
        tvis = self.G
        if self.masked:
            vis_rows, mask_rows = mask_row_ids(mask)
            nv, nm = vis_rows.numel(), mask_rows.numel()
            tvis = nv // self.B
            slot['vis'][:nv].copy_(vis_rows)
            slot['msk'][:nm].copy_(mask_rows)
            slot['vis32'][:nv].copy_(vis_rows)
            slot['msk32'][:nm].copy_(mask_rows)
            torch.cat([vis_rows.view(self.B, tvis), mask_rows.view(self.B, self.G - tvis)], dim=1,
                      out=slot['order'].view(self.B, self.G))
        self.stage.copy_(slot['raw'], non_blocking=True)       # ONE copy
        slot['done'] = torch.cuda.Event()
        slot['done'].record()
        return tvis

    def _gather(self, idx, written=()):
        """Gradients are produced as fresh tensors (autograd ASSIGNS them: no 203 accumulate-add
        launches, no memset of the flat buffer) and gathered into the flat gradient buffer with one
        multi-tensor copy.  written: indices of parameters whose gradient a Function already put
        into its flat view (the armed sink of nn_ops._sink_views): nothing to copy, nothing to zero."""
        m = self.model
        if written:
            idx = [i for i in idx if i not in written or m.params[i].grad is not None]
            for i in written:
                if m.params[i].grad is None:
                    m.params[i].grad = m.grad_views[i]
        _copy_into_views([(m.grad_views[i], m.params[i].grad) for i in idx], m.sink_strided)
        m.sink_strided = []
        for i in idx:
            m.params[i].grad = m.grad_views[i]

    def _phase1(self, tvis, cut=None):
        """forward + loss + backward; with `cut` (a dict) the backward stops at the patch tokens."""
        nv = self.B * tvis
        nm = self.B * (self.G - tvis)
        m = self.model
        for p in m.params:
            p.grad = None
        enc = self.net.MAE_encoder
        enc.grad_cut = cut
        # the blocks' weight gradients go straight into THIS model's flat buffer (nn_ops._sink_views): every .grad
        # is None here and nothing else touches the flat gradient views until the gather below
        from . import nn_ops
        m.sink_armed, m.sink_written, m.sink_strided = SINK, set(), []
        if WGRAD_SIDE and m.wgrad_stream is None:
            m.wgrad_stream = torch.cuda.Stream()
        try:
            lx, ln = m(self.pts, self.pts, steps=self.steps,
                       rows=(self.vis[:nv], self.msk[:nm], self.vis32[:nv], self.msk32[:nm], self.order))
            loss = self._mix(lx, ln)
            # the 34 LayerNorm backward launches park their parameter-gradient partials; ONE launch adds them
            # all after the backward (include/pdae.h: deferred reductions) -- nothing reads those gradients
            # before the gather below
            if SINK:
                _lib.deferred_begin()
            try:
                # (the seed of the backward pass is a constant: autograd's own ones_like is a fill launch per step)
                loss.backward(self.seed if self.seed.shape == loss.shape and self.seed.dtype == loss.dtype else None)
                nn_ops.flush_wgrad_queue(m)        # (stacks flush themselves; this catches a queue left by a cut)
                nn_ops.join_wgrad_stream(m)
            finally:
                m.wgrad_queue = []
                if SINK:
                    _lib.deferred_flush(lx)
        finally:
            enc.grad_cut = None
            m.sink_armed = False
        self._gather(self.early_idx if cut is not None else range(len(m.params)), m.sink_written)
        return lx.detach(), ln.detach()

    def _phase2(self, cut):
        """the patch embedder's backward from the token gradient phase 1 left in cut['leaf'].grad"""
        if SINK:
            _lib.deferred_begin()
        try:
            cut['tokens'].backward(cut['leaf'].grad)
        finally:
            if SINK:
                _lib.deferred_flush(self.pts)
        self._gather(self.late_idx)
        cut.clear()

    def _fwd_bwd(self, tvis):
        if not self.split:
            return self._phase1(tvis)
        cut = {}
        out = self._phase1(tvis, cut)
        self._phase2(cut)
        return out

    def _capture(self, tvis):
        sync = self.model.require_sync
        self.model.require_sync = False
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), _KeepBNState(self.net):  # warm-up on a side stream (PyTorch recipe)
            self._fwd_bwd(tvis)
        torch.cuda.current_stream().wait_stream(side)
        if self.pool is None:
            self.pool = torch.cuda.graph_pool_handle()
        # thread_local: only this thread's calls are checked against the capture (the RCCL
        # watchdog thread of torch.distributed queries events while we capture)
        g = torch.cuda.CUDAGraph()
        if not self.split:
            with torch.cuda.graph(g, pool=self.pool, capture_error_mode='thread_local'):
                out = self._phase1(tvis)
        else:
            # two graphs out of one pool, always replayed back to back in capture order: what
            # graph 2 reads (the embedder's saved activations, the token gradient) stays allocated
            # in the pool between the captures because `cut` holds it
            cut = {}
            with torch.cuda.graph(g, pool=self.pool, capture_error_mode='thread_local'):
                out = self._phase1(tvis, cut)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=self.pool, capture_error_mode='thread_local'):
                self._phase2(cut)
            g = (g, g2)
        self.model.require_sync = sync
        self.graphs[tvis], self.outputs[tvis] = g, out
        return g

    def _step_split(self, run1, run2, closing=True):
        """phase 1, start the big all-reduce, phase 2 under it, the two small slices, wait.  closing False: a
        gradient-accumulation micro-step (nothing is reduced)."""
        m = self.model
        out = run1()
        works = []
        reduce = closing and m.world_size > 1
        if self.accum is not None and closing:
            a, b = m.early_range
            m.flat_grad[a:b].add_(self.accum[a:b])
        if reduce:
            works.append((m.early_range, _start_average(m, *m.early_range)))
        run2()
        if self.accum is not None and closing:
            for a, b in m.late_ranges:
                m.flat_grad[a:b].add_(self.accum[a:b])
        if reduce:
            works += [((a, b), _start_average(m, a, b)) for a, b in m.late_ranges]
            for (a, b), (w, div) in works:
                w.wait()
                if div:                    # this slice came back as a SUM (no ReduceOp.AVG): divide IT, not the buffer
                    m.flat_grad[a:b].div_(m.world_size)
        return out

    def __call__(self, points, gt=None):
        self.pts.copy_(points[:, :, :3], non_blocking=True)
        tvis = self._draw()
        self.last_tvis = tvis
        self.micro += 1
        closing = self.micro >= self.spu
        sync = self.model.require_sync
        self.model.require_sync = False                        # the bucket hooks stay out of this path
        try:
            if self.eager_left > 0:                            # first steps eager: library init
                self.eager_left -= 1
                if self.split:
                    cut = {}
                    out = self._step_split(lambda: self._phase1(tvis, cut), lambda: self._phase2(cut), closing)
                else:
                    out = self._phase1(tvis)
            else:
                g = self.graphs.get(tvis)
                if g is None:
                    g = self._capture(tvis)
                if self.split:
                    self._step_split(g[0].replay, g[1].replay, closing)
                else:
                    g.replay()
                out = self.outputs[tvis]
        finally:
            self.model.require_sync = sync
        if not closing:
            self.accum.add_(self.model.flat_grad)              # every replay ASSIGNS the flat gradient: sum it here
            return out
        self.micro = 0
        if not self.split:
            if self.accum is not None:
                self.model.flat_grad.add_(self.accum)
            if self.model.world_size > 1:
                _average_gradients(self.model)
        if self.accum is not None:
            self.accum.zero_()
        self.optimizer.step()
        return out


class GraphedStaticStep:
    """Same idea for models whose step has static shapes and no host-side random
    draws (Point_CAE_PointNetv2: the corruption is applied by the data loader):
    ONE captured forward+loss+backward graph, replayed every step."""

    def __init__(self, model, optimizer, loss_mix, batch_size, npoints, warmup_eager=2, step_per_update=1):
        assert isinstance(model, FlatDataParallel)
        if getattr(model.module, 'draws_in_forward', False):
            raise NotImplementedError('GraphedStaticStep: this model draws a corruption on the host inside forward '
                                      '(dropout_global): a captured graph would replay one frozen draw; step it eagerly')
        self.model, self.optimizer, self.loss_mix = model, optimizer, loss_mix
        dev = model.flat_param.device
        self.corrupted = torch.zeros(batch_size, npoints, 3, device=dev)
        self.clean = torch.zeros(batch_size, npoints, 3, device=dev)
        self.graph, self.out = None, None
        _warn_if_null_stream()
        self.eager_left = warmup_eager
        self.spu, self.micro = int(step_per_update), 0
        self.accum = torch.zeros_like(model.flat_grad) if self.spu > 1 else None

    def invalidate(self):
        """Drop the captured graph (misc.BNMomentumScheduler: the momentum is a kernel argument); re-captured by the next step."""
        self.graph, self.out = None, None

    def reset_micro(self):
        self.micro = 0

    def _fwd_bwd(self):
        m = self.model
        for p in m.params:
            p.grad = None
        l1, l2 = m(self.corrupted, self.clean)
        self.loss_mix(l1, l2).backward()
        _copy_into_views([(v, p.grad) for p, v in zip(m.params, m.grad_views)])
        for p, v in zip(m.params, m.grad_views):
            p.grad = v
        return l1.detach(), l2.detach()

    def __call__(self, corrupted, clean):
        self.corrupted.copy_(corrupted[:, :, :3], non_blocking=True)
        self.clean.copy_(clean[:, :, :3], non_blocking=True)
        sync = self.model.require_sync
        self.model.require_sync = False
        if self.eager_left > 0:
            self.eager_left -= 1
            out = self._fwd_bwd()
        else:
            if self.graph is None:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side), _KeepBNState(self.model.module):
                    self._fwd_bwd()
                torch.cuda.current_stream().wait_stream(side)
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                    self.out = self._fwd_bwd()
            self.graph.replay()
            out = self.out
        self.model.require_sync = sync
        self.micro += 1
        if self.micro < self.spu:                                  # gradient-accumulation micro-step
            self.accum.add_(self.model.flat_grad)
            return out
        self.micro = 0
        if self.accum is not None:
            self.model.flat_grad.add_(self.accum)
            self.accum.zero_()
        if self.model.world_size > 1:
            _average_gradients(self.model)
        self.optimizer.step()
        return out
