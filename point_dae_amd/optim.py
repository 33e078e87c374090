"""AdamW over the flat parameter buffer of FlatDataParallel.

Same update as torch.optim.AdamW with the reference's two parameter groups
(tools/builder.py:41-101: no weight decay for 1-D tensors, '.bias' and 'token'),
run as two launches of the fused kernel (csrc/adamw.hip) instead of ~35
multi-tensor launches over 203 tensors.  Exposes `param_groups` (so the
reference's schedulers drive `lr` unchanged), `step`, `zero_grad`,
`state_dict` / `load_state_dict` in torch.optim.AdamW's layout: per-parameter `step` /
`exp_avg` / `exp_avg_sq` entries numbered the way the reference's groups number them
(module.named_parameters() order inside each group), so a checkpoint moves between this
optimiser and torch.optim.AdamW over tools/builder.py's groups, and does not depend on how
FlatDataParallel lays the flat buffer out.
"""
import torch

from . import _lib


_TORCH_GROUP_DEFAULTS = dict(amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False,
                             fused=None, decoupled_weight_decay=True)


class FlatAdamW:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.model = model
        flat = model.flat_param
        if not flat.is_cuda:
            raise RuntimeError('FlatAdamW needs the parameters on the GPU (no CPU path)')
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.steps = 0
        nd0, nd1 = model.no_decay_range
        d0, d1 = model.decay_range
        groups = model.param_groups(weight_decay)
        self.param_groups = [
            dict(groups[0], lr=lr, betas=betas, eps=eps, range=(nd0, nd1)),
            dict(groups[1], lr=lr, betas=betas, eps=eps, range=(d0, d1)),
        ]
        for g in self.param_groups:
            g.setdefault('initial_lr', lr)

    def step(self):
        self.steps += 1
        m = self.model
        for g in self.param_groups:
            a, b = g['range']
            if b <= a:
                continue
            _lib.call('pdae_adamw_step', m.flat_param, b - a, m.flat_param[a:].data_ptr(),
                      m.flat_grad[a:].data_ptr(), self.exp_avg[a:].data_ptr(), self.exp_avg_sq[a:].data_ptr(),
                      float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']),
                      float(g['weight_decay']), self.steps)

    def zero_grad(self, set_to_none=False):
        self.model.zero_grad()

    def _torch_order(self):
        """[(torch id, flat offset, numel, shape)] -- ids follow the reference's groups: no-decay
        parameters first, then the decayed ones, each in named_parameters() order."""
        m = self.model
        where = {n: (off, cnt, p.shape) for n, (off, cnt), p in zip(m.names, m.offsets, m.params)}
        named = [n for n, p in m.module.named_parameters() if p.requires_grad]
        k = len(self.param_groups[0]['params'])
        nd_names = set(m.names[:k])
        order = [n for n in named if n in nd_names] + [n for n in named if n not in nd_names]
        return [(i,) + where[n] for i, n in enumerate(order)], k

    def state_dict(self):
        order, k = self._torch_order()
        state = {}
        if self.steps > 0:
            for i, off, cnt, shape in order:
                state[i] = {'step': torch.tensor(float(self.steps)),
                            'exp_avg': self.exp_avg[off:off + cnt].reshape(shape).clone(),
                            'exp_avg_sq': self.exp_avg_sq[off:off + cnt].reshape(shape).clone()}
        groups = []
        for gi, g in enumerate(self.param_groups):
            d = {key: v for key, v in g.items() if key not in ('params', 'range')}
            for key, v in _TORCH_GROUP_DEFAULTS.items():      # so torch.optim.AdamW can load the groups
                d.setdefault(key, v)
            d['params'] = list(range(0, k)) if gi == 0 else list(range(k, len(order)))
            groups.append(d)
        return {'state': state, 'param_groups': groups}

    def load_state_dict(self, sd):
        order, _ = self._torch_order()
        state = sd['state']
        if state and 'exp_avg' in state and not isinstance(state['exp_avg'], dict):
            raise RuntimeError('FlatAdamW.load_state_dict: flat-layout optimiser state of an earlier build; '
                               'checkpoints now use torch.optim.AdamW\'s per-parameter layout')
        self.exp_avg.zero_(), self.exp_avg_sq.zero_()
        steps = set()
        for i, off, cnt, shape in order:
            st = state.get(i, state.get(str(i)))
            if st is None:
                continue
            self.exp_avg[off:off + cnt].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[off:off + cnt].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(st['step']))
        if len(steps) > 1:
            raise RuntimeError('FlatAdamW.load_state_dict: parameters with different step counts %s '
                               '(the fused update keeps one counter)' % sorted(steps))
        self.steps = steps.pop() if steps else 0
        for g, saved in zip(self.param_groups, sd['param_groups']):
            g.update({key: v for key, v in saved.items()
                      if key not in ('params', 'range') and key not in _TORCH_GROUP_DEFAULTS})
