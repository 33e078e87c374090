"""AdamW over the flat parameter buffer of FlatDataParallel.

Same update as torch.optim.AdamW with the reference's two parameter groups
(tools/builder.py:41-101: no weight decay for 1-D tensors, '.bias' and 'token'),
run as two launches of the fused kernel (csrc/adamw.hip) instead of ~35
multi-tensor launches over 203 tensors.  Exposes `param_groups` (so the
reference's schedulers drive `lr` unchanged), `step`, `zero_grad`,
`state_dict` / `load_state_dict`.
"""
import torch

from . import _lib


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

    def state_dict(self):
        return {'state': {'step': self.steps, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq},
                'param_groups': [{k: v for k, v in g.items() if k != 'params'} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.steps = int(sd['state']['step'])
        self.exp_avg.copy_(sd['state']['exp_avg'])
        self.exp_avg_sq.copy_(sd['state']['exp_avg_sq'])
        for g, s in zip(self.param_groups, sd['param_groups']):
            g.update({k: v for k, v in s.items() if k != 'range'})
