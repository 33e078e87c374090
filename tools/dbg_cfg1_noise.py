"""How far are the cfg1 fixture gradients from (a) the row-GEMM path, (b) the library-GEMM path, and how far
are (a) and (b) from each other?  (B = 2 through six training-mode BatchNorms: ill-conditioned.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch, torch.nn.functional as F
from golden_util import load_fixture, fill_state, grad_sample
from point_dae_amd import nn_ops
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.point_cae_pointnetv2 import Point_CAE_PointNetv2
from point_dae_amd.graph_step import use_created_stream
use_created_stream()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fx = load_fixture('pointnetv2_cfg1_b2.npz')
cfg = cfg_from_yaml_file(os.path.join(root, 'cfgs', 'pretrain_PointCAE_clean.yaml')).model


def run(lib):
    """lib: False = row GEMMs everywhere, True = library everywhere, 'enc' = library in the SA encoder only,
    'fold' = library in the folding heads only"""
    orig = nn_ops.linear_any
    libf = lambda x, w, b=None, relu=False: (F.relu(F.linear(x, w, b)) if relu else F.linear(x, w, b))
    if lib is True:
        nn_ops.linear_any = libf
    elif lib == 'noise':
        import inspect
        g = torch.Generator(device='cuda').manual_seed(1)

        def pick(x, w, b=None, relu=False):
            y = libf(x, w, b, relu)
            if any(f.function == 'rows' for f in inspect.stack()[1:4]):      # library result, last bit jittered
                y = y * (1 + 6e-8 * (torch.randint(0, 3, y.shape, device=y.device, generator=g) - 1))
            return y
        nn_ops.linear_any = pick
    elif lib in ('fwdmine', 'bwdmine'):
        class Mixed(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                ctx.save_for_backward(x, w)
                return nn_ops.rows_gemm(x.contiguous(), w) if lib == 'fwdmine' else F.linear(x, w)

            @staticmethod
            def backward(ctx, dy):
                x, w = ctx.saved_tensors
                dy = dy.contiguous()
                if lib == 'fwdmine':
                    return (dy @ w if ctx.needs_input_grad[0] else None), dy.t() @ x
                dx = nn_ops.rows_gemm(dy, w, True) if ctx.needs_input_grad[0] else None
                return dx, nn_ops.rows_wgrad([dy], [x], [False])[0][0]
        import inspect

        def pick(x, w, b=None, relu=False):
            in_enc = any(f.function == 'rows' for f in inspect.stack()[1:4])
            return Mixed.apply(x, w) if in_enc else libf(x, w, b, relu)
        nn_ops.linear_any = pick
    elif lib in ('enc', 'fold'):
        import inspect

        def pick(x, w, b=None, relu=False):
            in_enc = any(f.function == 'rows' for f in inspect.stack()[1:4])
            return libf(x, w, b, relu) if in_enc == (lib == 'enc') else orig(x, w, b, relu)
        nn_ops.linear_any = pick
    model = fill_state(Point_CAE_PointNetv2(cfg), int(fx['seed'])).cuda().train()
    lc, lf = model(torch.from_numpy(fx['corrupted']).cuda(), torch.from_numpy(fx['clean']).cuda())
    (lc + 0.5 * lf).backward()
    nn_ops.linear_any = orig
    return {n: p.grad.detach().clone() for n, p in model.named_parameters()}, lc.item(), lf.item()


mode = sys.argv[1] if len(sys.argv) > 1 else 'rows'
ga, la, _ = run({'rows': False}.get(mode, mode))
gb, lb, _ = run(True)
print('mode', mode)
print('loss coarse: rows %.9f lib %.9f fixture %.9f' % (la, lb, float(fx['loss_coarse'])))
rows = []
for n in ga:
    key = 'grad/' + n
    if key + '/full' in fx:
        ref = fx[key + '/full']; a = ga[n].cpu().numpy(); b = gb[n].cpu().numpy()
    else:
        ref = fx[key + '/sample']; a = grad_sample(ga[n]); b = grad_sample(gb[n])
    sc = max(np.abs(ref).max(), 2e-5)
    rows.append((np.abs(a - ref).max() / sc, np.abs(b - ref).max() / sc, np.abs(a - b).max() / sc, n))
rows.sort(reverse=True)
print('  rows-vs-fixture  lib-vs-fixture  rows-vs-lib   parameter')
for r in rows[:10]:
    print('  %.2e         %.2e        %.2e    %s' % r)
