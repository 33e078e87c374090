"""Helpers shared by the model parity tests (CPU oracle side and GPU product side)."""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
from weights import fill_state  # noqa: E402,F401


def load_fixture(name):
    return dict(np.load(os.path.join(HERE, 'golden', name), allow_pickle=False))


def model_cfg(fx):
    """cfg3 model section with the fixture's overrides applied."""
    from point_dae_amd.config import cfg_from_yaml_file
    root = os.path.dirname(HERE)
    cfg = cfg_from_yaml_file(os.path.join(
        root, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml')).model
    for k, v in ast.literal_eval(str(fx['overrides'])):
        node = cfg
        parts = k.split('.')
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    return cfg


def grad_sample(g, n=256):
    flat = g.detach().reshape(-1).cpu()
    idx = np.linspace(0, flat.numel() - 1, min(n, flat.numel())).astype(np.int64)
    return flat[idx].numpy()


def check_grads(model, fx, rtol, what, atol=2e-5, spike=None, max_spikes=0, named=None):
    """Every parameter gradient against the reference's: L2 norm, plus the full
    tensor (small parameters) or 256 evenly spaced entries (large ones).  `atol`
    covers gradients that are analytically zero (a conv bias feeding a training
    mode BatchNorm) and therefore pure rounding noise on both sides.
    spike / max_spikes: models whose max-pools sit on exact ties (the set-abstraction levels: ReLU clamps
    many rows of a group to 0) re-route one gradient element when a last-bit difference flips a tied
    arg-max -- in the reference as much as here.  Up to `max_spikes` tensors may then miss `rtol` in the
    max norm as long as they stay within `spike`; the L2 norms keep `rtol`.
    named: {parameter name: its own rtol} for tensors with a stated reason to be looser than the rest."""
    worst, spikes = 0.0, []
    base_rtol = rtol
    for name, p in model.named_parameters():
        key = 'grad/' + name
        rtol = (named or {}).get(name, base_rtol)
        ref_norm = float(fx[key + '/norm'])
        if p.grad is None:                      # an unused parameter: the reference leaves it without a gradient too
            p.grad = torch.zeros_like(p)
        got_norm = p.grad.double().norm().item()
        assert abs(got_norm - ref_norm) <= rtol * ref_norm + atol, (what, name, got_norm, ref_norm)
        if key + '/full' in fx:
            ref, got = fx[key + '/full'], p.grad.detach().cpu().numpy()
        else:
            ref, got = fx[key + '/sample'], grad_sample(p.grad)
        scale = np.abs(ref).max()
        diff = np.abs(got - ref).max()
        worst = max(worst, diff / max(scale, atol))
        if diff > rtol * scale + atol and spike is not None and diff <= spike * scale + atol:
            spikes.append((name, float(diff / scale)))
            continue
        assert diff <= rtol * scale + atol, (what, name, diff, scale)
    assert len(spikes) <= max_spikes, (what, spikes)
    return worst
