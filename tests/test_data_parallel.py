"""CPU tests of the multi-rank path: world_size-2 gloo processes drive
FlatDataParallel (flat parameter / gradient buffers, bucketed async all-reduce)
and must reproduce single-process training on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(16, 64), nn.LayerNorm(64), nn.GELU(), nn.Linear(64, 64), nn.GELU(),
                         nn.Linear(64, 8))


def _data():
    g = torch.Generator().manual_seed(1)
    return torch.randn(5, 8, 16, generator=g), torch.randn(5, 8, 8, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', GLOO_SOCKET_IFNAME='lo', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from point_dae_amd.data_parallel import FlatDataParallel
    net = _net()
    if rank == 1:                      # rank 1 starts from different weights: broadcast must fix it
        for p in net.parameters():
            p.data.add_(1.0)
    model = FlatDataParallel(net, bucket_mb=0.004)      # several buckets
    assert len(model.buckets) > 2
    opt = torch.optim.AdamW(model.param_groups(0.05), lr=1e-2)
    x, y = _data()
    per = x.shape[1] // world
    for step in range(x.shape[0]):
        xs, ys = x[step, rank * per:(rank + 1) * per], y[step, rank * per:(rank + 1) * per]
        loss = ((model(xs) - ys) ** 2).mean()
        loss.backward()
        model.finish()
        opt.step()
        model.zero_grad()
    if rank == 0:
        torch.save(model.flat_param.clone(), out)
    flat = [torch.empty_like(model.flat_param) for _ in range(world)]
    dist.all_gather(flat, model.flat_param)
    assert torch.equal(flat[0], flat[1])                 # replicas stay identical
    dist.destroy_process_group()


def test_flat_data_parallel_world2_matches_single_process(tmp_path):
    out = str(tmp_path / 'flat.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    from point_dae_amd.data_parallel import FlatDataParallel
    net = _net()
    model = FlatDataParallel(net)
    opt = torch.optim.AdamW(model.param_groups(0.05), lr=1e-2)
    x, y = _data()
    for step in range(x.shape[0]):
        # mean of per-rank means == mean over the whole batch (equal shards)
        loss = ((model(x[step]) - y[step]) ** 2).mean()
        loss.backward()
        model.finish()
        opt.step()
        model.zero_grad()
    got = torch.load(out)
    assert torch.allclose(got, model.flat_param, rtol=1e-5, atol=1e-6)


def test_flat_views_and_groups():
    from point_dae_amd.data_parallel import FlatDataParallel
    net = _net()
    ref = [p.detach().clone() for p in net.parameters()]
    model = FlatDataParallel(net)
    for p, r in zip(net.parameters(), ref):
        assert torch.equal(p, r)                                  # values preserved
        assert p.data_ptr() >= model.flat_param.data_ptr()
    groups = model.param_groups(0.05)
    assert groups[0]['weight_decay'] == 0. and all(p.dim() == 1 for p in groups[0]['params'])
    assert sum(p.numel() for p in groups[0]['params']) == model.no_decay_numel
    loss = net(torch.randn(4, 16)).sum()
    loss.backward()
    assert model.flat_grad.abs().sum() > 0
    for p in net.parameters():
        assert p.grad.data_ptr() >= model.flat_grad.data_ptr()     # grads are views of the flat buffer
    model.zero_grad()
    assert model.flat_grad.abs().sum() == 0


def test_late_parameters_sit_at_the_ends():
    """late_prefixes (the patch embedder in the product model): [late no-decay | early | late decay], the
    early range contiguous across the AdamW group boundary, optimiser groups unchanged as sets."""
    from point_dae_amd.data_parallel import FlatDataParallel
    net = _net()
    ref = {n: p.detach().clone() for n, p in net.named_parameters()}
    model = FlatDataParallel(net, late_prefixes=('0.', '1.'))      # the first Linear and the LayerNorm
    where = dict(zip(model.names, model.offsets))
    e0, e1 = model.early_range
    for n, (off, cnt) in where.items():
        late = n.startswith(('0.', '1.'))
        assert (e0 <= off and off + cnt <= e1) != late, n
        assert any(a <= off and off + cnt <= b for a, b in model.late_ranges) == late, n
        assert torch.equal(dict(net.named_parameters())[n], ref[n])
    assert model.late_ranges == [(0, 64 + 64 + 64), (model.flat_param.numel() - 16 * 64, model.flat_param.numel())]
    nd0, nd1 = model.no_decay_range
    d0, d1 = model.decay_range
    for n, (off, cnt) in where.items():
        one_d = dict(net.named_parameters())[n].dim() == 1
        assert (nd0 <= off and off + cnt <= nd1) if one_d else (d0 <= off and off + cnt <= d1), n
    groups = model.param_groups(0.05)
    assert all(p.dim() == 1 for p in groups[0]['params']) and all(p.dim() == 2 for p in groups[1]['params'])
    # default: the module's own declaration
    net2 = _net()
    net2.late_grad_prefixes = ('5.',)
    m2 = FlatDataParallel(net2)
    assert sum(b - a for a, b in m2.late_ranges) == 64 * 8 + 8
    assert FlatDataParallel(_net()).late_ranges == []


def _slices_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', GLOO_SOCKET_IFNAME='lo', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.graph_step import _average_gradients, _start_average
    model = FlatDataParallel(_net(), late_prefixes=('0.',))
    model.require_sync = False
    x, y = _data()
    ((model(x[0, rank * 4:(rank + 1) * 4]) - y[0, rank * 4:(rank + 1) * 4]) ** 2).mean().backward()
    whole = model.flat_grad.clone()
    # the split step's order: the early range first, then the late ends, one division at the end
    works = [_start_average(model, *model.early_range)] + [_start_average(model, a, b) for a, b in model.late_ranges]
    for w, _ in works:
        w.wait()
    assert all(div for _, div in works)          # gloo sums; RCCL would average in the collective
    model.flat_grad.div_(world)
    sliced = model.flat_grad.clone()
    model.flat_grad.copy_(whole)
    _average_gradients(model)
    if rank == 0:
        torch.save({'sliced': sliced, 'whole': model.flat_grad.clone()}, out)
    dist.destroy_process_group()


def test_sliced_allreduce_equals_one_flat_allreduce(tmp_path):
    out = str(tmp_path / 's.pt')
    mp.spawn(_slices_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert torch.equal(r['sliced'], r['whole']) and r['whole'].abs().sum() > 0


def _accum_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', GLOO_SOCKET_IFNAME='lo', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from point_dae_amd.data_parallel import FlatDataParallel
    model = FlatDataParallel(_net(), bucket_mb=0.004)
    x, y = _data()
    # two micro-steps per update (step_per_update = 2): the first only accumulates locally
    for micro in range(2):
        model.require_sync = micro == 1
        xs, ys = x[micro, rank * 4:(rank + 1) * 4], y[micro, rank * 4:(rank + 1) * 4]
        ((model(xs) - ys) ** 2).mean().backward()
    model.finish()
    if rank == 0:
        torch.save(model.flat_grad.clone(), out)
    dist.destroy_process_group()


def test_gradient_accumulation_world2(tmp_path):
    """step_per_update (tools/runner_pretrain.py:188-197): gradients of the micro-steps add up; only the last
    one reduces.  Two ranks x two micro-steps == the average over ranks of the summed micro-step gradients."""
    out = str(tmp_path / 'grad.pt')
    mp.spawn(_accum_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    from point_dae_amd.data_parallel import FlatDataParallel
    model = FlatDataParallel(_net())
    x, y = _data()
    for micro in range(2):
        for rank in range(2):
            xs, ys = x[micro, rank * 4:(rank + 1) * 4], y[micro, rank * 4:(rank + 1) * 4]
            (((model(xs) - ys) ** 2).mean() / 2).backward()
    assert torch.allclose(torch.load(out), model.flat_grad, rtol=1e-5, atol=1e-7)


def test_pretrain_key_remap():
    """models/Point_MAE.py:643-656: 'module.' stripped, 'MAE_encoder.' / 'base_model.' prefixes dropped."""
    from point_dae_amd.builder import remap_pretrain_keys
    sd = {'module.MAE_encoder.blocks.blocks.0.norm1.weight': 1, 'MAE_encoder.encoder.first_conv.0.weight': 2,
          'base_model.cls_head.weight': 3, 'mask_token': 4, 'module.MAE_decoder.norm.bias': 5}
    got = remap_pretrain_keys(sd)
    assert got == {'blocks.blocks.0.norm1.weight': 1, 'encoder.first_conv.0.weight': 2, 'cls_head.weight': 3,
                   'mask_token': 4, 'MAE_decoder.norm.bias': 5}
