"""Soak: N optimisation steps, eager vs hipGraph replay, same seeds; prints the loss every 25 steps.
    python tools/soak.py graph|eager STEPS [B]      env SOAK_POKE=cpu|save_opt|...: a host action at step 50
    SOAK_NULL=1 SOAK_POKE=cpu python tools/soak.py graph 100   -> the NULL-stream failure (garbage losses)
    SOAK_MODEL=<registered model name>: another Transformer variant on the same YAML"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_dae_amd import builder
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.data_parallel import FlatDataParallel
from point_dae_amd.graph_step import GraphedTrainStep
from point_dae_amd.runner_pretrain import train_step
from point_dae_amd.synthetic import shapenet_like_clouds
from point_dae_amd.misc import set_random_seed

mode, steps = sys.argv[1], int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
cfg.npoints = 1024
if os.environ.get('SOAK_MODEL'):                 # e.g. PointCAE_transformer_fc_global_folding_local (the published variant)
    cfg.model.NAME = os.environ['SOAK_MODEL']
dev = torch.device('cuda')
set_random_seed(0)
model = FlatDataParallel(builder.model_builder(cfg.model).to(dev))
opt, _ = builder.build_opti_sche(model, cfg)
model.train()
model.zero_grad()
pool = torch.from_numpy(shapenet_like_clouds(B * 16, 1024, seed=7)).to(dev).split(B)
if not os.environ.get('SOAK_NULL'):          # SOAK_NULL=1 reproduces the NULL-stream failure
    from point_dae_amd.graph_step import use_created_stream
    use_created_stream()
step = GraphedTrainStep(model, opt, cfg, B, 1024) if mode == 'graph' else None
acc = torch.zeros((), device=dev)
POKE = os.environ.get('SOAK_POKE', '')
for i in range(steps):
    if i == 50 and POKE:
        if POKE == 'sd': sd = model.module.state_dict()
        if POKE == 'save_model': torch.save(model.module.state_dict(), '/tmp/x.pth')
        if POKE == 'save_opt': torch.save(opt.state_dict(), '/tmp/x.pth')
        if POKE == 'cpu': z = model.flat_param.cpu()
        if POKE.startswith('cpuMB'): z = model.flat_param[:int(POKE[5:]) * 262144].cpu()
        if POKE == 'clonecpu': z = model.flat_param.clone().cpu()
        if POKE == 'pinned': z = torch.empty(model.flat_param.shape, pin_memory=True); z.copy_(model.flat_param); torch.cuda.synchronize()
        if POKE == 'other': z = torch.zeros(29000000, device=dev).cpu()
        if POKE == 'sync': torch.cuda.synchronize()
        if POKE == 'nullcpu':
            with torch.cuda.stream(torch.cuda.default_stream()): z = model.flat_param.cpu()
        if POKE == 'nullkernel':
            with torch.cuda.stream(torch.cuda.default_stream()): z = torch.zeros(1 << 20, device=dev) + 1
        if POKE == 'nullkernel_sync':
            with torch.cuda.stream(torch.cuda.default_stream()): z = torch.zeros(1 << 20, device=dev) + 1
            torch.cuda.synchronize()
        if POKE == 'sleep': import time; time.sleep(2)
        if POKE == 'alloc': z = torch.empty(1 << 28, device=dev); del z
        if POKE == 'empty': torch.cuda.empty_cache()
        print('poked', POKE, flush=True)
    x = pool[i % len(pool)]
    lx, _ = step(x) if step is not None else train_step(model, opt, cfg, x, x)
    acc += lx.reshape(())
    if os.environ.get('SOAK_TRACE') and 45 <= i < 62:
        print('  step', i, 'T', getattr(step, 'last_tvis', None), 'loss %.5f' % lx.item(), flush=True)
    if (i + 1) % 25 == 0:
        print(mode, i + 1, 'loss*1000 = %.3f' % (acc.item() / 25 * 1000), 'param absmax %.3f' % model.flat_param.abs().max().item(), flush=True)
        acc.zero_()
