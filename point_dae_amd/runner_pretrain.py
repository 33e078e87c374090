"""Pretraining loop (tools/runner_pretrain.py:50-288 of the reference).

Same control flow -- model(points, gt) -> (loss_xyz, loss_normal), loss mixing
by `loss_type` (:161-186), backward, optimiser step every `step_per_update`,
epoch-granular scheduler -- without its per-step host syncs: the reference
calls .item() twice and cuda.synchronize() every step (:201-217); here losses
accumulate on the device and are read back every `log_every` steps.
"""
import os
import time

import torch

from . import builder, dist_utils
from . import datasets  # noqa: F401  (registers the synthetic ShapeNet set)
from .data_parallel import FlatDataParallel
from .registry import DATASETS


def mix_loss(config, loss_xyz, loss_normal, gradual_weight):
    lt = config.loss_type
    w = float(config.normal_weight)
    if lt == 'xyz':
        return loss_xyz
    if lt == 'normal':
        return w * loss_normal
    if lt == 'xyznormal':
        return loss_xyz + w * loss_normal
    if lt in ('xyznormal_gradual', 'xyznormal_warm'):
        return loss_xyz + w * loss_normal * gradual_weight
    raise NotImplementedError(lt)


def gradual_weight_of(config, epoch):
    if config.loss_type == 'xyznormal_gradual':
        return float(epoch) / float(config.max_epoch)
    if config.loss_type == 'xyznormal_warm':
        r = float(epoch) / float(config.max_epoch)
        return r * 3 if r < 1.0 / 3.0 else 1.0
    return 0


def train_step(model, optimizer, config, points, gt, gradual_weight=0., update=True):
    """One optimisation step: the unit bench.py times.  update=False: a gradient-accumulation
    micro-step (runner_pretrain.py:188-197, `step_per_update`): backward only, gradients stay in the
    flat buffer, no collective; the step that closes the group reduces and updates."""
    sync = getattr(model, 'require_sync', True)
    if isinstance(model, FlatDataParallel) and not update:
        model.require_sync = False
    loss_xyz, loss_normal = model(points, gt)
    loss = mix_loss(config, loss_xyz, loss_normal.sum(), gradual_weight)
    loss.backward()
    if isinstance(model, FlatDataParallel):
        model.require_sync = sync
        if update:
            model.finish()
    if update:
        optimizer.step()
        model.zero_grad()
    return loss_xyz.detach(), loss_normal.detach()


def run_net(args, config, train_writer=None, val_writer=None, log=print, log_every=50):
    rank, world = dist_utils.get_dist_info()
    device = torch.device('cuda', torch.cuda.current_device())
    # Everything this process does on the GPU -- steps, logging reads, checkpoint copies -- goes to
    # ONE created stream (graph_step.use_created_stream explains why).
    from .graph_step import use_created_stream
    use_created_stream(device)
    ds_cfg = dict(config.dataset.train._base_)          # NAME, N_POINTS, PC_PATH, DATA_PATH (reference schema)
    ds_cfg.update(dict(config.dataset.train.others))
    ds_cfg.update(seed=args.seed + rank, shared_seed=args.seed, device=device, steps_per_epoch=getattr(args, 'steps_per_epoch', None),
                  rank=rank, world=world)
    ds_cfg.setdefault('bs', config.total_bs // world)
    train_loader = DATASETS.build(ds_cfg)

    base_model = builder.model_builder(config.model).to(device)
    start_epoch, best_metric = 0, 0.
    if args.resume:
        start_epoch, best_metric = builder.resume_model(base_model, args)   # runner_pretrain.py:69-72
    elif args.start_ckpts is not None:
        builder.load_model(base_model, args.start_ckpts)
    sync_bn = bool(args.sync_bn) and world > 1
    if sync_bn:
        # runner_pretrain.py:81-83.  The converted modules run the layer-by-layer embedder / set-abstraction paths
        # (patch_embed.patch_embed_layerwise), whose SyncBatchNorm modules issue their own collectives: eager steps.
        base_model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(base_model)
    spu = int(config.get('step_per_update', 1))     # gradient accumulation (runner_pretrain.py:188-197): eager steps
    model = FlatDataParallel(base_model)
    optimizer, scheduler = builder.build_opti_sche(model, config)
    model.zero_grad()
    # the step is replayed as hipGraphs (graph_step.py); eager launches only as a fallback
    # for configurations the graphed steps do not cover
    from .graph_step import GraphedStaticStep, GraphedTrainStep
    from .point_cae_dgcnn import Point_CAE_DGCNN_FCOnly
    from .point_cae_pointnetv2 import Point_CAE_PointNetv2
    from .point_cae_transformer import PointCAE_transformer
    bs = ds_cfg['bs']
    gw_dev = torch.zeros((), device=device)                 # gradual weight, read inside the captured graph
    graphed = None
    if sync_bn:
        step_fn = None
    elif isinstance(base_model, PointCAE_transformer):
        # every branch of the step body replays (graph_step.GraphedTrainStep): all loss types through a device
        # scalar, the un-masked model, gradient accumulation
        graphed = GraphedTrainStep(model, optimizer, config, bs, config.npoints, step_per_update=spu)
        step_fn = lambda corrupted, clean: graphed(clean)   # noqa: E731  (corrupted input unused on this path)
    elif isinstance(base_model, (Point_CAE_PointNetv2, Point_CAE_DGCNN_FCOnly)) and not base_model.draws_in_forward:
        w = float(config.normal_weight)
        mixes = {'xyz': lambda a, b: a, 'normal': lambda a, b: w * b, 'xyznormal': lambda a, b: a + w * b,
                 'xyznormal_gradual': lambda a, b: a + w * b * gw_dev, 'xyznormal_warm': lambda a, b: a + w * b * gw_dev}
        graphed = GraphedStaticStep(model, optimizer, mixes[config.loss_type], bs, config.npoints, step_per_update=spu)
        step_fn = graphed
    else:
        # a model that draws a corruption on the host inside forward (`dropout_global`) cannot be captured -- the
        # draw would be frozen into the graph, and the surviving point count changes from step to step
        step_fn = None
    if graphed is not None:
        for item in (scheduler if isinstance(scheduler, list) else []):
            if hasattr(item, 'listeners'):                  # misc.BNMomentumScheduler: a new momentum re-captures the graphs
                item.listeners.append(graphed.invalidate)
    if rank == 0:
        log('step: %s' % ('eager' if step_fn is None else 'hipGraph replay (%s, step_per_update %d, loss_type %s)' % (
            type(graphed).__name__, spu, config.loss_type)))

    # SVM-probe validation (runner_pretrain.py:263-270): labelled loaders, when the experiment has them
    val = None
    dcfg = config.dataset
    if dcfg.get('extra_train') is not None and dcfg.get('val') is not None and hasattr(base_model, 'forward'):
        import inspect
        if 'return_feat' in inspect.signature(base_model.forward).parameters:
            def _loader(node):
                c = dict(node._base_)
                c.update(dict(node.others))
                c.update(device=device, seed=args.seed, rank=rank, world=world)   # sharded: validate() all_gathers
                c.setdefault('bs', 32)
                return DATASETS.build(c)
            val = (_loader(dcfg.extra_train), _loader(dcfg.val))
    from .svm_probe import Acc_Metric, validate
    best_metrics, metrics = Acc_Metric(best_metric), Acc_Metric(0.)
    num_iter = 0

    for epoch in range(start_epoch, config.max_epoch + 1):
        model.train()
        num_iter = 0                                   # runner_pretrain.py:112: the micro-step counter restarts every epoch
        if graphed is not None and hasattr(graphed, 'reset_micro'):
            graphed.reset_micro()                      # (gradients accumulated so far are kept, as in the reference)
        if hasattr(train_loader, 'set_epoch'):
            train_loader.set_epoch(epoch)              # :115 sampler.set_epoch(epoch)
        gw = gradual_weight_of(config, epoch)
        gw_dev.fill_(float(gw))
        if isinstance(graphed, GraphedTrainStep):
            graphed.set_gradual_weight(gw)
        acc = torch.zeros(2, device=device)
        t0 = time.time()
        n = 0
        for idx, (_, _, corrupted, clean) in enumerate(train_loader):
            if step_fn is not None:
                lx, ln = step_fn(corrupted, clean)
            else:
                num_iter += 1
                lx, ln = train_step(model, optimizer, config, corrupted, clean, gw, update=num_iter == spu)
                if num_iter == spu:
                    num_iter = 0
            acc += torch.stack([lx.reshape(()), ln.sum().reshape(())])
            n += 1
            if (idx + 1) % log_every == 0 or idx + 1 == len(train_loader):
                vals = acc.clone()
                if world > 1:
                    vals = dist_utils.reduce_tensor(vals, args)
                vals = (vals / n * 1000).tolist()          # one host sync per log line
                dt = time.time() - t0
                if rank == 0:
                    log('[Epoch %d/%d][Batch %d/%d] %.1f clouds/s Lossxyz = %.4f Lossnormal = %.4f lr = %.6f' % (
                        epoch, config.max_epoch, idx + 1, len(train_loader),
                        n * clean.shape[0] * world / dt, vals[0], vals[1], optimizer.param_groups[0]['lr']))
        for item in (scheduler if isinstance(scheduler, list) else [scheduler]):      # runner_pretrain.py:237-241
            if item is not None:
                item.step(epoch)
        if val is not None and epoch % max(int(getattr(args, 'val_freq', 1)), 1) == 0:
            metrics = validate(base_model, val[0], val[1], epoch, config, log=log)
            if metrics.better_than(best_metrics):
                best_metrics = metrics
                builder.save_checkpoint(model, optimizer, epoch, metrics, best_metrics, 'ckpt-best', args)
        builder.save_checkpoint(model, optimizer, epoch, metrics, best_metrics, 'ckpt-last', args)
    return model
