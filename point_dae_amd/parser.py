"""Command line of main.py: the flags of the reference's utils/parser.py:7-104
that matter on the pretraining path, same names and defaults."""
import argparse
import os
from pathlib import Path


def get_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--config', type=str, required=True, help='yaml config file')
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none', help='job launcher')
    p.add_argument('--local_rank', type=int, default=0)
    p.add_argument('--num_workers', type=int, default=8)
    p.add_argument('--seed', type=int, default=0, help='random seed')
    p.add_argument('--deterministic', action='store_true')
    p.add_argument('--sync_bn', action='store_true', default=False, help='whether to use sync bn')
    p.add_argument('--exp_name', type=str, default='default', help='experiment name')
    p.add_argument('--root_folder', type=str, default='experiments')
    p.add_argument('--model_name', type=str, default='none', help='overrides model.NAME')
    p.add_argument('--start_ckpts', type=str, default=None, help='reload used ckpt path')
    p.add_argument('--ckpts', type=str, default=None)
    p.add_argument('--val_freq', type=int, default=1, help='test freq')
    p.add_argument('--resume', action='store_true', default=False)
    p.add_argument('--total_bs', type=int, default=-1)
    # synthetic-data controls (no dataset ships with this repo)
    p.add_argument('--max_epoch', type=int, default=-1, help='override config.max_epoch')
    p.add_argument('--steps_per_epoch', type=int, default=None,
                   help='batches per epoch; default for a listed .npy set: the reference\'s DataLoader length -- '
                        'floor(ceil(N / world) / bs) for the training subset (drop_last, tools/builder.py:21,28), ceil '
                        'otherwise; 50 for the synthetic set')
    args = p.parse_args(argv)
    if args.resume and args.start_ckpts is not None:
        raise ValueError('--resume and --start_ckpts cannot be both activate')
    if 'LOCAL_RANK' not in os.environ:
        os.environ['LOCAL_RANK'] = str(args.local_rank)
    else:
        args.local_rank = int(os.environ['LOCAL_RANK'])
    stem = Path(args.config).stem + args.model_name
    args.experiment_path = os.path.join('./' + args.root_folder, stem, Path(args.config).parent.stem, args.exp_name)
    args.tfboard_path = os.path.join('./' + args.root_folder, stem, Path(args.config).parent.stem, 'TFBoard', args.exp_name)
    args.log_name = Path(args.config).stem
    if args.local_rank == 0:
        os.makedirs(args.experiment_path, exist_ok=True)
    return args
