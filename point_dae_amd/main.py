"""Entry point: `python -m point_dae_amd.main --config cfgs/X.yaml [--launcher pytorch]`
(main.py:16-111 of the reference, pretraining dispatch only)."""
import torch

from . import dist_utils, parser
from .config import get_config
from .misc import set_random_seed
from .runner_pretrain import run_net


def main(argv=None):
    args = parser.get_args(argv)
    args.use_gpu = torch.cuda.is_available()
    if not args.use_gpu:
        raise RuntimeError('point_dae_amd needs an MI355X: there is no CPU path')
    if args.launcher == 'none':
        args.distributed = False
        args.world_size = 1
        torch.cuda.set_device(args.local_rank)
    else:
        args.distributed = True
        import os
        dist_utils.init_dist(args.launcher, backend=os.environ.get('PDAE_DIST_BACKEND', 'nccl'))
        _, args.world_size = dist_utils.get_dist_info()
    config = get_config(args)
    if args.model_name != 'none':
        config.model.NAME = args.model_name
    if args.total_bs != -1:
        config.total_bs = args.total_bs
    if args.max_epoch != -1:
        config.max_epoch = args.max_epoch
    if len(config.model['corrupt_type']) == 0:          # main.py:51-55
        config.model['corrupt_type'] = config.dataset['train']['others']['corrupt_type']
    assert config.total_bs % args.world_size == 0
    config.dataset.train.others.bs = config.total_bs // args.world_size
    set_random_seed(args.seed + args.local_rank, deterministic=args.deterministic)   # main.py:78-81
    run_net(args, config)


if __name__ == '__main__':
    main()
