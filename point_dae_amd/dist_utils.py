"""Process-group helpers (utils/dist_utils.py of the reference): one process per
GPU, `backend='nccl'` is RCCL over xGMI on ROCm; `gloo` is used by the CPU
tests of the multi-rank path."""
import os

import torch
from torch import distributed as dist


def init_dist(launcher, backend='nccl', **kwargs):
    if launcher != 'pytorch':
        raise ValueError(f'Invalid launcher type: {launcher}')
    local_rank = int(os.environ['LOCAL_RANK'])
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, **kwargs)


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def reduce_tensor(tensor, args):
    """mean over ranks of a (0-d) tensor, for logging (dist_utils.py:46-53)."""
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= args.world_size
    return rt


def gather_tensor(tensor, args):
    out = [tensor.clone() for _ in range(args.world_size)]
    dist.all_gather(out, tensor)
    return torch.cat(out, dim=0)
