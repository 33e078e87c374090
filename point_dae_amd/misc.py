"""utils/misc.py of the reference: fps(), set_random_seed()."""
import random

import numpy as np
import torch

from . import pointnet2_utils


def fps(data, number):
    """data (B,N,3|6) -> (idx (B,number) i32, data[idx] (B,number,C)); utils/misc.py:13-20."""
    xyz = data[:, :, :3].contiguous()
    if data.shape[2] == 3:
        return pointnet2_utils.furthest_point_sample_with_centres(xyz, number)
    idx = pointnet2_utils.furthest_point_sample(xyz, number)
    out = pointnet2_utils.gather_operation(data.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    return idx, out


def set_random_seed(seed, deterministic=False):
    """utils/misc.py:42-66: python, numpy, torch (CPU + all GPUs)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    if deterministic:
        # the reference's --deterministic pins cuDNN (utils/misc.py:56-62); here the hand-written kernels'
        # atomic reductions become ordered ones (include/pdae.h: pdae_set_deterministic)
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
        if torch.cuda.is_available():
            from . import _lib
            if not _lib.deterministic():
                _lib.set_deterministic(True)
