"""Synthetic stand-in of the ShapeNet-55 pretraining set.

The reference item is (taxonomy_id, model_id, corrupted (N,C), clean (N,C))
(datasets/ShapeNet55Dataset.py:90-119).  No dataset ships with the image, so
batches are generated on the host once and kept resident on the device; each
rank seeds with seed + rank like the reference's per-rank seeding
(main.py:78-81)."""
import torch

from .registry import DATASETS
from .synthetic import shapenet_like_clouds


@DATASETS.register_module()
class ShapeNet:
    def __init__(self, config):
        self.npoints = config.get('npoints', 1024)
        self.bs = config.get('bs', 128)
        self.steps = config.get('steps_per_epoch', 50)
        self.seed = config.get('seed', 0)
        self.device = config.get('device', 'cuda')
        self.pool = config.get('pool', 4)
        self._batches = None

    def _materialise(self):
        clouds = shapenet_like_clouds(self.bs * self.pool, self.npoints, seed=self.seed)
        x = torch.from_numpy(clouds).to(self.device)
        self._batches = list(x.split(self.bs))

    def __len__(self):
        return self.steps

    def __iter__(self):
        if self._batches is None:
            self._materialise()
        for i in range(self.steps):
            clean = self._batches[i % self.pool]
            yield 'synthetic', i, clean, clean
