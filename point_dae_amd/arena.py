"""Pre-zeroed scratch for the small reduction outputs of one optimisation step."""
import torch


class ZeroArena:
    """One pre-zeroed device buffer per step for the small reduction outputs of
    the backward (LayerNorm dgamma/dbeta, bias gradients): `reset()` is ONE
    memset at the start of a step, kernels then accumulate into `take()`n slices
    (accumulate=1) instead of each issuing its own memset node (~110 per step)."""

    def __init__(self, numel=1 << 20):
        self.numel, self.buf, self.used = numel, None, 0

    def reset(self, device):
        if self.buf is None or self.buf.device != device:
            self.buf = torch.zeros(self.numel, device=device)
        else:
            self.buf.zero_()
        self.used = 0

    def take(self, n, like):
        """-> (zeroed float tensor of n elements, came_from_arena)"""
        n_pad = (n + 63) & ~63
        if self.buf is None or self.buf.device != like.device or self.used + n_pad > self.numel:
            return torch.zeros(n, device=like.device), True
        t = self.buf[self.used:self.used + n]
        self.used += n_pad
        return t, True


arena = ZeroArena()


def begin_step(device):
    """Call once at the start of a forward pass (the models do)."""
    arena.reset(device)
