"""Pre-zeroed scratch for the small reduction outputs of one optimisation step.

Aliasing contract.  A slice handed out by `take()` is valid until the next `begin_step()`: it is
zeroed there and handed out again by the next backward.  The autograd Functions return such slices
as parameter gradients (bias, LayerNorm, BatchNorm gradients), and autograd may keep the returned
tensor itself as `p.grad`.  That is only sound when the consumer copies the gradients out before the
next forward -- FlatDataParallel does (its .grad are views of the flat buffer, autograd adds into
them; the graphed steps gather with one multi-tensor copy).  So the shared buffer is used only for
forwards entered through FlatDataParallel.forward (`lease()`); a bare model gets fresh zero-filled
tensors from `take()` -- one memset each, ~110 per step, but `p.grad` then owns its memory and
`zero_grad(set_to_none=False)`, gradient accumulation or a late read of `.grad` behave like torch.
"""
import torch


class ZeroArena:
    """One pre-zeroed device buffer per step for the small reduction outputs of
    the backward (LayerNorm dgamma/dbeta, bias gradients): `reset()` is ONE
    memset at the start of a step, kernels then accumulate into `take()`n slices
    (accumulate=1) instead of each issuing its own memset node (~110 per step)."""

    def __init__(self, numel=1 << 20):
        self.numel, self.buf, self.used = numel, None, 0
        self.shared = False            # this step's slices come from the shared buffer
        self._leased = False           # the next begin_step() may use the shared buffer

    def lease(self):
        """The caller copies every gradient out before its next forward (FlatDataParallel)."""
        self._leased = True

    def reset(self, device):
        self.shared, self._leased = self._leased, False
        self.used = 0
        if not self.shared:
            return
        if self.buf is None or self.buf.device != device:
            self.buf = torch.zeros(self.numel, device=device)
        else:
            self.buf.zero_()

    def take(self, n, like):
        """-> (zeroed float tensor of n elements, True)"""
        n_pad = (n + 63) & ~63
        if (not self.shared or self.buf is None or self.buf.device != like.device
                or self.used + n_pad > self.numel):
            return torch.zeros(n, device=like.device), True
        t = self.buf[self.used:self.used + n]
        self.used += n_pad
        return t, True


arena = ZeroArena()


def begin_step(device):
    """Call once at the start of a forward pass (the models do)."""
    arena.reset(device)
