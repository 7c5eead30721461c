"""Optional (off by default) Fourier amplitude perturbation layer -- the north_star's "multi-resolution Fourier
perturbation (amplitude-phase mix)".  The reference model contains no FFT, so this is a
build-defined extension: it never runs unless a caller attaches it (`model.fourier_perturb = FourierAmplitudeMix()`),
which keeps the default path reference-exact."""
import random

import torch
from torch import nn

from . import ops


class FourierAmplitudeMix(nn.Module):
    """Swap / blend the low- (or high-) band spectral amplitude of every feature map with that of another sample of
    the batch, keeping the phase.  radius in frequency bins (the reference's image-level HPF/LPF use r = 16,
    dataloaders.py:33, 68)."""

    def __init__(self, radius=16.0, lam=1.0, high=False, p=0.5):
        super().__init__()
        self.radius, self.lam, self.high, self.p = radius, lam, high, p

    def forward(self, x, perm=None):
        if perm is None:
            if not self.training or random.random() >= self.p:
                return x
            perm = torch.randperm(x.shape[0])
        return ops.fourier_amplitude_mix(x, perm, self.radius, self.lam, self.high)
