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


class MultiResolutionFourier(nn.Module):
    """The north_star's "multi-resolution Fourier perturbation (amplitude-phase mix) injected into the encoder": one
    FourierAmplitudeMix per encoder resolution -- after the stem (1/4 of the input: 192 x 192 planes at 768 x 768), after
    layer1 (1/4) and after layer2 (1/8: 96 x 96) -- all mixing with the SAME partner permutation of the batch in one
    forward (the partner's low-band amplitude, i.e. its "style", replaces the sample's at every scale; the phase, i.e. the
    layout, stays).  radii default to the reference's image-level r = 16 (dataloaders.py:33, 68) at 1/4 resolution and
    r = 8 at 1/8 (the same fraction of the spectrum).  BUILD-DEFINED, parity unpinned: the reference model has no FFT
    (SURVEY.md section 0); the oracle restates it with torch.fft (oracle/mrfp_oracle.py::mrfp_forward(fourier=...)).

    Attach with `model.fourier_perturb = MultiResolutionFourier()`; MRFPPlus.forward calls begin() once and
    at(level, x) at each attach point.  Like the other perturbations it only acts when forward is called with
    training=True."""

    LEVELS = ("stem", "layer1", "layer2")

    def __init__(self, radii=(16.0, 16.0, 8.0), lam=1.0, high=False, p=0.5, levels=LEVELS):
        super().__init__()
        self.levels = tuple(levels)
        self.radii = dict(zip(self.LEVELS, radii))
        self.lam, self.high, self.p = lam, high, p
        self.perm = None            # injected permutation (tests / benchmarks); None = drawn per forward
        self._cur = None

    def begin(self, batch, training):
        """Draws (or takes the injected) partner permutation for this forward; None = perturbation off this time."""
        if not training:
            self._cur = None
        elif self.perm is not None:
            self._cur = self.perm
        elif random.random() < self.p:
            self._cur = torch.randperm(batch)
        else:
            self._cur = None

    def at(self, level, x):
        if self._cur is None or level not in self.levels:
            return x
        return ops.fourier_amplitude_mix(x, self._cur, self.radii[level], self.lam, self.high)

    def spec(self):
        """{level: (radius, lam, high)} for the oracle."""
        return {lv: (self.radii[lv], self.lam, self.high) for lv in self.levels}
