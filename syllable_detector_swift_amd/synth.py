"""Synthetic multichannel audio of BASELINE.md section 3 (no dataset is available offline).

Per channel c: fp32, `fs` Hz, 0.05*N(0,1) noise plus an 80 ms FM-tone burst
0.3*sin(2*pi*(3000 + 1500*sin(2*pi*3*t))*t) every 500 ms, clipped to [-1, 1],
RNG seed 1000 + c.  Channel-major [C][S].
"""
from __future__ import annotations

import numpy as np

NOISE_RMS = 0.05
BURST_AMPLITUDE = 0.3
BURST_SECONDS = 0.080
BURST_PERIOD_SECONDS = 0.500
SEED_BASE = 1000


def _burst(t):
    return BURST_AMPLITUDE * np.sin(2.0 * np.pi * (3000.0 + 1500.0 * np.sin(2.0 * np.pi * 3.0 * t)) * t)


def channel(n_samples: int, c: int = 0, fs: float = 44100.0) -> np.ndarray:
    """One channel, generated on the host with numpy's PCG64 (used by tests and fixtures)."""
    rng = np.random.default_rng(SEED_BASE + c)
    x = NOISE_RMS * rng.standard_normal(n_samples)
    t = np.arange(n_samples, dtype=np.float64) / fs
    on = np.mod(t, BURST_PERIOD_SECONDS) < BURST_SECONDS
    x[on] += _burst(t[on])
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def channels(n_channels: int, n_samples: int, first: int = 0, fs: float = 44100.0) -> np.ndarray:
    return np.stack([channel(n_samples, first + c, fs) for c in range(n_channels)], axis=0)


def channels_on_device(n_channels: int, n_samples: int, device, first: int = 0, fs: float = 44100.0,
                       block: int = 8):
    """Same signal model generated directly in HBM with torch (bench sizes: GiBs).
    The noise comes from torch's generator seeded per channel block, so values differ from
    `channel()`; the distribution and burst schedule are the same."""
    import torch
    out = torch.empty((n_channels, n_samples), dtype=torch.float32, device=device)
    t = torch.arange(n_samples, dtype=torch.float64, device=device) / fs
    on = torch.remainder(t, BURST_PERIOD_SECONDS) < BURST_SECONDS
    burst = torch.where(on, BURST_AMPLITUDE * torch.sin(2.0 * torch.pi * (3000.0 + 1500.0 * torch.sin(2.0 * torch.pi * 3.0 * t)) * t),
                        torch.zeros((), dtype=torch.float64, device=device)).to(torch.float32)
    del t, on
    gen = torch.Generator(device=device)
    for c0 in range(0, n_channels, block):
        c1 = min(n_channels, c0 + block)
        gen.manual_seed(SEED_BASE + first + c0)
        noise = torch.randn((c1 - c0, n_samples), generator=gen, dtype=torch.float32, device=device)
        out[c0:c1] = torch.clamp(noise * NOISE_RMS + burst, -1.0, 1.0)
        del noise
    return out


def syllable(template: np.ndarray, hop: int, window: int, f0: int, fourier_length: int, rng, amplitude: float = 0.5) -> np.ndarray:
    """Audio whose band-limited spectrogram follows `template` [T][F]: one sinusoid per bin
    f0+f with the template column values as a piecewise-linear envelope over frame centres."""
    T, F = template.shape
    n = (T - 1) * hop + window
    t = np.arange(n, dtype=np.float64)
    centres = np.arange(T) * hop + window / 2.0
    x = np.zeros(n)
    for f in range(F):
        env = np.interp(t, centres, template[:, f])
        x += env * np.sin(2.0 * np.pi * (f0 + f) / fourier_length * t + rng.uniform(0.0, 2.0 * np.pi))
    return amplitude * x / np.abs(x).max()


def syllable_channel(n_samples: int, template: np.ndarray, seed: int, hop: int = 132, window: int = 256, f0: int = 12,
                     fourier_length: int = 256, every: int = 22050, noise: float = 0.01) -> np.ndarray:
    """Noise plus a template syllable roughly every `every` samples at jittered positions and
    amplitudes, so a detector trained on the template fires on some and not on others."""
    rng = np.random.default_rng(seed)
    x = noise * rng.standard_normal(n_samples)
    pos = int(rng.integers(0, every))
    while True:
        s = syllable(template, hop, window, f0, fourier_length, rng, amplitude=float(rng.uniform(0.02, 0.6)))
        if pos + s.size > n_samples:
            break
        x[pos:pos + s.size] += s
        pos += every + int(rng.integers(-every // 4, every // 4))
    return np.clip(x, -1.0, 1.0).astype(np.float32)
