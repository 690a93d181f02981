"""Synthetic AWGN frames for the parity tests and bench.py.

Clean frame: deterministic smooth + edges pattern in 0..255; frame t is frame 0
shifted by 2 px. Noise: the reference tool chain's own generator — 64-bit Knuth
LCG, uniform = (s >> 32) / UINT_MAX, Box-Muller cosine branch
(reference: lib/imscript-lite/src/random.c:19-31,50-53,68-75; awgn.c:24-26),
vectorised here with numpy by LCG jump-ahead. Pure host code, no oracle import.
"""
import numpy as np

_A = np.uint64(6364136223846793005)
_C = np.uint64(1442695040888963407)


def lcg_stream(n, seed):
    """First n outputs (s >> 32 after each step) of the LCG started at `seed`."""
    s = np.empty(n, np.uint64)
    with np.errstate(over="ignore"):
        s[0] = np.uint64(seed) * _A + _C
        have = 1
        a, c = _A, _C  # jump by `have` steps: s -> a*s + c
        while have < n:
            m = min(have, n - have)
            s[have:have + m] = s[:m] * a + c
            have += m
            c = a * c + c
            a = a * a
    return (s >> np.uint64(32)).astype(np.uint32)


def awgn(clean, sigma, seed):
    """clean + sigma * N(0,1) with the reference's LCG / Box-Muller sequence."""
    clean = np.ascontiguousarray(clean, np.float32)
    n = clean.size
    u = lcg_stream(2 * n, seed).astype(np.float64) / 4294967295.0
    with np.errstate(divide="ignore"):
        g = np.sqrt(-2.0 * np.log(u[0::2])) * np.cos(2.0 * np.pi * u[1::2])
    out = clean.reshape(-1).astype(np.float64) + float(np.float32(sigma)) * g
    return out.astype(np.float32).reshape(clean.shape)


def clean_frame(w, h, ch, t=0):
    """Sinusoids + checker blocks + a few hard edges, values in 0..255."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    x = x + 2.0 * t
    im = (128.0 + 50.0 * np.sin(x / 9.0) * np.cos(y / 13.0)
          + 30.0 * np.sin((x + 2.0 * y) / 37.0)
          + 35.0 * ((((x // 24) + (y // 24)) % 2) - 0.5)
          + 25.0 * (((x - 0.6 * y) % 97.0) < 12.0))
    im = np.clip(im, 0.0, 255.0)
    gains = np.array([1.0, 0.85, 0.7])[:ch] if ch <= 3 else np.ones(ch)
    out = im[:, :, None] * gains[None, None, :]
    if ch == 3:
        out[:, :, 1] = np.clip(out[:, :, 1] + 20.0 * np.cos(y / 21.0), 0, 255)
        out[:, :, 2] = np.clip(out[:, :, 2] + 15.0 * np.sin(x / 17.0), 0, 255)
    return out.astype(np.float32)


def noisy_pair(w, h, ch, sigma, seed):
    """(noisy frame 0, noisy frame 1, clean frame 1) in RGB / gray, HWC float32."""
    c0, c1 = clean_frame(w, h, ch, 0), clean_frame(w, h, ch, 1)
    return awgn(c0, sigma, seed), awgn(c1, sigma, seed + 1000), c1


def psnr(a, b):
    """PSNR = 20 log10(255 / RMSE) over all samples (reference: scripts/psnr.sh:9-11)."""
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else 20.0 * np.log10(255.0 / np.sqrt(mse))
