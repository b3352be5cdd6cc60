"""Deterministic synthetic meteorological fields.

Host generators use a counter-based SplitMix64 hash in numpy (no rand(), no
global RNG state), so the same (seed, shape) gives the same bits everywhere --
the golden fixtures under tests/golden/ depend on that.  ``device_*`` helpers
build the large benchmark batches directly in HBM with torch ops from the same
closed-form waves (the noise term there comes from torch's generator: bench
inputs only need to be deterministic per run, they are never compared with a
host copy bit for bit).
"""
import numpy as np

UNDEF = np.float32(1.0e35)
OMEGA = 7.292e-5


def splitmix64(idx, seed):
    """Vectorised SplitMix64 finaliser of (seed + idx * golden) -> uint64."""
    with np.errstate(over="ignore"):
        z = (np.asarray(idx, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(shape, seed, lo=0.0, hi=1.0):
    n = int(np.prod(shape))
    bits = splitmix64(np.arange(n, dtype=np.uint64), seed)
    u01 = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return (lo + (hi - lo) * u01).reshape(shape)


def grid_maps(nx, ny, h=27800.0):
    """xmapr = 1/(2 h cos(phi_j)), ymapr = 1/(2 h), fcoriolis = 2 Omega sin(phi)
    on a global lat-lon grid with |phi| clipped to 85 degrees."""
    j = np.arange(ny, dtype=np.float64)
    phi = np.deg2rad(np.clip(-90.0 + (j + 0.5) * (180.0 / ny), -85.0, 85.0))
    xm = (1.0 / (2.0 * h * np.cos(phi)))[:, None] * np.ones((1, nx))
    ym = np.full((ny, nx), 1.0 / (2.0 * h))
    fc = (2.0 * OMEGA * np.sin(phi))[:, None] * np.ones((1, nx))
    fc = np.where(np.abs(fc) < 1e-5, np.where(fc < 0, -1e-5, 1e-5), fc)
    return xm.astype(np.float32), ym.astype(np.float32), fc.astype(np.float32)


def wind(nx, ny, seed, nlev=None):
    """u, v: large-scale waves (20 / 15 m/s) + +-0.5 m/s noise."""
    shape = (ny, nx) if nlev is None else (nlev, ny, nx)
    i = np.arange(nx, dtype=np.float64)[None, :]
    j = np.arange(ny, dtype=np.float64)[:, None]
    lev = np.zeros((1, 1, 1)) if nlev is None else np.arange(nlev, dtype=np.float64)[:, None, None]
    kx, ky = 0.013, 0.017
    u = 20.0 * np.sin(kx * i + 0.1 * lev) * np.cos(ky * j) + uniform(shape, seed, -0.5, 0.5)
    v = 15.0 * np.cos(0.011 * i) * np.sin(0.019 * j + 0.07 * lev) + uniform(shape, seed + 7919, -0.5, 0.5)
    return u.reshape(shape).astype(np.float32), v.reshape(shape).astype(np.float32)


def scalar_field(nx, ny, seed, base=5500.0, amp=120.0, noise=2.0):
    """geopotential-height-like smooth field + noise."""
    i = np.arange(nx, dtype=np.float64)[None, :]
    j = np.arange(ny, dtype=np.float64)[:, None]
    z = base + amp * np.sin(0.012 * i) * np.cos(0.015 * j) + uniform((ny, nx), seed, -noise, noise)
    return z.astype(np.float32)


def thermo(nx, ny, seed, nlev=None):
    """t in [220,310] K, q in [1e-5,0.02] kg/kg, ps in [500,1050] hPa (ps is 2-D)."""
    shape = (ny, nx) if nlev is None else (nlev, ny, nx)
    t = uniform(shape, seed + 1, 220.0, 310.0).astype(np.float32)
    q = uniform(shape, seed + 2, 1e-5, 0.02).astype(np.float32)
    ps = uniform((ny, nx), seed + 3, 500.0, 1050.0).astype(np.float32)
    return t, q, ps


def hybrid_levels(nlev):
    """(a, b) ramp with a >= 0, 0 <= b <= 1, never both zero (bad_hlevel, FieldCalculations.cc:298)."""
    k = (np.arange(nlev, dtype=np.float64) + 0.5) / nlev
    a = 200.0 * (1.0 - k) * k * 4.0 + 1.0
    b = k ** 1.5
    return a.astype(np.float32), b.astype(np.float32)


def sprinkle_undef(field, seed, frac=0.01, undef=UNDEF, nan_every=5):
    """~frac of the cells become undef; every nan_every-th of those becomes NaN instead."""
    f = np.array(field, dtype=np.float32, copy=True)
    flat = f.reshape(-1)
    r = uniform(flat.shape, seed + 4242)
    idx = np.nonzero(r < frac)[0]
    flat[idx] = undef
    if nan_every:
        flat[idx[::nan_every]] = np.nan
    return f


# ------------------------------------------------------------------ device side
def device_wind(nx, ny, nlev, seed, device):
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    i = torch.arange(nx, dtype=torch.float32, device=device)[None, None, :]
    j = torch.arange(ny, dtype=torch.float32, device=device)[None, :, None]
    lev = torch.arange(nlev, dtype=torch.float32, device=device)[:, None, None]
    u = torch.rand((nlev, ny, nx), generator=g, device=device, dtype=torch.float32).sub_(0.5)
    u.add_(20.0 * torch.sin(0.013 * i + 0.1 * lev) * torch.cos(0.017 * j))
    v = torch.rand((nlev, ny, nx), generator=g, device=device, dtype=torch.float32).sub_(0.5)
    v.add_(15.0 * torch.cos(0.011 * i) * torch.sin(0.019 * j + 0.07 * lev))
    return u.contiguous(), v.contiguous()


def device_thermo(nx, ny, nlev, seed, device):
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(int(seed) + 1)
    t = torch.rand((nlev, ny, nx), generator=g, device=device, dtype=torch.float32).mul_(90.0).add_(220.0)
    q = torch.rand((nlev, ny, nx), generator=g, device=device, dtype=torch.float32).mul_(0.02 - 1e-5).add_(1e-5)
    ps = torch.rand((ny, nx), generator=g, device=device, dtype=torch.float32).mul_(550.0).add_(500.0)
    return t, q, ps
