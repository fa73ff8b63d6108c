#!/usr/bin/env python3
"""Derivation and accuracy check of the polynomial kernels of the fast build (slam_amd/csrc/device_math.h):
  * atan2_poly: least-squares/minimax fit of atan(t)/t in z = t^2 on [0,1], degree 8, evaluated the way the device
    does (float32 Horner with FMA) -> max 1.1 ulp
  * sincos_cw: 2-constant Cody-Waite reduction by pi/2 + Cephes single-precision polynomials -> max 1.5 ulp, |x| <= 20
float32 FMA is emulated as round(float64(a*b+c)), exact for these magnitudes.  CPU only, numpy only."""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def fit_atan(deg=8):
    z = 0.5 * (1 - np.cos(np.linspace(0, np.pi, 4001)))
    z[0] = 1e-12
    t = np.sqrt(z)
    g = np.arctan(t) / t
    w = np.ones_like(z)
    for _ in range(60):  # iteratively re-weighted least squares towards the minimax solution, c0 pinned to 1
        A1 = np.vander(z, deg + 1, increasing=True)[:, 1:] / g[:, None]
        rhs = (g - 1.0) / g
        c, *_ = np.linalg.lstsq(A1 * w[:, None], rhs * w, rcond=None)
        err = np.abs(A1 @ c - rhs)
        w = w * (1 + 4 * err / err.max())
        w /= w.mean()
    return np.concatenate([[1.0], c]).astype(f32), err.max()


def check_atan(c32):
    t = np.random.default_rng(0).uniform(0, 1, 2_000_000).astype(f32)
    z = (t * t).astype(f32)
    p = np.full_like(z, c32[-1])
    for k in range(len(c32) - 2, 0, -1):
        p = fma(p, z, c32[k])
    r = fma((t * z).astype(f32), p, t)
    ref = np.arctan(t.astype(np.float64))
    return np.max(np.abs(r - ref) / np.spacing(ref.astype(f32)))


def check_sincos():
    x = np.random.default_rng(1).uniform(-20, 20, 4_000_000).astype(f32)
    hi = f32(1.5707963705062866)
    lo = f32(np.pi / 2 - np.float64(hi))
    q = np.rint((x * f32(0.6366197723675814)).astype(f32)).astype(f32)
    r = fma(-q, hi, x)
    r = fma(-q, lo, r)
    z = (r * r).astype(f32)
    p = fma(z, f32(-1.9515295891e-4), f32(8.3321608736e-3))
    p = fma(p, z, f32(-1.6666654611e-1))
    s = fma((r * z).astype(f32), p, r)
    p = fma(z, f32(2.443315711809948e-5), f32(-1.388731625493765e-3))
    p = fma(p, z, f32(4.166664568298827e-2))
    c = fma((z * z).astype(f32), p, fma(z, f32(-0.5), f32(1.0)))
    n = q.astype(np.int64) & 3
    sn = np.where(n & 1, c, s)
    cs = np.where(n & 1, s, c)
    sn = np.where(n & 2, -sn, sn)
    cs = np.where((n == 1) | (n == 2), -cs, cs)
    xs = x.astype(np.float64)
    out = {}
    for name, got, ref in (("sin", sn, np.sin(xs)), ("cos", cs, np.cos(xs))):
        e = np.abs(got - ref)
        out[name] = (e.max(), (e / np.spacing(np.abs(ref).astype(f32))).max())
    return out


if __name__ == "__main__":
    c32, fit_err = fit_atan()
    print("atan(t)/t coefficients in z=t^2 (c0..c8):", [float(v) for v in c32], "fit rel err %.2e" % fit_err)
    print("atan float32 evaluation: max %.2f ulp" % check_atan(c32))
    for k, (a, u) in check_sincos().items():
        print("%s: max abs %.2e, max %.2f ulp" % (k, a, u))
