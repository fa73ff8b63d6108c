#!/usr/bin/env python3
"""float64 evaluation of ONE FastSLAM2 observation update (fastslam2.cpp:290-368: sampleProposal + likelihoodGivenXv),
vectorised over particles, used as the yardstick when the float32 reference and a float32 GPU build disagree about a
weight: whichever is closer to this is the better float32 evaluation of the formula.  numpy only; diagnostic."""
import numpy as np


def wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def jac(xv, xf, Pf, R):
    dx, dy = xf[:, 0] - xv[:, 0], xf[:, 1] - xv[:, 1]
    d2 = dx * dx + dy * dy
    d = np.sqrt(d2)
    zp = np.stack([d, wrap(np.arctan2(dy, dx) - xv[:, 2])], 1)
    N = xv.shape[0]
    Hv = np.zeros((N, 2, 3)); Hf = np.zeros((N, 2, 2))
    Hv[:, 0, 0], Hv[:, 0, 1] = -dx / d, -dy / d
    Hv[:, 1, 0], Hv[:, 1, 1], Hv[:, 1, 2] = dy / d2, -dx / d2, -1.0
    Hf[:, 0, 0], Hf[:, 0, 1] = dx / d, dy / d
    Hf[:, 1, 0], Hf[:, 1, 1] = -dy / d2, dx / d2
    Sf = Hf @ Pf @ Hf.transpose(0, 2, 1) + R
    return zp, Hv, Hf, Sf


def gauss(v, S):
    """gaussEvaluate with the reference's constant: (2 pi)^(D/2 as INTEGER division) = 2 pi for D = 2 and 3."""
    L = np.linalg.cholesky(S)
    n = np.linalg.solve(L, v[..., None])[..., 0]
    E = -0.5 * (n * n).sum(-1)
    return np.exp(E) / (2 * np.pi * np.prod(np.diagonal(L, axis1=-2, axis2=-1), -1))


def update_weights(pre, zf, idf, R, g):
    """pre: dict xv[N,3] Pv[N,3,3] w[N] xf[N,nf,2] Pf[N,nf,2,2] (any float dtype) -> (xs[N,3], w_post[N]) in float64"""
    xv = pre["xv"].astype(np.float64); Pv = pre["Pv"].astype(np.float64); w = pre["w"].astype(np.float64)
    Pv = 0.5 * (Pv + Pv.transpose(0, 2, 1))
    xf = pre["xf"].astype(np.float64); Pf = pre["Pf"].astype(np.float64)
    R = np.asarray(R, np.float64); g = np.asarray(g, np.float64)
    xv0, Pv0 = xv.copy(), Pv.copy()
    for k, j in enumerate(idf):
        zp, Hv, Hf, Sf = jac(xv, xf[:, j], Pf[:, j], R)
        v = np.stack([zf[k][0] - zp[:, 0], wrap(zf[k][1] - zp[:, 1])], 1)
        S = Hv @ Pv @ Hv.transpose(0, 2, 1) + Sf
        K = Pv @ Hv.transpose(0, 2, 1) @ np.linalg.inv(S)
        xv = xv + (K @ v[..., None])[..., 0]
        Pv = Pv - K @ Hv @ Pv
        Pv = 0.5 * (Pv + Pv.transpose(0, 2, 1))
    L = np.linalg.cholesky(Pv)
    xs = xv + (L @ g[..., None])[..., 0]
    lik = np.ones_like(w)
    for k, j in enumerate(idf):
        zp, Hv, Hf, Sf = jac(xs, xf[:, j], Pf[:, j], R)
        v = np.stack([zf[k][0] - zp[:, 0], wrap(zf[k][1] - zp[:, 1])], 1)
        lik = lik * gauss(v, Sf)
    a = xv0 - xs; a[:, 2] = wrap(a[:, 2])
    b = xv - xs; b[:, 2] = wrap(b[:, 2])
    return xs, w * lik * gauss(a, Pv0) / gauss(b, Pv)


def gauss_log(v, S):
    """log of gauss(): the reference's gaussEvaluate(v, S, logflag = 1) (fastslam2.cpp:154-160)"""
    L = np.linalg.cholesky(S)
    n = np.linalg.solve(L, v[..., None])[..., 0]
    E = -0.5 * (n * n).sum(-1)
    return E - np.log(2 * np.pi * np.prod(np.diagonal(L, axis1=-2, axis2=-1), -1))


def update_log_weights(pre, zf, idf, R, g):
    """update_weights for log-weight particle sets (pre["w"] = log-weights), hundreds of landmarks per step: the same update,
    the likelihood accumulated as a sum of logs.  -> (xs[N,3], logw_post[N]) in float64"""
    xv = pre["xv"].astype(np.float64); Pv = pre["Pv"].astype(np.float64); lw = pre["w"].astype(np.float64)
    Pv = 0.5 * (Pv + Pv.transpose(0, 2, 1))
    xf = pre["xf"].astype(np.float64); Pf = pre["Pf"].astype(np.float64)
    Pf = 0.5 * (Pf + Pf.transpose(0, 1, 3, 2))
    R = np.asarray(R, np.float64); g = np.asarray(g, np.float64)
    xv0, Pv0 = xv.copy(), Pv.copy()
    for k, j in enumerate(idf):
        zp, Hv, Hf, Sf = jac(xv, xf[:, j], Pf[:, j], R)
        v = np.stack([zf[k][0] - zp[:, 0], wrap(zf[k][1] - zp[:, 1])], 1)
        S = Hv @ Pv @ Hv.transpose(0, 2, 1) + Sf
        K = Pv @ Hv.transpose(0, 2, 1) @ np.linalg.inv(S)
        xv = xv + (K @ v[..., None])[..., 0]
        Pv = Pv - K @ Hv @ Pv
        Pv = 0.5 * (Pv + Pv.transpose(0, 2, 1))
    L = np.linalg.cholesky(Pv)
    xs = xv + (L @ g[..., None])[..., 0]
    ll = np.zeros_like(lw)
    for k, j in enumerate(idf):
        zp, Hv, Hf, Sf = jac(xs, xf[:, j], Pf[:, j], R)
        v = np.stack([zf[k][0] - zp[:, 0], wrap(zf[k][1] - zp[:, 1])], 1)
        ll = ll + gauss_log(v, Sf)
    a = xv0 - xs; a[:, 2] = wrap(a[:, 2])
    b = xv - xs; b[:, 2] = wrap(b[:, 2])
    return xs, lw + ll + gauss_log(a, Pv0) - gauss_log(b, Pv)
