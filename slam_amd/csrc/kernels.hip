// gfx950 kernels of the FastSLAM observation update: one particle per work-item, SoA state in HBM,
// wave64 shuffles for the weight prefix-sum / reductions, device-resident resample decision.
// Compiled twice (see Makefile): -DSLAM_KNS=slam_strict -ffp-contract=off and -DSLAM_KNS=slam_fast.
#include "kernels.h"
#include "device_math.h"

namespace SLAM_KNS {
using namespace slamgpu;

constexpr int kBlock = 256;

SLAM_DEV const int32_t *pkt_idf(const ObsPacket *p) { return reinterpret_cast<const int32_t *>(p + 1); }

// ---------------------------------------------------------------------------------------------------
// K1: per-particle observation update.  FastSLAM2::update body (fastslam2.cpp:26-45): sampleProposal
// (:290-368) + likelihoodGivenXv (:370-400) fused with featureUpdate (core.cpp:132-175, the Jacobians
// of both are evaluated at the same sampled pose) + addFeature (core.cpp:479-509); or FastSLAM1::update
// body (fastslam1.cpp:21-32).  Ends with the in-wave inclusive prefix of the raw weights and the wave
// totals of w and w^2 (resampleParticles' normalisation / Neff inputs, core.cpp:726-729,781-788).
// ---------------------------------------------------------------------------------------------------
template <int METHOD>
__global__ void __launch_bounds__(kBlock) update_kernel(Buffers B, const ObsPacket *__restrict__ pkt, int m, int n,
                                                         int nf, RngArgs rng, WeightScratch ws) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const size_t S = (size_t) B.ncap;
    const int cur = B.ctrl->cur;
    float *__restrict__ pose = B.pose[cur];
    float *__restrict__ lmk = B.lmk[cur];
    const bool active = i < B.n;
    float w = 0.0f;

    if (active) {
        const int32_t *__restrict__ idf = pkt_idf(pkt);
        const float *__restrict__ zf = reinterpret_cast<const float *>(idf + m);
        const float *__restrict__ zn = zf + 2 * m;
        const float r00 = pkt->R[0], r01 = pkt->R[1], r10 = pkt->R[2], r11 = pkt->R[3];

        float x = pose[0 * S + i], y = pose[1 * S + i], th = pose[2 * S + i];
        w = pose[9 * S + i];

        if (METHOD == 2) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (m > 0 || n > 0) {
                if (rng.mode == 0) {
                    g0 = rng.normals[0 * S + i];
                    g1 = rng.normals[1 * S + i];
                    g2 = rng.normals[2 * S + i];
                } else {
                    U4 r = philox4x32((uint32_t) (rng.first_particle + i), rng.step, 0u, 0u, rng.k0, rng.k1);
                    box_muller3(r, g0, g1, g2);
                }
            }
            // Pv lower triangle as stored
            float q00 = pose[3 * S + i], q10 = pose[4 * S + i], q11 = pose[5 * S + i];
            float q20 = pose[6 * S + i], q21 = pose[7 * S + i], q22 = pose[8 * S + i];
            if (m > 0) {
                const float x0 = x, y0 = y, th0 = th;
                // running proposal covariance, full 3x3 (the reference's Pv stays a full matrix inside the loop)
                float P[9] = {q00, q10, q20, q10, q11, q21, q20, q21, q22};
                for (int k = 0; k < m; k++) {
                    const size_t lb = (size_t) idf[k] * kLmkRows * S + i;
                    const float fx = lmk[lb], fy = lmk[lb + S], p00 = lmk[lb + 2 * S], p10 = lmk[lb + 3 * S],
                                p11 = lmk[lb + 4 * S];
                    // Jacobians at the running mean (fastslam2.cpp:320,:348)
                    Jac j = jacobian(x, y, th, fx, fy, p00, p10, p11, r00, r01, r10, r11);
                    float s00, s01, s10, s11;
                    inverse2(j.s00, j.s01, j.s10, j.s11, s00, s01, s10, s11);  // Sfi (:324)
                    const float v0 = zf[2 * k] - j.zp0;
                    const float v1 = trig_offset(zf[2 * k + 1] - j.zp1);
                    float Pinv[9];
                    llt_solve_identity3(llt3(P[0], P[3], P[4], P[6], P[7], P[8]), Pinv);  // (:335)
                    // T1 = Hv^T * Sfi (3x2), T2 = T1 * Hv (3x3); Hv = [[hv00 hv01 0],[hv10 hv11 -1]]
                    const float t00 = j.hv00 * s00 + j.hv10 * s10, t01 = j.hv00 * s01 + j.hv10 * s11;
                    const float t10 = j.hv01 * s00 + j.hv11 * s10, t11 = j.hv01 * s01 + j.hv11 * s11;
                    const float t20 = -s10, t21 = -s11;
                    P[0] = (t00 * j.hv00 + t01 * j.hv10) + Pinv[0];
                    P[1] = (t00 * j.hv01 + t01 * j.hv11) + Pinv[1];
                    P[2] = (-t01) + Pinv[2];
                    P[3] = (t10 * j.hv00 + t11 * j.hv10) + Pinv[3];
                    P[4] = (t10 * j.hv01 + t11 * j.hv11) + Pinv[4];
                    P[5] = (-t11) + Pinv[5];
                    P[6] = (t20 * j.hv00 + t21 * j.hv10) + Pinv[6];
                    P[7] = (t20 * j.hv01 + t21 * j.hv11) + Pinv[7];
                    P[8] = (-t21) + Pinv[8];
                    llt_solve_identity3(llt3(P[0], P[3], P[4], P[6], P[7], P[8]), P);  // (:341)
                    // xv += ((Pv * Hv^T) * Sfi) * v   (:345)
                    float c[3];
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        const float a0 = P[3 * r] * j.hv00 + P[3 * r + 1] * j.hv01;
                        const float a1 = (P[3 * r] * j.hv10 + P[3 * r + 1] * j.hv11) + P[3 * r + 2] * -1.0f;
                        const float b0 = a0 * s00 + a1 * s10;
                        const float b1 = a0 * s01 + a1 * s11;
                        c[r] = b0 * v0 + b1 * v1;
                    }
                    x = x + c[0];
                    y = y + c[1];
                    th = th + c[2];
                }
                // sample from the proposal (:353) ; weight terms (:360-367)
                const L3 Lp = llt3(P[0], P[3], P[4], P[6], P[7], P[8]);
                float xs = x, ys = y, ths = th;
                mvgauss3(xs, ys, ths, Lp, g0, g1, g2);
                const float a0 = x0 - xs, a1 = y0 - ys, a2 = trig_offset(th0 - ths);
                const float b0 = x - xs, b1 = y - ys, b2 = trig_offset(th - ths);
                float lik = 1.0f;
                for (int k = 0; k < m; k++) {
                    const size_t lb = (size_t) idf[k] * kLmkRows * S + i;
                    float fx = lmk[lb], fy = lmk[lb + S], p00 = lmk[lb + 2 * S], p10 = lmk[lb + 3 * S],
                          p11 = lmk[lb + 4 * S];
                    Jac j = jacobian(xs, ys, ths, fx, fy, p00, p10, p11, r00, r01, r10, r11);
                    const float v0 = zf[2 * k] - j.zp0;
                    const float v1 = trig_offset(zf[2 * k + 1] - j.zp1);
                    lik = lik * gauss2(v0, v1, j.s00, j.s10, j.s11);
                    cholesky_update2(fx, fy, p00, p10, p11, v0, v1, r00, r01, r10, r11, j.hf00, j.hf01, j.hf10, j.hf11);
                    lmk[lb] = fx;
                    lmk[lb + S] = fy;
                    lmk[lb + 2 * S] = p00;
                    lmk[lb + 3 * S] = p10;
                    lmk[lb + 4 * S] = p11;
                }
                const float prior = gauss3(a0, a1, a2, q00, q10, q11, q20, q21, q22);
                const float prop = gauss3(b0, b1, b2, P[0], P[3], P[4], P[6], P[7], P[8]);
                w = w * lik * prior / prop;
                x = xs;
                y = ys;
                th = ths;
                q00 = q10 = q11 = q20 = q21 = q22 = 0.0f;
            } else if (n > 0) {
                // no re-observed landmark: sample the pose from the predicted Gaussian (fastslam2.cpp:36-42)
                mvgauss3(x, y, th, llt3(q00, q10, q11, q20, q21, q22), g0, g1, g2);
                q00 = q10 = q11 = q20 = q21 = q22 = 0.0f;
            }
            if (m > 0 || n > 0) {
                pose[0 * S + i] = x;
                pose[1 * S + i] = y;
                pose[2 * S + i] = th;
                pose[3 * S + i] = q00;
                pose[4 * S + i] = q10;
                pose[5 * S + i] = q11;
                pose[6 * S + i] = q20;
                pose[7 * S + i] = q21;
                pose[8 * S + i] = q22;
            }
        } else {
            // FastSLAM 1: computeWeight (fastslam1.cpp:91-118) + featureUpdate at the particle pose
            if (m > 0) {
                float wp = 1.0f;
                for (int k = 0; k < m; k++) {
                    const size_t lb = (size_t) idf[k] * kLmkRows * S + i;
                    float fx = lmk[lb], fy = lmk[lb + S], p00 = lmk[lb + 2 * S], p10 = lmk[lb + 3 * S],
                          p11 = lmk[lb + 4 * S];
                    Jac j = jacobian(x, y, th, fx, fy, p00, p10, p11, r00, r01, r10, r11);
                    const float v0 = zf[2 * k] - j.zp0;
                    const float v1 = trig_offset(zf[2 * k + 1] - j.zp1);
                    const float den = (float) (2 * kPi * (double) sqrtf(determinant2(j.s00, j.s01, j.s10, j.s11)));
                    float i00, i01, i10, i11;
                    inverse2(j.s00, j.s01, j.s10, j.s11, i00, i01, i10, i11);
                    const float t0 = -0.5f * (v0 * i00 + v1 * i10);
                    const float t1 = -0.5f * (v0 * i01 + v1 * i11);
                    const float num = expf(t0 * v0 + t1 * v1);
                    wp = wp * num / den;
                    cholesky_update2(fx, fy, p00, p10, p11, v0, v1, r00, r01, r10, r11, j.hf00, j.hf01, j.hf10, j.hf11);
                    lmk[lb] = fx;
                    lmk[lb + S] = fy;
                    lmk[lb + 2 * S] = p00;
                    lmk[lb + 3 * S] = p10;
                    lmk[lb + 4 * S] = p11;
                }
                w = w * wp;
            }
        }
        // addFeature (core.cpp:479-509): new landmarks appended at nf, nf+1, ...
        for (int k = 0; k < n; k++) {
            float fx, fy, p00, p10, p11;
            add_feature(x, y, th, zn[2 * k], zn[2 * k + 1], r00, r01, r10, r11, fx, fy, p00, p10, p11);
            const size_t lb = (size_t) (nf + k) * kLmkRows * S + i;
            lmk[lb] = fx;
            lmk[lb + S] = fy;
            lmk[lb + 2 * S] = p00;
            lmk[lb + 3 * S] = p10;
            lmk[lb + 4 * S] = p11;
        }
        pose[9 * S + i] = w;
    }

    // in-wave inclusive prefix of w, wave totals of w and w^2
    float s = w;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const float t = __shfl_up(s, d, kWave);
        if (lane >= d) s += t;
    }
    float s2 = w * w;
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) s2 += __shfl_xor(s2, d, kWave);
    if (i < B.ncap) ws.lcum[i] = s;
    const int wave = i / kWave;
    if (lane == kWave - 1 && wave < ws.nwaves) {
        ws.wave_w[wave] = s;
        ws.wave_w2[wave] = s2;
    }
}

// ---------------------------------------------------------------------------------------------------
// K2: one block.  Exclusive prefix of the wave totals (double), sum w, sum w^2, Neff and the resample
// decision `doResample && Neff < nMin` (core.cpp:739), all left in device memory.
// ---------------------------------------------------------------------------------------------------
constexpr int kFinBlock = 1024;

__global__ void __launch_bounds__(kFinBlock) finalize_kernel(Buffers B, WeightScratch ws, int do_resample,
                                                              int n_effective) {
    __shared__ double sh_sum[kFinBlock];
    __shared__ double sh_sq[kFinBlock];
    const int t = threadIdx.x;
    const int P = ws.nwaves;
    const int per = (P + kFinBlock - 1) / kFinBlock;
    const int lo = t * per, hi = min(P, lo + per);
    double a = 0.0, q = 0.0;
    for (int k = lo; k < hi; k++) {
        a += (double) ws.wave_w[k];
        q += (double) ws.wave_w2[k];
    }
    sh_sum[t] = a;
    sh_sq[t] = q;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 1024 thread totals
    for (int d = 1; d < kFinBlock; d <<= 1) {
        double va = 0.0, vq = 0.0;
        if (t >= d) {
            va = sh_sum[t - d];
            vq = sh_sq[t - d];
        }
        __syncthreads();
        sh_sum[t] += va;
        sh_sq[t] += vq;
        __syncthreads();
    }
    double run = sh_sum[t] - a;  // exclusive prefix of this thread's segment
    for (int k = lo; k < hi; k++) {
        ws.wave_off[k] = run;
        run += (double) ws.wave_w[k];
    }
    if (t == kFinBlock - 1) {
        const double W = sh_sum[t], Q = sh_sq[t];
        ws.wave_off[P] = W;
        Ctrl *c = B.ctrl;
        c->wsum = W;
        c->wsq = Q;
        // Neff = 1 / sum((w/W)^2)  (core.cpp:784-788)
        const float neff = (float) ((W * W) / Q);
        c->neff = neff;
        c->resampled = (do_resample && (neff < (float) n_effective)) ? 1 : 0;
        c->done = 0;
    }
}

// ---------------------------------------------------------------------------------------------------
// K3: normalise or resample.  No resample: w_i /= sum(w) (core.cpp:726-729).  Resample: stratified
// ancestor of output particle k = min{ i : select_k < cumsum_i } (core.cpp:800-806), gather-copy of the
// whole particle (pose + nf landmarks) from the live buffer into the other one, w = 1/N (:744-747); the
// last block to finish flips Ctrl.cur.  blockIdx.y splits the landmark rows.
// ---------------------------------------------------------------------------------------------------
constexpr int kLmkPerBlockY = 32;

__global__ void __launch_bounds__(kBlock) resample_kernel(Buffers B, WeightScratch ws, RngArgs rng, int nf) {
    Ctrl *ctrl = B.ctrl;
    const int cur = ctrl->cur;
    const int k = blockIdx.x * kBlock + threadIdx.x;
    const size_t S = (size_t) B.ncap;
    if (!ctrl->resampled) {
        if (blockIdx.y == 0 && k < B.n) {
            float *pose = B.pose[cur];
            pose[9 * S + k] = pose[9 * S + k] / (float) ctrl->wsum;
        }
        return;
    }
    if (k < B.n) {
        const int64_t gid = rng.first_particle + k;
        float sel;
        if (rng.mode == 0) {
            sel = rng.strata[gid];
        } else {
            U4 r = philox4x32((uint32_t) gid, rng.step, 1u, 0u, rng.k0, rng.k1);
            const double u = ((double) (r.x >> 8) + 0.5) * (1.0 / 16777216.0);
            sel = (float) (((double) gid + u) / (double) rng.n_global);
        }
        const double target = (double) sel * ctrl->wsum;
        // wave containing the ancestor: first b with wave_off[b+1] > target
        int lo = 0, hi = ws.nwaves;  // answer in [lo, hi]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (ws.wave_off[mid + 1] > target) hi = mid; else lo = mid + 1;
        }
        int a;
        if (lo >= ws.nwaves) {
            a = B.n - 1;  // select beyond the last cumulative weight: undefined upstream (keep = -1), clamp
        } else {
            const double off = ws.wave_off[lo];
            const float *lc = ws.lcum + (size_t) lo * kWave;
            int l0 = 0, l1 = kWave - 1;  // first lane with off + lc > target; lane 63 if rounding hides it
            while (l0 < l1) {
                const int mid = (l0 + l1) >> 1;
                if (off + (double) lc[mid] > target) l1 = mid; else l0 = mid + 1;
            }
            a = min(lo * kWave + l0, B.n - 1);
        }
        const float *__restrict__ sp = B.pose[cur];
        float *__restrict__ dp = B.pose[cur ^ 1];
        if (blockIdx.y == 0) {
#pragma unroll
            for (int c = 0; c < 9; c++) dp[c * S + k] = sp[c * S + a];
            dp[9 * S + k] = ctrl->inv_n;
            ws.keep[k] = a;
        }
        const float *__restrict__ sl = B.lmk[cur];
        float *__restrict__ dl = B.lmk[cur ^ 1];
        const int j0 = blockIdx.y * kLmkPerBlockY, j1 = min(nf, j0 + kLmkPerBlockY);
        for (int j = j0; j < j1; j++) {
            const size_t lb = (size_t) j * kLmkRows * S;
#pragma unroll
            for (int c = 0; c < kLmkRows; c++) dl[lb + c * S + k] = sl[lb + c * S + a];
        }
    }
    // last block flips the live buffer (every other block has finished reading Ctrl.cur's buffers by then)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned total = gridDim.x * gridDim.y;
        if (atomicAdd(&ctrl->done, 1u) == total - 1) {
            ctrl->cur = cur ^ 1;
            ctrl->done = 0;
            __threadfence();
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Predict: FastSLAM2::predictState (fastslam2.cpp:70-105) [+ observeHeading -> josephUpdate
// (fastslam2.cpp:113-125, core.cpp:294-317)] or FastSLAM1::predictState (fastslam1.cpp:37-54), up to
// kMaxFusedPredict consecutive control steps with the state held in registers.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) predict_kernel(Buffers B, PredictArgs A, RngArgs rng) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= B.n) return;
    const size_t S = (size_t) B.ncap;
    float *__restrict__ pose = B.pose[B.ctrl->cur];
    float x = pose[0 * S + i], y = pose[1 * S + i], th = pose[2 * S + i];
    float P[9];
    const bool fs2 = A.method == 2;
    if (fs2) {
        const float q00 = pose[3 * S + i], q10 = pose[4 * S + i], q11 = pose[5 * S + i];
        const float q20 = pose[6 * S + i], q21 = pose[7 * S + i], q22 = pose[8 * S + i];
        P[0] = q00; P[1] = q10; P[2] = q20;
        P[3] = q10; P[4] = q11; P[5] = q21;
        P[6] = q20; P[7] = q21; P[8] = q22;
    }
    const float dt = A.dt, wb = A.wheel_base;
    const float Q00 = A.Q[0], Q01 = A.Q[1], Q10 = A.Q[2], Q11 = A.Q[3];
    for (int s = 0; s < A.nsteps; s++) {
        float V = A.steps[s].V, G = A.steps[s].G;
        if (fs2) {
            // Gv, Gu (fastslam2.cpp:78-79)
            const float gv02 = -V * dt * sinf(G + th), gv12 = V * dt * cosf(G + th);
            const float gu00 = dt * cosf(G + th), gu01 = -V * dt * sinf(G + th);
            const float gu10 = dt * sinf(G + th), gu11 = V * dt * cosf(G + th);
            const float gu20 = dt * sinf(G) / wb, gu21 = V * dt * cosf(G) / wb;
            // T = Gv * Pv ; A = T * Gv^T   (Gv = [[1,0,gv02],[0,1,gv12],[0,0,1]])
            float T[9], N9[9];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                T[c] = P[c] + gv02 * P[6 + c];
                T[3 + c] = P[3 + c] + gv12 * P[6 + c];
                T[6 + c] = P[6 + c];
            }
#pragma unroll
            for (int r = 0; r < 3; r++) {
                N9[3 * r] = T[3 * r] + T[3 * r + 2] * gv02;
                N9[3 * r + 1] = T[3 * r + 1] + T[3 * r + 2] * gv12;
                N9[3 * r + 2] = T[3 * r + 2];
            }
            // B = (Gu * Q) * Gu^T
            const float u00 = gu00 * Q00 + gu01 * Q10, u01 = gu00 * Q01 + gu01 * Q11;
            const float u10 = gu10 * Q00 + gu11 * Q10, u11 = gu10 * Q01 + gu11 * Q11;
            const float u20 = gu20 * Q00 + gu21 * Q10, u21 = gu20 * Q01 + gu21 * Q11;
            P[0] = N9[0] + (u00 * gu00 + u01 * gu01);
            P[1] = N9[1] + (u00 * gu10 + u01 * gu11);
            P[2] = N9[2] + (u00 * gu20 + u01 * gu21);
            P[3] = N9[3] + (u10 * gu00 + u11 * gu01);
            P[4] = N9[4] + (u10 * gu10 + u11 * gu11);
            P[5] = N9[5] + (u10 * gu20 + u11 * gu21);
            P[6] = N9[6] + (u20 * gu00 + u21 * gu01);
            P[7] = N9[7] + (u20 * gu10 + u21 * gu11);
            P[8] = N9[8] + (u20 * gu20 + u21 * gu21);
        }
        if (A.add_noise) {
            // multivariateGauss((V,G), Q, 1) (core.cpp:452): L = chol(Q), (V,G) = L*g + (V,G)
            float g0, g1, g2;
            if (rng.mode == 0) {
                g0 = rng.normals[((size_t) s * 2 + 0) * S + i];
                g1 = rng.normals[((size_t) s * 2 + 1) * S + i];
            } else {
                U4 r = philox4x32((uint32_t) (rng.first_particle + i), A.steps[s].step, 2u, 0u, rng.k0, rng.k1);
                box_muller3(r, g0, g1, g2);
            }
            const L2 L = llt2(Q00, Q10, Q11);
            const float nv = (L.l00 * g0 + 0.0f * g1) + V;
            const float ng = (L.l10 * g0 + L.l11 * g1) + G;
            V = nv;
            G = ng;
        }
        const float nx = x + V * dt * cosf(G + th);
        const float ny = y + V * dt * sinf(G + th);
        const float nth = trig_offset(th + V * dt * sinf(G / wb));  // sin(G/wheelBase): upstream quirk (:103)
        x = nx;
        y = ny;
        th = nth;
        if (A.use_heading) {
            // josephUpdate with H = [0 0 1] (core.cpp:294-317)
            const float v = trig_offset(A.steps[s].phi_true - th);
            const float R = (float) ((double) A.sigma_phi * (double) A.sigma_phi);
            const float ph0 = fs2 ? P[2] : 0.0f, ph1 = fs2 ? P[5] : 0.0f, ph2 = fs2 ? P[8] : 0.0f;
            const float Sx = ph2 + R;
            const float Si = 1.0f / Sx;
            const float w0 = ph0 * Si, w1 = ph1 * Si, w2 = ph2 * Si;
            x = x + w0 * v;
            y = y + w1 * v;
            th = th + w2 * v;
            if (fs2) {
                // C = I - W*H ; P = C P C^T + W R W^T + eps I
                const float C[9] = {1.0f, 0.0f, 0.0f - w0, 0.0f, 1.0f, 0.0f - w1, 0.0f, 0.0f, 1.0f - w2};
                float CP[9], CPC[9];
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        CP[3 * r + c] = (C[3 * r] * P[c] + C[3 * r + 1] * P[3 + c]) + C[3 * r + 2] * P[6 + c];
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        CPC[3 * r + c] = (CP[3 * r] * C[3 * c] + CP[3 * r + 1] * C[3 * c + 1]) + CP[3 * r + 2] * C[3 * c + 2];
                const float wv[3] = {w0, w1, w2};
                const float eps = (float) (2.2204 * 1e-16);
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float p = CPC[3 * r + c] + wv[c] * (wv[r] * R);
                        P[3 * r + c] = p + ((r == c) ? 1.0f : 0.0f) * eps;
                    }
            }
        }
    }
    pose[0 * S + i] = x;
    pose[1 * S + i] = y;
    pose[2 * S + i] = th;
    if (fs2) {
        pose[3 * S + i] = P[0];
        pose[4 * S + i] = P[3];
        pose[5 * S + i] = P[4];
        pose[6 * S + i] = P[6];
        pose[7 * S + i] = P[7];
        pose[8 * S + i] = P[8];
    }
}

// ---------------------------------------------------------------------------------------------------
// Pose estimate (ParticleSLAMWrapper.cpp:56-77): sum x, sum y (double), heading of the first particle
// with the strictly greatest weight.  Block partials, reduced by the last block to arrive.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) estimate_kernel(Buffers B, double *partials, unsigned *counter) {
    __shared__ double sx[kBlock], sy[kBlock];
    __shared__ float sw[kBlock], st[kBlock];
    __shared__ int si[kBlock];
    __shared__ bool last;
    const int t = threadIdx.x;
    const int i = blockIdx.x * kBlock + t;
    const size_t S = (size_t) B.ncap;
    const float *pose = B.pose[B.ctrl->cur];
    const bool act = i < B.n;
    sx[t] = act ? (double) pose[0 * S + i] : 0.0;
    sy[t] = act ? (double) pose[1 * S + i] : 0.0;
    sw[t] = act ? pose[9 * S + i] : -3.0e38f;
    st[t] = act ? pose[2 * S + i] : 0.0f;
    si[t] = act ? i : 0x7fffffff;
    __syncthreads();
    for (int d = kBlock / 2; d > 0; d >>= 1) {
        if (t < d) {
            sx[t] += sx[t + d];
            sy[t] += sy[t + d];
            if (sw[t + d] > sw[t] || (sw[t + d] == sw[t] && si[t + d] < si[t])) {
                sw[t] = sw[t + d];
                st[t] = st[t + d];
                si[t] = si[t + d];
            }
        }
        __syncthreads();
    }
    if (t == 0) {
        double *p = partials + (size_t) blockIdx.x * 4;
        p[0] = sx[0];
        p[1] = sy[0];
        p[2] = (double) st[0];
        p[3] = (double) sw[0];
        __threadfence();
        last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && t == 0) {
        __threadfence();
        double ax = 0.0, ay = 0.0, bt = 0.0, bw = -1e300;
        for (unsigned b = 0; b < gridDim.x; b++) {  // block order == particle order: first strict maximum wins
            const volatile double *p = partials + (size_t) b * 4;
            ax += p[0];
            ay += p[1];
            if (p[3] > bw) {
                bw = p[3];
                bt = p[2];
            }
        }
        Ctrl *c = B.ctrl;
        c->est[0] = ax;
        c->est[1] = ay;
        c->est[2] = bt;
        c->est[3] = bw;
        *counter = 0;
    }
}

// ---------------------------------------------------------------------------------------------------
// Seam 1: batched computeJacobians in the AcceleratorHandler window layout (core.cpp:586-664).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) jacobians_kernel(const float *__restrict__ in, uint32_t n,
                                                            float *__restrict__ out) {
    const uint32_t f = blockIdx.x * kBlock + threadIdx.x;
    if (f >= n) return;
    const float x = in[0], y = in[1], th = in[2];
    // R(i) linear = column-major: R00, R10, R01, R11
    const float r00 = in[3], r10 = in[4], r01 = in[5], r11 = in[6];
    const float *p = in + 7 + 6 * (size_t) f;
    // Pf column-major: P00, P10, P01, P11 (the packed kernels assume symmetry: use the lower entry)
    Jac j = jacobian(x, y, th, p[0], p[1], p[2], p[3], p[5], r00, r01, r10, r11);
    float *o = out + 16 * (size_t) f;
    o[0] = j.zp0; o[1] = j.zp1;
    o[2] = j.hf00; o[3] = j.hf01; o[4] = j.hf10; o[5] = j.hf11;
    o[6] = j.hv00; o[7] = j.hv01; o[8] = 0.0f; o[9] = j.hv10; o[10] = j.hv11; o[11] = -1.0f;
    o[12] = j.s00; o[13] = j.s01; o[14] = j.s10; o[15] = j.s11;
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static inline int blocks_for(int n) { return (n + kBlock - 1) / kBlock; }

static void launch_update(hipStream_t st, const Buffers &B, int method, const ObsPacket *pkt, int m, int n, int nf,
                          const RngArgs &rng, const WeightScratch &ws) {
    const int grid = blocks_for(B.ncap);
    if (method == 2)
        hipLaunchKernelGGL(update_kernel<2>, dim3(grid), dim3(kBlock), 0, st, B, pkt, m, n, nf, rng, ws);
    else
        hipLaunchKernelGGL(update_kernel<1>, dim3(grid), dim3(kBlock), 0, st, B, pkt, m, n, nf, rng, ws);
}

static void launch_finalize(hipStream_t st, const Buffers &B, const WeightScratch &ws, int do_resample,
                            int n_effective) {
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kFinBlock), 0, st, B, ws, do_resample, n_effective);
}

static void launch_resample(hipStream_t st, const Buffers &B, const WeightScratch &ws, const RngArgs &rng, int nf) {
    const int gy = nf > 0 ? (nf + kLmkPerBlockY - 1) / kLmkPerBlockY : 1;
    hipLaunchKernelGGL(resample_kernel, dim3(blocks_for(B.n), gy), dim3(kBlock), 0, st, B, ws, rng, nf);
}

static void launch_predict(hipStream_t st, const Buffers &B, const PredictArgs &A, const RngArgs &rng) {
    hipLaunchKernelGGL(predict_kernel, dim3(blocks_for(B.n)), dim3(kBlock), 0, st, B, A, rng);
}

static void launch_estimate(hipStream_t st, const Buffers &B, double *partials, int nblocks) {
    unsigned *counter = reinterpret_cast<unsigned *>(partials + (size_t) nblocks * 4);
    hipLaunchKernelGGL(estimate_kernel, dim3(nblocks), dim3(kBlock), 0, st, B, partials, counter);
}

static void launch_jacobians(hipStream_t st, const float *in, uint32_t n, float *out) {
    hipLaunchKernelGGL(jacobians_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, in, n, out);
}

static const KernelTable kTable = {launch_update, launch_finalize, launch_resample, launch_predict,
                                   launch_estimate, launch_jacobians};

}  // namespace SLAM_KNS

namespace slamgpu {
#define SLAM_CAT2(a, b) a##b
#define SLAM_CAT(a, b) SLAM_CAT2(a, b)
const KernelTable *SLAM_CAT(kernels_, SLAM_TABLE)() { return &SLAM_KNS::kTable; }
}  // namespace slamgpu
