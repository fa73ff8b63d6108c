// gfx950 kernels of the FastSLAM observation update: one particle per work-item, 16-byte-chunk state in
// HBM, wave64 shuffles for the weight prefix-sum / reductions, device-resident resample decision.
// Compiled twice (see Makefile): -DSLAM_KNS=slam_strict -ffp-contract=off and -DSLAM_KNS=slam_fast.
#include "kernels.h"
#include <type_traits>
#include "device_math.h"

namespace SLAM_KNS {
using namespace slamgpu;

// The queued controls of a step (PredictArgs::steps as 8 floats per step) where the predict loops read them: ALWAYS in LDS (the
// per-step launch parks them there, the persistent loop's queue entry lives there) or null (big maps: read from the kernel
// arguments).  A typed LDS pointer: as a generic `const float *` the loops read them with flat_load + s_waitcnt vmcnt(0) -- every
// step of a predict loop waited for whatever else was in flight (the genealogy chunks of a resampling launch) and paid a flat
// load's latency instead of a ds_read's (round 5, found in the ISA of the heading-observed predicts).
typedef const __attribute__((address_space(3))) float *CtlP;

// Diagnostic build only (make stamps -> libslamgpu_stamps.so, tools/stamps.py): thread 0 of every compute block drains its
// wave's outstanding memory operations and records the 100 MHz wall clock at the levels of the update kernel's
// dependent-load chain.  The drain is the point (the stamp is the time the level's data has ARRIVED); it perturbs the
// overlap a little, so the instrumented kernel is slower than the product's.  Compiled out of the product libraries.
// SLAM_STAMPS=2 (make stamps_flow): no drain -- the stamp is the time the block's first wave REACHED the point in the product's
// own schedule (the waits it would have made anyway included): where the time of an undisturbed launch goes.
#ifdef SLAM_STAMPS
#if SLAM_STAMPS == 2
#define SLAM_STAMP_DRAIN() asm volatile("" ::: "memory")
#else
#define SLAM_STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#endif
#define SLAM_STAMP(k)                                                                                      \
    do {                                                                                                   \
        if (U.stamps && threadIdx.x == 0 && bid < ws.nblocks) {                                            \
            SLAM_STAMP_DRAIN();                                                                            \
            U.stamps[(size_t) bid * kStampSlots + (k)] = wall_clock64();                                   \
        }                                                                                                  \
    } while (0)
#else
#define SLAM_STAMP(k)
#endif

// Streaming (nontemporal) stores for what a launch leaves for the NEXT launch (records, pose, genealogy, weight prefix): the
// L2 does not survive the kernel boundary, and whatever is still dirty in it when the kernel ends is written back ON the
// boundary (MI355X_MICROARCH.md: + bytes / 6 TB/s).  Big maps (2.6 GB of records per launch): 1.63 -> 1.58 ms; small maps
// (10 MB per launch): 14.65 -> 14.35-14.45 us per step in round 3 (round 2 had measured the opposite, 16.0 -> 16.5 us, on
// the kernel of that time).
typedef float nt_v4f __attribute__((ext_vector_type(4)));
SLAM_DEV void nt_store(float4 *p, float4 v) { __builtin_nontemporal_store((nt_v4f){v.x, v.y, v.z, v.w}, reinterpret_cast<nt_v4f *>(p)); }
SLAM_DEV void nt_store(float *p, float v) { __builtin_nontemporal_store(v, p); }

// ---------------------------------------------------------------------------------------------------
// predictState x nsteps with the pose in registers: FastSLAM2::predictState (fastslam2.cpp:70-105)
// [+ observeHeading -> josephUpdate (fastslam2.cpp:113-125, core.cpp:294-317)] or
// FastSLAM1::predictState (fastslam1.cpp:37-54).  P is the full 3x3 (the reference's Pv is not kept
// symmetric by its own float arithmetic; only the stored form is packed).
// ---------------------------------------------------------------------------------------------------
SLAM_DEV void predict_steps(float &x, float &y, float &th, float P[9], const PredictArgs &A, const RngArgs &rng, int i,
                            size_t S) {
    const bool fs2 = A.method == 2;
    const float dt = A.dt, wb = A.wheel_base;
    const float Q00 = A.Q[0], Q01 = A.Q[1], Q10 = A.Q[2], Q11 = A.Q[3];
    for (int s = 0; s < A.nsteps; s++) {
        float V = A.steps[s].V, G = A.steps[s].G;
        float sn, cs;
        sincosf(G + th, &sn, &cs);  // one range reduction for both
        if (fs2) {
            // Gv, Gu (fastslam2.cpp:78-79); sin(G), cos(G) are particle-independent: host-evaluated
            const float gv02 = -V * dt * sn, gv12 = V * dt * cs;
            const float gu00 = dt * cs, gu01 = -V * dt * sn;
            const float gu10 = dt * sn, gu11 = V * dt * cs;
            // (the third row of Gu and its Q products are particle-independent; evaluating them on the host in the same operations
            // was built in round 5: no measurable gain, and the 160 bytes it added to PredictArgs tipped a distributed variant into a
            // 20-byte spill)
            const float gu20 = dt * A.steps[s].sinG / wb, gu21 = V * dt * A.steps[s].cosG / wb;
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
            sincosf(G + th, &sn, &cs);
        }
        const float sgw = A.add_noise ? sinf(G / wb) : A.steps[s].sinGw;
        const float nx = x + V * dt * cs;
        const float ny = y + V * dt * sn;
        const float nth = trig_offset(th + V * dt * sgw);  // sin(G/wheelBase): upstream quirk (:103)
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
}

#ifdef SLAM_FAST_MATH
#ifdef SLAM_FAST_MATH
// FastSLAM1::predictState x nsteps (fastslam1.cpp:37-54) in the fast build's arithmetic: every step samples its own
// (V, G) ~ N((V, G), Q) per particle, so the steps cannot be composed -- but they need not cost ~500 instructions each:
// chol(Q) is step- and particle-independent, the first sincos of predict_steps is dead when the controls are resampled,
// Box-Muller and the trigonometry use the bounded-angle polynomials / hardware transcendentals of this build
// (1.5 ulp, tools/check_fast_math.py).  BASELINE config 2 (1 000 particles): 19.8 -> see DESIGN.md section 5.
// ctl (optional): LDS copy of A.steps (8 dwords per step: V, G first), so that the loop does not fetch a cold line of the
// kernel-argument segment per iteration (scalar cache misses at every launch: ~0.3 us each on the step's critical chain)
// the (V, G) normals of steps s0 .. s0 + nb - 1 (nb <= W) of predict_steps_fs1_fast, made TOGETHER (see there)
template <int W>
SLAM_DEV void draw_batch_fs1_fast(float (&g0)[W], float (&g1)[W], const PredictArgs &A, const RngArgs &rng, int i, size_t S, CtlP ctl,
                                  int s0, int nb) {
    if (rng.mode == 0) {
#pragma unroll
        for (int q = 0; q < W; q++) {
            const int s = s0 + min(q, nb - 1);
            g0[q] = rng.normals[((size_t) s * 2 + 0) * S + i];
            g1[q] = rng.normals[((size_t) s * 2 + 1) * S + i];
        }
    } else {
        uint32_t c1[W];
        U4 r[W];
#pragma unroll
        for (int q = 0; q < W; q++) {
            const int s = s0 + min(q, nb - 1);  // (past the last step: a draw nobody uses)
            c1[q] = ctl ? __float_as_uint(ctl[8 * s + 3]) : A.steps[s].step;
        }
        philox4x32_n<W>(r, (uint32_t) (rng.first_particle + i), c1, 2u, 0u, rng.k0, rng.k1);
#pragma unroll
        for (int q = 0; q < W; q++) {
            float g2;
            box_muller3_fast(r[q], g0[q], g1[q], g2);
        }
    }
}

// ... and the steps applied, one after the other
template <int W>
SLAM_DEV void apply_batch_fs1_fast(float &x, float &y, float &th, const float (&g0)[W], const float (&g1)[W], const PredictArgs &A, CtlP ctl,
                                   float dt, float iwb, const L2 &L, int s0, int nb) {
    // a step's heading needs only the headings before it (a chain of one FMA and one wrap per step); its sine and cosine need
    // only that heading: in a full batch the headings are run first and the W sincos pairs are independent
    float vd[W], gs[W], th_at[W];
    auto heading = [&](int q) {
        const int s = s0 + q;
        const float V = ffma(L.l00, g0[q], ctl ? ctl[8 * s] : A.steps[s].V);
        const float G = ffma(L.l11, g1[q], ffma(L.l10, g0[q], ctl ? ctl[8 * s + 1] : A.steps[s].G));
        float sgw, cgw;
        sincos_cw(G * iwb, sgw, cgw);  // sin(G / wheelBase): upstream quirk (fastslam1.cpp:52)
        vd[q] = V * dt;
        gs[q] = G;
        th_at[q] = th;
        th = wrap_pi(ffma(vd[q], sgw, th));
    };
    auto position = [&](int q) {
        float sn, cs;
        sincos_cw(gs[q] + th_at[q], sn, cs);
        x = ffma(vd[q], cs, x);
        y = ffma(vd[q], sn, y);
    };
    if (nb == W) {
#pragma unroll
        for (int q = 0; q < W; q++) heading(q);
#pragma unroll
        for (int q = 0; q < W; q++) position(q);
    } else {
#pragma unroll
        for (int q = 0; q < W; q++)
            if (q < nb) {
                heading(q);
                position(q);
            }
    }
}

// The same steps in two halves (persistent loop): what needs no pose -- the perturbed controls, V dt, sin(G / wheelBase) -- is made
// an iteration ahead by a drawer workgroup (controls_batch_fs1_fast); the pose's own chain applies it (apply_controls_fs1_fast).
// The operations of apply_batch_fs1_fast's `heading` and `position`, each on the same values: identical bits.
template <int W>
SLAM_DEV void controls_batch_fs1_fast(float (&vd)[W], float (&gs)[W], float (&sgw)[W], const float (&g0)[W], const float (&g1)[W], const PredictArgs &A,
                                      CtlP ctl, float dt, float iwb, const L2 &L, int nb) {
#pragma unroll
    for (int q = 0; q < W; q++) {
        const int s = min(q, nb - 1);  // (past the last step: values nobody uses)
        const float V = ffma(L.l00, g0[q], ctl ? ctl[8 * s] : A.steps[s].V);
        const float G = ffma(L.l11, g1[q], ffma(L.l10, g0[q], ctl ? ctl[8 * s + 1] : A.steps[s].G));
        float cgw;
        sincos_cw(G * iwb, sgw[q], cgw);  // sin(G / wheelBase): upstream quirk (fastslam1.cpp:52)
        vd[q] = V * dt;
        gs[q] = G;
    }
}
template <int W>
SLAM_DEV void apply_controls_fs1_fast(float &x, float &y, float &th, const float (&vd)[W], const float (&gs)[W], const float (&sgw)[W], int nb) {
    float th_at[W];
    auto heading = [&](int q) {
        th_at[q] = th;
        th = wrap_pi(ffma(vd[q], sgw[q], th));
    };
    auto position = [&](int q) {
        float sn, cs;
        sincos_cw(gs[q] + th_at[q], sn, cs);
        x = ffma(vd[q], cs, x);
        y = ffma(vd[q], sn, y);
    };
    if (nb == W) {
#pragma unroll
        for (int q = 0; q < W; q++) heading(q);
#pragma unroll
        for (int q = 0; q < W; q++) position(q);
    } else {
#pragma unroll
        for (int q = 0; q < W; q++)
            if (q < nb) {
                heading(q);
                position(q);
            }
    }
}

template <int W>
SLAM_DEV void predict_batch_fs1_fast(float &x, float &y, float &th, const PredictArgs &A, const RngArgs &rng, int i, size_t S, CtlP ctl,
                                     float dt, float iwb, const L2 &L, int s0, int nb) {
    float g0[W], g1[W];
    draw_batch_fs1_fast<W>(g0, g1, A, rng, i, S, ctl, s0, nb);
    apply_batch_fs1_fast<W>(x, y, th, g0, g1, A, ctl, dt, iwb, L, s0, nb);
}

SLAM_DEV void predict_steps_fs1_fast(float &x, float &y, float &th, const PredictArgs &A, const RngArgs &rng, int i, size_t S,
                                     CtlP ctl = nullptr) {
    const float dt = A.dt, iwb = 1.0f / A.wheel_base;
    const L2 L = llt2(A.Q[0], A.Q[2], A.Q[3]);  // multivariateGauss((V,G), Q, 1) (core.cpp:452)
    // The draws of up to eight steps are made TOGETHER (they depend on counters only), then the steps are applied one after the
    // other (each needs the heading the one before left): for a wave that has its SIMD to itself -- BASELINE config 2: 16 waves on
    // 16 SIMDs -- a dependent instruction issues every ~9.6 cycles, four independent chains at ~2.9 each
    // (tools/microbench/valu_latency.hip), and one Philox4x32-10 + Box-Muller is ~130 instructions of mostly ONE chain.  Same
    // counters, same integers, same float operations per draw and step: bit-identical to drawing inside the step.
    // Measured at config 2 (eight steps per update), one box: draws inside the step 13.02-13.10 us per step, batches of four
    // 12.34, of eight 12.13-12.24.
    // (Drawing at the HEAD of the launch, behind the loads in flight, was built and measured in round 4 and was slower, 14.55
    // against 14.22 us per step: the compiler sinks the head's loads below the inserted loop and the values take an LDS hop.)
    for (int s0 = 0; s0 < A.nsteps;) {
        const int rem = A.nsteps - s0;
        if (rem > 4) {
            predict_batch_fs1_fast<8>(x, y, th, A, rng, i, S, ctl, dt, iwb, L, s0, min(rem, 8));
            s0 += 8;
        } else {
            predict_batch_fs1_fast<4>(x, y, th, A, rng, i, S, ctl, dt, iwb, L, s0, rem);
            s0 += 4;
        }
    }
}
#endif

// FastSLAM2::predictState + observeHeading -> josephUpdate (fastslam2.cpp:70-125, core.cpp:294-317) x nsteps in the fast build's
// arithmetic (round 5): three of the four bundled maps observe the heading (SWITCH_HEADING_KNOWN), which makes every step depend
// on the particle's own heading, so the queued steps cannot be folded into one (predict_composite) -- and predict_steps replays
// the reference's full 3x3 products, ~190 instructions per step.  Here: symmetric-packed covariance; Gv Pv Gv^T as the rank-one
// form of predict_composite; Gu Q Gu^T from the host-evaluated sin(G), cos(G); the heading update with H = (0 0 1) in gain form,
// P - K (H P) (equal to the Joseph form in exact arithmetic for the optimal gain the reference computes; the reference's
// "+ 2.2204e-16 I" kept): ~60 instructions per step.  Same tolerances as the rest of the fast build (tests/test_gpu_parity.py).
SLAM_DEV void predict_steps_heading_fast(float &x, float &y, float &th, Sym3 &P, const PredictArgs &A, CtlP ctl) {
    const float dt = A.dt, iwb = 1.0f / A.wheel_base;
    const float q00 = A.Q[0], q10 = 0.5f * (A.Q[1] + A.Q[2]), q11 = A.Q[3];
    const float Rphi = A.sigma_phi * A.sigma_phi;
    const float eps = (float) (2.2204 * 1e-16);
    for (int s = 0; s < A.nsteps; s++) {
        const float V = ctl ? ctl[8 * s] : A.steps[s].V, G = ctl ? ctl[8 * s + 1] : A.steps[s].G;
        const float phi = ctl ? ctl[8 * s + 2] : A.steps[s].phi_true;
        const float sinG = ctl ? ctl[8 * s + 4] : A.steps[s].sinG, cosG = ctl ? ctl[8 * s + 5] : A.steps[s].cosG;
        const float sgw = ctl ? ctl[8 * s + 6] : A.steps[s].sinGw;
        float sn, cs;
        sincos_cw(G + th, sn, cs);
        const float vd = V * dt;
        // Gv Pv Gv^T, Gv = I + (f0, f1, 0)^T e3^T
        const float f0 = -vd * sn, f1 = vd * cs;
        const float n20 = ffma(f0, P.p22, P.p20), n21 = ffma(f1, P.p22, P.p21);
        const float n00 = ffma(f0, P.p20 + n20, P.p00);
        const float n10 = ffma(f1, P.p20, ffma(f0, n21, P.p10));
        const float n11 = ffma(f1, P.p21 + n21, P.p11);
        // Gu Q Gu^T, Gu = [[dt cs, -vd sn], [dt sn, vd cs], [dt sinG / wb, vd cosG / wb]]
        const float g00 = dt * cs, g01 = f0, g10 = dt * sn, g11 = f1, g20 = dt * sinG * iwb, g21 = vd * cosG * iwb;
        const float u00 = ffma(g00, q00, g01 * q10), u01 = ffma(g00, q10, g01 * q11);
        const float u10 = ffma(g10, q00, g11 * q10), u11 = ffma(g10, q10, g11 * q11);
        const float u20 = ffma(g20, q00, g21 * q10), u21 = ffma(g20, q10, g21 * q11);
        float p00 = n00 + ffma(u00, g00, u01 * g01);
        float p10 = n10 + ffma(u10, g00, u11 * g01);
        float p11 = n11 + ffma(u10, g10, u11 * g11);
        float p20 = n20 + ffma(u20, g00, u21 * g01);
        float p21 = n21 + ffma(u20, g10, u21 * g11);
        float p22 = P.p22 + ffma(u20, g20, u21 * g21);
        x = ffma(vd, cs, x);
        y = ffma(vd, sn, y);
        th = wrap_pi(ffma(vd, sgw, th));  // sin(G / wheelBase): upstream quirk (fastslam2.cpp:103)
        // observeHeading: v = wrap(phi - th), S = P22 + R, K = P(:, 2) / S
        const float v = wrap_pi(phi - th);
        const float si = __builtin_amdgcn_rcpf(p22 + Rphi);
        const float k0 = p20 * si, k1 = p21 * si, k2 = p22 * si;
        x = ffma(k0, v, x);
        y = ffma(k1, v, y);
        th = ffma(k2, v, th);
        P.p00 = ffma(-k0, p20, p00) + eps;
        P.p10 = ffma(-k1, p20, p10);
        P.p11 = ffma(-k1, p21, p11) + eps;
        P.p20 = ffma(-k2, p20, p20);
        P.p21 = ffma(-k2, p21, p21);
        P.p22 = ffma(-k2, p22, p22) + eps;
    }
}

// All queued predicts as one step (PredictComposite, kernels.h): one sincos and ~60 FMAs per particle instead of
// ~190 instructions per queued predict.
SLAM_DEV void predict_composite(float &x, float &y, float &th, Sym3 &P, const PredictComposite &C) {
    float s, c;
    sincos_cw(th, s, c);
    const float dx = ffma(c, C.ax, -s * C.ay), dy = ffma(s, C.ax, c * C.ay);
    const float f0 = -dy, f1 = dx;  // F = I + (f0, f1, 0)^T e3^T
    // F P F^T
    const float n20 = ffma(f0, P.p22, P.p20), n21 = ffma(f1, P.p22, P.p21);
    const float n00 = ffma(f0, P.p20 + n20, P.p00);
    const float n10 = ffma(f1, P.p20, ffma(f0, n21, P.p10));
    const float n11 = ffma(f1, P.p21 + n21, P.p11);
    // T M T^T, T = diag(R(th), 1)
    const float t00 = ffma(c, C.m00, -s * C.m10), t01 = ffma(c, C.m10, -s * C.m11);
    const float t10 = ffma(s, C.m00, c * C.m10), t11 = ffma(s, C.m10, c * C.m11);
    P.p00 = n00 + ffma(t00, c, -t01 * s);
    P.p10 = n10 + ffma(t10, c, -t11 * s);
    P.p11 = n11 + ffma(t10, s, t11 * c);
    P.p20 = n20 + ffma(c, C.m20, -s * C.m21);
    P.p21 = n21 + ffma(s, C.m20, c * C.m21);
    P.p22 = P.p22 + C.m22;
    x += dx;
    y += dy;
    th = wrap_pi(th + C.dth);
}
#endif

__global__ void __launch_bounds__(kBlock) predict_kernel(Buffers B, PredictArgs A, RngArgs rng) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= B.n) return;
    const int cur = B.ctrl->live[B.slot];
    float4 a = B.poseA[cur][i];
#ifdef SLAM_FAST_MATH
    if (A.comp.valid) {
        const float4 b = B.poseB[cur][i];
        const float2 c = B.poseC[cur][i];
        Sym3 P = {b.x, b.y, b.z, b.w, c.x, c.y};
        predict_composite(a.x, a.y, a.z, P, A.comp);
        B.poseA[cur][i] = a;
        B.poseB[cur][i] = make_float4(P.p00, P.p10, P.p11, P.p20);
        B.poseC[cur][i] = make_float2(P.p21, P.p22);
        return;
    }
    if (A.method == 2 && A.use_heading && !A.add_noise) {  // (the same arithmetic as inside the update launch)
        const float4 b = B.poseB[cur][i];
        const float2 c = B.poseC[cur][i];
        Sym3 P = {b.x, b.y, b.z, b.w, c.x, c.y};
        predict_steps_heading_fast(a.x, a.y, a.z, P, A, nullptr);
        B.poseA[cur][i] = a;
        B.poseB[cur][i] = make_float4(P.p00, P.p10, P.p11, P.p20);
        B.poseC[cur][i] = make_float2(P.p21, P.p22);
        return;
    }
    if (A.method == 1 && A.add_noise && !A.use_heading) {  // (the same arithmetic as inside the update launch)
        predict_steps_fs1_fast(a.x, a.y, a.z, A, rng, i, (size_t) B.ncap);
        B.poseA[cur][i] = a;
        return;
    }
#endif
    float P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool fs2 = A.method == 2;
    if (fs2) {
        const float4 b = B.poseB[cur][i];
        const float2 c = B.poseC[cur][i];
        P[0] = b.x; P[1] = b.y; P[2] = b.w;
        P[3] = b.y; P[4] = b.z; P[5] = c.x;
        P[6] = b.w; P[7] = c.x; P[8] = c.y;
    }
    predict_steps(a.x, a.y, a.z, P, A, rng, i, (size_t) B.ncap);
    B.poseA[cur][i] = a;
    if (fs2) {
        B.poseB[cur][i] = make_float4(P[0], P[3], P[4], P[6]);
        B.poseC[cur][i] = make_float2(P[7], P[8]);
    }
}

// ---------------------------------------------------------------------------------------------------
// block-level helpers
// ---------------------------------------------------------------------------------------------------
struct EstItem {
    double sx, sy;
    float w, th;
    int idx;
};

SLAM_DEV void est_combine(EstItem &a, const EstItem &b) {
    a.sx += b.sx;
    a.sy += b.sy;
    if (b.w > a.w || (b.w == a.w && b.idx < a.idx)) {
        a.w = b.w;
        a.th = b.th;
        a.idx = b.idx;
    }
}

// reduce over the 256 threads of a block; result valid in thread 0.  Sums on the DPP path (device_math.h); the first
// strictly greatest weight of a wave = the lowest lane holding the wave's maximum (indices ascend with the lane).
// ... in two halves, so that a caller with a barrier of its own can put it between them: the wave's part (every lane returns the
// wave's result; lane 0 parks it in sh[wave]) ...
// DPP: the maximum by DPP moves inside the VALU (wave_max_f) instead of six LDS-crossbar shuffles (~0.2 us of a launch's tail for a
// wave that has its SIMD to itself; a maximum is exact whatever the pairing: same value).  Not in the distributed variants of
// update_kernel: they sit at the register limit, and one of them (FastSLAM 1, compact, strict build) spills 20 bytes with it.
template <bool DPP = true>
SLAM_DEV EstItem wave_reduce_est(EstItem v, EstItem *sh) {
    v.sx = wave_sum_d(v.sx);
    v.sy = wave_sum_d(v.sy);
    float wm = v.w;
    if constexpr (DPP) {
        wm = wave_max_f(v.w);
    } else {
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) wm = fmaxf(wm, __shfl_xor(wm, d, kWave));
    }
    const unsigned long long holders = __ballot(v.w == wm);
    if (holders) {
        const int src = __ffsll((long long) holders) - 1;
        v.w = wm;
        v.th = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.th), src));
        v.idx = __builtin_amdgcn_readlane(v.idx, src);
    }
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    if (lane == 0) sh[wv] = v;
    return v;
}
// ... and, behind a barrier, thread 0's combination of the waves' results (v: what wave_reduce_est returned to it)
SLAM_DEV EstItem combine_waves_est(EstItem v, const EstItem *sh) {
#pragma unroll
    for (int k = 1; k < kBlock / kWave; k++) est_combine(v, sh[k]);
    return v;
}
SLAM_DEV EstItem block_reduce_est(EstItem v, EstItem *sh) {
    v = wave_reduce_est(v, sh);
    __syncthreads();
    if (threadIdx.x == 0) v = combine_waves_est(v, sh);
    return v;
}

// One block: parallel reduction of the per-block pose-estimate partials (doubles, tree order fixed by the launch
// geometry) into Ctrl.est and a history slot.
SLAM_DEV void finish_estimate(const Buffers &B, const WeightScratch &ws, int par, double *hist, EstItem *sh) {
    EstItem v{0.0, 0.0, -3.0e38f, 0.0f, 0x7fffffff};
    for (int b = threadIdx.x; b < ws.nblocks; b += kBlock) {
        const double *p = ws.est_part[par] + (size_t) b * 4;
        EstItem o{p[0], p[1], (float) p[3], (float) p[2], b};
        est_combine(v, o);
    }
    __syncthreads();
    v = block_reduce_est(v, sh);
    if (threadIdx.x == 0) {
        Ctrl *c = B.ctrl;
        c->est[0] = v.sx;
        c->est[1] = v.sy;
        c->est[2] = (double) v.th;
        c->est[3] = (double) v.w;
        if (hist) {
            hist[0] = v.sx;
            hist[1] = v.sy;
            hist[2] = (double) v.th;
            hist[3] = (double) v.w;
            // the resampling record of the update these partials belong to: left behind the partials by whoever
            // planned it (Ctrl.neff may already belong to a later update)
            hist[4] = ws.est_part[par][4 * (size_t) ws.nblocks];
            hist[5] = ws.est_part[par][4 * (size_t) ws.nblocks + 1];
        }
    }
}

// Exclusive prefix (double) of the block totals into LDS `off[0..nb]`, plus sum w and sum w^2.  Every block of
// every kernel (and every rank of a sharded run, which sees the same all-gathered totals) executes exactly
// this association, so W, Q, Neff, the resample decision and all ancestors are identical everywhere.
// Layout of the totals: shard-major records [w(nbl) | w2(nbl)] (what one all-gather of each shard's contiguous
// [w | w2] block produces): total k of shard k/nbl sits at (k/nbl)*2*nbl + k%nbl, its square sum nbl further.
// Log-weight contexts (logw): row 3 of the totals holds each block's largest log-weight M_b and rows 1 / 2 are sums of
// exp(l - M_b) and its square; a first pass finds M = max M_b and every total is rescaled by exp(M_b - M) (double), so W, Q
// and the prefix are those of exp(l - M).  With linear weights the factor is exactly 1.0: same bits as without it.
SLAM_DEV double block_scale(float mb, double M) {
    return mb == -INFINITY ? 0.0 : exp((double) mb - M);
}

struct NoOp {
    SLAM_DEV void operator()() const {}
};

// after_loads(): called once this function's own global loads have been requested (contexts of at most 512 blocks: at
// most two totals per thread) -- the place for a caller to request data it needs AFTER the scan, so that it queues behind
// the totals on the in-order return path instead of in front of them.
// The loads of the scan (contexts of at most 512 blocks: at most two totals per thread), separated from the arithmetic so
// that a caller can request them before anything else (update_kernel: at kernel entry, from preloaded arguments).
constexpr int kScanBatch = 4;  // totals of a thread's segment requested together by the scan of a wide table (scan_finish)
struct ScanLoads {
    float tv0, tv1, qv0, qv1, mv0, mv1;
};
SLAM_DEV int scan_at(int k, int nb, int nbl, bool logw) {
    // (one shard: nbl == nb and the index is k itself; the general form costs two integer divisions per entry)
    return nbl == nb ? k : (k / nbl) * (logw ? 3 : 2) * nbl + (k % nbl);
}
// BYP: the totals were stored by other workgroups of the RUNNING launch (the persistent step loop's tiles, log-weight contexts:
// linear ones take scan_issue_small): read them past the vector cache (device_math.h: ldg) -- a tile skips the cache invalidate
// at the loop's meeting, so a plain load could be served a line of two iterations ago
template <bool BYP = false>
SLAM_DEV ScanLoads scan_issue(const float *__restrict__ tot, int nb, int nbl, bool logw) {
    const int t = threadIdx.x;
    const int per = (nb + kBlock - 1) / kBlock;
    const int lo = min(nb, t * per), hi = min(nb, lo + per);
    // (named scalars, not arrays: a register array indexed by k - lo would be demoted to scratch)
    ScanLoads L{0.0f, 0.0f, 0.0f, 0.0f, -INFINITY, -INFINITY};
    if (per <= 2) {
        if (lo < hi) {
            const int at = scan_at(lo, nb, nbl, logw);
            L.tv0 = ldg<BYP>(tot + at);
            L.qv0 = ldg<BYP>(tot + at + nbl);
            if (logw) L.mv0 = ldg<BYP>(tot + at + 2 * nbl);
        }
        if (lo + 1 < hi) {
            const int at = scan_at(lo + 1, nb, nbl, logw);
            L.tv1 = ldg<BYP>(tot + at);
            L.qv1 = ldg<BYP>(tot + at + nbl);
            if (logw) L.mv1 = ldg<BYP>(tot + at + 2 * nbl);
        }
    }
    return L;
}

// Wide tables (the gathered totals of several shards: more than two per thread), linear weights: the whole table by LDS-DMA
// (global_load_lds: memory -> LDS, no registers) into `stab` -- w[nb] then, kScanPad further, q[nb] -- every request of the block
// in flight at once, where scan_finish's batches of kScanBatch are ceil(per / kScanBatch) dependent trips at the head of every
// launch (profiles/dist_width_r04.txt: ~+4 us at 8 shards).  Lane l of a wave lands at base + l: instruction u of thread t
// carries entry k = t + 256 u, so the table arrives in index order.
SLAM_DEV int scan_pad(int nb) { return (nb + 255) & ~255; }
SLAM_DEV void scan_issue_dma(const float *__restrict__ tot, int nb, int nbl, float *stab) {
    const int t = threadIdx.x, wbase = (t / kWave) * kWave;
    float *const qtab = stab + scan_pad(nb);
    int sh = t / nbl, r = t - sh * nbl;  // entry k = t + 256 u lives at [shard sh][row][r]
    for (int k0 = 0; k0 < nb; k0 += kBlock) {
        if (k0 + t < nb) {
            const float *src = tot + (size_t) sh * 2 * nbl + r;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *) src,
                                             (void __attribute__((address_space(3))) *) (stab + k0 + wbase), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *) (src + nbl),
                                             (void __attribute__((address_space(3))) *) (qtab + k0 + wbase), 4, 0, 0);
        }
        r += kBlock;
        while (r >= nbl) {
            r -= nbl;
            sh++;
        }
    }
}

SLAM_DEV void scan_finish(const ScanLoads L, const float *__restrict__ tot, int nb, int nbl, bool logw, double *off, double *sh_a,
                          double *sh_q, double &W, double &Q, double &M, const float *stab = nullptr) {
    const int t = threadIdx.x, lane = t & (kWave - 1), wv = t / kWave;
    const int per = (nb + kBlock - 1) / kBlock;
    const int lo = min(nb, t * per), hi = min(nb, lo + per);
    const bool two = per <= 2;
    auto at_of = [&](int k) { return scan_at(k, nb, nbl, logw); };
    M = 0.0;
    if (logw) {
        float mx = -INFINITY;
        if (two) mx = fmaxf(L.mv0, L.mv1);
        else
            for (int k = lo; k < hi; k++) mx = fmaxf(mx, tot[at_of(k) + 2 * nbl]);
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, kWave));
        if (lane == 0) sh_a[wv] = (double) mx;
        __syncthreads();
        M = fmax(fmax(sh_a[0], sh_a[1]), fmax(sh_a[2], sh_a[3]));
        __syncthreads();
    }
    double a = 0.0, q = 0.0;
    auto acc = [&](int k, float tk_f, float qk_f, float mk_f) {
        const double sc = logw ? block_scale(mk_f, M) : 1.0;
        const double tk = (double) tk_f * sc;
        off[k] = tk;  // this thread's own segment: read back below
        a += tk;
        q += (double) qk_f * (tk * tk);  // second row: sum (w_i / T)^2 of the block (update_kernel's tail)
    };
    if (two) {
        if (lo < hi) acc(lo, L.tv0, L.qv0, L.mv0);
        if (lo + 1 < hi) acc(lo + 1, L.tv1, L.qv1, L.mv1);
    } else if (stab) {
        // the table is in LDS (scan_issue_dma): every wave's requests must have landed before anybody reads.  (The wait is spelt
        // out: gfx950's barrier does not imply it, and a workgroup-scope fence need not wait on vmcnt: ADVICE r4)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const float *qtab = stab + scan_pad(nb);
        for (int k = lo; k < hi; k++) acc(k, stab[k], qtab[k], 0.0f);
        // (stab is the memory of the ancestor windows: the barriers below stand between these reads and the first window)
    } else {
        // (gathered tables of several shards: [shard][w(nbl) | q(nbl) (| m(nbl))]: walk the index instead of dividing per
        // entry, and request kScanBatch totals of the thread's segment before using any of them.  Round 4 priced this loop for
        // the table of 8 shards x 100 096 particles (13 totals per thread = four dependent trips at the head of every launch:
        // tools/dist_width.py, profiles/dist_width_r04.txt: ~+4 us per launch against one shard) and tried larger batches and a
        // burst of the whole segment into registers: the distributed kernels sit at the scalar-register limit and one or two of
        // their variants then spill (tests/test_host_frontend.py::test_no_kernel_spills_to_scratch), chaotically in the batch
        // size (6, 7, 8, 16 all do, in different variants).  Left at four; the remedy that needs no registers is LDS-DMA
        // (global_load_lds) of the table into `off`: DESIGN.md section 10)
        const int rows = logw ? 3 : 2;
        int sh = lo < hi ? lo / nbl : 0, r = lo < hi ? lo - sh * nbl : 0;
        for (int k0 = lo; k0 < hi; k0 += kScanBatch) {
            float tk[kScanBatch], qk[kScanBatch], mk[kScanBatch];
            int shq = sh, rq = r;
#pragma unroll
            for (int u = 0; u < kScanBatch; u++) {
                const bool on = k0 + u < hi;
                const int at = on ? shq * rows * nbl + rq : 0;
                tk[u] = on ? tot[at] : 0.0f;
                qk[u] = on ? tot[at + nbl] : 0.0f;
                mk[u] = (on && logw) ? tot[at + 2 * nbl] : 0.0f;
                if (++rq == nbl) {
                    rq = 0;
                    shq++;
                }
            }
#pragma unroll
            for (int u = 0; u < kScanBatch; u++)
                if (k0 + u < hi) acc(k0 + u, tk[u], qk[u], mk[u]);
            sh = shq;
            r = rq;
        }
    }
    const double sa = wave_scan_d(a);
    const double sq = wave_sum_d(q);
    if (lane == kWave - 1) sh_a[wv] = sa;
    if (lane == 0) sh_q[wv] = sq;
    __syncthreads();
    double base = 0.0;
#pragma unroll
    for (int k = 0; k < kBlock / kWave; k++)
        if (k < wv) base += sh_a[k];
    double run = base + sa - a;  // exclusive prefix of this thread's segment
    for (int k = lo; k < hi; k++) {
        const double tk = off[k];
        off[k] = run;
        run += tk;
    }
    W = ((sh_a[0] + sh_a[1]) + sh_a[2]) + sh_a[3];
    Q = ((sh_q[0] + sh_q[1]) + sh_q[2]) + sh_q[3];
    if (t == 0) off[nb] = W;
}

template <class After = NoOp>
SLAM_DEV void scan_block_totals(const float *__restrict__ tot, int nb, int nbl, bool logw, double *off, double *sh_a,
                                double *sh_q, double &W, double &Q, double &M, After after_loads = After()) {
    const ScanLoads L = scan_issue(tot, nb, nbl, logw);
    after_loads();
    scan_finish(L, tot, nb, nbl, logw, off, sh_a, sh_q, W, Q, M);
}

// scan_finish for a table of at most 64 totals (one per lane of ONE wave: at most 16 384 particles), linear weights, by every wave
// for itself: no barrier.  In scan_finish the first wave holds all the totals and the other waves hold zeros, so W is the first
// wave's sum plus three zeros and the exclusive prefix is that wave's; here every wave redoes exactly those operations on those
// values (identical bits) and all of them store the same prefix into `off` (LDS: a wave reads back what it wrote itself).
SLAM_DEV ScanLoads scan_issue_small(const float *__restrict__ tot, int nb) {  // (lane l of EVERY wave: total l)
    const int lane = threadIdx.x & (kWave - 1);
    ScanLoads L{0.0f, 0.0f, 0.0f, 0.0f, -INFINITY, -INFINITY};
    if (lane < nb) {
        L.tv0 = ldg<true>(tot + lane);  // (another workgroup's stores of this launch: past the vector cache, device_math.h: ldg)
        L.qv0 = ldg<true>(tot + nb + lane);
    }
    return L;
}
SLAM_DEV void scan_small(const ScanLoads L, int nb, double *off, double &W, double &Q) {
    const int lane = threadIdx.x & (kWave - 1);
    double a = 0.0, q = 0.0;
    if (lane < nb) {
        const double tk = (double) L.tv0 * 1.0;
        a += tk;
        q += (double) L.qv0 * (tk * tk);
    }
    const double sa = wave_scan_d(a);
    const double sq = wave_sum_d(q);
    const double s0 = wave_last_d(sa);
    W = ((s0 + 0.0) + 0.0) + 0.0;
    Q = ((sq + 0.0) + 0.0) + 0.0;
    if (lane < nb) off[lane] = (0.0 + sa) - a;
    if (lane == 0) off[nb] = W;
    __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave execute in order)
}

// Neff = W^2 / Q, the same expression in every kernel, block and shard (the decision must be identical everywhere): one
// v_rcp_f64 + a Newton step instead of the ~40-instruction IEEE double division on the step's critical path
SLAM_DEV float neff_of(double W, double Q) {
    double r = __builtin_amdgcn_rcp(Q);
    r = fma(fma(-Q, r, 1.0), r, r);
    return (float) ((W * W) * r);
}

// kStatus* bits of a resampling stage from the two global sums (NaN-safe comparisons)
SLAM_DEV int weight_status(double W, double Q) {
    return (W > 0.0 && W < 1.0e300 && Q > 0.0 && Q < 1.0e300) ? 0 : kStatusDegenerate;
}

// stratum of global output particle gid: the caller's tape, or (gid + u)/N with u from Philox stream 1
SLAM_DEV float stratum(const RngArgs &rng, int64_t gid) {
    if (rng.mode == 0) return rng.strata[gid];
    U4 r = philox4x32((uint32_t) gid, rng.step, 1u, 0u, rng.k0, rng.k1);
    const double u = ((double) (r.x >> 8) + 0.5) * (1.0 / 16777216.0);
    return (float) (((double) gid + u) / (double) rng.n_global);
}

// the same for the PREVIOUS update (inline planning inside the next update launch)
SLAM_DEV float stratum_prev(const RngArgs &rng, int64_t gid) {
    if (rng.mode == 0) return rng.strata_prev[gid];
    U4 r = philox4x32((uint32_t) gid, rng.prev_step, 1u, 0u, rng.k0, rng.k1);
    const double u = ((double) (r.x >> 8) + 0.5) * (1.0 / 16777216.0);
    return (float) (((double) gid + u) / (double) rng.n_global);
}

// ancestor (global particle index) of a stratum: min{ i : target < cumsum_i }, two-level binary search
// blk_m != null (log-weight contexts): the in-block prefix of block b is that of exp(l - M_b); it is rescaled by
// exp(M_b - M) like the block totals (scan_block_totals).  Linear weights: factor 1.0, same bits as without it.
SLAM_DEV int64_t find_ancestor(double target, const double *off, int nb, const float *__restrict__ lcum_local,
                               int first_block, int nb_local, int64_t n_global, const float *__restrict__ blk_m = nullptr,
                               double M = 0.0, const PeerPtrs *peers = nullptr, int par = 0) {
    int b0 = 0, b1 = nb;
    while (b0 < b1) {
        const int mid = (b0 + b1) >> 1;
        if (off[mid + 1] > target) b1 = mid; else b0 = mid + 1;
    }
    if (b0 >= nb) return n_global - 1;  // select beyond the last cumulative weight: undefined upstream (keep = -1), clamp
    // the in-block prefix is only resident for this shard's blocks; callers only ask for strata they own
    const int lb = min(max(b0 - first_block, 0), nb_local - 1);
    const double o = off[b0];
    const double sc = blk_m ? block_scale(blk_m[b0], M) : 1.0;
    // (distributed contexts: the in-block prefix of a block of another shard is read out of that GPU's memory)
    const float *lc = peers ? peers[b0 / nb_local].lcum[par] + (size_t) (b0 % nb_local) * kBlock : lcum_local + (size_t) lb * kBlock;
    // first slot with o + lc > target; the last slot if rounding hides it.  The prefix is non-decreasing, so instead of
    // 8 dependent probes: 16 pivots in flight together (every 16th entry), then the 16 entries of the pivot's segment
    float pv[16];
#pragma unroll
    for (int q = 0; q < 16; q++) pv[q] = lc[16 * q + 15];
    int seg = 15;
#pragma unroll
    for (int q = 14; q >= 0; q--)
        if (o + (double) pv[q] * sc > target) seg = q;
    const float4 *l4 = reinterpret_cast<const float4 *>(lc + 16 * seg);
    const float4 e0 = l4[0], e1 = l4[1], e2 = l4[2], e3 = l4[3];
    const float ev[16] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w, e2.x, e2.y, e2.z, e2.w, e3.x, e3.y, e3.z, e3.w};
    int r = 15;
#pragma unroll
    for (int q = 14; q >= 0; q--)
        if (o + (double) ev[q] * sc > target) r = q;
    const int l0 = 16 * seg + r;
    return min((int64_t) b0 * kBlock + l0, n_global - 1);
}

// The same search, cooperatively by a WAVE (update_kernel, inline plan).  A wave's 64 consecutive outputs draw strata that lie
// next to each other, so their ancestors live in one or two source blocks: the wave stages those blocks' in-block prefixes
// (1 KB each: ONE coalesced 16-byte load per lane and block) in its own LDS window and every lane searches there -- instead
// of every lane fetching a 64-byte segment of its own, found through a pivot table of ALL blocks (25 KB at 10^5 particles)
// that every block had to pull into LDS at kernel entry, resampling step or not (round 2; measured equal in time on one GPU,
// 16.00 against 16.02 us per step, at a thirteenth of the search's traffic and without the table).  Same comparisons on the same values as
// find_ancestor: the same ancestors, bit for bit.  Lanes whose block falls outside the window (kWinBlocks source blocks:
// never seen on the bundled maps; possible with degenerate weights) take the per-lane path.  ALL lanes of the wave must call
// (`valid` = the lane has an output particle); distributed contexts stage peer blocks straight out of the owning GPU's memory.
constexpr int kWinBlocks = 4;
// PRE: the caller has already staged blocks 0 .. nb - 1 (nb <= kWinBlocks) in `win` (persistent loop: requested at the head of the
// iteration, with the block totals: one trip less between the decision and the ancestor)
// BYP: the prefixes were stored by other workgroups of the running launch: read them past the vector cache (device_math.h: ldg)
// OFFL: `off_g` points into LDS (the block prefix every block builds for itself: everything but the contexts of more than
// scan_min_blocks tiles, whose prefix scan_kernel leaves in global memory): read through a typed LDS pointer -- as a generic
// pointer every probe of the search below was a flat load with a full s_waitcnt (vmcnt and lgkmcnt) behind it.
template <bool PRE = false, bool BYP = false, bool OFFL = false>
SLAM_DEV int64_t find_ancestor_win(double target, bool valid, int guess, const double *off_g, int nb, float *win, const float *__restrict__ lcum_local,
                                   int nb_local, int64_t n_global, const float *__restrict__ blk_m, double M, const PeerPtrs *peers, int par) {
    using OffP = std::conditional_t<OFFL, const __attribute__((address_space(3))) double *, const double *>;
    const OffP off = (OffP) off_g;
    const int lane = threadIdx.x & (kWave - 1);
    // The source block: the first b with off[b + 1] > target (nb: none).  Stratified ancestors sit near their offspring, so
    // the search starts at the block the caller names (the particle's own) and gallops outwards -- two or three dependent LDS
    // reads in the common case instead of the nine of a bisection over all blocks -- then bisects the bracket it found: the
    // same answer, the predicate is monotone.
    int b0 = 0;
    if (valid) {
        const int g = min(max(guess, 0), nb - 1);
        int s_lo, s_hi;
        if (off[g + 1] > target) {  // the answer is at or below g
            s_hi = g;
            s_lo = g - 1;
            for (int step = 1; s_lo >= 0 && off[s_lo + 1] > target; step <<= 1) {
                s_hi = s_lo;
                s_lo -= step;
            }
            s_lo = max(s_lo, -1) + 1;
        } else {                    // above g
            s_lo = s_hi = g + 1;
            for (int step = 1; s_hi < nb && !(off[s_hi + 1] > target); step <<= 1) {
                s_lo = s_hi + 1;
                s_hi += step;
            }
            s_hi = min(s_hi, nb);
        }
        while (s_lo < s_hi) {
            const int mid = (s_lo + s_hi) >> 1;
            if (off[mid + 1] > target) s_hi = mid; else s_lo = mid + 1;
        }
        b0 = s_lo;
    }
    const bool use = valid && b0 < nb;  // (beyond the last cumulative weight: undefined upstream, clamped below)
    const int lo = PRE ? 0 : wave_min_i(use ? b0 : 0x7fffffff), hi = PRE ? -1 : wave_max_i(use ? b0 : -1);
    auto block_ptr = [&](int b) -> const float * {
        return peers ? peers[b / nb_local].lcum[par] + (size_t) (b % nb_local) * kBlock : lcum_local + (size_t) b * kBlock;
    };
    // (named scalars, not an array: a conditionally written register array is demoted to scratch)
    static_assert(kWinBlocks == 4, "four named staging registers below");
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0, v2 = v0, v3 = v0;
    const int nw = hi < 0 ? 0 : hi - lo + 1;  // source blocks this wave's outputs draw from (0: no lane has an output)
    if (nw > 0) v0 = ldg<BYP>(reinterpret_cast<const float4 *>(block_ptr(lo)) + lane);
    if (nw > 1) v1 = ldg<BYP>(reinterpret_cast<const float4 *>(block_ptr(lo + 1)) + lane);
    if (nw > 2) v2 = ldg<BYP>(reinterpret_cast<const float4 *>(block_ptr(lo + 2)) + lane);
    if (nw > 3) v3 = ldg<BYP>(reinterpret_cast<const float4 *>(block_ptr(lo + 3)) + lane);
    if (nw > 0) reinterpret_cast<float4 *>(win)[lane] = v0;
    if (nw > 1) reinterpret_cast<float4 *>(win + kBlock)[lane] = v1;
    if (nw > 2) reinterpret_cast<float4 *>(win + 2 * kBlock)[lane] = v2;
    if (nw > 3) reinterpret_cast<float4 *>(win + 3 * kBlock)[lane] = v3;
    __builtin_amdgcn_wave_barrier();  // (LDS operations of one wave execute in order: the reads below see the stores above)
    if (!use) return n_global - 1;
    (void) block_ptr;
    const double o = off[b0];
    const double sc = blk_m ? block_scale(ldg<BYP>(blk_m + b0), M) : 1.0;  // (another tile's block maximum: as the prefixes)
    float pv[16], ev[16];
    int seg = 15, r = 15;
    if (b0 - lo < kWinBlocks) {
        const float *lc = win + (b0 - lo) * kBlock;
#pragma unroll
        for (int q = 0; q < 16; q++) pv[q] = lc[16 * q + 15];
#pragma unroll
        for (int q = 14; q >= 0; q--)
            if (o + (double) pv[q] * sc > target) seg = q;
        const float4 *l4 = reinterpret_cast<const float4 *>(lc + 16 * seg);
        const float4 e0 = l4[0], e1 = l4[1], e2 = l4[2], e3 = l4[3];
        ev[0] = e0.x; ev[1] = e0.y; ev[2] = e0.z; ev[3] = e0.w; ev[4] = e1.x; ev[5] = e1.y; ev[6] = e1.z; ev[7] = e1.w;
        ev[8] = e2.x; ev[9] = e2.y; ev[10] = e2.z; ev[11] = e2.w; ev[12] = e3.x; ev[13] = e3.y; ev[14] = e3.z; ev[15] = e3.w;
    } else {
        const float *lc = block_ptr(b0);
#pragma unroll
        for (int q = 0; q < 16; q++) pv[q] = ldg<BYP>(lc + 16 * q + 15);
#pragma unroll
        for (int q = 14; q >= 0; q--)
            if (o + (double) pv[q] * sc > target) seg = q;
        const float4 *l4 = reinterpret_cast<const float4 *>(lc + 16 * seg);
        const float4 e0 = ldg<BYP>(l4), e1 = ldg<BYP>(l4 + 1), e2 = ldg<BYP>(l4 + 2), e3 = ldg<BYP>(l4 + 3);
        ev[0] = e0.x; ev[1] = e0.y; ev[2] = e0.z; ev[3] = e0.w; ev[4] = e1.x; ev[5] = e1.y; ev[6] = e1.z; ev[7] = e1.w;
        ev[8] = e2.x; ev[9] = e2.y; ev[10] = e2.z; ev[11] = e2.w; ev[12] = e3.x; ev[13] = e3.y; ev[14] = e3.z; ev[15] = e3.w;
    }
#pragma unroll
    for (int q = 14; q >= 0; q--)
        if (o + (double) ev[q] * sc > target) r = q;
    return min((int64_t) b0 * kBlock + 16 * seg + r, n_global - 1);
}

// ---------------------------------------------------------------------------------------------------
// Lazy gather, copy role: after a resample nothing is moved until the next update kernel, whose compute blocks
// read their particle's pose and genealogy through keep[] and write them to the other buffer set; these blocks
// compose the genealogy rows still in use (gen_out[e][k] = gen[e][anc(k)]; 4 B per particle and LIVE ROW -- the
// 20-byte landmark records themselves stay where they are, kernels.h), kRowsPerRole rows x 256 particles per block,
// all loads in flight before the first store.  The row this update opens is written by the compute blocks (own slot).
// They run beside the compute blocks / the planning blocks, whose waves spend most of their time waiting on
// dependent loads.
// ---------------------------------------------------------------------------------------------------
constexpr int kLmkPerBlockY = 8;    // landmark rows per block in the kernels that move records (flatten, pack, unpack)

template <class AncOf>
SLAM_DEV void copy_genealogy(const Buffers &B, const int32_t *__restrict__ rows, int n_rows, int per_role, const WeightScratch &ws,
                             int cur, int role, AncOf anc_of) {
    const int bx = role % ws.nblocks, by = role / ws.nblocks;
    const int k = bx * kBlock + threadIdx.x;
    int anc = anc_of(k, k < B.n);  // (every lane: the inline plan's search is a wave's joint effort)
    if (k >= B.n) return;
    const size_t S = (size_t) B.ncap;
    if (anc < 0) return;  // legacy shards: arrived from another shard, genealogy already in place
    const int32_t *__restrict__ src = cur ? B.gen[1] : B.gen[0];
    int32_t *__restrict__ dst = cur ? B.gen[0] : B.gen[1];
    if (B.n_shards > 1) {  // distributed context: anc is a global index; the ancestor's rows may live on another GPU
        const int h = (int) __umul64hi((unsigned long long) (unsigned) anc, B.div_n);
        anc -= h * B.ncap;
        if (h != B.shard) src = B.peers[h].gen[cur];
    }
    const int r0 = by * per_role, r1 = min(n_rows, r0 + per_role);
    // sixteen rows per trip, all loads in flight before the first store: a role is a chain of dependent round trips, and
    // with ~1 000 live rows (config 5) the copy roles are a quarter of the launch's work
    constexpr int kTrip = 16;
    for (int r = r0; r < r1; r += kTrip) {
        int e[kTrip], q[kTrip];
#pragma unroll
        for (int t = 0; t < kTrip; t++) e[t] = rows[min(r + t, r1 - 1)];
#pragma unroll
        for (int t = 0; t < kTrip; t++) q[t] = src[(size_t) e[t] * S + anc];
#pragma unroll
        for (int t = 0; t < kTrip; t++)
            if (r + t < r1) dst[(size_t) e[t] * S + k] = q[t];
    }
}

// device-resident observation packet (kernels.h: ObsPacket), dense (written by the host) or fixed layout (written by the
// device front end): where its arrays start, in 4-byte units behind the header
struct PacketView {
    int m, n, nf, e_new, n_rows, rows_per_role, n_cons;
    const int32_t *idf, *row, *rows;
    const float *zf, *zn;
};
SLAM_DEV PacketView packet_view(const UpdateArgs &U) {
    PacketView V;
    const int32_t *base = reinterpret_cast<const int32_t *>(U.big + 1);
    int cap = 0;
    if (U.dev_packet) {
        // (the header through the constant address space: scalar loads, like the per-landmark entries)
        const auto *h = (const __attribute__((address_space(4))) ObsPacket *) reinterpret_cast<uintptr_t>(U.big);
        V.m = h->m;
        V.n = h->n;
        V.nf = h->nf;
        V.e_new = h->e_new;
        V.n_rows = h->n_rows;
        cap = h->cap;
        V.rows_per_role = max(16, ((V.n_rows + 3) / 4 + 3) / 4 * 4);  // (the host's rule, slamgpu.cpp: do_update)
    } else {
        V.m = U.m;
        V.n = U.n;
        V.nf = U.nf;
        V.e_new = U.e_new;
        V.n_rows = U.n_rows;
        V.rows_per_role = U.rows_per_role;
    }
    // (host-made packets: the landmarks the launch consolidates ride behind the re-observed ones in idf[] and row[])
    V.n_cons = U.dev_packet ? ((const __attribute__((address_space(4))) ObsPacket *) reinterpret_cast<uintptr_t>(U.big))->pad : U.n_cons;
    const int a = cap ? cap : V.m, ai = cap ? cap : V.m + V.n_cons, b = cap ? cap : V.n;
    V.idf = base;
    V.zf = reinterpret_cast<const float *>(base + ai);
    V.zn = V.zf + 2 * a;
    V.row = reinterpret_cast<const int32_t *>(V.zn + 2 * b);
    V.rows = V.row + ai;
    return V;
}

// ---------------------------------------------------------------------------------------------------
// Observation front end of a compact context, inside the update launch (kernels.h: FrontArgs).  Called by the first wave of a
// block; lane t owns landmark t of the map (state `s`, read from the copy the previous launch left).  Same arithmetic, same
// order of the visible landmarks and of the new features as observe_book_kernel (and so as the reference's
// get_observations + add_observation_noise + dataAssociationKnown: core.cpp:185-273, :438-449, :91-120); the genealogy
// bookkeeping is the host's for a packet of its own (slamgpu.cpp: do_update), row consolidation included.
// ---------------------------------------------------------------------------------------------------
// the two normals of one landmark's sensor noise (device draws): Box-Muller on one Philox block; the fast build takes the
// hardware transcendentals (the precise sinf / cosf / logf cost the launch 1.4 us on its critical path)
SLAM_DEV void sensor_normals(U4 r, float &g0, float &g1) {
    float g2;
#ifdef SLAM_FAST_MATH
    box_muller3_fast(r, g0, g1, g2);
#else
    box_muller3(r, g0, g1, g2);
#endif
}

// Two halves, so that the arithmetic (double-precision sqrt / atan2, the Philox draw) runs while the state is still in flight:
// front_observe needs the kernel arguments only, front_book the state.
struct FrontObs {
    bool v;          // this lane's landmark is visible
    int c, nz;       // its rank among the visible ones; how many there are
    float z0, z1;    // range, bearing (noise applied)
};
// A single wave runs all of this as one dependent chain of ~500 instructions (1.4 us at the head of the launch, measured), so
// the block's four waves share it: wave 0 the geometry (double-precision sqrt / atan2: needs the map), wave 1 the device
// draw of the sensor noise, wave 2 the heading's sine and cosine (both need the kernel arguments only), wave 3 clears the
// packet's LDS words; all of it while the loads of the head are in flight; one barrier; then wave 0 alone: ballots and the book.
struct FrontGeom {
    float dx, dy, z0, z1;
    double d2;
};
SLAM_DEV FrontGeom front_geometry(const FrontArgs &F, float lx, float ly) {
    FrontGeom g;
    g.dx = lx - F.x;
    g.dy = ly - F.y;
    g.d2 = (double) g.dx * (double) g.dx + (double) g.dy * (double) g.dy;
    g.z0 = (float) sqrt(g.d2);
    g.z1 = (float) atan2((double) g.dy, (double) g.dx) - F.phi;
    return g;
}
SLAM_DEV void front_draw(const FrontArgs &F, int t, float *aux) {  // aux[t], aux[kWave + t]: the two normals of landmark t
    float g0 = 0.f, g1 = 0.f;
    if (F.noise == 2) {
        U4 r = philox4x32((uint32_t) t, F.step, 3u, 0u, F.k0, F.k1);
        sensor_normals(r, g0, g1);
    }
    aux[t] = g0;
    aux[kWave + t] = g1;
}
SLAM_DEV FrontObs front_observe(const FrontArgs &F, const FrontGeom g, const float *aux, const __attribute__((address_space(4))) float *tape) {
    const int t = (int) threadIdx.x;  // < kWave
    const unsigned long long lt = (1ull << t) - 1ull;
    const float cph = aux[2 * kWave], sph = aux[2 * kWave + 1];
    FrontObs o;
    o.v = t < F.nlm && (fabsf(g.dx) < F.max_range) && (fabsf(g.dy) < F.max_range) && ((g.dx * cph + g.dy * sph) > 0.0f) &&
          (g.d2 < (double) F.max_range * (double) F.max_range);
    const unsigned long long V = __ballot(o.v);
    o.c = __popcll(V & lt);
    o.nz = __popcll(V);
    o.z0 = g.z0;
    o.z1 = g.z1;
    if (o.v && F.noise) {  // sensor noise (core.cpp:438-449)
        float g0 = aux[t], g1 = aux[kWave + t];
        if (F.noise == 1) {
            g0 = tape[o.c];
            g1 = tape[kSmallObs + o.c];
        }
        o.z0 = o.z0 + g0 * F.sr;
        o.z1 = o.z1 + g1 * F.sb;
    }
    return o;
}

// sets of the genealogy rows (< 64) named by the lanes with `a` / with `b`, through four LDS words: one atomic round trip
// instead of two six-level shuffle reductions
SLAM_DEV void wave_row_sets(bool a, bool b, int r, uint32_t *sh4, unsigned long long &sa, unsigned long long &sb) {
    if (threadIdx.x < 4) sh4[threadIdx.x] = 0u;
    if (a) atomicOr(&sh4[r >> 5], 1u << (r & 31));
    if (b) atomicOr(&sh4[2 + (r >> 5)], 1u << (r & 31));
    __builtin_amdgcn_wave_barrier();
    const volatile uint32_t *v = sh4;
    const uint32_t a0 = v[0], a1 = v[1], b0 = v[2], b1 = v[3];
    sa = ((unsigned long long) a1 << 32) | a0;
    sb = ((unsigned long long) b1 << 32) | b0;
}

SLAM_DEV void front_book(const FrontArgs &F, const FrontObs ob, const FrontLm s, const FrontHdr hd, int32_t *pk, uint32_t *sh4, bool writer,
                         FrontLm *next_lm = nullptr, FrontHdr *next_hd = nullptr) {
    const int t = (int) threadIdx.x;  // < kWave
    const unsigned long long lt = (1ull << t) - 1ull;
    const bool has_lm = t < F.nlm, v = ob.v;
    const int nf0 = hd.nf, fresh = hd.fresh_row, C = F.nlm, c = ob.c;
    const float z0 = ob.z0, z1 = ob.z1;
    // dataAssociationKnown: split by the table, in order; new landmarks get indices nf0, nf0 + 1, ... (beyond the
    // context's capacity: dropped and flagged)
    const bool has = has_lm && s.idf >= 0;
    const bool known = v && has, unseen = v && !has;
    const unsigned long long Mk = __ballot(known), Mn = __ballot(unseen);
    const int m = __popcll(Mk), room = F.cap_nf - nf0;
    int n = __popcll(Mn);
    const int dropped = n > room ? n - room : 0;
    n -= dropped;
    const int ko = __popcll(Mk & lt), kn = __popcll(Mn & lt);
    const bool added = unseen && kn < room;
    // genealogy rows: in use before; consolidation of the landmarks not observed; the row this update opens (the lowest
    // unused one); in use after, without it
    const int r = s.row & kRowMask;
    // (`rest`: rows of the landmarks not re-observed.  With a consolidation every one of those moves too -- a compact context
    // has fewer landmarks than a packet has entries -- and no old row stays in use; should that ever not hold, a second pass)
    unsigned long long before, rest;
    wave_row_sets(has, has && !known, r, sh4, before, rest);
    const bool cons_on = F.cons_above >= 0 && __popcll(before) > F.cons_above;
    const bool cand = cons_on && has && !known;
    const unsigned long long Mc = __ballot(cand);
    const int qc = __popcll(Mc & lt);
    const bool is_cons = cand && (m + qc < kSmallObs);
    const int nc = min(__popcll(Mc), max(kSmallObs - m, 0));
    const bool any = m + n + nc > 0;
    const int e_new = any ? (int) __ffsll(~before) - 1 : -1;
    const bool moved = known || is_cons;
    unsigned long long after = cons_on ? 0ull : rest;
    if (cons_on && __popcll(Mc) > nc) {
        unsigned long long dummy;
        wave_row_sets(has && !moved, false, r, sh4, after, dummy);
    }
    const int top = max(e_new, after ? 63 - (int) __clzll(after) : -1);
    const int word = s.row | (r == fresh ? kRowFreshBit : 0);
    const unsigned long long stale = __ballot(known && ko < 8 && !(word & kRowFreshBit));
    int32_t *p_idf = pk + offsetof(SmallObs, idf) / 4, *p_row = pk + offsetof(SmallObs, row) / 4;
    float *p_zf = reinterpret_cast<float *>(pk + offsetof(SmallObs, zf) / 4), *p_zn = reinterpret_cast<float *>(pk + offsetof(SmallObs, zn) / 4);
    if (known) {
        p_idf[ko] = s.idf;
        p_row[ko] = word;
        p_zf[2 * ko] = z0;
        p_zf[2 * ko + 1] = z1;
    }
    if (is_cons) {
        p_idf[m + qc] = s.idf;
        p_row[m + qc] = word;
    }
    if (added) {
        p_zn[2 * kn] = z0;
        p_zn[2 * kn + 1] = z1;
    }
    const int status = dropped ? kStatusCapacity : 0;
    if (t == 0) {
        int32_t *h = pk + offsetof(SmallObs, head) / 4;
        h[kFrontHeadM] = m;
        h[kFrontHeadN] = n;
        h[kFrontHeadNf] = nf0;
        h[kFrontHeadENew] = e_new;
        h[kFrontHeadChunks] = (top >> 2) + 1;  // (top = -1: 0 chunks)
        h[kFrontHeadFresh] = stale == 0 ? 1 : 0;
        h[kFrontHeadCons] = nc;
        h[kFrontHeadStatus] = status;
        pk[offsetof(SmallObs, magic) / 4] = (int32_t) kSmallMagic;
    }
    // the successor state: moved landmarks live in the row opened now, their records in the row's other buffer; a new
    // landmark's first records go to buffer 0
    FrontLm o = s;
    if (moved) o.row = e_new | ((s.row & kRowLiveBit) ^ kRowLiveBit);
    if (added) {
        o.idf = nf0 + kn;
        o.row = e_new;
    }
    if (next_lm) {  // (persistent loop: every workgroup keeps the state in its first wave's registers)
        *next_lm = o;
        *next_hd = FrontHdr{nf0 + n, e_new, hd.status | status, 0};
    }
    if (writer) {
        if (has_lm) F.state_out->lm[t] = o;
        if (t == 0) F.state_out->hdr = FrontHdr{nf0 + n, e_new, hd.status | status, 0};
        // ... and the observation itself, for slamgpu_observe_fetch (observe_kernel's block + a fixed-layout packet)
        float *oz = reinterpret_cast<float *>(F.out + 1);
        int32_t *ovis = reinterpret_cast<int32_t *>(oz + 2 * (size_t) C);
        ObsPacket *P = F.pkt;
        int32_t *base = reinterpret_cast<int32_t *>(P + 1);
        float *fzf = reinterpret_cast<float *>(base + C), *fzn = fzf + 2 * (size_t) C;
        int32_t *frow = reinterpret_cast<int32_t *>(fzn + 2 * (size_t) C);
        if (v) {
            oz[2 * c] = z0;
            oz[2 * c + 1] = z1;
            ovis[c] = t;
        }
        if (known) {
            base[ko] = s.idf;
            fzf[2 * ko] = z0;
            fzf[2 * ko + 1] = z1;
            frow[ko] = word;
        }
        if (added) {
            fzn[2 * kn] = z0;
            fzn[2 * kn + 1] = z1;
        }
        if (t == 0) {
            *F.out = ObserveOut{ob.nz, m, n, nf0 + n};
            *P = ObsPacket{m, n, nf0, (int32_t) __popcll(after), e_new, status, C, 0};
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// K1: [pending predicts] + per-particle observation update.  FastSLAM2::update body
// (fastslam2.cpp:26-45): sampleProposal (:290-368) + likelihoodGivenXv (:370-400) fused with
// featureUpdate (core.cpp:132-175; both evaluate their Jacobians at the same sampled pose) + addFeature
// (core.cpp:479-509); or FastSLAM1::update body (fastslam1.cpp:21-32).  Ends with the in-block inclusive
// prefix of the raw weights and the block totals of w and w^2 (inputs of resampleParticles'
// normalisation / Neff / cumulative sum, core.cpp:726-729,781-788,813-824).
// ---------------------------------------------------------------------------------------------------

// ARR: the context is a shard whose particles may have ARRIVED from other shards (keep[i] < 0, arrival-pool records).
// Single contexts instantiate ARR = false: per-lane buffer and pool selects cost them 6 % of the step for nothing.
// BIG: the observation packet lives in device memory (more than kSmallObs re-observed or new landmarks: synthetic maps,
// BASELINE config 5 re-observes ~1.3 k per step).  Every landmark loop then runs as a software pipeline over chunks of
// kBigChunk landmarks: while a chunk is computed out of LDS the next chunk's records are in flight into registers and
// the genealogy slots of the chunk after that behind them, so a wave has ~2 x kBigChunk x 20 B per lane outstanding
// instead of one dependent slot -> record round trip per landmark.
// (measured at config 5, gpurun_out/var: 4 and 8 landmarks per chunk run the same 1.61 ms per step, 12 and 16 are slower;
// 4 keeps the kernel at 103 / 110 VGPRs (fast / strict build: 4 waves per SIMD) and 20 KB of LDS per block, 8 needs 134 / 143)
// (round 4, same question again with the scalar packet reads in place, tools/gpu_r04_c5.sh: chunks of 2 / 4 / 8 landmarks with
// 1 / 2 / 3 chunks of records in flight: 1.155 .. 1.193 ms per step, all within 3 % of each other: not what bounds the launch)
constexpr int kBigChunk = 4;
constexpr int kStage = 8;  // landmarks per particle kept in LDS between the two passes of a small packet

// LDS of the per-wave ancestor windows of a launch that plans inline (host and device agree on the dynamic LDS layout)
__host__ __device__ constexpr size_t update_window_bytes() { return sizeof(float) * kWinBlocks * kBlock * (kBlock / kWave); }

// staged landmark slots per thread of an update launch (host and device agree on the dynamic LDS layout)
__host__ __device__ inline int staging_slots(int method, bool big, int m) {
    if (big) return kBigChunk;
    if (m <= 0) return 0;
    return m <= kStage / 2 ? kStage / 2 : kStage;
}

// MODE 0: single context.  MODE 1 (ARR): legacy shard context (arrival pool).  MODE 2 (DIST): distributed context: the
// particle set spans several GPUs whose state arrays are all mapped here (Buffers::peers); the plan runs over the
// all-gathered block totals of every shard, and whatever an ancestor owns on another GPU is read in place.
// The leading scalar parameters repeat what the HEAD of the kernel's dependent chain needs (the previous step's block
// totals, the Ctrl words, the launch shape) as plain pointers / ints, fetched with the first scalar load of the
// kernel: the loads of the scan and the packet are requested at once, before any field of the argument
// structs is looked at (the compiler fetches those where they are first used: five to six dependent scalar round trips
// stood between kernel entry and the first vector load before; 16.7 -> 16.05 us per step at 10^5 particles).
// Tried on top and measured as no better (gpurun_out/ab, 16.28 / 16.10 / 16.04 us): preloading these arguments into SGPRs
// (-mllvm -amdgpu-kernarg-preload-count=16) and touching every 64-byte line of the argument segment at entry.
//   h_flags: bit 0 plan_inline, bit 1 scan_global, bit 2 logw, bit 3 lazy, bit 4 front end inside the launch (h_front),
//            bit 5 the gathered totals table travels by LDS-DMA (scan_issue_dma), bit 6 count the remote ancestors (Ctrl::remote_reads)
// FastSLAM 1, fast build: do the (V, G) normals of this iteration's predicts come in one batch of eight (update_step: early_draws)?
SLAM_DEV bool persist_batch_draws(int method, const PredictArgs &PA) {
#ifdef SLAM_FAST_MATH
    return method == 1 && PA.nsteps > 4 && PA.nsteps <= 8 && PA.add_noise && !PA.use_heading && !PA.comp.valid;
#else
    return false;
#endif
}

// (Doing the scan at the END of the update launch instead -- no launch of its own -- was built three ways in round 5 and removed,
// 10^6 particles, us per step against 89.8 with scan_kernel: the last tile to arrive at ONE counter scans: 153.8 (3 907 agent-scope
// atomics on one address serialise at ~17 ns); 64 counters and one above them: 90.8 (every tile then ends with a drain of its
// stores and a returning atomic, ~3 us it stays resident for); the helper block polls totals the tiles store tagged and written
// through: 91.2 (rocprofv3: the launch grows by 9.8 us where scan_kernel takes 7.9: a trip to the memory side is 2-3 us, and
// "store lands, sweep sees it, sums, stores" is three of them -- what a kernel boundary and one HBM trip cost).)
constexpr int kScanMaxPer = kMaxScanBlocks / kBlock;
// P = the unrolled segment length (>= the totals per thread).  Every load is unconditional, at a clamped index (a first version
// guarded each load with `lo + j < hi`: the compiler made a branch, a load and an s_waitcnt vmcnt(0) of every one -- 32 dependent
// round trips, 9.4 us); a lane's surplus loads re-read the table's last entry and are masked out of the sums.
template <int P>
SLAM_DEV void scan_segments_sum(const float (&fw)[P], const float (&fq)[P], int nb, double *__restrict__ out, double *sh_a, double *sh_q, double &W,
                                double &Q) {
    const int t = threadIdx.x, lane = t & (kWave - 1), wv = t / kWave;
    const int per = (nb + kBlock - 1) / kBlock;
    const int lo = min(nb, t * per), hi = min(nb, lo + per);
    double tk[P];
    double a = 0.0, q = 0.0;
#pragma unroll
    for (int j = 0; j < P; j++) {
        const bool on = lo + j < hi;
        tk[j] = on ? (double) fw[j] * 1.0 : 0.0;
        if (on) {  // (selects: the sums of the lanes' live entries in ascending order, as scan_finish adds them)
            a += tk[j];
            q += (double) fq[j] * (tk[j] * tk[j]);
        }
    }
    const double sa = wave_scan_d(a);
    const double sq = wave_sum_d(q);
    if (lane == kWave - 1) sh_a[wv] = sa;
    if (lane == 0) sh_q[wv] = sq;
    __syncthreads();
    double base = 0.0;
#pragma unroll
    for (int k = 0; k < kBlock / kWave; k++)
        if (k < wv) base += sh_a[k];
    double run = base + sa - a;
#pragma unroll
    for (int j = 0; j < P; j++)
        if (lo + j < hi) {
            out[lo + j] = run;
            run += tk[j];
        }
    W = ((sh_a[0] + sh_a[1]) + sh_a[2]) + sh_a[3];
    Q = ((sh_q[0] + sh_q[1]) + sh_q[2]) + sh_q[3];
    if (t == 0) out[nb] = W;
}

template <int P>
SLAM_DEV void scan_segments(const float *__restrict__ tw, const float *__restrict__ tq, int nb, double *__restrict__ out, double *sh_a, double *sh_q,
                            double &W, double &Q) {
    const int per = (nb + kBlock - 1) / kBlock;
    const int lo = min(nb, (int) threadIdx.x * per);
    float fw[P], fq[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
        const int at = min(lo + j, nb - 1);
        fw[j] = tw[at];
        fq[j] = tq[at];
    }
    scan_segments_sum<P>(fw, fq, nb, out, sh_a, sh_q, W, Q);
}

// PERSIST (update_persist_kernel, kernels.h: PersistArgs): the same step as ONE ITERATION of a launch that runs K of them.  What
// the launch boundary gives a per-step launch comes from elsewhere: the iteration's arguments from `qe` (an LDS copy of its queue
// entry), the live buffer and the front end's state from `carry` (registers; no workgroup re-reads a Ctrl word or the state
// another one wrote), the helper's role from workgroup nb, tile = workgroup.  Same operations on the same values: bit-identical
// to K launches (tests/test_gpu_observe.py).
struct StepCarry {
    int cur;                  // live pose / genealogy buffer
    bool pend_word;           // first iteration only: Ctrl.pend as the launch found it
    const int32_t *pk_src;    // the iteration's observation packet (SmallObs words), made by the helper workgroup
    // drawn while the workgroups were meeting (persist_predraw): this particle's resampling stratum and its FastSLAM 2 normals --
    // counters only, nothing another workgroup wrote
    float strat, hg0, hg1, hg2;
    const float4 *draw_src;   // FastSLAM 1, fast build: this iteration's (V, G) normals, made by the drawer workgroups (or null)
};

// (PPT: per-particle association, kernels.h: PerParticle -- plain rows, single contexts; every other instantiation compiles the text it always did)
template <int METHOD, int MODE, bool BIG, bool PPT = false>
__global__ void __launch_bounds__(kBlock) update_kernel(const float *__restrict__ h_tot, Ctrl *h_ctrl,
                                                         const FrontState *h_front, int h_nb, int h_slot, int h_grid, int h_flags, Buffers B, PredictArgs PA,
                                                         UpdateArgs U, RngArgs rng, WeightScratch ws, PerParticle ppa) {
    static_assert(!PPT || (BIG && MODE == 0), "per-particle association: plain rows, single contexts");
    constexpr bool PERSIST = false, PP = PPT;
    const PersistStep *const qe = nullptr;
    StepCarry carry;  // (unused by a per-step launch)
#define STEP_WPAR ws.wpar
#define STEP_PLAN (U.plan_inline != 0)
#define STEP_FRONT U.front
#include "update_step.inl"
#undef STEP_WPAR
#undef STEP_PLAN
#undef STEP_FRONT
}

// The same step for sets of more than kWideBlocks tiles on one GPU (compact layout, single context), compiled for THREE waves per
// SIMD (168 vector registers, 4 of them spilled: update_kernel takes 179 and gets two).  At 10^5 particles the launch is a single
// round of tiles and the registers buy latency; from ~2 x 10^5 on the tiles queue for the CUs and a third resident tile per CU
// hides more of each tile's dependent chain than the spills cost.  Same box, us per step, update_kernel -> this: 125 000
// particles 15.0 -> 15.2 (not used there), 250 000: 29.0 -> 28.0, 500 000: 55.1 -> 48.1, 10^6 (BASELINE config 4 on one GPU):
// 103.0 -> 88.1; four waves per SIMD (128 registers, 30 spilled): 120.9.  Same source text, same contraction rule: bit-identical.
template <int METHOD>
__global__ void __launch_bounds__(kBlock, 3) update_kernel_wide(const float *__restrict__ h_tot, Ctrl *h_ctrl, const FrontState *h_front, int h_nb,
                                                                 int h_slot, int h_grid, int h_flags, Buffers B, PredictArgs PA, UpdateArgs U,
                                                                 RngArgs rng, WeightScratch ws) {
    constexpr int MODE = 0;
    constexpr bool BIG = false, PERSIST = false, PP = false;
    [[maybe_unused]] const PerParticle ppa{};
    const PersistStep *const qe = nullptr;
    StepCarry carry;  // (unused by a per-step launch)
#define STEP_WPAR ws.wpar
#define STEP_PLAN (U.plan_inline != 0)
#define STEP_FRONT U.front
#include "update_step.inl"
#undef STEP_WPAR
#undef STEP_PLAN
#undef STEP_FRONT
}

// one iteration of the persistent loop: the same text, its per-iteration arguments taken from `qe` and `carry`
template <int METHOD>
SLAM_DEV void persist_step(const float *__restrict__ h_tot, Ctrl *h_ctrl, const FrontState *h_front, int h_nb, int h_slot, int h_grid, int h_flags,
                           const Buffers &B, const UpdateArgs &U, const RngArgs &rng_k, const WeightScratch &ws, const PersistStep *qe,
                           StepCarry &carry) {
    constexpr bool PERSIST = true, BIG = false, PP = false;
    [[maybe_unused]] const PerParticle ppa{};
    constexpr int MODE = 0;
    const PredictArgs &PA = qe->PA;
    const int wpar = __builtin_amdgcn_readfirstlane(qe->wpar);
    const bool plan_inline = __builtin_amdgcn_readfirstlane(qe->plan_inline) != 0;
    RngArgs rng = rng_k;
    rng.step = (uint32_t) __builtin_amdgcn_readfirstlane((int) qe->rng_step);
    rng.prev_step = (uint32_t) __builtin_amdgcn_readfirstlane((int) qe->rng_prev_step);
    FrontArgs F_l = U.front;
    F_l.x = qe->fx;
    F_l.y = qe->fy;
    F_l.phi = qe->fphi;
    F_l.step = (uint32_t) __builtin_amdgcn_readfirstlane((int) qe->fstep);
#define STEP_WPAR wpar
#define STEP_PLAN plan_inline
#define STEP_FRONT F_l
#include "update_step.inl"
#undef STEP_WPAR
#undef STEP_PLAN
#undef STEP_FRONT
}

// ---------------------------------------------------------------------------------------------------
// The persistent small-N step loop (kernels.h: PersistArgs).  Same leading arguments and argument layout as update_kernel (the
// front end reads the map out of the kernel-argument segment at update_kernel's offsets).
//
// One iteration, tile workgroups:   [packet + block totals of the previous iteration arrive from L2] -> plan (scan, Neff,
//   decision, ancestor) -> pose / records -> predicts -> landmark pass -> stores -> weight prefix + totals -> ARRIVE at the
//   counter -> (while the others arrive) the next iteration's queue entry into LDS and everything of it that needs nobody
//   else's data: the resampling stratum and the particle's normals (Philox + Box-Muller: ~1 us of arithmetic) -> PASS.
// helper workgroup:  the decision again (for the Ctrl words), the estimate reduction of two iterations ago, and the observation
//   packet of the NEXT iteration (get_observations + dataAssociationKnown + the genealogy book: the front end's state lives in
//   its first wave's registers), stored for the tiles to pick up behind the barrier -> ARRIVE -> PASS.
// ---------------------------------------------------------------------------------------------------
// All workgroups of the loop meet, in two halves.  arrive: stores drained, one arrival each at the counter.  pass: a bounded
// poll, then this CU's vector cache is invalidated (the other workgroups' stores are in the XCD's L2; a workgroup on another
// XCD -- never seen, checked at the start of the launch -- also writes its L2 back before it arrives: the release of the
// memory model).  pass returns false when the launch is being abandoned (somebody waited longer than `max_spins` polls: the
// abort word is set and everybody leaves).
SLAM_DEV void persist_arrive(uint32_t *sync, bool cross_xcd) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        if (cross_xcd) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __hip_atomic_fetch_add(sync + kPersistSyncCounter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// invalidate: the workgroup goes on to read other workgroups' stores with plain loads (the helper; everybody when the placement
// spans XCDs); the tiles read them with ldg<true> (device_math.h) and need no invalidate -- 1.2-1.7 us per iteration otherwise
// The wait is bounded twice: in polls (max_spins: what the tests force) and in TIME (max_ticks of the 100 MHz constant clock,
// read every 256 polls so that the clock's scalar-memory trip stays off the path of a wait that ends in a microsecond): a poll
// count alone is some fraction of a second that depends on the L2's mood.
SLAM_DEV bool persist_pass(uint32_t *sync, uint32_t target, uint32_t max_spins, unsigned long long max_ticks, int *sh_ok, bool invalidate,
                           bool lazy = false) {
    if (threadIdx.x == 0) {
        int ok = 1;
        uint32_t spins = 0;
        unsigned long long t0 = 0;
        while ((int32_t) (__hip_atomic_load(sync + kPersistSyncCounter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            bool late = ++spins > max_spins;
            if ((spins & 255u) == 0u) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (t0 == 0) t0 = now;
                late = late || now - t0 > max_ticks;
            }
            if (__hip_atomic_load(sync + kPersistSyncAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || late) {
                __hip_atomic_store(sync + kPersistSyncAbort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            // (the helper and the drawers have microseconds to spare: they poll at leisure and leave the counter's line to the tiles)
            if (lazy) __builtin_amdgcn_s_sleep(24);
            else __builtin_amdgcn_s_sleep(1);
        }
        if (ok && __hip_atomic_load(sync + kPersistSyncAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ok = 0;
        if (invalidate) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (holds the barrier below until the invalidate has completed)
        }
        *sh_ok = ok;
    }
    __syncthreads();
    return *sh_ok != 0;
}

// what an iteration can work out before it has seen anybody else's data (StepCarry): the draws update_step would make at the
// places named there, from the same counters, with the same functions
template <int METHOD>
SLAM_DEV void persist_predraw(const PersistStep *qe, const RngArgs &rng_k, const Buffers &B, int i, StepCarry &carry) {
    RngArgs rng = rng_k;
    rng.step = (uint32_t) __builtin_amdgcn_readfirstlane((int) qe->rng_step);
    rng.prev_step = (uint32_t) __builtin_amdgcn_readfirstlane((int) qe->rng_prev_step);
    carry.strat = stratum_prev(rng, (int64_t) i);
    carry.hg0 = carry.hg1 = carry.hg2 = 0.f;
    if (METHOD == 2 && rng.mode != 0) {
        U4 r = philox4x32((uint32_t) (rng.first_particle + i), rng.step, 0u, 0u, rng.k0, rng.k1);
#ifdef SLAM_FAST_MATH
        box_muller3_fast(r, carry.hg0, carry.hg1, carry.hg2);
#else
        box_muller3(r, carry.hg0, carry.hg1, carry.hg2);
#endif
    }
}

// A drawer workgroup: everything of the predicts of tile `tile`'s particles that needs no pose, for the iteration whose queue entry
// is `qe`: the (V, G) normals (~1.9 us of Philox + Box-Muller per iteration that need nothing but counters), the perturbed
// controls, V dt and sin(G / wheelBase) of the eight steps -- made on a CU that would otherwise idle, an iteration ahead; the tile
// picks them up with the block totals (update_step.inl).  The same functions on the same values as the tile would call.
SLAM_DEV void persist_draw(const PersistStep *qe, const RngArgs &rng, const Buffers &B, int tile, float4 *dst) {
#ifdef SLAM_FAST_MATH
    const PredictArgs &PA = qe->PA;
    if (!persist_batch_draws(1, PA)) return;
    const int i = tile * kBlock + (int) threadIdx.x;
    const size_t S = (size_t) B.ncap;
    float g0[8], g1[8], vd[8], gs[8], sgw[8];
    const CtlP ctl = (CtlP) reinterpret_cast<const float *>(qe->PA.steps);  // (the queue entries are LDS copies)
    draw_batch_fs1_fast<8>(g0, g1, PA, rng, i, S, ctl, 0, PA.nsteps);
    const L2 Lq = llt2(PA.Q[0], PA.Q[2], PA.Q[3]);
    controls_batch_fs1_fast<8>(vd, gs, sgw, g0, g1, PA, ctl, PA.dt, 1.0f / PA.wheel_base, Lq, PA.nsteps);
    dst[i] = make_float4(vd[0], vd[1], vd[2], vd[3]);
    dst[S + i] = make_float4(vd[4], vd[5], vd[6], vd[7]);
    dst[2 * S + i] = make_float4(gs[0], gs[1], gs[2], gs[3]);
    dst[3 * S + i] = make_float4(gs[4], gs[5], gs[6], gs[7]);
    dst[4 * S + i] = make_float4(sgw[0], sgw[1], sgw[2], sgw[3]);
    dst[5 * S + i] = make_float4(sgw[4], sgw[5], sgw[6], sgw[7]);
#endif
}

// (The (V, G) normals of FastSLAM 1's predicts stay where update_step draws them, all eight in one interleaved batch behind the pose
// request.  Drawing the first four between an iteration's last stores and the wait for them was built and measured in round 5:
// 10.39 against 9.72 us per step -- two batches of four lose the instruction-level parallelism of one batch of eight, and the
// compiler is free to sink arithmetic whose results are needed an iteration later below the wait it was meant to fill.)

// the observation packet of one iteration, by the helper workgroup (the front end of update_step, same functions in the same
// roles): SmallObs words into `pk` (LDS), then into `dst`; the front end's state moves on in the first wave's registers
SLAM_DEV void persist_front(const FrontArgs &F, float f_x, float f_y, FrontLm &lm, FrontHdr &hd, const __attribute__((address_space(4))) float *tape,
                            int32_t *pk, uint32_t *f_sets, float *f_aux, int32_t *dst) {
    const int fw = threadIdx.x / kWave, ft = threadIdx.x & (kWave - 1);
    FrontGeom fg{};
    if (fw == 0) {
        fg = front_geometry(F, f_x, f_y);
    } else if (fw == 1) {
        front_draw(F, ft, f_aux);
    } else if (fw == 2) {
        const float cph = cosf(F.phi), sph = sinf(F.phi);
        if (ft == 0) {
            f_aux[2 * kWave] = cph;
            f_aux[2 * kWave + 1] = sph;
        }
    } else {
        for (int w = ft; w < kSmallWords; w += kWave) pk[w] = 0;
    }
    __syncthreads();
    if (fw == 0) {
        const FrontObs ob = front_observe(F, fg, f_aux, tape);
        FrontLm nl;
        FrontHdr nh;
        front_book(F, ob, lm, hd, pk, f_sets, true, &nl, &nh);
        lm = nl;
        hd = nh;
    }
    __syncthreads();
    if (threadIdx.x < kSmallWords) dst[threadIdx.x] = pk[threadIdx.x];
}

template <int METHOD>
__global__ void __launch_bounds__(kBlock) update_persist_kernel(const float *__restrict__ h_tot, Ctrl *h_ctrl,
                                                                 const FrontState *h_front, int h_nb, int h_slot, int h_grid, int h_flags, Buffers B,
                                                                 PredictArgs PA, UpdateArgs U, RngArgs rng, WeightScratch ws) {
    if (blockIdx.x % kPersistStride != 0) return;  // (the workgroups that stay were dealt to ONE XCD)
    const int bid = (int) blockIdx.x / kPersistStride, nb = h_nb;
    const bool helper = bid == nb;
    const PersistArgs &P = U.persist;
    const bool drawer = bid > nb;          // (P.drawers of them: drawer nb + 1 + t serves tile t)
    const int dtile = bid - nb - 1;
    __shared__ PersistStep qes[2];  // the queue entries of this iteration and the next
    __shared__ int sh_ok;
    static_assert(kPersistStepWords <= kBlock, "one dword of a queue entry per thread");
    const uint32_t members = (uint32_t) (nb + 1 + P.drawers);  // the tiles' workgroups + the helper + the drawers
    uint32_t *sync = P.sync;
    // which XCD is this?  All members say; one barrier with the memory model's full protocol; all read
    if (threadIdx.x == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_store(sync + kPersistSyncXcc + bid, (xcc & 15u) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the queue lives in pinned host memory and is never written during the launch.  Entry 0: everybody reads it there, once; the
    // others reach the tiles and drawers through a ring in device memory that the helper workgroup fills two iterations ahead
    const uint32_t *__restrict__ q0 = reinterpret_cast<const uint32_t *>(P.queue);
    uint32_t *const ring = reinterpret_cast<uint32_t *>(P.ring);
    if (P.K > 0 && threadIdx.x < kPersistStepWords) reinterpret_cast<uint32_t *>(&qes[0])[threadIdx.x] = q0[threadIdx.x];
    if (helper && P.K > 1 && threadIdx.x < kPersistStepWords) ring[kPersistStepWords + threadIdx.x] = q0[kPersistStepWords + threadIdx.x];
    __syncthreads();
    StepCarry carry;
    carry.cur = h_ctrl->live[h_slot];
    carry.pend_word = h_ctrl->pend[h_slot] != 0;
    carry.pk_src = P.packets;
    const int i = bid * kBlock + (int) threadIdx.x;  // (tile = workgroup)
    // helper: the front end's state, the map (kernel arguments, update_kernel's layout) and the first packet
    FrontLm f_lm{-1, 0};
    FrontHdr f_hd{0, -1, 0, 0};
    float f_x = 0.f, f_y = 0.f;
    constexpr size_t ka0 = (40 + sizeof(Buffers) + alignof(PredictArgs) - 1) / alignof(PredictArgs) * alignof(PredictArgs);
    constexpr size_t ka1 = (ka0 + sizeof(PredictArgs) + alignof(UpdateArgs) - 1) / alignof(UpdateArgs) * alignof(UpdateArgs);
    constexpr size_t ka_small = (ka1 + offsetof(UpdateArgs, small)) / 4;
    const auto *kf = (const __attribute__((address_space(4))) float *) __builtin_amdgcn_kernarg_segment_ptr();
    __shared__ int32_t h_pk[kSmallWords];
    __shared__ uint32_t h_sets[4];
    __shared__ float h_aux[2 * kWave + 2];
    __shared__ double hs_a[kBlock / kWave], hs_q[kBlock / kWave];
    __shared__ EstItem hs_est[kBlock / kWave];
    extern __shared__ __align__(16) unsigned char dyn_lds[];  // (helper: the prefix of the block totals, as in update_step)
    auto front_args = [&](const PersistStep *q) {
        FrontArgs F = U.front;
        F.x = q->fx;
        F.y = q->fy;
        F.phi = q->fphi;
        F.step = (uint32_t) __builtin_amdgcn_readfirstlane((int) q->fstep);
        return F;
    };
    if (helper) {
        if (threadIdx.x < kWave) {
            const int t = min((int) threadIdx.x, kSmallObs - 1);
            f_x = kf[ka_small + offsetof(SmallObs, zn) / 4 + t];
            f_y = kf[ka_small + offsetof(SmallObs, zn) / 4 + kSmallObs + t];
            f_hd = h_front->hdr;
            f_lm = h_front->lm[threadIdx.x];
        }
        if (P.K > 0) persist_front(front_args(&qes[0]), f_x, f_y, f_lm, f_hd, kf + ka_small + offsetof(SmallObs, zf) / 4, h_pk, h_sets, h_aux, P.packets);
    }
    const size_t draw_words = 6 * (size_t) B.ncap;  // float4s per buffer of PersistArgs::draws
    if (drawer && P.K > 0) persist_draw(&qes[0], rng, B, dtile, P.draws);
    persist_arrive(sync, true);
    if (!helper && !drawer && P.K > 0) persist_predraw<METHOD>(&qes[0], rng, B, i, carry);
    bool alive = persist_pass(sync, members, P.max_spins, P.max_ticks, &sh_ok, true);
    bool cross = false;
    {
        const uint32_t mine = __hip_atomic_load(sync + kPersistSyncXcc + bid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int b = 0; b < (int) members; b++) cross |= __hip_atomic_load(sync + kPersistSyncXcc + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != mine;
    }
    const bool logw = (h_flags & 4) != 0;
    int it = 0, done = 0;
    for (; alive && it < P.K; it++) {
        SLAM_STAMP(12);  // (diagnostic build) this iteration begins: the previous barrier has been passed
        const bool more = it + 1 < P.K;
        const PersistStep &qe = qes[it & 1];
        PersistStep *const qn = &qes[(it + 1) & 1];  // (free since the previous barrier)
        // the next iteration's entry (from the ring: an L2 hit): requested now, parked in LDS behind the step's code; the helper also
        // fetches the entry after it from the host (a PCIe trip it has the time for) and puts it into the ring before the barrier
        uint32_t qv = 0, qv2 = 0;
        if (more && threadIdx.x < kPersistStepWords) qv = ldg_u32<true>(ring + (size_t) ((it + 1) & 3) * kPersistStepWords + threadIdx.x);
        const bool more2 = helper && it + 2 < P.K;
        if (more2 && threadIdx.x < kPersistStepWords) qv2 = q0[(size_t) (it + 2) * kPersistStepWords + threadIdx.x];
        if (drawer) {
            // (nothing of this iteration: the next one's draws, below)
        } else if (!helper) {
            carry.pk_src = P.packets + (size_t) (it & 1) * kSmallWords;
            carry.draw_src = P.drawers ? P.draws + (size_t) (it & 1) * draw_words : nullptr;
            persist_step<METHOD>(h_tot, h_ctrl, h_front, h_nb, h_slot, h_grid, h_flags, B, U, rng, ws, &qe, carry);
        } else {
            // the decision (for the live buffer the launch leaves), as update_step's helper block works it out
            const int wpar = __builtin_amdgcn_readfirstlane(qe.wpar);
            bool pend = (h_flags & 8) && carry.pend_word;
            if (__builtin_amdgcn_readfirstlane(qe.plan_inline)) {
                const float *__restrict__ tot = wpar ? ws.blk_w[0] : ws.blk_w[1];
                const ScanLoads scl = scan_issue(tot, nb, nb, logw);
                double W, Q, Mx;
                scan_finish(scl, tot, nb, nb, logw, reinterpret_cast<double *>(dyn_lds), hs_a, hs_q, W, Q, Mx);
                pend = U.do_resample && (neff_of(W, Q) < (float) U.n_effective);
            }
            carry.cur = pend ? carry.cur ^ 1 : carry.cur;
            if (__builtin_amdgcn_readfirstlane(qe.finalize)) finish_estimate(B, ws, __builtin_amdgcn_readfirstlane(qe.finalize_par), qe.finalize_hist, hs_est);
        }
        carry.pend_word = false;
        SLAM_STAMP(11);  // the step's code is done
        if (more2 && threadIdx.x < kPersistStepWords) ring[(size_t) ((it + 2) & 3) * kPersistStepWords + threadIdx.x] = qv2;
        if (more) {
            if (threadIdx.x < kPersistStepWords) reinterpret_cast<uint32_t *>(qn)[threadIdx.x] = qv;
            __syncthreads();
            // helper: the NEXT iteration's observation packet, in place before this iteration's barrier; drawers: its draws
            if (helper)
                persist_front(front_args(qn), f_x, f_y, f_lm, f_hd, kf + ka_small + offsetof(SmallObs, zf) / 4, h_pk, h_sets, h_aux,
                              P.packets + (size_t) ((it + 1) & 1) * kSmallWords);
            else if (drawer)
                persist_draw(qn, rng, B, dtile, P.draws + (size_t) ((it + 1) & 1) * draw_words);
        }
        // (tests of the abandon path: the helper gives up in iteration abort_at, as a workgroup that waited too long would)
        if (helper && it == P.abort_at && threadIdx.x == 0) __hip_atomic_store(sync + kPersistSyncAbort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        persist_arrive(sync, cross);
        if (more && !helper && !drawer) persist_predraw<METHOD>(qn, rng, B, i, carry);  // (while the arrival travels)
        alive = persist_pass(sync, members * (uint32_t) (it + 2), P.max_spins, P.max_ticks, &sh_ok, helper || cross, helper || drawer);
        if (alive) done = it + 1;  // (this barrier passed: EVERY workgroup has completed iteration `it`)
        SLAM_STAMP(13);  // barrier passed
    }
    // where the launch leaves things, for the host and the next launch: the Ctrl words (both slots: the set is plain, in
    // `cur`) and the front end's state
    if (helper && threadIdx.x == 0) {
        h_ctrl->live[0] = h_ctrl->live[1] = carry.cur;
        h_ctrl->pend[0] = h_ctrl->pend[1] = 0;
        sync[kPersistSyncDone] = (uint32_t) done;      // iterations every workgroup completed (= K unless the launch was abandoned)
        sync[kPersistSyncCross] = cross ? 1u : 0u;     // (diagnostic: the placement was not one XCD)
        // An abandoned launch says how far it got: the helper workgroup alone writes the host's words, and only for
        // the FIRST abandoned launch -- the abort word is sticky, the launches queued behind this one leave at their first meeting
        // with nothing done and must not overwrite the account.  (The helper always gets here: every wait is bounded.)
        if (!alive && __hip_atomic_load(P.host_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
            __hip_atomic_store(P.host_status + 1, (uint32_t) done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(P.host_status + 2, (uint32_t) P.serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(P.host_status + 3, (uint32_t) P.K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(P.host_status, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (helper && threadIdx.x < kWave) {
        if ((int) threadIdx.x < U.front.nlm) P.state_final->lm[threadIdx.x] = f_lm;
        if (threadIdx.x == 0) P.state_final->hdr = f_hd;
    }
}

// ---------------------------------------------------------------------------------------------------
// resampleParticles (core.cpp:718-749), planning only, + the per-step pose estimate (ParticleSLAMWrapper.cpp:56-77) as a
// launch of its own: used when something needs the outcome of the last update before the next update launch (which
// otherwise does the same work inline, update_kernel: plan_inline).  Every block redundantly scans the block totals
// (LDS, double) => sum w, sum w^2, Neff and the decision `doResample && Neff < nMin`.
//   no resample: w_i /= sum(w) (core.cpp:726-729)
//   resample   : ancestor of output k = min{ i : select_k < cumsum_i } (core.cpp:800-806) into keep[slot ^ 1]; NOTHING
//                is moved here: the next update launch (or gather_kernel) gathers through keep[].
// ---------------------------------------------------------------------------------------------------
// VectorXf::sum() of Eigen 3.1.3 as the reference's build runs it (Redux.h:200-240, SSE packets of four, unrolled by two, then the
// SSE2 horizontal add; restated in oracle/slam_oracle.c: orc_eigen_sum and pinned there to the reference objects): v[i], or
// v[i]^2 (= (float) pow((double) v[i], 2): the double product of two floats is exact, so one rounding either way), i < n.
template <bool SQ>
SLAM_DEV float eigen_order_sum(const float *v, int n) {
    auto at = [&](int i) { return SQ ? v[i] * v[i] : v[i]; };
    const int aligned2 = (n / 8) * 8, aligned = (n / 4) * 4;
    float res;
    if (aligned) {
        float p00 = at(0), p01 = at(1), p02 = at(2), p03 = at(3);
        if (aligned > 4) {
            float p10 = at(4), p11 = at(5), p12 = at(6), p13 = at(7);
            for (int i = 8; i < aligned2; i += 8) {
                p00 = p00 + at(i);
                p01 = p01 + at(i + 1);
                p02 = p02 + at(i + 2);
                p03 = p03 + at(i + 3);
                p10 = p10 + at(i + 4);
                p11 = p11 + at(i + 5);
                p12 = p12 + at(i + 6);
                p13 = p13 + at(i + 7);
            }
            p00 = p00 + p10;
            p01 = p01 + p11;
            p02 = p02 + p12;
            p03 = p03 + p13;
            if (aligned > aligned2) {
                p00 = p00 + at(aligned2);
                p01 = p01 + at(aligned2 + 1);
                p02 = p02 + at(aligned2 + 2);
                p03 = p03 + at(aligned2 + 3);
            }
        }
        res = (p00 + p02) + (p01 + p03);
        for (int i = aligned; i < n; i++) res = res + at(i);
    } else {
        res = at(0);
        for (int i = 1; i < n; i++) res = res + at(i);
    }
    return res;
}

// resampleParticles' plan in the reference's order of operations (kernels.h: kRefResampleMax): ONE block.  The sums and the
// running prefix are the reference's serial chains and run in one thread (eight independent accumulators in the sums); the
// divisions, the squares, the write-back and the ancestor of every stratum are the block's.  `keep[ctr] = i while select[ctr] <
// cum[i]` (core.cpp:800-806) assigns stratum ctr the first i with cum[i] > max(select[0..ctr]) -- the loop never moves back --
// so every thread searches that prefix maximum in the prefix (both in LDS).  Dynamic LDS: two arrays of n floats.
__global__ void __launch_bounds__(kBlock) resample_ref_kernel(Buffers B, WeightScratch ws, RngArgs rng, ResampleArgs ra) {
#pragma clang fp contract(off)
    extern __shared__ float ref_lds[];
    __shared__ float sh_f[4];
    __shared__ int sh_i[2];
    const int n = B.n, t = threadIdx.x;
    float *w = ref_lds, *sel = ref_lds + n;
    Ctrl *ctrl = B.ctrl;
    const int cur = ctrl->live[B.slot];
    float4 *__restrict__ poseA = B.poseA[cur];
    for (int i = t; i < n; i += kBlock) w[i] = poseA[i].w;
    __syncthreads();
    if (t == 0) sh_f[0] = eigen_order_sum<false>(w, n);  // ws = w.sum() (core.cpp:726; the same value again at :782)
    __syncthreads();
    const float wsum = sh_f[0];
    for (int i = t; i < n; i += kBlock) w[i] = w[i] / wsum;  // (core.cpp:727-729 and :785: the same division)
    __syncthreads();
    if (t == 0) {
        const float neff = 1 / eigen_order_sum<true>(w, n);  // (core.cpp:786-788)
        const bool resample = ra.do_resample && (neff < (float) ra.n_effective);  // (core.cpp:739: float < int)
        sh_f[1] = neff;
        sh_i[0] = resample ? 1 : 0;
        const double W = (double) wsum, Q = (double) wsum * (double) wsum / (double) neff;
        ctrl->wsum = W;
        ctrl->wsq = Q;
        ctrl->wmax = 0.0;
        ctrl->neff = neff;
        ctrl->resampled = resample ? 1 : 0;
        ctrl->status = weight_status(W, Q);
        ws.est_part[ws.wpar][4 * (size_t) ws.nblocks] = (double) neff;
        ws.est_part[ws.wpar][4 * (size_t) ws.nblocks + 1] = (double) ((resample ? 1 : 0) | (weight_status(W, Q) << 1));
    }
    __syncthreads();
    if (!sh_i[0]) {  // Neff >= nMin: the particles keep the normalised weights (core.cpp:727-729)
        for (int i = t; i < n; i += kBlock) poseA[i].w = w[i];
        return;
    }
    // the strata and their running maximum (a block scan: a maximum is exact whatever the order)
    for (int i = t; i < n; i += kBlock) sel[i] = rng.strata[i];
    if (t == 0) {  // cumulativeSum (core.cpp:813-824): the serial float32 running sum, in place
        float run = 0;
        for (int i = 0; i < n; i++) {
            run += w[i];
            w[i] = run;
        }
    }
    __syncthreads();
    if (t == 0) {  // (n <= 5 000: ~n compare-and-stores; a parallel scan would save ~10 us of the stage's ~40)
        float m = sel[0];
        for (int i = 1; i < n; i++) {
            m = fmaxf(m, sel[i]);
            sel[i] = m;
        }
    }
    __syncthreads();
    int32_t *__restrict__ keep = ws.keep[B.slot ^ 1];
    for (int c = t; c < n; c += kBlock) {
        const float s = sel[c];
        int lo = 0, hi = n;  // first i with s < w[i]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s < w[mid]) hi = mid; else lo = mid + 1;
        }
        keep[c] = lo < n ? lo : n - 1;  // (beyond the last cumulative weight: keep = -1 upstream, undefined when used; clamped as everywhere)
    }
}

__global__ void __launch_bounds__(kBlock) resample_kernel(Buffers B, WeightScratch ws, RngArgs rng, ResampleArgs ra,
                                                           UpdateArgs U) {
    extern __shared__ double off[];  // [nblocks + 1] exclusive prefix of the block totals
    __shared__ double sh_a[kBlock / kWave], sh_q[kBlock / kWave];
    __shared__ EstItem sh_est[kBlock / kWave];
    Ctrl *ctrl = B.ctrl;
    const int t = threadIdx.x;
    const int nb = ws.nblocks;
    // every update launch publishes where it left the set (pend = 0) and the host flips its slot: plain read here
    const int cur = ctrl->live[B.slot];

    double W, Q, Mx;
    const bool logw = ra.logw != 0;
    if (ra.planned) {
        // resample_ref_kernel has made the plan (reference order of operations): the decision is in Ctrl, the ancestors are in
        // keep[], the weights are normalised; what is left to this launch: the state for the next one and the estimate partials
        const bool resample = ctrl->resampled != 0;
        if (blockIdx.x == 0 && t == 0) {
            ctrl->live[B.slot ^ 1] = cur;
            ctrl->pend[B.slot ^ 1] = resample ? 1 : 0;
        }
        const int k = blockIdx.x * kBlock + t;
        EstItem ei{0.0, 0.0, -3.0e38f, 0.0f, 0x7fffffff};
        if (k < B.n) {
            const float4 pa = B.poseA[cur][resample ? ws.keep[B.slot ^ 1][k] : k];
            ei = EstItem{(double) pa.x, (double) pa.y, resample ? ctrl->inv_n : pa.w, pa.z, k};
        }
        ei = block_reduce_est(ei, sh_est);
        if (t == 0) {
            double *p = ws.est_part[ws.wpar] + (size_t) blockIdx.x * 4;
            p[0] = ei.sx;
            p[1] = ei.sy;
            p[2] = (double) ei.th;
            p[3] = (double) ei.w;
        }
        return;
    }
    scan_block_totals(ws.blk_w[ws.wpar], nb, nb, logw, off, sh_a, sh_q, W, Q, Mx);  // one shard: [w(nb) | w2(nb)] contiguous
    // Neff = 1 / sum((w/W)^2)  (core.cpp:784-788)
    const float neff = neff_of(W, Q);
    const bool resample = ra.do_resample && (neff < (float) ra.n_effective);
    if (blockIdx.x == 0 && t == 0) {
        ctrl->wsum = W;
        ctrl->wsq = Q;
        ctrl->wmax = Mx;
        ctrl->neff = neff;
        ctrl->resampled = resample ? 1 : 0;
        ctrl->status = weight_status(W, Q);
        ws.est_part[ws.wpar][4 * (size_t) nb] = (double) neff;  // travels with the partials into the history
        ws.est_part[ws.wpar][4 * (size_t) nb + 1] = (double) ((resample ? 1 : 0) | (weight_status(W, Q) << 1));
        // state for the next launch goes to the OTHER slot (see Ctrl): the set now lives in `cur`, and after a
        // resample it is defined through keep[] until somebody gathers it
        ctrl->live[B.slot ^ 1] = cur;
        ctrl->pend[B.slot ^ 1] = resample ? 1 : 0;
    }
    __syncthreads();

    const int k = blockIdx.x * kBlock + t;
    const bool active = k < B.n;
    EstItem ei{0.0, 0.0, -3.0e38f, 0.0f, 0x7fffffff};
    if (!resample) {
        if (active) {
            float4 pa = B.poseA[cur][k];
            pa.w = logw ? pa.w - (float) (Mx + log(W)) : pa.w / (float) W;
            B.poseA[cur][k] = pa;
            ei = EstItem{(double) pa.x, (double) pa.y, pa.w, pa.z, k};
        }
    } else if (active) {
        const double target = (double) stratum(rng, (int64_t) k) * W;
        const int anc = (int) min(find_ancestor(target, off, nb, ws.lcum[ws.wpar], 0, nb, (int64_t) B.n,
                                                logw ? ws.blk_w[ws.wpar] + 2 * nb : nullptr, Mx), (int64_t) B.n - 1);
        ws.keep[B.slot ^ 1][k] = anc;
        const float4 pa = B.poseA[cur][anc];
        ei = EstItem{(double) pa.x, (double) pa.y, ctrl->inv_n, pa.z, k};
    }

    // ---- estimate partial of this block; reduced by the next update launch's helper block or by finish_kernel
    // (no in-kernel hand-off: an agent-scope release per block = an L2 write-back per block on gfx950,
    //  which costs far more than the kernel boundary it would save)
    ei = block_reduce_est(ei, sh_est);
    if (t == 0) {
        double *p = ws.est_part[ws.wpar] + (size_t) blockIdx.x * 4;
        p[0] = ei.sx;
        p[1] = ei.sy;
        p[2] = (double) ei.th;
        p[3] = (double) ei.w;
    }
}

// Materialise a pending lazy gather: pose and the live genealogy rows (B.rows) of keep[k] into slot k of the other buffer
// set, w = 1/N (core.cpp:744-747); the landmark records stay where they are (kernels.h: gen).  Needed before anything
// but the next update reads the set (download, stand-alone predict / estimate).  Every launch publishes the
// resulting state in the other Ctrl slot; the host flips its slot afterwards.  blockIdx.y = group of kRowsPerRole rows.
__global__ void __launch_bounds__(kBlock) gather_kernel(Buffers B, WeightScratch ws) {
    Ctrl *ctrl = B.ctrl;
    const int cur = ctrl->live[B.slot];
    const bool pend = ctrl->pend[B.slot] != 0;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        ctrl->live[B.slot ^ 1] = pend ? cur ^ 1 : cur;
        ctrl->pend[B.slot ^ 1] = 0;
    }
    if (!pend) return;
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= B.n) return;
    const int anc = ws.keep[B.slot][k];
    if (anc < 0) return;  // sharded runs: arrived from another shard: pose and genealogy already in place, w = 1/N
    const size_t S = (size_t) B.ncap;
    if (blockIdx.y == 0) {
        float4 pa = B.poseA[cur][anc];
        pa.w = ctrl->inv_n;
        B.poseA[cur ^ 1][k] = pa;
        B.poseB[cur ^ 1][k] = B.poseB[cur][anc];
        B.poseC[cur ^ 1][k] = B.poseC[cur][anc];
    }
    const int32_t *__restrict__ src = B.gen[cur];
    int32_t *__restrict__ dst = B.gen[cur ^ 1];
    if (B.compact) {  // every chunk of the particle (grid.y == 1)
        const int4 *__restrict__ s4 = reinterpret_cast<const int4 *>(src);
        int4 *__restrict__ d4 = reinterpret_cast<int4 *>(dst);
        for (int c = 0; c < ((B.cap_rows + 3) >> 2); c++) d4[(size_t) c * S + k] = s4[(size_t) c * S + anc];
        return;
    }
    const int r0 = blockIdx.y * kRowsPerRole, r1 = min(B.n_rows, r0 + kRowsPerRole);
    for (int r = r0; r < r1; r++) {
        const size_t e = (size_t) B.rows[r];
        dst[e * S + k] = src[e * S + anc];
    }
}

SLAM_DEV void read_through_genealogy(const Buffers &B, const int32_t *__restrict__ live, int cur, size_t S, int l, int anc,
                                     float4 &la, float &lb);

// Flatten the genealogy: every landmark record into its particle's own slot of the row's other buffer, genealogy row 0
// = identity (the host points every landmark at row 0 afterwards and flips every landmark row's live flag in its
// table).  Requires a plain set (no pending gather), B.erow and B.lmk_live.  Used by download
// and before records from other shards are put in place.  blockIdx.y = group of 8 landmarks.  Row 0 may be in use as a
// source row: it is rewritten by a second launch (identity_kernel), after every block of this one has read it.
__global__ void __launch_bounds__(kBlock) flatten_kernel(Buffers B, int nf) {
    const int cur = B.ctrl->live[B.slot];
    const int32_t *__restrict__ live = B.lmk_live;
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= B.n) return;
    const size_t S = (size_t) B.ncap;
    const int32_t *__restrict__ gen = B.gen[cur];
    const int j0 = blockIdx.y * kLmkPerBlockY, j1 = min(nf, j0 + kLmkPerBlockY);
    for (int j = j0; j < j1; j++) {
        const int sl = gen[gen_index(B.compact, S, B.erow[j], (size_t) k)];
        const int b = live[j];
        if (B.n_shards > 1) {  // distributed context: global slot ids
            float4 la;
            float lb;
            read_through_genealogy(B, live, cur, S, j, k, la, lb);
            B.lmkA[b ^ 1][(size_t) j * S + k] = la;
            B.lmkB[b ^ 1][(size_t) j * S + k] = lb;
        } else if (sl < 0) {  // arrival pool
            const size_t at = (size_t) j * B.pool_cap + (sl & ~kPoolBit);
            B.lmkA[b ^ 1][(size_t) j * S + k] = B.poolA[at];
            B.lmkB[b ^ 1][(size_t) j * S + k] = B.poolB[at];
        } else {
            B.lmkA[b ^ 1][(size_t) j * S + k] = B.lmkA[b][(size_t) j * S + sl];
            B.lmkB[b ^ 1][(size_t) j * S + k] = B.lmkB[b][(size_t) j * S + sl];
        }
    }
}

// One block: the scan every block of a small context does for itself (scan_block_totals: same association, so the
// results are bit-identical), once, into global memory: [0..nb] exclusive prefix, [nb+1] sum w, [nb+2] sum w^2.
// Round 5: with the prefix built in global memory every thread re-read its own 16 stores one dependent trip at a time, and its
// totals came in four dependent batches: 12.9 us for 3 912 totals (11.6 % of a config-4 step).  reg_mode (linear weights): a
// thread's segment (at most 32 totals: 8 192 blocks) is read in one burst -- contiguous, 64 bytes per thread at config 4 -- and
// kept in REGISTERS (a first version staged table and prefix in LDS: 10.2 us, the segments' stride made every access a 32- or
// 64-way bank conflict); the prefix goes straight out.  The operations of scan_finish in its order (k ascending inside a
// segment, the wave scan, the waves' sums left to right): the same bits.
__global__ void __launch_bounds__(kBlock) scan_kernel(WeightScratch ws, int logw, int reg_mode) {
    __shared__ double sh_a[kBlock / kWave], sh_q[kBlock / kWave];
    double W, Q, Mx = 0.0;
    double *out = ws.scan[ws.wpar];
    if (reg_mode && ws.nblocks > 0) {
        const int per = (ws.nblocks + kBlock - 1) / kBlock;
        const float *tw = ws.blk_w[ws.wpar], *tq = tw + ws.nblocks;
        if (per <= 4) scan_segments<4>(tw, tq, ws.nblocks, out, sh_a, sh_q, W, Q);
        else if (per <= 8) scan_segments<8>(tw, tq, ws.nblocks, out, sh_a, sh_q, W, Q);
        else if (per <= 16) scan_segments<16>(tw, tq, ws.nblocks, out, sh_a, sh_q, W, Q);
        else scan_segments<kScanMaxPer>(tw, tq, ws.nblocks, out, sh_a, sh_q, W, Q);
    } else {
        scan_block_totals(ws.blk_w[ws.wpar], ws.nblocks, ws.nblocks, logw != 0, out, sh_a, sh_q, W, Q, Mx);
    }
    if (threadIdx.x == 0) {
        out[ws.nblocks + 1] = W;
        out[ws.nblocks + 2] = Q;
        out[ws.nblocks + 3] = Mx;
    }
}

// One block: reduces the estimate partials (-> Ctrl.est, history slot) on demand.
__global__ void __launch_bounds__(kBlock) finish_kernel(Buffers B, WeightScratch ws, double *hist, int par) {
    __shared__ EstItem sh_est[kBlock / kWave];
    finish_estimate(B, ws, par, hist, sh_est);
}

// ---------------------------------------------------------------------------------------------------
// Stand-alone pose estimate (used when the particle set changed since the last update, e.g. after predicts)
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) estimate_kernel(Buffers B, WeightScratch ws) {
    __shared__ EstItem sh_est[kBlock / kWave];
    const int t = threadIdx.x;
    const int i = blockIdx.x * kBlock + t;
    EstItem ei{0.0, 0.0, -3.0e38f, 0.0f, 0x7fffffff};
    if (i < B.n) {
        const float4 pa = B.poseA[B.ctrl->live[B.slot]][i];
        ei = EstItem{(double) pa.x, (double) pa.y, pa.w, pa.z, i};
    }
    ei = block_reduce_est(ei, sh_est);
    if (t == 0) {
        double *p = ws.est_part[ws.wpar] + (size_t) blockIdx.x * 4;
        p[0] = ei.sx;
        p[1] = ei.sy;
        p[2] = (double) ei.th;
        p[3] = (double) ei.w;
        if (blockIdx.x == 0) {  // the resampling record that goes with a history entry: the last update's
            ws.est_part[ws.wpar][4 * (size_t) ws.nblocks] = (double) B.ctrl->neff;
            ws.est_part[ws.wpar][4 * (size_t) ws.nblocks + 1] = (double) (B.ctrl->resampled | (B.ctrl->status << 1));
        }
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

// The MULTIPARTICLE_ACCELERATOR form of the window (fastslam2.cpp:172-286; AcceleratorHandler.h:17-21): `records` records back to
// back, each self-describing: [n][xv 3][R 4][n x (xf 2, Pf 4)][n x 16 output floats].  One thread per feature; tab[f] = (offset
// of the record's first float, feature index inside the record, the record's n).  Outputs in place, as the FPGA wrote them.
__global__ void __launch_bounds__(kBlock) jacobians_multi_kernel(float *__restrict__ win, const uint32_t *__restrict__ tab, uint32_t nfeat) {
    const uint32_t f = blockIdx.x * kBlock + threadIdx.x;
    if (f >= nfeat) return;
    const uint32_t base = tab[3 * (size_t) f], k = tab[3 * (size_t) f + 1], n = tab[3 * (size_t) f + 2];
    const float *h = win + base + 1;
    const float x = h[0], y = h[1], th = h[2];
    const float r00 = h[3], r10 = h[4], r01 = h[5], r11 = h[6];
    const float *p = h + 7 + 6 * (size_t) k;
    Jac j = jacobian(x, y, th, p[0], p[1], p[2], p[3], p[5], r00, r01, r10, r11);
    float *o = win + base + 8 + 6 * (size_t) n + 16 * (size_t) k;
    o[0] = j.zp0; o[1] = j.zp1;
    o[2] = j.hf00; o[3] = j.hf01; o[4] = j.hf10; o[5] = j.hf11;
    o[6] = j.hv00; o[7] = j.hv01; o[8] = 0.0f; o[9] = j.hv10; o[10] = j.hv11; o[11] = -1.0f;
    o[12] = j.s00; o[13] = j.s01; o[14] = j.s10; o[15] = j.s11;
}

// Known-answer entry point (slamgpu_kat): the scalar device functions of THIS build against the reference's edge-case
// vectors (tests/golden/kat_functions.npz): op 0 trigonometricOffset (core.cpp:460-477; the fast build's wrap_pi),
// op 1 / 2 gaussEvaluate for D = 2 / 3 (fastslam2.cpp:127-163) in the form the update kernel evaluates it.
__global__ void __launch_bounds__(kBlock) kat_kernel(int op, const float *__restrict__ in, int n, float *__restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (op == 0) {
#ifdef SLAM_FAST_MATH
        out[i] = wrap_pi(in[i]);
#else
        out[i] = trig_offset(in[i]);
#endif
    } else if (op == 1) {
        const float *p = in + 5 * (size_t) i;  // v0 v1 s00 s10 s11
#ifdef SLAM_FAST_MATH
        const Gauss2 g = gauss2_parts(p[2], p[3], p[4], p[0], p[1]);
        out[i] = __expf(g.E) * g.norm;
#else
        out[i] = gauss2(p[0], p[1], p[2], p[3], p[4]);
#endif
    } else {
        const float *p = in + 9 * (size_t) i;  // v0 v1 v2 a00 a10 a11 a20 a21 a22
#ifdef SLAM_FAST_MATH
        const L3r L = llt3r(Sym3{p[3], p[4], p[5], p[6], p[7], p[8]});
        out[i] = __expf(gauss3_exponent(L, p[0], p[1], p[2])) * (0.15915494309189535f * ((L.r0 * L.r1) * L.r2));
#else
        out[i] = gauss3(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]);
#endif
    }
}

// ---------------------------------------------------------------------------------------------------
// Observation front end on the device: getObservations = findVisibleLandmarks +
// computeRangeBearing (core.cpp:185-273), addObservationNoise (:438-449), dataAssociationKnown (:91-120), restated
// operation by operation from the reference's float / double mix (the product's host front end, host/frontend.cpp, is
// the CPU twin and reproduces the reference's tape bit for bit).  One block: the landmark list is scanned in chunks of 256
// with an ordered block-wide compaction (visibility order = landmark order, as upstream), then the visible ones are split
// into re-observed (zf, idf) and new (zn) by the device-resident association table, again in order, and the new ones get
// feature indices nf, nf + 1, ...  The bearing is atan2 evaluated in double and rounded once (glibc's atan2f, which the
// reference calls, is within 1 ulp of that).
// ---------------------------------------------------------------------------------------------------
SLAM_DEV int block_exclusive_count(int flag, int *sh, int &total) {
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const unsigned long long mask = __ballot(flag);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) sh[wv] = __popcll(mask);
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < kBlock / kWave; k++)
        if (k < wv) base += sh[k];
    total = ((sh[0] + sh[1]) + sh[2]) + sh[3];
    __syncthreads();
    return base + before;
}

__global__ void __launch_bounds__(kBlock) observe_kernel(ObserveArgs A) {
    __shared__ int sh[kBlock / kWave];
    float *z = reinterpret_cast<float *>(A.out + 1);
    int32_t *vis = reinterpret_cast<int32_t *>(z + 2 * (size_t) A.nlm);
    float *zf = reinterpret_cast<float *>(vis + A.nlm);
    int32_t *idf = reinterpret_cast<int32_t *>(zf + 2 * (size_t) A.nlm);
    float *zn = reinterpret_cast<float *>(idf + A.nlm);
    const float cph = cosf(A.phi), sph = sinf(A.phi);
    // pass 1: visibility + range / bearing, compacted in landmark order
    int nz = 0;
    for (int j0 = 0; j0 < A.nlm; j0 += kBlock) {
        const int j = j0 + (int) threadIdx.x;
        bool v = false;
        float dx = 0.f, dy = 0.f;
        if (j < A.nlm) {
            dx = A.lm[j] - A.x;
            dy = A.lm[(size_t) A.nlm + j] - A.y;
            const double d2 = (double) dx * (double) dx + (double) dy * (double) dy;
            v = (fabsf(dx) < A.max_range) && (fabsf(dy) < A.max_range) && ((dx * cph + dy * sph) > 0.0f) &&
                (d2 < (double) A.max_range * (double) A.max_range);
        }
        int tot;
        const int at = nz + block_exclusive_count(v ? 1 : 0, sh, tot);
        if (v) {
            const double d2 = (double) dx * (double) dx + (double) dy * (double) dy;
            vis[at] = j;
            z[2 * at] = (float) sqrt(d2);
            z[2 * at + 1] = (float) atan2((double) dy, (double) dx) - A.phi;
        }
        nz += tot;
    }
    __syncthreads();
    // sensor noise: z(0,c) += randn * sqrt(R00); z(1,c) += randn * sqrt(R11)   (core.cpp:438-449)
    if (A.noise) {
        for (int c = threadIdx.x; c < nz; c += kBlock) {
            float g0, g1;
            if (A.noise == 1) {
                g0 = A.r1[c];
                g1 = A.r2[c];
            } else {
                U4 r = philox4x32((uint32_t) vis[c], A.step, 3u, 0u, A.k0, A.k1);
                sensor_normals(r, g0, g1);
            }
            z[2 * c] = z[2 * c] + g0 * A.sr;
            z[2 * c + 1] = z[2 * c + 1] + g1 * A.sb;
        }
        __syncthreads();
    }
    // pass 2: dataAssociationKnown: split by the table, in order; new landmarks get indices nf, nf + 1, ...
    int m = 0, n = 0;
    for (int c0 = 0; c0 < nz; c0 += kBlock) {
        const int c = c0 + (int) threadIdx.x;
        const bool in = c < nz;
        const int lmk = in ? vis[c] : 0;
        const int t = in ? A.table[lmk] : 0;
        const bool is_new = in && t < 0;
        int tot_new, tot_old;
        const int an = n + block_exclusive_count(is_new ? 1 : 0, sh, tot_new);
        const int ao = m + block_exclusive_count((in && !is_new) ? 1 : 0, sh, tot_old);
        if (is_new) {
            zn[2 * an] = z[2 * c];
            zn[2 * an + 1] = z[2 * c + 1];
            A.table[lmk] = A.nf + an;
        } else if (in) {
            zf[2 * ao] = z[2 * c];
            zf[2 * ao + 1] = z[2 * c + 1];
            idf[ao] = t;
        }
        n += tot_new;
        m += tot_old;
    }
    if (threadIdx.x == 0) {
        A.out->nz = nz;
        A.out->m = m;
        A.out->n = n;
        A.out->nf_after = A.nf + n;
    }
}

// The same front end for the step loop (slamgpu_step_observe): the observation goes straight into the device-resident packet
// the update launch reads (kernels.h: ObsPacket, fixed layout) -- and with it the genealogy bookkeeping the host does for a
// packet of its own (slamgpu.cpp: do_update): the row every re-observed landmark leaves (| live buffer | fresh), the row this
// update opens (the lowest unused one), reference counts, the live-buffer flips, the rows a pending gather still has to
// compose.  One block of 1 024 threads; the host sends the true pose and learns nothing about the observation.
constexpr int kObsThreads = 1024;
SLAM_DEV int block_exclusive_count_n(int flag, int *sh, int &total) {  // block_exclusive_count for kObsThreads threads
    constexpr int NW = kObsThreads / kWave;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const unsigned long long mask = __ballot(flag);
    const int before = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) sh[wv] = __popcll(mask);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const int c = sh[k];
        if (k < wv) base += c;
        tot += c;
    }
    total = tot;
    __syncthreads();
    return base + before;
}

SLAM_DEV int block_exclusive_sum_n(int v, int *sh, int &total) {  // the same for values (wave scan by shuffles)
    constexpr int NW = kObsThreads / kWave;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    int inc = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int o = __shfl_up(inc, d, kWave);
        if (lane >= d) inc += o;
    }
    if (lane == kWave - 1) sh[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
        const int c = sh[k];
        if (k < wv) base += c;
        tot += c;
    }
    total = tot;
    __syncthreads();
    return base + inc - v;
}

__global__ void __launch_bounds__(kObsThreads) observe_book_kernel(ObserveArgs A) {
    __shared__ int sh[kObsThreads / kWave];
    __shared__ int sh_min;
    const int T = kObsThreads, t = threadIdx.x, C = A.nlm;
    ObsPacket *P = A.pkt;
    int32_t *base = reinterpret_cast<int32_t *>(P + 1);
    int32_t *idf = base;
    float *zf = reinterpret_cast<float *>(base + C);
    float *zn = zf + 2 * (size_t) C;
    int32_t *row = reinterpret_cast<int32_t *>(zn + 2 * (size_t) C);
    int32_t *rows = row + C;
    float *z = reinterpret_cast<float *>(A.out + 1);
    int32_t *vis = reinterpret_cast<int32_t *>(z + 2 * (size_t) C);
    const int nf0 = A.book->nf, fresh = A.book->fresh_row;
    const float cph = cosf(A.phi), sph = sinf(A.phi);
    if (t == 0) sh_min = 0x7fffffff;
    // pass 1: visibility + range / bearing, compacted in landmark order (observe_kernel, core.cpp:185-273)
    int nz = 0;
    for (int j0 = 0; j0 < C; j0 += T) {
        const int j = j0 + t;
        bool v = false;
        float dx = 0.f, dy = 0.f;
        if (j < C) {
            dx = A.lm[j] - A.x;
            dy = A.lm[(size_t) C + j] - A.y;
            const double d2 = (double) dx * (double) dx + (double) dy * (double) dy;
            v = (fabsf(dx) < A.max_range) && (fabsf(dy) < A.max_range) && ((dx * cph + dy * sph) > 0.0f) &&
                (d2 < (double) A.max_range * (double) A.max_range);
        }
        int tot;
        const int at = nz + block_exclusive_count_n(v ? 1 : 0, sh, tot);
        if (v) {
            const double d2 = (double) dx * (double) dx + (double) dy * (double) dy;
            vis[at] = j;
            z[2 * at] = (float) sqrt(d2);
            z[2 * at + 1] = (float) atan2((double) dy, (double) dx) - A.phi;
        }
        nz += tot;
    }
    __syncthreads();
    // sensor noise (core.cpp:438-449)
    if (A.noise) {
        for (int c = t; c < nz; c += T) {
            float g0, g1;
            if (A.noise == 1) {
                g0 = A.r1[c];
                g1 = A.r2[c];
            } else {
                U4 r = philox4x32((uint32_t) vis[c], A.step, 3u, 0u, A.k0, A.k1);
                sensor_normals(r, g0, g1);
            }
            z[2 * c] = z[2 * c] + g0 * A.sr;
            z[2 * c + 1] = z[2 * c + 1] + g1 * A.sb;
        }
        __syncthreads();
    }
    // pass 2: dataAssociationKnown (core.cpp:91-120): split by the table, in order; new landmarks get indices nf0, nf0 + 1, ...
    // (new landmarks beyond the context's capacity are dropped and flagged)
    const int room = A.cap_nf - nf0;
    int m = 0, n = 0, dropped = 0;
    for (int c0 = 0; c0 < nz; c0 += T) {
        const int c = c0 + t;
        const bool in = c < nz;
        const int lmk = in ? vis[c] : 0;
        const int tb = in ? A.table[lmk] : 0;
        const bool is_new = in && tb < 0;
        int tot_new, tot_old;
        const int an = n + block_exclusive_count_n(is_new ? 1 : 0, sh, tot_new);
        const int ao = m + block_exclusive_count_n((in && !is_new) ? 1 : 0, sh, tot_old);
        if (is_new) {
            if (an < room) {
                zn[2 * an] = z[2 * c];
                zn[2 * an + 1] = z[2 * c + 1];
                A.table[lmk] = nf0 + an;
            }
        } else if (in) {
            zf[2 * ao] = z[2 * c];
            zf[2 * ao + 1] = z[2 * c + 1];
            idf[ao] = tb;
        }
        n += tot_new;
        m += tot_old;
    }
    if (n > room) {
        dropped = n - room;
        n = room;
    }
    __syncthreads();
    // Row consolidation (as slamgpu.cpp: do_update does for host-made packets): past cons_target rows in use, the landmarks of
    // stale rows -- taken in row order while they fit the budget -- are handed to the update launch behind the re-observed ones
    // (a no-op update that rewrites their records into the particles' own slots) and join the row this update opens.
    int n_live = 0;
    if (A.cons_target >= 0) {
        for (int r0 = 0; r0 < A.cap_rows; r0 += T) {
            const int r = r0 + t;
            int tot;
            block_exclusive_count_n((r < A.cap_rows && A.refcnt[r] > 0) ? 1 : 0, sh, tot);
            n_live += tot;
        }
    }
    const bool want_cons = A.cons_target >= 0 && n_live > A.cons_target;
    // the row this update opens: the lowest one no landmark uses
    int e_new = -1;
    if (m + n > 0 || want_cons) {
        int mine = 0x7fffffff;
        for (int r = t; r < A.cap_rows; r += T)
            if (A.refcnt[r] == 0) mine = min(mine, r);
#pragma unroll
        for (int d = kWave / 2; d > 0; d >>= 1) mine = min(mine, __shfl_xor(mine, d, kWave));
        if ((t & (kWave - 1)) == 0 && mine != 0x7fffffff) atomicMin(&sh_min, mine);
        __syncthreads();
        e_new = sh_min;  // (cap_rows = cap_nf + 1 rows for at most cap_nf landmarks: one is always free)
    }
    // the rows the re-observed landmarks leave; they move to e_new and their records to the other buffer
    for (int k = t; k < m; k += T) {
        const int j = idf[k];
        const int r = A.erow[j];
        row[k] = r | (A.live[j] ? kRowLiveBit : 0) | (r == fresh ? kRowFreshBit : 0);
        A.live[j] ^= 1;
        atomicSub(&A.refcnt[r], 1);
        A.erow[j] = e_new;
    }
    for (int k = t; k < n; k += T) {
        A.erow[nf0 + k] = e_new;
        A.live[nf0 + k] = 0;  // a new row's first records go to buffer 0
    }
    __syncthreads();
    int nc = 0;
    if (want_cons) {
        // (the re-observed landmarks have left their rows: what the counts say now is what stays behind)
        const int budget = max(A.cons_budget, nz / 16), spare = n_live - A.cons_target;
        int cum = 0, taken = 0;
        for (int r0 = 0; r0 < A.cap_rows; r0 += T) {
            const int r = r0 + t;
            const int cnt = (r < A.cap_rows && r != e_new) ? A.refcnt[r] : 0;
            int tot_c, tot_r;
            const int before_c = cum + block_exclusive_sum_n(cnt, sh, tot_c);
            const int before_r = taken + block_exclusive_count_n(cnt > 0 ? 1 : 0, sh, tot_r);
            if (r < A.cap_rows) A.take[r] = (cnt > 0 && before_c + cnt <= budget && before_r < spare) ? 1 : 0;
            cum += tot_c;
            taken += tot_r;
        }
        __syncthreads();
        for (int j0 = 0; j0 < nf0; j0 += T) {
            const int j = j0 + t;
            const int r = j < nf0 ? A.erow[j] : 0;
            const bool mv = j < nf0 && r != e_new && A.take[r] != 0;
            int tot;
            const int at = m + nc + block_exclusive_count_n(mv ? 1 : 0, sh, tot);
            if (mv) {
                idf[at] = j;
                row[at] = r | (A.live[j] ? kRowLiveBit : 0) | (r == fresh ? kRowFreshBit : 0);
                A.live[j] ^= 1;
                atomicSub(&A.refcnt[r], 1);
                A.erow[j] = e_new;
            }
            nc += tot;
        }
        __syncthreads();
    }
    // the rows still in use, without e_new: what a pending gather composes
    int n_rows = 0;
    for (int r0 = 0; r0 < A.cap_rows; r0 += T) {
        const int r = r0 + t;
        const bool on = r < A.cap_rows && r != e_new && A.refcnt[r] > 0;
        int tot;
        const int at = n_rows + block_exclusive_count_n(on ? 1 : 0, sh, tot);
        if (on) rows[at] = r;
        n_rows += tot;
    }
    if (t == 0) {
        if (e_new >= 0) A.refcnt[e_new] = m + n + nc;
        P->m = m;
        P->n = n;
        P->nf = nf0;
        P->n_rows = n_rows;
        P->e_new = e_new;
        P->status = dropped ? kStatusCapacity : 0;
        P->cap = C;
        P->pad = nc;  // landmarks consolidated by the update launch: idf[m .. m + nc), row[m .. m + nc)
        A.book->nf = nf0 + n;
        A.book->fresh_row = e_new;
        if (dropped) A.book->status |= kStatusCapacity;
        A.out->nz = nz;
        A.out->m = m;
        A.out->n = n;
        A.out->nf_after = nf0 + n;
    }
}

// ---------------------------------------------------------------------------------------------------
// Per-particle gated nearest-neighbour data association.  The reference only has it for EKF-SLAM:
// EKFSLAM::dataAssociate (algorithms/ekfslam.cpp:151-189) over ekfComputeAssociation (:131-149): for every observation z,
// over the Nf landmarks j: v = wrap(z - h_j), S = H P H^T + R, nis = v^T S^-1 v, nd = nis + log det S;
//     if (nis < gate1 && nd < nbest) { nbest = nd; jbest = j; } else if (nis < outer) outer = nis;
// then: jbest found -> associate; else outer > gate2 -> new feature; else the observation is dropped.
// For a FastSLAM particle the pose is given, so P = blockdiag(0, Pf_1, .., Pf_Nf) and H P H^T = Hf Pf Hf^T: exactly the Sf
// of computeJacobians (core.cpp:682-704).  One particle per work-item; the particle's landmarks are read once per batch of
// kAssocBatch observations, through the genealogy (needs Buffers::erow / lmk_live; plain set).
// ---------------------------------------------------------------------------------------------------
SLAM_DEV void read_through_genealogy(const Buffers &B, const int32_t *__restrict__ live, int cur, size_t S, int l, int anc,
                                     float4 &la, float &lb);

// What one particle's estimate of one landmark contributes to the gates of EKFSLAM::dataAssociate (ekfslam.cpp:160-176): the
// predicted observation, S^-1 and log det S with S = Hf Pf Hf^T + R; then per observation nis = v^T S^-1 v and nd = nis + log det S.
// strict build: the reference's operations (computeJacobians, the LU inverse and determinant of the dynamic 2x2, core.cpp:579-715).
// fast build (round 6): the restructured arithmetic of the update's second pass (device_math.h: observe2 -- rsq, the polynomial
// atan2, FMAs; the closed-form inverse on one v_rcp_f32; v_log_f32; wrap_pi): ~110 instead of ~435 VALU instructions per
// evaluated triple on the 10 000-landmark map (SQ counters: profiles/gated_association_r06.txt), the same decisions wherever a gate
// is not within rounding of its bound (tests/test_association.py holds both builds to the reference's decision vectors).
struct AssocLm {
    float zp0, zp1, i00, i01, i10, i11, ldet;
};
SLAM_DEV AssocLm assoc_landmark(const float4 &pa, const float4 &la, float lb, float r00, float r01, float r10, float r11) {
    AssocLm A;
#ifdef SLAM_FAST_MATH
    const Obs2 o = observe2(pa.x, pa.y, pa.z, la.x, la.y, la.z, la.w, lb, r00, 0.5f * (r01 + r10), r11);
    const float det = ffma(o.s00, o.s11, -o.s10 * o.s10);
    const float rdet = __builtin_amdgcn_rcpf(det);
    A.zp0 = o.zp0;
    A.zp1 = o.zp1;  // (not wrapped: assoc_gate wraps the residual, which is the same angle modulo 2 pi)
    A.i00 = o.s11 * rdet;
    A.i01 = A.i10 = -o.s10 * rdet;
    A.i11 = o.s00 * rdet;
    A.ldet = 0.69314718055994531f * __builtin_amdgcn_logf(det);  // v_log_f32 is log2
#else
    const Jac jc = jacobian(pa.x, pa.y, pa.z, la.x, la.y, la.z, la.w, lb, r00, r01, r10, r11);
    inverse2(jc.s00, jc.s01, jc.s10, jc.s11, A.i00, A.i01, A.i10, A.i11);
    A.ldet = logf(determinant2(jc.s00, jc.s01, jc.s10, jc.s11));
    A.zp0 = jc.zp0;
    A.zp1 = jc.zp1;
#endif
    return A;
}
SLAM_DEV void assoc_gate(const AssocLm &A, float zr, float zb, float &nis, float &nd) {
    const float v0 = zr - A.zp0;
#ifdef SLAM_FAST_MATH
    const float v1 = wrap_pi(zb - A.zp1);
#else
    const float v1 = trig_offset(zb - A.zp1);
#endif
    const float t0 = v0 * A.i00 + v1 * A.i10, t1 = v0 * A.i01 + v1 * A.i11;  // v^T S^-1
    nis = t0 * v0 + t1 * v1;
    nd = nis + A.ldet;
}

// EXCL (slamgpu_particle_assoc::excl_*; slamgpu_associate never): the EXCLUSION rule of a particle's own map.  The gates measure an
// observation against S = Hf Pf Hf^T + R -- for a converged landmark that is R, half a metre at five sigma -- and know nothing of the
// particle's own pose error (the EKF's S carries it, ekfslam.cpp:160-176; a particle's pose is a point).  So a particle a metre off
// calls an observation of a mapped landmark NEW and opens a duplicate next to it.  With the rule on, an observation no landmark
// gates is placed in the world from the particle's pose; if a landmark of the particle lies within excl_base + excl_per_m * range of
// that point the observation cannot be new: it is matched with that landmark when no other is within unique_ratio times the distance
// (the update then pulls pose and landmark together, at the price of the innovation's likelihood), and discarded otherwise.
template <bool EXCL>
__global__ void __launch_bounds__(kBlock) associate_kernel(Buffers B, int nf, const float *__restrict__ z, int nz, float r00, float r01,
                                                            float r10, float r11, float gate1, float gate2, float excl_base, float excl_per_m, float unique_ratio,
                                                            const uint32_t *__restrict__ retired, int32_t *__restrict__ labels, int by_obs) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= B.n) return;
    const int cur = B.ctrl->live[B.slot];
    const size_t S = (size_t) B.ncap;
    const float4 pa = B.poseA[cur][i];
    for (int q0 = 0; q0 < nz; q0 += kAssocBatch) {
        float nbest[kAssocBatch], outer[kAssocBatch];
        int jbest[kAssocBatch];
        [[maybe_unused]] float wx[kAssocBatch], wy[kAssocBatch], d1[kAssocBatch], d2[kAssocBatch];
        [[maybe_unused]] int j1[kAssocBatch];
#pragma unroll
        for (int q = 0; q < kAssocBatch; q++) {
            nbest[q] = INFINITY;  // the reference's `float nbest = 1e60` is +inf in float32
            outer[q] = INFINITY;
            jbest[q] = -1;
            if constexpr (EXCL) {
                const int qq = min(q0 + q, nz - 1);
                float sn, cs;
                sincosf(pa.z + z[2 * qq + 1], &sn, &cs);
                wx[q] = pa.x + z[2 * qq] * cs;  // where this particle's pose puts the observation
                wy[q] = pa.y + z[2 * qq] * sn;
                d1[q] = d2[q] = INFINITY;
                j1[q] = -1;
            }
        }
        // Several landmarks per trip through the loop (kTogether).  The scan is bound by dependent issue, not by memory: 12 700 VALU instructions per wave on
        // example_webmap's 35 landmarks with a wave and a half per SIMD, and a wave alone on its SIMD issues a DEPENDENT instruction every
        // ~9.6 cycles but two interleaved chains at ~5.6 each (tools/microbench/valu_latency.hip; SQ counters:
        // profiles/rocprof_sq_r06_assoc_exhaustive_c3.txt).  The gate arithmetic of landmarks j and j + 1 is independent; only the
        // comparisons run through both, in landmark order: the same decisions.
        auto take = [&](int q, int j, float nis, float nd, const float4 &la) {
            if (nis < gate1 && nd < nbest[q]) {
                nbest[q] = nd;
                jbest[q] = j;
            } else if (nis < outer[q]) {
                outer[q] = nis;
            }
            if constexpr (EXCL) {  // (an absent record compares false twice: it is nobody's neighbour)
                const float ex = la.x - wx[q], ey = la.y - wy[q], dd = ex * ex + ey * ey;
                if (dd < d1[q]) {
                    d2[q] = d1[q];
                    d1[q] = dd;
                    j1[q] = j;
                } else if (dd < d2[q]) {
                    d2[q] = dd;
                }
            }
        };
        constexpr int kTogether = 4;
        for (int j = 0; j < nf; j += kTogether) {
            // (a retired landmark -- slamgpu_retire_landmarks: a duplicate the caller's policy has given up -- takes no part; uniform)
            int jj[kTogether];
            bool onj[kTogether];
            float4 las[kTogether];
            float lbs[kTogether];
            AssocLm As[kTogether];
#pragma unroll
            for (int t = 0; t < kTogether; t++) {
                jj[t] = min(j + t, nf - 1);
                onj[t] = j + t < nf && !(retired && ((retired[jj[t] >> 5] >> (jj[t] & 31)) & 1u));
                read_through_genealogy(B, B.lmk_live, cur, S, jj[t], i, las[t], lbs[t]);
            }
#pragma unroll
            for (int t = 0; t < kTogether; t++) As[t] = assoc_landmark(pa, las[t], lbs[t], r00, r01, r10, r11);
#pragma unroll
            for (int q = 0; q < kAssocBatch; q++) {
                if (q0 + q < nz) {
                    float nis[kTogether], nd[kTogether];
#pragma unroll
                    for (int t = 0; t < kTogether; t++) assoc_gate(As[t], z[2 * (q0 + q)], z[2 * (q0 + q) + 1], nis[t], nd[t]);
#pragma unroll
                    for (int t = 0; t < kTogether; t++)
                        if (onj[t]) take(q, jj[t], nis[t], nd[t], las[t]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < kAssocBatch; q++)
            if (q0 + q < nz) {
                int label = jbest[q] > -1 ? jbest[q] : (outer[q] > gate2 ? kAssocNew : kAssocDiscard);
                if constexpr (EXCL) {
                    const float rho = excl_base + excl_per_m * z[2 * (q0 + q)];
                    if (jbest[q] < 0 && d1[q] < rho * rho) label = d2[q] > unique_ratio * unique_ratio * d1[q] ? j1[q] : kAssocDiscard;
                }
                labels[by_obs ? (size_t) (q0 + q) * S + i : (size_t) i * nz + q0 + q] = label;  // (by_obs: [nz][ncap], what the per-particle update reads)
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same association with a spatial prefilter (kernels.h: LmkBox / AssocGeom): O(N * nz * k) landmark evaluations instead
// of O(N * nz * Nf), k = the landmarks one grid cell holds.  Which landmarks a (particle, observation) pair may skip is
// decided by bounds that hold for every particle (below), so the labels are those of associate_kernel, decision for decision.
// ---------------------------------------------------------------------------------------------------
// one block per listed landmark: box of its estimates and largest covariance trace over all particle slots of the live buffer
// (a resample only removes particles from that set: the box stays valid until the landmark is written again)
__global__ void __launch_bounds__(kBlock) lmk_box_kernel(Buffers B, const int32_t *__restrict__ ids, int count, const uint32_t *__restrict__ retired,
                                                          LmkBox *__restrict__ box) {
    __shared__ float sh[5][kBlock / kWave];
    const int j = ids[blockIdx.x];
    if (retired && ((retired[j >> 5] >> (j & 31)) & 1u)) {  // the empty box: no cell (assoc_cells), so the landmark is never evaluated
        if (threadIdx.x == 0) box[j] = LmkBox{INFINITY, -INFINITY, INFINITY, -INFINITY, 0.0f, {0.0f, 0.0f, 0.0f}};
        return;
    }
    const int b = B.lmk_live[j];
    const float4 *__restrict__ a = B.lmkA[b] + (size_t) j * B.ncap;
    const float *__restrict__ c = B.lmkB[b] + (size_t) j * B.ncap;
    float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY, t = 0.0f;
    for (int k = threadIdx.x; k < B.n; k += kBlock) {
        const float4 v = a[k];
        x0 = fminf(x0, v.x);
        x1 = fmaxf(x1, v.x);
        y0 = fminf(y0, v.y);
        y1 = fmaxf(y1, v.y);
        t = fmaxf(t, fabsf(v.z) + fabsf(c[k]));  // tr Pf >= its largest eigenvalue
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, d, kWave));
        x1 = fmaxf(x1, __shfl_xor(x1, d, kWave));
        y0 = fminf(y0, __shfl_xor(y0, d, kWave));
        y1 = fmaxf(y1, __shfl_xor(y1, d, kWave));
        t = fmaxf(t, __shfl_xor(t, d, kWave));
    }
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    if (lane == 0) {
        sh[0][wv] = x0; sh[1][wv] = x1; sh[2][wv] = y0; sh[3][wv] = y1; sh[4][wv] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        LmkBox o{};
        o.xmin = fminf(fminf(sh[0][0], sh[0][1]), fminf(sh[0][2], sh[0][3]));
        o.xmax = fmaxf(fmaxf(sh[1][0], sh[1][1]), fmaxf(sh[1][2], sh[1][3]));
        o.ymin = fminf(fminf(sh[2][0], sh[2][1]), fminf(sh[2][2], sh[2][3]));
        o.ymax = fmaxf(fmaxf(sh[3][0], sh[3][1]), fmaxf(sh[3][2], sh[3][3]));
        o.tmax = fmaxf(fmaxf(sh[4][0], sh[4][1]), fmaxf(sh[4][2], sh[4][3]));
        box[j] = o;
    }
}

// one block: bounding box of the particle poses, largest observed range => the grid over the region the observations can
// point into (pose box grown by the largest range)
// (the poses are scanned by kGeomBlocks workgroups -- one workgroup walking 10^5 poses alone took 0.21 ms of a 3.6 ms association --
// whose partial boxes the one-block kernel behind them joins)
constexpr int kGeomBlocks = 64;
__global__ void __launch_bounds__(kBlock) assoc_geom_partial_kernel(Buffers B, float *__restrict__ part) {
    __shared__ float sh[6][kBlock / kWave];
    const int cur = B.ctrl->live[B.slot];
    float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY, t0 = INFINITY, t1 = -INFINITY;
    const float th_ref = B.n > 0 ? B.poseA[cur][0].z : 0.0f;
    for (int k = blockIdx.x * kBlock + threadIdx.x; k < B.n; k += kGeomBlocks * kBlock) {
        const float4 v = B.poseA[cur][k];
        x0 = fminf(x0, v.x);
        x1 = fmaxf(x1, v.x);
        y0 = fminf(y0, v.y);
        y1 = fmaxf(y1, v.y);
        // heading relative to particle 0's, wrapped to [-pi, pi] (exactly: remainder): the set's headings are th_ref + [t0, t1] modulo 2 pi
        const float dt = remainderf(v.z - th_ref, 6.28318530717958648f);
        t0 = fminf(t0, dt);
        t1 = fmaxf(t1, dt);
    }
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, d, kWave));
        x1 = fmaxf(x1, __shfl_xor(x1, d, kWave));
        y0 = fminf(y0, __shfl_xor(y0, d, kWave));
        y1 = fmaxf(y1, __shfl_xor(y1, d, kWave));
        t0 = fminf(t0, __shfl_xor(t0, d, kWave));
        t1 = fmaxf(t1, __shfl_xor(t1, d, kWave));
    }
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    if (lane == 0) {
        sh[0][wv] = x0; sh[1][wv] = x1; sh[2][wv] = y0; sh[3][wv] = y1; sh[4][wv] = t0; sh[5][wv] = t1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float *o = part + 8 * blockIdx.x;
        o[0] = fminf(fminf(sh[0][0], sh[0][1]), fminf(sh[0][2], sh[0][3]));
        o[1] = fmaxf(fmaxf(sh[1][0], sh[1][1]), fmaxf(sh[1][2], sh[1][3]));
        o[2] = fminf(fminf(sh[2][0], sh[2][1]), fminf(sh[2][2], sh[2][3]));
        o[3] = fmaxf(fmaxf(sh[3][0], sh[3][1]), fmaxf(sh[3][2], sh[3][3]));
        o[4] = fminf(fminf(sh[4][0], sh[4][1]), fminf(sh[4][2], sh[4][3]));
        o[5] = fmaxf(fmaxf(sh[5][0], sh[5][1]), fmaxf(sh[5][2], sh[5][3]));
    }
}
__global__ void __launch_bounds__(kBlock) assoc_geom_kernel(Buffers B, AssocGridArgs A, const float *__restrict__ part) {
    __shared__ float sh[7][kBlock / kWave];
    const int cur = B.ctrl->live[B.slot];
    float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY, zm = 0.0f, t0 = INFINITY, t1 = -INFINITY;
    const float th_ref = B.n > 0 ? B.poseA[cur][0].z : 0.0f;
    if (threadIdx.x < kGeomBlocks) {
        const float *o = part + 8 * threadIdx.x;
        x0 = o[0]; x1 = o[1]; y0 = o[2]; y1 = o[3]; t0 = o[4]; t1 = o[5];
    }
    for (int q = threadIdx.x; q < A.nz; q += kBlock) zm = fmaxf(zm, fabsf(A.z[2 * q]));
#pragma unroll
    for (int d = kWave / 2; d > 0; d >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, d, kWave));
        x1 = fmaxf(x1, __shfl_xor(x1, d, kWave));
        y0 = fminf(y0, __shfl_xor(y0, d, kWave));
        y1 = fmaxf(y1, __shfl_xor(y1, d, kWave));
        zm = fmaxf(zm, __shfl_xor(zm, d, kWave));
        t0 = fminf(t0, __shfl_xor(t0, d, kWave));
        t1 = fmaxf(t1, __shfl_xor(t1, d, kWave));
    }
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    if (lane == 0) {
        sh[0][wv] = x0; sh[1][wv] = x1; sh[2][wv] = y0; sh[3][wv] = y1; sh[4][wv] = zm; sh[5][wv] = t0; sh[6][wv] = t1;
    }
    __syncthreads();
    for (int cidx = threadIdx.x; cidx < kAssocMaxCells * kAssocMaxCells + 1; cidx += kBlock) {
        A.cell_start[cidx] = 0;
        if (cidx < kAssocMaxCells * kAssocMaxCells) A.cell_fill[cidx] = 0;
    }
    if (threadIdx.x == 0) {
        AssocGeom g{};
        g.px0 = fminf(fminf(sh[0][0], sh[0][1]), fminf(sh[0][2], sh[0][3]));
        g.px1 = fmaxf(fmaxf(sh[1][0], sh[1][1]), fmaxf(sh[1][2], sh[1][3]));
        g.py0 = fminf(fminf(sh[2][0], sh[2][1]), fminf(sh[2][2], sh[2][3]));
        g.py1 = fmaxf(fmaxf(sh[3][0], sh[3][1]), fmaxf(sh[3][2], sh[3][3]));
        g.zmax = fmaxf(fmaxf(sh[4][0], sh[4][1]), fmaxf(sh[4][2], sh[4][3]));
        g.th_ref = th_ref;
        g.dth0 = fminf(fminf(sh[5][0], sh[5][1]), fminf(sh[5][2], sh[5][3]));
        g.dth1 = fmaxf(fmaxf(sh[6][0], sh[6][1]), fmaxf(sh[6][2], sh[6][3]));
        if (!(g.dth0 <= g.dth1)) {  // (no particle)
            g.dth0 = -3.14159274f;
            g.dth1 = 3.14159274f;
        }
        const float margin = 1.0f + 1e-3f * (fabsf(g.px0) + fabsf(g.px1) + fabsf(g.py0) + fabsf(g.py1) + g.zmax);
        g.x0 = g.px0 - g.zmax - margin;
        g.y0 = g.py0 - g.zmax - margin;
        const float ex = (g.px1 + g.zmax + margin) - g.x0, ey = (g.py1 + g.zmax + margin) - g.y0;
        g.cs = fmaxf(fmaxf(ex, ey) / (float) kAssocMaxCells, 0.5f);
        g.inv_cs = 1.0f / g.cs;
        g.nx = min(kAssocMaxCells, (int) (ex * g.inv_cs) + 1);
        g.ny = min(kAssocMaxCells, (int) (ey * g.inv_cs) + 1);
        g.total = 0;
        g.overflow = 0;
        g.pairs = 0ull;
        *A.geom = g;
    }
}

// A particle's estimate l of landmark j can pass a gate (nis < G) for an observation (r, b) only if it lies within rho of the
// world point p the observation implies for that particle.  With d = |l - pose|, e = |d - r| and D the wrapped bearing residual:
//   |l - p|^2 = (d - r)^2 + 2 d r (1 - cos D) <= e^2 + d r D^2,  so  |l - p| <= e + sqrt(d r) |D| <= e + d |D| + (e / 2) |D|
//   (r <= d + e; sqrt(d r) <= d + e / 2), and gate by gate (nis >= v_i^2 / S_ii for a positive definite S):
//   e = |v0| < sqrt(G S00) <= sqrt(G (t + R00)),   d |D| = d |v1| < d sqrt(G S11) <= sqrt(G (t + d^2 R11)),   |D| <= pi,
// t = tr Pf (>= its largest eigenvalue; the rows of Hf have norms 1 and 1 / d).  Over all particles: t <= tmax_j and
// d <= the largest distance between the pose box and the landmark's box.
// (round 6) |D| <= pi is the bound of a landmark the poses stand ON; a landmark whose box is dmin away from the pose box passes the
// bearing gate only with |D| < sqrt(G (t / dmin^2 + R11)) (S11 <= t / d^2 + R11, d >= dmin): a few hundredths of a radian beyond a
// few metres, so the third term is e |D| / 2 with THAT |D|, not e pi / 2 -- the radius of a landmark 30 m away falls from 4.9 to
// 3.7 m on BASELINE config 5's map and a cell holds half as many entries.
SLAM_DEV float assoc_radius(const LmkBox &bx, const AssocGeom &g, float r00, float r11, float G) {
    const float dx = fmaxf(fabsf(bx.xmax - g.px0), fabsf(g.px1 - bx.xmin)), dy = fmaxf(fabsf(bx.ymax - g.py0), fabsf(g.py1 - bx.ymin));
    const float D2 = dx * dx + dy * dy;
    const float gx = fmaxf(fmaxf(bx.xmin - g.px1, g.px0 - bx.xmax), 0.0f), gy = fmaxf(fmaxf(bx.ymin - g.py1, g.py0 - bx.ymax), 0.0f);
    const float dmin2 = gx * gx + gy * gy;
    const float Dcap = dmin2 > 0.0f ? fminf(3.14159274f, sqrtf(G * (bx.tmax / dmin2 + r11)) * 1.01f) : 3.14159274f;
    const float rho = (1.0f + 0.5f * Dcap) * sqrtf(G * (bx.tmax + r00)) + sqrtf(G * (bx.tmax + D2 * r11));
    return rho * 1.01f + 1e-3f;  // (rounding of the bound itself and of the kernels' own arithmetic)
}

// cells the grown box of landmark j overlaps (clipped to the grid); false: none
SLAM_DEV bool assoc_cells(const LmkBox &bx, const AssocGeom &g, float rho, int &cx0, int &cx1, int &cy0, int &cy1) {
    if (!(bx.xmin <= bx.xmax)) return false;  // (no particle slot: n = 0)
    const float fx0 = (bx.xmin - rho - g.x0) * g.inv_cs, fx1 = (bx.xmax + rho - g.x0) * g.inv_cs;
    const float fy0 = (bx.ymin - rho - g.y0) * g.inv_cs, fy1 = (bx.ymax + rho - g.y0) * g.inv_cs;
    if (fx1 < 0.0f || fy1 < 0.0f || fx0 >= (float) g.nx || fy0 >= (float) g.ny) return false;
    cx0 = max(0, (int) floorf(fx0) - 1);  // (one cell of slack on every side: the point's own cell index is rounded too)
    cy0 = max(0, (int) floorf(fy0) - 1);
    cx1 = min(g.nx - 1, (int) floorf(fx1) + 1);
    cy1 = min(g.ny - 1, (int) floorf(fy1) + 1);
    return true;
}

__global__ void __launch_bounds__(kBlock) assoc_count_kernel(AssocGridArgs A, int fill) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= A.nf) return;
    const AssocGeom g = *A.geom;
    const LmkBox bx = A.box[j];
    // (the radial bound of this call's gates for landmark j rides in its entries, once per call instead of once per (particle,
    // observation, entry): a gate needs |d - r| = |v0| < sqrt(G S00) <= sqrt(G (t + R00)): nis >= v0^2 / S00, the range row of Hf has norm 1)
    int cx0, cx1, cy0, cy1;
    if (!assoc_cells(bx, g, assoc_radius(bx, g, A.r00, A.r11, A.G), cx0, cx1, cy0, cy1)) return;
    for (int cy = cy0; cy <= cy1; cy++)
        for (int cx = cx0; cx <= cx1; cx++) {
            const int cell = cy * g.nx + cx;
            if (!fill) {
                atomicAdd(&A.cell_start[cell + 1], 1);
            } else {
                const int at = A.cell_start[cell] + atomicAdd(&A.cell_fill[cell], 1);
                if (at < A.cap_items) {
                    A.items[2 * (size_t) at] = make_float4(bx.xmin, bx.xmax, bx.ymin, bx.ymax);
                    A.items[2 * (size_t) at + 1] = make_float4(1.01f * sqrtf(A.G * (bx.tmax + A.r00)) + 1e-3f, __int_as_float(j), bx.tmax,
                                                               1.01f * sqrtf(A.G1 * (bx.tmax + A.r00)) + 1e-3f);
                }
            }
        }
}

// one block: cell populations -> exclusive prefix (cell_start[c + 1] held the population of cell c)
__global__ void __launch_bounds__(kBlock) assoc_scan_kernel(AssocGridArgs A) {
    __shared__ int sh[kBlock / kWave];
    __shared__ int carry;
    const AssocGeom g = *A.geom;
    const int nc = g.nx * g.ny;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nc; c0 += kBlock) {
        const int c = c0 + (int) threadIdx.x;
        const int v = c < nc ? A.cell_start[c + 1] : 0;
        // inclusive scan of v over the block
        int s = v;
        const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int o = __shfl_up(s, d, kWave);
            if (lane >= d) s += o;
        }
        if (lane == kWave - 1) sh[wv] = s;
        __syncthreads();
        int base = carry;
        for (int k = 0; k < wv; k++) base += sh[k];
        const int incl = base + s;
        __syncthreads();
        if (c < nc) A.cell_start[c + 1] = incl;
        if (threadIdx.x == kBlock - 1) carry = incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        A.cell_start[0] = 0;
        A.geom->total = carry;
        A.geom->overflow = carry > A.cap_items ? 1 : 0;
    }
}

// Candidate lists per observation (round 6), in place of the grid when there are few enough observations.  A grid cell serves every
// observation whose point falls into it, so it holds every landmark within (cell + 2 radii) -- ~33 entries on the 10 000-landmark map
// at MAX_RANGE 60, where the bearing gate alone is 5 m wide at 60 m -- and each (particle, observation) pair walks all of them to
// evaluate ~3.  One block per observation knows the observation: the landmarks it keeps are those whose box lies within the cell
// bound (assoc_radius) of the box of the points the observation implies over ALL particle poses (pose box + range x the arc of the
// set's headings) AND within the radial ring of the range -- the entry's own pre-test, taken over the pose box instead of one pose.
// Both are necessary for a gate to pass for any particle, so the labels stay those of the exhaustive scan.
SLAM_DEV void arc_range(float a0, float a1, bool cosine, float &lo, float &hi) {
    // min / max of cos (or sin) over [a0, a1], a1 >= a0
    const float sh = cosine ? 0.0f : 1.57079632679489662f;  // sin x = cos(x - pi / 2)
    const float b0 = a0 - sh, b1 = a1 - sh;
    const float c0 = cosf(b0), c1 = cosf(b1);
    lo = fminf(c0, c1);
    hi = fmaxf(c0, c1);
    if (b1 - b0 >= 6.28318530717958648f) {
        lo = -1.0f;
        hi = 1.0f;
        return;
    }
    // a maximum (angle = 2 pi k) or a minimum (angle = pi + 2 pi k) inside the interval?
    const float inv = 0.159154943091895336f;
    if (floorf(b1 * inv) > floorf(b0 * inv) || b0 * inv == floorf(b0 * inv)) hi = 1.0f;
    if (floorf((b1 - 3.14159265358979324f) * inv) > floorf((b0 - 3.14159265358979324f) * inv)) lo = -1.0f;
}
__global__ void __launch_bounds__(kBlock) assoc_lists_kernel(AssocGridArgs A, const int32_t *__restrict__ erow, const int32_t *__restrict__ live) {
    __shared__ int32_t sh_n;
    const int q = blockIdx.x;
    const AssocGeom g = *A.geom;
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    const float zr = A.z[2 * q], zb = A.z[2 * q + 1];
    // the points this observation implies: pose + zr (cos, sin)(theta + zb), theta in th_ref + [dth0, dth1] (a little wider: rounding)
    const float a0 = g.th_ref + g.dth0 + zb - 1e-3f, a1 = g.th_ref + g.dth1 + zb + 1e-3f;
    float cl, ch, sl, shh;
    arc_range(a0, a1, true, cl, ch);
    arc_range(a0, a1, false, sl, shh);
    const float slack = 1e-3f * (1.0f + fabsf(zr)) + 1e-4f * (fabsf(g.px0) + fabsf(g.px1) + fabsf(g.py0) + fabsf(g.py1));
    const float qx0 = g.px0 + fminf(zr * cl, zr * ch) - slack, qx1 = g.px1 + fmaxf(zr * cl, zr * ch) + slack;
    const float qy0 = g.py0 + fminf(zr * sl, zr * shh) - slack, qy1 = g.py1 + fmaxf(zr * sl, zr * shh) + slack;
    for (int j = threadIdx.x; j < A.nf; j += kBlock) {
        const LmkBox bx = A.box[j];
        if (!(bx.xmin <= bx.xmax)) continue;  // (no estimate / retired: the empty box)
        const float rho = assoc_radius(bx, g, A.r00, A.r11, A.G);
        const float gx = fmaxf(fmaxf(bx.xmin - qx1, qx0 - bx.xmax), 0.0f), gy = fmaxf(fmaxf(bx.ymin - qy1, qy0 - bx.ymax), 0.0f);
        if (gx * gx + gy * gy > rho * rho) continue;
        // the radial ring over the pose box: some pose's distance to some estimate in the box must be within e of the range
        const float e = 1.01f * sqrtf(A.G * (bx.tmax + A.r00)) + 1e-3f;
        const float ex = fmaxf(fmaxf(bx.xmin - g.px1, g.px0 - bx.xmax), 0.0f), ey = fmaxf(fmaxf(bx.ymin - g.py1, g.py0 - bx.ymax), 0.0f);
        const float fx = fmaxf(fabsf(bx.xmax - g.px0), fabsf(g.px1 - bx.xmin)), fy = fmaxf(fabsf(bx.ymax - g.py0), fabsf(g.py1 - bx.ymin));
        const float dmin2 = ex * ex + ey * ey, dmax2 = fx * fx + fy * fy;
        const float hi = zr + e, lo = zr - e;
        if ((hi < 0.0f || hi * hi < dmin2 * 0.998f) || (lo > 0.0f && lo * lo > dmax2 * 1.002f)) continue;
        const int at = atomicAdd(&sh_n, 1);
        if (at < A.lcap) {
            const size_t w = 2 * ((size_t) q * A.lcap + at);
            A.items[w] = make_float4(bx.xmin, bx.xmax, bx.ymin, bx.ymax);
            // (fourth word: where the landmark's records are found -- genealogy row | live buffer << 30 -- so that the walk need not look them up)
            A.items[w + 1] = make_float4(e, __int_as_float(j), bx.tmax, __int_as_float(erow[j] | (live[j] ? kRowLiveBit : 0)));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        A.cell_start[q] = min(sh_n, A.lcap);
        atomicAdd(&A.geom->total, min(sh_n, A.lcap));
        if (sh_n > A.lcap) atomicOr(&A.geom->overflow, 1);  // (the caller takes the grid)
    }
}


// one vote into the table of an observation: open addressing, linear probing; returns false if the table is full
SLAM_DEV bool vote_add(VoteSlot *tab, int label, float w) {
    unsigned h = ((unsigned) label * 2654435761u) >> 27;  // 5 bits
    for (int p = 0; p < kVoteSlots; p++) {
        VoteSlot *sl = tab + ((h + p) & (kVoteSlots - 1));
        int k = __hip_atomic_load(&sl->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == kVoteEmpty) {
            int expected = kVoteEmpty;
            if (__hip_atomic_compare_exchange_strong(&sl->key, &expected, label, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) k = label;
            else k = expected;
        }
        if (k == label) {
            atomicAdd(&sl->w, w);
            return true;
        }
    }
    return false;
}

template <bool LISTS>
__global__ void __launch_bounds__(kBlock) associate_grid_kernel(Buffers B, AssocGridArgs A, float r00, float r01, float r10, float r11, float gate1,
                                                                 float gate2, int32_t *__restrict__ labels) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const AssocGeom g = *A.geom;
    if (g.overflow & 1) return;  // (the entry buffer was too small: the caller runs the exhaustive scan)
    unsigned long long pairs = 0;
    const bool on = i < B.n;
    const int cur = B.ctrl->live[B.slot];
    const size_t S = (size_t) B.ncap;
    const float4 pa = on ? B.poseA[cur][i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float wi = on ? (A.logw ? expf(pa.w) : pa.w) : 0.0f;
    const int lane = threadIdx.x & (kWave - 1);
    bool full = false;
    // blockIdx.y = a group of A.obs_per_block observations: a particle's observations are independent of each other, and one
    // thread walking all of them alone (1 300 on the 10 000-landmark map, ~65 dependent record reads each) left the
    // machine two thirds empty and every read exposed: 225 ms per call at 10^5 particles
    const int q_lo = blockIdx.y * A.obs_per_block, q_hi = min(A.nz, q_lo + A.obs_per_block);
    for (int q = q_lo; q < q_hi; q++) {
        int label = kVoteEmpty;
        [[maybe_unused]] int pbest = -1;  // (LISTS: position of jbest in the observation's list: where its vote goes)
        if (on) {
            // (uniform and written before the launch: scalar loads)
            const auto *zc = (const __attribute__((address_space(4))) float *) reinterpret_cast<uintptr_t>(A.z);
            const float zr = zc[2 * q], zb = zc[2 * q + 1];
            float sn, cs;
#ifdef SLAM_FAST_MATH
            sincos_cw(pa.z + zb, sn, cs);  // (a bounded angle; the cell the point falls in has a cell of slack on every side)
#else
            sincosf(pa.z + zb, &sn, &cs);
#endif
            const float px = pa.x + zr * cs, py = pa.y + zr * sn;
            int cell = 0;
            if constexpr (!LISTS) {
                const int cx = min(max((int) floorf((px - g.x0) * g.inv_cs), 0), g.nx - 1);
                const int cy = min(max((int) floorf((py - g.y0) * g.inv_cs), 0), g.ny - 1);
                cell = cy * g.nx + cx;
            }
            float nbest = INFINITY, outer = INFINITY;
            int jbest = -1;
            [[maybe_unused]] int pcur = 0;  // (LISTS: position of the entry being visited in the observation's list)
            // one entry of the cell: the radial pre-test on the landmark's box alone (every estimate of j lies in the box, so its
            // distance d from this pose is within [dmin, dmax] of the box; a gate needs |d - r| < the entry's bound), on SQUARED
            // distances (round 6: no square root per entry, and the bound without the factor (1 + pi / 2) the CELL radius needs and
            // this test never did -- |d - r| is |v0| exactly: 2.6 times fewer entries reach the gates), then the gates
            // Gb / e: the gate bound of this pass of the walk and the entry's radial bound for it (see the passes below)
            auto visit = [&](const float4 bb, const float4 bt, const float Gb, const float e) {
                const float ex = fmaxf(fmaxf(bb.x - pa.x, pa.x - bb.y), 0.0f), ey = fmaxf(fmaxf(bb.z - pa.y, pa.y - bb.w), 0.0f);
                const float fx = fmaxf(fabsf(bb.x - pa.x), fabsf(bb.y - pa.x)), fy = fmaxf(fabsf(bb.z - pa.y), fabsf(bb.w - pa.y));
                const float dmin2 = ex * ex + ey * ey, dmax2 = fx * fx + fy * fy;
                const float hi = zr + e, lo = zr - e;
                if ((hi < 0.0f || hi * hi < dmin2 * 0.998f) || (lo > 0.0f && lo * lo > dmax2 * 1.002f)) return;
                {
                    // ... and, for the entries the ring lets through, the same bound as the cell radius (assoc_radius) with THIS pose's
                    // distances instead of the worst over all poses: an estimate l of j that passes a gate lies within
                    //   e + sqrt(G (t + d^2 R11)) + (e / 2) min(pi, sqrt(G (t / d^2 + R11)))
                    // of the point p this observation implies (d in [dmin, dmax] of the box from this pose), and l is in the box: the
                    // distance from p to the box must not exceed it.  Cuts the entries whose record is fetched (two dependent trips
                    // each: what the kernel waits for) by the bearing as well as the range
                    const float t = bt.z;
                    const float Dg = dmin2 > 0.0f ? fminf(3.14159274f, sqrtf(Gb * (t * __builtin_amdgcn_rcpf(dmin2) * 1.001f + r11))) : 3.14159274f;
                    const float rho = (e * (1.0f + 0.5f * Dg) + sqrtf(Gb * (t + dmax2 * r11))) * 1.01f + 1e-3f;
                    const float qx = fmaxf(fmaxf(bb.x - px, px - bb.y), 0.0f), qy = fmaxf(fmaxf(bb.z - py, py - bb.w), 0.0f);
                    if (qx * qx + qy * qy > rho * rho) return;
                }
                const int j = __float_as_int(bt.y);
                float4 la;
                float lb;
                if constexpr (LISTS) {
                    // (single contexts without arrivals: the slot is this GPU's; row and live buffer ride in the entry)
                    const int rw = __float_as_int(bt.w);
                    const int sl = B.gen[cur][gen_index(B.compact, S, rw & kRowMask, (size_t) i)];
                    const size_t at = (size_t) j * S + (size_t) sl;
                    la = B.lmkA[(rw >> 30) & 1][at];
                    lb = B.lmkB[(rw >> 30) & 1][at];
                } else {
                    read_through_genealogy(B, B.lmk_live, cur, S, j, i, la, lb);
                }
                const AssocLm L = assoc_landmark(pa, la, lb, r00, r01, r10, r11);
                float nis, nd;
                assoc_gate(L, zr, zb, nis, nd);
                pairs++;
                // (the cell's landmarks come in no particular order: ties go to the lower index, as in the ascending scan)
                if (nis < gate1 && (nd < nbest || (nd == nbest && j < jbest))) {
                    nbest = nd;
                    jbest = j;
                    if constexpr (LISTS) pbest = pcur;
                } else if (nis < outer) {
                    outer = nis;
                }
            };
            // the walk: entries are self-contained (box + bound + id: ONE contiguous 32-byte read each, where an id and the box behind
            // it were two dependent trips), two in flight at a time; SQ counters before: 87 % of the wave cycles waiting, the SIMDs
            // a third busy (profiles/gated_association_r06.txt)
            // (LISTS: the observation's own candidate list, assoc_lists_kernel)
            // (LISTS: the list is the same for every lane -- q is the block's -- and was written before this launch: read it through the constant
            // address space, as SCALAR loads that sit outside the vector memory pipeline the record fetches use)
            using ItemP = std::conditional_t<LISTS, const __attribute__((address_space(4))) float *, const float *>;
            using CntP = std::conditional_t<LISTS, const __attribute__((address_space(4))) int32_t *, const int32_t *>;
            const ItemP items_f = (ItemP) reinterpret_cast<uintptr_t>(A.items);
            auto item_at = [&](size_t w) -> float4 { return make_float4(items_f[4 * w], items_f[4 * w + 1], items_f[4 * w + 2], items_f[4 * w + 3]); };
            const CntP counts = (CntP) reinterpret_cast<uintptr_t>(A.cell_start);
            const int c0 = LISTS ? q * A.lcap : counts[cell], c1 = LISTS ? c0 + counts[q] : counts[cell + 1];
            // Two passes.  Whatever an observation is MATCHED with passes gate_reject, so a first walk bounded by gate_reject alone (G1:
            // radii ~ sqrt(G1 / G) of the full ones) sees every landmark that can become jbest -- same candidates, same ties: the same
            // label.  The wider bound of gate_augment is only needed to tell "new" from "discard" for an observation NOTHING matched: a
            // second walk for those pairs alone (`outer` then runs over every candidate, as in a single pass).  With the grid this did
            // not pay (the walk over a cell's ~33 entries was the cost: 8.52 against 8.32 ms at config 5); with the lists' ~3 entries
            // the records fetched through the genealogy are the cost again, and the first pass fetches 40 % fewer.
            for (int at = c0; at < c1; at += 2) {
                const float4 b0 = item_at(2 * (size_t) at), t0 = item_at(2 * (size_t) at + 1);
                const int a1 = min(at + 1, c1 - 1);
                const float4 b1 = item_at(2 * (size_t) a1), t1 = item_at(2 * (size_t) a1 + 1);
                // (first-pass radial bound: the grid's entries carry it; a list's is derived from the bound for G -- 1 % of slack covers the rounding)
                pcur = at - c0;
                visit(b0, t0, A.G1, LISTS ? (t0.x - 1e-3f) * A.g1_ratio + 1.1e-3f : t0.w);
                pcur = at + 1 - c0;
                if (at + 1 < c1) visit(b1, t1, A.G1, LISTS ? (t1.x - 1e-3f) * A.g1_ratio + 1.1e-3f : t1.w);
            }
            if (jbest < 0 && A.G1 < A.G) {
                outer = INFINITY;
                for (int at = c0; at < c1; at += 2) {
                    const float4 b0 = item_at(2 * (size_t) at), t0 = item_at(2 * (size_t) at + 1);
                    const int a1 = min(at + 1, c1 - 1);
                    const float4 b1 = item_at(2 * (size_t) a1), t1 = item_at(2 * (size_t) a1 + 1);
                    pcur = at - c0;
                    visit(b0, t0, A.G, t0.x);
                    pcur = at + 1 - c0;
                    if (at + 1 < c1) visit(b1, t1, A.G, t1.x);
                }
            }
            label = jbest > -1 ? jbest : (outer > gate2 ? kAssocNew : kAssocDiscard);
            if (labels) labels[A.lab_by_obs ? (size_t) q * S + i : (size_t) i * A.nz + q] = label;
        }
        if (A.census_first) {
            // (slamgpu_update_particle: a wave's particles nearly always agree: one atomic per wave and distinct label, nobody waits for it)
            unsigned long long todo = __ballot(on && label >= 0);
            while (todo) {
                const int src = __ffsll((long long) todo) - 1;
                const int lab0 = __builtin_amdgcn_readlane(label, src);
                if (lane == src) atomicMin(A.census_first + lab0, q);
                todo &= ~__ballot(label == lab0);
            }
            const unsigned long long nw = __ballot(on && label == kAssocNew);
            if (nw && lane == (int) __ffsll((long long) nw) - 1) atomicAdd(A.census_news + q, (int) __popcll(nw));
        }
        bool direct = false;
        if constexpr (LISTS) direct = A.vote_w != nullptr;
        if (direct) {
            // (candidate lists: the label's place in the observation's table is known -- no look-up, nothing to wait for)
            const int place = label >= 0 ? 2 + pbest : (label == kAssocNew ? 0 : 1);
            unsigned long long todo = __ballot(on);
            while (todo) {
                const int src = __ffsll((long long) todo) - 1;
                const int p0 = __builtin_amdgcn_readlane(place, src);
                const bool mine = on && place == p0;
                const float ws = wave_sum_f(mine ? wi : 0.0f);
                if (lane == src) atomicAdd(A.vote_w + (size_t) q * (A.lcap + 2) + p0, ws);
                todo &= ~__ballot(mine);
            }
        } else if (A.votes) {
            // a wave's particles nearly always agree: one atomic per wave and distinct label, not one per particle
            unsigned long long todo = __ballot(on);
            while (todo) {
                const int src = __ffsll((long long) todo) - 1;
                const int lab0 = __builtin_amdgcn_readlane(label, src);
                const bool mine = on && label == lab0;
                const float ws = wave_sum_f(mine ? wi : 0.0f);
                if (lane == src && !vote_add(A.votes + (size_t) q * kVoteSlots, lab0, ws)) full = true;
                todo &= ~__ballot(mine);
            }
        }
    }
    if (__ballot(full) && lane == 0) atomicOr(&A.geom->overflow, 2);
    pairs = (unsigned long long) wave_sum_d((double) pairs);
    if (lane == 0 && pairs) atomicAdd(&A.geom->pairs, pairs);
}

// ---------------------------------------------------------------------------------------------------
// Sharded resampling (particles partitioned over contexts / GPUs in contiguous blocks of 256).  The host
// layer all-gathers the per-block totals (4 B + 4 B per 256 particles); every shard then runs the same
// scan, so the decision and every ancestor are independent of the number of shards.
//   shard_plan   : W, Q, Neff, decision and K[r] = first output particle whose ancestor lives on shard r
//   shard_pack   : offspring k in [K[g], K[g+1]) of this shard, gathered into per-destination blocks
//                  [dst][field][slot] (field-major => coalesced on both sides), fields = 10 + 5*nf floats
//   shard_unpack : blocks received from each source shard scattered into the spare buffers, w = 1/N
//   shard_normalize : no resample: w /= W
// ---------------------------------------------------------------------------------------------------
// `out` may be pinned host memory (the host polls `seq_out` instead of synchronising the stream): everything is
// written, fenced at system scope, and only then the sequence number.
__global__ void __launch_bounds__(kBlock) shard_plan_kernel(ShardPlanArgs A, RngArgs rng, ShardPlan *out,
                                                             volatile uint32_t *seq_out, uint32_t seq) {
    extern __shared__ double off[];
    __shared__ double sh_a[kBlock / kWave], sh_q[kBlock / kWave];
    double W, Q;
    double Mx;
    scan_block_totals(A.gblk, A.nb_global, A.nb_per_shard, false, off, sh_a, sh_q, W, Q, Mx);  // shards: linear weights only
    __syncthreads();
    const int t = threadIdx.x;
    if (t == 0) {
        const float neff = neff_of(W, Q);
        out->wsum = W;
        out->wsq = Q;
        out->neff = neff;
        out->resampled = (A.do_resample && (neff < (float) A.n_effective)) ? 1 : 0;
        out->status = weight_status(W, Q);
        out->pad = 0;
    }
    // K[r] = #{ k : stratum_k * W < C_r },  C_r = off[r * nb_per_shard]; strata are increasing in k
    for (int r = t; r <= A.n_shards; r += kBlock) {
        int64_t lo = 0, hi = rng.n_global;
        if (r == 0) {
            hi = 0;
        } else if (r == A.n_shards) {
            lo = hi;
        } else {
            const double C = off[min(r * A.nb_per_shard, A.nb_global)];
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if ((double) stratum(rng, mid) * W < C) lo = mid + 1; else hi = mid;
            }
        }
        out->K[r] = lo;
    }
    if (seq_out) {
        __threadfence_system();
        __syncthreads();
        if (t == 0) {
            __threadfence_system();
            *seq_out = seq;
        }
    }
}

// landmark l of the particle in slot `anc` of genealogy buffer `cur`: through the genealogy, the row's live flag, or the
// arrival pool
SLAM_DEV void read_through_genealogy(const Buffers &B, const int32_t *__restrict__ live, int cur, size_t S, int l, int anc,
                                     float4 &la, float &lb) {
    int sl = B.gen[cur][gen_index(B.compact, S, B.erow[l], (size_t) anc)];
    if (B.n_shards > 1) {  // distributed context: global slot id, possibly another GPU's record
        const int h = (int) __umul64hi((unsigned long long) (unsigned) sl, B.div_n);
        const int b = live[l];
        const size_t at = (size_t) l * S + (size_t) (sl - h * B.ncap);
        la = (h == B.shard ? B.lmkA[b] : B.peers[h].lmkA[b])[at];
        lb = (h == B.shard ? B.lmkB[b] : B.peers[h].lmkB[b])[at];
        return;
    }
    if (sl < 0) {
        const size_t at = (size_t) l * B.pool_cap + (sl & ~kPoolBit);
        la = B.poolA[at];
        lb = B.poolB[at];
    } else {
        const int b = live[l];
        la = B.lmkA[b][(size_t) l * S + sl];
        lb = B.lmkB[b][(size_t) l * S + sl];
    }
}

__global__ void __launch_bounds__(kBlock) shard_pack_kernel(Buffers B, WeightScratch ws, ShardPackArgs A, RngArgs rng) {
    extern __shared__ double off[];
    __shared__ double sh_a[kBlock / kWave], sh_q[kBlock / kWave];
    double W, Q;
    double Mx;
    scan_block_totals(A.gblk, A.nb_global, A.nb_per_shard, false, off, sh_a, sh_q, W, Q, Mx);  // shards: linear weights only
    __syncthreads();
    const int64_t j = (int64_t) blockIdx.x * kBlock + threadIdx.x;  // offspring slot of this shard
    const int64_t k = A.k_lo + j;
    if (k >= A.k_hi) return;
    const int cur = B.ctrl->live[B.slot] ^ (B.ctrl->pend[B.slot] ? 1 : 0);  // the buffers this step's update wrote
    const size_t S = (size_t) B.ncap;
    const double target = (double) stratum(rng, k) * W;
    const int64_t ganc = find_ancestor(target, off, A.nb_global, ws.lcum[ws.wpar], A.first_block, ws.nblocks, rng.n_global);
    const int anc = (int) min(max(ganc - rng.first_particle, (int64_t) 0), (int64_t) B.n - 1);
    // destination block: d = k / n_per_shard ; slot within it = k - max(K_lo, d*n_per_shard)
    const int d = (int) (k / A.n_per_shard);
    const int64_t blk_lo = max(A.k_lo, (int64_t) d * A.n_per_shard);
    const int64_t blk_hi = min(A.k_hi, (int64_t) (d + 1) * A.n_per_shard);
    const int64_t cnt = blk_hi - blk_lo, slot = k - blk_lo;
    if (d == A.shard) {
        // the output slot lives on this shard too: nothing moves now -- record the (local) ancestor; the next update
        // launch gathers while it computes (lazy gather, as in the single-context pipeline).  With balanced weights
        // this is almost every offspring.
        ws.keep[B.slot ^ 1][(int) (k - (int64_t) d * A.n_per_shard)] = anc;
        return;
    }
    // records before this block in the send buffer = offspring before it, minus the ones kept local
    const int64_t self_lo = max(A.k_lo, (int64_t) A.shard * A.n_per_shard);
    const int64_t self_hi = min(A.k_hi, (int64_t) (A.shard + 1) * A.n_per_shard);
    const int64_t kept_before = (d > A.shard && self_hi > self_lo) ? (self_hi - self_lo) : 0;
    float *__restrict__ dst = A.send + (size_t) (blk_lo - A.k_lo - kept_before) * A.fields + slot;
    {
        const float4 pa = B.poseA[cur][anc], pb = B.poseB[cur][anc];
        const float2 pc = B.poseC[cur][anc];
        dst[0 * cnt] = pa.x; dst[1 * cnt] = pa.y; dst[2 * cnt] = pa.z;
        dst[3 * cnt] = pb.x; dst[4 * cnt] = pb.y; dst[5 * cnt] = pb.z; dst[6 * cnt] = pb.w;
        dst[7 * cnt] = pc.x; dst[8 * cnt] = pc.y;
        dst[9 * cnt] = __int_as_float((int) ganc);  // ancestor id (keep[]); the weight is reset to 1/N on arrival
    }
    // the few offspring that leave the shard carry all their landmarks (one thread each: they are a fraction of a
    // percent of the particles; a second grid dimension would make every block redo the scan above for nothing)
    const int32_t *__restrict__ live = B.lmk_live;
    for (int l = 0; l < A.nf; l++) {
        float4 la;
        float lb;
        read_through_genealogy(B, live, cur, S, l, anc, la, lb);
        float *f = dst + (size_t) (10 + 5 * l) * cnt;
        f[0] = la.x; f[cnt] = la.y; f[2 * cnt] = la.z; f[3 * cnt] = la.w; f[4 * cnt] = lb;
    }
}

// Records that arrive from other shards need a place no sibling shares.
//   A.pool_base >= 0 (normal): arrival number a of this step gets arrival-pool slot pool_base + a: its landmark records
//     go there, its pose and a genealogy that points at the pool go to slot i of the buffers the NEXT update launch writes
//     (which reads them in place: keep[i] < 0); local offspring stay a lazy gather.  Only arrivals do any work.
//   A.pool_base < 0 (pool full): settle the whole shard: every output particle is written physically into the other pose /
//     genealogy buffers and into the other buffer of every landmark row -- local offspring from their ancestor (keep[],
//     through the ancestor's genealogy), arrivals from the receive buffer -- with identity genealogy, and every row's
//     live flag flips (in the host's table).  Nothing references the pool afterwards.
// blockIdx.y = group of 8 landmarks.
__global__ void __launch_bounds__(kBlock) shard_unpack_kernel(Buffers B, WeightScratch ws, ShardUnpackArgs A) {
    const bool settle = A.pool_base < 0;
    const int32_t *__restrict__ live = B.lmk_live;
    // local output particle: every one when settling, only the arrivals ([0, own_lo) and [own_hi, n)) otherwise
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (!settle && i >= A.own_lo) i += A.own_hi - A.own_lo;
    if (i >= B.n) return;
    const int cur = B.ctrl->live[B.slot] ^ (B.ctrl->pend[B.slot] ? 1 : 0);  // the buffers this step's update wrote
    const size_t S = (size_t) B.ncap;
    // source block: the s with src_lo[s] <= i < src_lo[s+1] (local output index boundaries, increasing)
    int s = 0;
    while (s + 1 < A.n_shards && i >= A.src_lo[s + 1]) s++;
    const int j0 = blockIdx.y * kLmkPerBlockY, j1 = min(A.nf, j0 + kLmkPerBlockY);
    if (s == A.shard) {
        if (!settle) return;  // lazy gather through keep[] (recorded by this shard's own pack kernel)
        const int anc = ws.keep[B.slot ^ 1][i];
        if (blockIdx.y == 0) {
            B.gen[cur ^ 1][gen_index(B.compact, S, 0, (size_t) i)] = i;  // settled: every landmark in genealogy row 0, own slot
            float4 pa = B.poseA[cur][anc];
            pa.w = B.ctrl->inv_n;
            B.poseA[cur ^ 1][i] = pa;
            B.poseB[cur ^ 1][i] = B.poseB[cur][anc];
            B.poseC[cur ^ 1][i] = B.poseC[cur][anc];
        }
        for (int l = j0; l < j1; l++) {
            float4 la;
            float lb;
            read_through_genealogy(B, live, cur, S, l, anc, la, lb);
            const int b = live[l];
            B.lmkA[b ^ 1][(size_t) l * S + i] = la;
            B.lmkB[b ^ 1][(size_t) l * S + i] = lb;
        }
        return;
    }
    const int64_t cnt = A.src_lo[s + 1] - A.src_lo[s], slot = i - A.src_lo[s];
    // records before this block in the receive buffer = outputs before it, minus the locally produced ones
    const int64_t own = A.src_lo[A.shard + 1] - A.src_lo[A.shard];
    const int64_t before = A.src_lo[s] - (s > A.shard ? own : 0);
    const float *__restrict__ src = A.recv + (size_t) before * A.fields + slot;
    if (blockIdx.y == 0) {
        B.poseA[cur ^ 1][i] = make_float4(src[0], src[cnt], src[2 * cnt], B.ctrl->inv_n);
        B.poseB[cur ^ 1][i] = make_float4(src[3 * cnt], src[4 * cnt], src[5 * cnt], src[6 * cnt]);
        B.poseC[cur ^ 1][i] = make_float2(src[7 * cnt], src[8 * cnt]);
        // negative = "came from another shard"; the global ancestor id is -(keep + 1)
        ws.keep[B.slot ^ 1][i] = -(__float_as_int(src[9 * cnt]) + 1);
    }
    if (settle) {
        if (blockIdx.y == 0) B.gen[cur ^ 1][gen_index(B.compact, S, 0, (size_t) i)] = i;  // row 0, own slot
        for (int l = j0; l < j1; l++) {
            const float *f = src + (size_t) (10 + 5 * l) * cnt;
            const int b = live[l];
            B.lmkA[b ^ 1][(size_t) l * S + i] = make_float4(f[0], f[cnt], f[2 * cnt], f[3 * cnt]);
            B.lmkB[b ^ 1][(size_t) l * S + i] = f[4 * cnt];
        }
    } else {
        const int p = A.pool_base + (int) (before + slot);  // arrival number within this step
        const int ref = kPoolBit | p;
        // every live genealogy row of this particle points at its pool slot (rows spread over the blockIdx.y groups)
        for (int r = blockIdx.y; r < B.n_rows; r += gridDim.y) B.gen[cur ^ 1][gen_index(B.compact, S, B.rows[r], (size_t) i)] = ref;
        for (int l = j0; l < j1; l++) {
            const float *f = src + (size_t) (10 + 5 * l) * cnt;
            const size_t at = (size_t) l * B.pool_cap + p;
            B.poolA[at] = make_float4(f[0], f[cnt], f[2 * cnt], f[3 * cnt]);
            B.poolB[at] = f[4 * cnt];
        }
    }
}

// Last stage of a sharded update: normalise (no resample), this shard's pose-estimate partials, and the outcome in
// Ctrl.  mode 0: no resample, the set stays in the buffers this step's update wrote.  mode 1: resampled, nothing
// arrived from other shards: the set is defined through keep[] until the next update launch (or gather_kernel) moves
// it (lazy gather).  mode 2: resampled and settled by shard_unpack_kernel into the other buffers.
__global__ void __launch_bounds__(kBlock) shard_finalize_kernel(Buffers B, WeightScratch ws, double W, double Q, float neff,
                                                                 int mode) {
    const int resampled = mode != 0;
    __shared__ EstItem sh_est[kBlock / kWave];
    Ctrl *c = B.ctrl;
    const int cur = c->live[B.slot] ^ (c->pend[B.slot] ? 1 : 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        c->wsum = W;
        c->wsq = Q;
        c->neff = neff;
        c->resampled = resampled;
        c->status = weight_status(W, Q);
        ws.est_part[ws.wpar][4 * (size_t) ws.nblocks] = (double) neff;
        ws.est_part[ws.wpar][4 * (size_t) ws.nblocks + 1] = (double) (resampled | (weight_status(W, Q) << 1));
        // published in the other slot; the host flips its slot after this launch (see Ctrl)
        c->live[B.slot ^ 1] = mode == 2 ? cur ^ 1 : cur;
        c->pend[B.slot ^ 1] = mode == 1 ? 1 : 0;
    }
    const int i = blockIdx.x * kBlock + threadIdx.x;
    EstItem ei{0.0, 0.0, -3.0e38f, 0.0f, 0x7fffffff};
    if (i < B.n) {
        if (!resampled) {
            float4 pa = B.poseA[cur][i];
            pa.w = pa.w / (float) W;
            B.poseA[cur][i] = pa;
            ei = EstItem{(double) pa.x, (double) pa.y, pa.w, pa.z, i};
        } else {
            const int k = mode == 2 ? -1 : ws.keep[B.slot ^ 1][i];  // < 0: in place in the other buffers (settled / arrived)
            const float4 pa = k < 0 ? B.poseA[cur ^ 1][i] : B.poseA[cur][k];
            ei = EstItem{(double) pa.x, (double) pa.y, c->inv_n, pa.z, i};
        }
    }
    ei = block_reduce_est(ei, sh_est);
    if (threadIdx.x == 0) {
        double *p = ws.est_part[ws.wpar] + (size_t) blockIdx.x * 4;
        p[0] = ei.sx;
        p[1] = ei.sy;
        p[2] = (double) ei.th;
        p[3] = (double) ei.w;
    }
}

// ---------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------
static void launch_shard_plan(hipStream_t st, const ShardPlanArgs &A, const RngArgs &rng, ShardPlan *out, uint32_t *seq_out,
                              uint32_t seq) {
    const size_t lds = sizeof(double) * ((size_t) A.nb_global + 1);
    hipLaunchKernelGGL(shard_plan_kernel, dim3(1), dim3(kBlock), lds, st, A, rng, out, seq_out, seq);
}

static void launch_shard_pack(hipStream_t st, const Buffers &B, const WeightScratch &ws, const ShardPackArgs &A,
                              const RngArgs &rng) {
    const int64_t cnt = A.k_hi - A.k_lo;
    if (cnt <= 0) return;
    const size_t lds = sizeof(double) * ((size_t) A.nb_global + 1);
    hipLaunchKernelGGL(shard_pack_kernel, dim3((unsigned) ((cnt + kBlock - 1) / kBlock)), dim3(kBlock), lds, st, B, ws, A, rng);
}

static void launch_shard_unpack(hipStream_t st, const Buffers &B, const WeightScratch &ws, const ShardUnpackArgs &A) {
    const int gy = A.nf > 0 ? (A.nf + kLmkPerBlockY - 1) / kLmkPerBlockY : 1;
    const int work = A.pool_base < 0 ? B.n : B.n - (A.own_hi - A.own_lo);  // everything when settling, else the arrivals
    if (work <= 0) return;
    hipLaunchKernelGGL(shard_unpack_kernel, dim3((work + kBlock - 1) / kBlock, gy), dim3(kBlock), 0, st, B, ws, A);
}

static void launch_shard_finish(hipStream_t st, const Buffers &B, const WeightScratch &ws, double W, double Q, float neff,
                                int resampled) {
    hipLaunchKernelGGL(shard_finalize_kernel, dim3(B.ncap / kBlock), dim3(kBlock), 0, st, B, ws, W, Q, neff, resampled);
}

static void launch_update_any(hipStream_t st, const Buffers &B, const PredictArgs &PA, const UpdateArgs &U,
                              const RngArgs &rng, const WeightScratch &ws, const PerParticle &ppa) {
    // compute blocks first (they are the long pole), then -- single-context pipeline only -- the copy blocks of a
    // pending lazy gather (they exit at once when nothing is pending: the host cannot know) and one helper block
    int grid = B.ncap / kBlock;
    if (U.lazy) grid += (U.copy_hi - U.copy_lo) + 1;
    // inline plan only: prefix of the previous step's block totals (launches that do not plan never touch off[])
    const size_t nbg = (size_t) ws.nblocks * (U.arrivals == 2 ? (size_t) B.n_shards : 1);  // distributed: blocks of all shards
    const size_t lds = (size_t) staging_slots(U.method, U.big != nullptr, U.m) * kBlock * (sizeof(float4) + sizeof(float)) +
                       ((U.plan_inline && !U.scan_global) ? sizeof(double) * ((nbg + 3) & ~(size_t) 1) : 0) +
                       (U.plan_inline ? update_window_bytes() : 0);
    const int sel = (U.method == 2 ? 6 : 0) + 2 * U.arrivals + (U.big ? 1 : 0);
    const float *h_tot = U.arrivals == 2 ? B.gtot[ws.wpar ^ 1] : ws.blk_w[ws.wpar ^ 1];
    // bit 5: distributed contexts: the gathered table is wider than two totals per thread and fits the LDS behind the block prefix
    // (ancestor windows + landmark staging): the scan fetches it by LDS-DMA (scan_issue_dma)
    const size_t stage_bytes = (size_t) staging_slots(U.method, U.big != nullptr, U.m) * kBlock * (sizeof(float4) + sizeof(float));
    const bool scan_dma = U.arrivals == 2 && U.plan_inline && !U.scan_global && !U.logw && nbg > 2 * (size_t) kBlock &&
                          2 * sizeof(float) * (((nbg + 255) / 256) * 256) <= update_window_bytes() + stage_bytes;
    const int h_flags = (U.plan_inline ? 1 : 0) | (U.scan_global ? 2 : 0) | (U.logw ? 4 : 0) | (U.lazy ? 8 : 0) | (U.front.on ? 16 : 0) | (scan_dma ? 32 : 0) | (U.count_remote ? 64 : 0);
#define SLAM_LAUNCH_UPDATE(M, A, G)                                                                                              \
    hipLaunchKernelGGL((update_kernel<M, A, G>), dim3(grid), dim3(kBlock), lds, st, h_tot, B.ctrl, U.front.state_in, ws.nblocks, B.slot, grid, \
                       h_flags, B, PA, U, rng, ws, ppa)
    if (update_is_wide(U.method, U.arrivals, U.big != nullptr, ws.nblocks)) {  // (FastSLAM 1, single context, compact layout, more tiles than two rounds of CUs)
        hipLaunchKernelGGL((update_kernel_wide<1>), dim3(grid), dim3(kBlock), lds, st, h_tot, B.ctrl, U.front.state_in, ws.nblocks, B.slot, grid,
                           h_flags, B, PA, U, rng, ws);
        return;
    }
    if (ppa.obs) {  // per-particle association (slamgpu.cpp: do_update_particle guarantees a single context on plain rows)
        // + the staged observation indices (update_step.inl: shJ) and, when they fit, the observations (shZ)
        PerParticle pq = ppa;
        pq.z_lds = pq.nz <= kPpLdsObs ? 1 : 0;
        const size_t lds_pp = lds + (size_t) kBigChunk * kBlock * sizeof(int32_t) + (pq.z_lds ? sizeof(float) * 2 * (size_t) pq.nz : 0);
        if (U.method == 2) hipLaunchKernelGGL((update_kernel<2, 0, true, true>), dim3(grid), dim3(kBlock), lds_pp, st, h_tot, B.ctrl, U.front.state_in, ws.nblocks, B.slot, grid, h_flags, B, PA, U, rng, ws, pq);
        else hipLaunchKernelGGL((update_kernel<1, 0, true, true>), dim3(grid), dim3(kBlock), lds_pp, st, h_tot, B.ctrl, U.front.state_in, ws.nblocks, B.slot, grid, h_flags, B, PA, U, rng, ws, pq);
        return;
    }
    switch (sel) {
        case 11: SLAM_LAUNCH_UPDATE(2, 2, true); break;
        case 10: SLAM_LAUNCH_UPDATE(2, 2, false); break;
        case 9: SLAM_LAUNCH_UPDATE(2, 1, true); break;
        case 8: SLAM_LAUNCH_UPDATE(2, 1, false); break;
        case 7: SLAM_LAUNCH_UPDATE(2, 0, true); break;
        case 6: SLAM_LAUNCH_UPDATE(2, 0, false); break;
        case 5: SLAM_LAUNCH_UPDATE(1, 2, true); break;
        case 4: SLAM_LAUNCH_UPDATE(1, 2, false); break;
        case 3: SLAM_LAUNCH_UPDATE(1, 1, true); break;
        case 2: SLAM_LAUNCH_UPDATE(1, 1, false); break;
        case 1: SLAM_LAUNCH_UPDATE(1, 0, true); break;
        default: SLAM_LAUNCH_UPDATE(1, 0, false); break;
    }
#undef SLAM_LAUNCH_UPDATE
}

static void launch_update(hipStream_t st, const Buffers &B, const PredictArgs &PA, const UpdateArgs &U, const RngArgs &rng, const WeightScratch &ws) {
    launch_update_any(st, B, PA, U, rng, ws, PerParticle{});
}
static void launch_update_particle(hipStream_t st, const Buffers &B, const PredictArgs &PA, const UpdateArgs &U, const RngArgs &rng,
                                   const WeightScratch &ws, const PerParticle &ppa) {
    launch_update_any(st, B, PA, U, rng, ws, ppa);
}

// K iterations in one launch (kernels.h: PersistArgs): kPersistStride x (tiles + 1 helper) workgroups, of which every
// kPersistStride-th stays; LDS as for a per-step launch that plans inline with the front end on
static void launch_update_persist(hipStream_t st, const Buffers &B, const PredictArgs &PA, const UpdateArgs &U, const RngArgs &rng,
                                  const WeightScratch &ws) {
    const int grid = kPersistStride * (ws.nblocks + 1 + U.persist.drawers);
    const size_t lds = (size_t) staging_slots(U.method, false, U.m) * kBlock * (sizeof(float4) + sizeof(float)) +
                       sizeof(double) * (((size_t) ws.nblocks + 3) & ~(size_t) 1) + update_window_bytes();
    const int h_flags = 1 | (U.logw ? 4 : 0) | 8 | 16;
    if (U.method == 2)
        hipLaunchKernelGGL((update_persist_kernel<2>), dim3(grid), dim3(kBlock), lds, st, ws.blk_w[ws.wpar ^ 1], B.ctrl, U.front.state_in, ws.nblocks, B.slot, grid,
                           h_flags, B, PA, U, rng, ws);
    else
        hipLaunchKernelGGL((update_persist_kernel<1>), dim3(grid), dim3(kBlock), lds, st, ws.blk_w[ws.wpar ^ 1], B.ctrl, U.front.state_in, ws.nblocks, B.slot, grid,
                           h_flags, B, PA, U, rng, ws);
}

static void launch_resample(hipStream_t st, const Buffers &B, const WeightScratch &ws, const RngArgs &rng,
                            const ResampleArgs &ra, const UpdateArgs &U) {
    const size_t lds = sizeof(double) * ((size_t) ws.nblocks + 1);
    hipLaunchKernelGGL(resample_kernel, dim3(ws.nblocks), dim3(kBlock), lds, st, B, ws, rng, ra, U);
}

static void launch_resample_ref(hipStream_t st, const Buffers &B, const WeightScratch &ws, const RngArgs &rng, const ResampleArgs &ra) {
    hipLaunchKernelGGL(resample_ref_kernel, dim3(1), dim3(kBlock), sizeof(float) * 2 * (size_t) B.n, st, B, ws, rng, ra);
}

static void launch_scan(hipStream_t st, const WeightScratch &ws, int logw) {
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(kBlock), 0, st, ws, logw, logw ? 0 : 1);
}

static void launch_gather(hipStream_t st, const Buffers &B, const WeightScratch &ws) {
    const int gy = (!B.compact && B.n_rows > 0) ? (B.n_rows + kRowsPerRole - 1) / kRowsPerRole : 1;
    hipLaunchKernelGGL(gather_kernel, dim3(ws.nblocks, gy), dim3(kBlock), 0, st, B, ws);
}

// identity ("own slot") in one genealogy row (upload, flatten)
__global__ void __launch_bounds__(kBlock) identity_kernel(int32_t *gen, int compact, int row, int ncap, int first) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k < ncap) gen[gen_index(compact, (size_t) ncap, row, (size_t) k)] = first + k;  // (distributed: global slot ids)
}

static void launch_identity(hipStream_t st, const Buffers &B, int which, int row) {
    hipLaunchKernelGGL(identity_kernel, dim3(B.ncap / kBlock), dim3(kBlock), 0, st, B.gen[which], B.compact, row, B.ncap, B.first);
}

// compact genealogy rows -> plain rows (slamgpu.cpp: demote_to_plain)
__global__ void __launch_bounds__(kBlock) decompact_kernel(const int32_t *__restrict__ src, int32_t *__restrict__ dst, int ncap, int rows) {
    const int k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= ncap) return;
    const int4 *__restrict__ s4 = reinterpret_cast<const int4 *>(src);
    for (int c = 0; c < (rows + 3) / 4; c++) {
        const int4 q = s4[(size_t) c * ncap + k];
        const int v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (4 * c + j < rows) dst[(size_t) (4 * c + j) * ncap + k] = v[j];
    }
}

static void launch_decompact(hipStream_t st, const int32_t *src, int32_t *dst, int ncap, int rows) {
    hipLaunchKernelGGL(decompact_kernel, dim3(ncap / kBlock), dim3(kBlock), 0, st, src, dst, ncap, rows);
}

static void launch_flatten(hipStream_t st, const Buffers &B, int nf) {
    const int gy = nf > 0 ? (nf + kLmkPerBlockY - 1) / kLmkPerBlockY : 1;
    hipLaunchKernelGGL(flatten_kernel, dim3(B.ncap / kBlock, gy), dim3(kBlock), 0, st, B, nf);
    // row 0 may have been a source row of that launch: it becomes the identity only now
    launch_identity(st, B, 0, 0);
    launch_identity(st, B, 1, 0);
}

static void launch_finish(hipStream_t st, const Buffers &B, const WeightScratch &ws, double *hist, int par) {
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(kBlock), 0, st, B, ws, hist, par);
}

static void launch_predict(hipStream_t st, const Buffers &B, const PredictArgs &A, const RngArgs &rng) {
    hipLaunchKernelGGL(predict_kernel, dim3(B.ncap / kBlock), dim3(kBlock), 0, st, B, A, rng);
}

static void launch_estimate(hipStream_t st, const Buffers &B, const WeightScratch &ws, double *hist) {
    hipLaunchKernelGGL(estimate_kernel, dim3(ws.nblocks), dim3(kBlock), 0, st, B, ws);
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(kBlock), 0, st, B, ws, hist, ws.wpar);
}

static void launch_jacobians(hipStream_t st, const float *in, uint32_t n, float *out) {
    hipLaunchKernelGGL(jacobians_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, in, n, out);
}

static void launch_jacobians_multi(hipStream_t st, float *win, const uint32_t *tab, uint32_t nfeat) {
    hipLaunchKernelGGL(jacobians_multi_kernel, dim3((nfeat + kBlock - 1) / kBlock), dim3(kBlock), 0, st, win, tab, nfeat);
}

static void launch_observe(hipStream_t st, const ObserveArgs &A) { hipLaunchKernelGGL(observe_kernel, dim3(1), dim3(kBlock), 0, st, A); }
static void launch_observe_book(hipStream_t st, const ObserveArgs &A) {
    hipLaunchKernelGGL(observe_book_kernel, dim3(1), dim3(kObsThreads), 0, st, A);
}

static void launch_associate(hipStream_t st, const Buffers &B, int nf, const float *z, int nz, const float *R4, float g1, float g2, const float *excl3,
                             const uint32_t *retired, int32_t *labels, int by_obs) {
    if (excl3 && excl3[0] + excl3[1] > 0.0f)
        hipLaunchKernelGGL(associate_kernel<true>, dim3(B.ncap / kBlock), dim3(kBlock), 0, st, B, nf, z, nz, R4[0], R4[1], R4[2], R4[3], g1, g2, excl3[0],
                           excl3[1], excl3[2], retired, labels, by_obs);
    else
        hipLaunchKernelGGL(associate_kernel<false>, dim3(B.ncap / kBlock), dim3(kBlock), 0, st, B, nf, z, nz, R4[0], R4[1], R4[2], R4[3], g1, g2, 0.0f, 0.0f,
                           0.0f, retired, labels, by_obs);
}

static void launch_kat(hipStream_t st, int op, const float *in, int n, float *out) {
    hipLaunchKernelGGL(kat_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, op, in, n, out);
}

__global__ void __launch_bounds__(kBlock) dist_gather_kernel(DistGatherArgs A) {
    const int h = blockIdx.x / A.n_shards, g = blockIdx.x % A.n_shards;
    const float *src = A.local[g];
    float *dst = A.gathered[h] + (size_t) g * A.floats_per_shard;
    for (int i = threadIdx.x; i < A.floats_per_shard; i += kBlock) dst[i] = src[i];
}

static void launch_dist_gather(hipStream_t st, const DistGatherArgs &A) {
    hipLaunchKernelGGL(dist_gather_kernel, dim3(A.n_shards * A.n_shards), dim3(kBlock), 0, st, A);
}

__global__ void __launch_bounds__(kWave) dist_flag_kernel(DistFlagArgs A) {
    const int t = threadIdx.x;
    if (t >= A.n_shards || t == A.shard) return;
    __hip_atomic_store(A.peer_flags[t] + A.shard, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // (an earlier barrier already gave up on a peer: the run is void and will be repeated: do not wait again)
    if (__hip_atomic_load(A.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
    uint32_t spins = 0;
    // (sequence numbers only grow: signed distance copes with the wrap)
    while ((int32_t) (__hip_atomic_load(A.my_flags + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - A.seq) < 0) {
        if (++spins > A.max_spins) {
            __hip_atomic_store(A.err, A.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

static void launch_dist_flags(hipStream_t st, const DistFlagArgs &A) {
    hipLaunchKernelGGL(dist_flag_kernel, dim3(1), dim3(kWave), 0, st, A);
}

// slamgpu_peek: see kernels.h: PeekArgs.  blockIdx.y = group of kLmkPerBlockY landmarks (group 0 also moves the pose).
__global__ void __launch_bounds__(kBlock) peek_kernel(Buffers B, WeightScratch ws, PeekArgs A) {
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= A.count) return;
    const int k = A.first + idx * A.stride;
    const Ctrl *ctrl = B.ctrl;
    const int cur = ctrl->live[B.slot];
    const bool pend = ctrl->pend[B.slot] != 0;
    const int anc = pend ? ws.keep[B.slot][k] : k;
    const size_t S = (size_t) B.ncap;
    if (blockIdx.y == 0) {
        float4 pa = B.poseA[cur][anc];
        if (pend) pa.w = ctrl->inv_n;  // resampled particles restart at 1/N (core.cpp:744-747)
        A.oa[idx] = pa;
        A.ob[idx] = B.poseB[cur][anc];
        A.oc[idx] = B.poseC[cur][anc];
    }
    const int j0 = blockIdx.y * kLmkPerBlockY, j1 = min(A.nf, j0 + kLmkPerBlockY);
    for (int j = j0; j < j1; j++) {
        float4 la;
        float lb;
        read_through_genealogy(B, B.lmk_live, cur, S, j, anc, la, lb);
        A.la[(size_t) j * A.count + idx] = la;
        A.lb[(size_t) j * A.count + idx] = lb;
    }
}

static void launch_peek(hipStream_t st, const Buffers &B, const WeightScratch &ws, const PeekArgs &A) {
    const int gy = A.nf > 0 ? (A.nf + kLmkPerBlockY - 1) / kLmkPerBlockY : 1;
    hipLaunchKernelGGL(peek_kernel, dim3((A.count + kBlock - 1) / kBlock, gy), dim3(kBlock), 0, st, B, ws, A);
}

static void launch_lmk_box(hipStream_t st, const Buffers &B, const int32_t *ids, int count, const uint32_t *retired, LmkBox *box) {
    if (count > 0) hipLaunchKernelGGL(lmk_box_kernel, dim3(count), dim3(kBlock), 0, st, B, ids, count, retired, box);
}

static void launch_assoc_grid(hipStream_t st, const Buffers &B, const AssocGridArgs &A) {
    hipLaunchKernelGGL(assoc_geom_partial_kernel, dim3(kGeomBlocks), dim3(kBlock), 0, st, B, A.geom_part);
    hipLaunchKernelGGL(assoc_geom_kernel, dim3(1), dim3(kBlock), 0, st, B, A, A.geom_part);
    const int gb = (A.nf + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(assoc_count_kernel, dim3(gb), dim3(kBlock), 0, st, A, 0);
    hipLaunchKernelGGL(assoc_scan_kernel, dim3(1), dim3(kBlock), 0, st, A);
    hipLaunchKernelGGL(assoc_count_kernel, dim3(gb), dim3(kBlock), 0, st, A, 1);
}

// the directly addressed votes of the candidate lists -> the per-observation tables the host reads (label, weight), at most kVoteSlots
// labels with a vote each (more: AssocGeom::overflow bit 1, as when the hash table fills up)
__global__ void __launch_bounds__(kWave) vote_compact_kernel(AssocGridArgs A) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const int cnt = A.cell_start[q] + 2;
    const float *w = A.vote_w + (size_t) q * (A.lcap + 2);
    VoteSlot *out = A.votes + (size_t) q * kVoteSlots;
    int used = 0;
    for (int p0 = 0; p0 < cnt; p0 += kWave) {
        const int p = p0 + lane;
        const float v = p < cnt ? w[p] : 0.0f;
        const bool has = v > 0.0f;
        const unsigned long long mk = __ballot(has);
        const int at = used + __popcll(mk & ((1ull << lane) - 1ull));
        if (has && at < kVoteSlots) {
            const int label = p == 0 ? kAssocNew : (p == 1 ? kAssocDiscard : __float_as_int(A.items[2 * ((size_t) q * A.lcap + (p - 2)) + 1].y));
            out[at].key = label;
            out[at].w = v;
        }
        used += __popcll(mk);
    }
    if (lane == 0 && used > kVoteSlots) atomicOr(&A.geom->overflow, 2);
}
static void launch_vote_compact(hipStream_t st, const AssocGridArgs &A) { hipLaunchKernelGGL(vote_compact_kernel, dim3(A.nz), dim3(kWave), 0, st, A); }

static void launch_associate_grid(hipStream_t st, const Buffers &B, const AssocGridArgs &A, const float *R4, float g1, float g2, int32_t *labels) {
    if (A.lcap > 0)
        hipLaunchKernelGGL(associate_grid_kernel<true>, dim3(B.ncap / kBlock, (A.nz + A.obs_per_block - 1) / A.obs_per_block), dim3(kBlock), 0, st, B, A,
                           R4[0], R4[1], R4[2], R4[3], g1, g2, labels);
    else
        hipLaunchKernelGGL(associate_grid_kernel<false>, dim3(B.ncap / kBlock, (A.nz + A.obs_per_block - 1) / A.obs_per_block), dim3(kBlock), 0, st, B, A,
                           R4[0], R4[1], R4[2], R4[3], g1, g2, labels);
}
static void launch_assoc_lists(hipStream_t st, const Buffers &B, const AssocGridArgs &A) {
    hipLaunchKernelGGL(assoc_geom_partial_kernel, dim3(kGeomBlocks), dim3(kBlock), 0, st, B, A.geom_part);
    hipLaunchKernelGGL(assoc_geom_kernel, dim3(1), dim3(kBlock), 0, st, B, A, A.geom_part);
    hipLaunchKernelGGL(assoc_lists_kernel, dim3(A.nz), dim3(kBlock), 0, st, A, B.erow, B.lmk_live);
}

// ---------------------------------------------------------------------------------------------------
// Per-particle association (kernels.h: PerParticle; slamgpu.cpp: do_update_particle): between the association's labels
// [n][nz] and the update launch.
// (the labels of this path are laid out BY OBSERVATION, [nz][ncap]: the association writes and these kernels read 256 consecutive
// particles of one observation at a time; particle-major, every access of a wave touched 64 lines)
// census: which landmark slots ANY particle matched (first[l] = the lowest observation index that names l, INT_MAX: none -- the
// host orders the packet's entries by it, so that particles which agree meet their landmarks in the order of the observations, as
// the reference's loop over zf does) and how many particles call each observation new.
constexpr int kCensusObs = 16;  // observations per workgroup of pp_census_kernel (blockIdx.y): one thread walking all 865 of a config-5 step
                                // alone was 865 dependent loads, ~1 ms of a 6.7 ms step
__global__ void __launch_bounds__(kBlock) pp_census_kernel(const int32_t *__restrict__ labels, int n, int nz, size_t S, int32_t *first, int32_t *__restrict__ news) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool on = i < n;
    const int j0 = blockIdx.y * kCensusObs, jn = min(kCensusObs, nz - j0);
    int labs[kCensusObs];
#pragma unroll
    for (int q = 0; q < kCensusObs; q++) labs[q] = (on && q < jn) ? labels[(size_t) (j0 + min(q, jn - 1)) * S + i] : kAssocDiscard;  // (all in flight together)
#pragma unroll
    for (int q = 0; q < kCensusObs; q++) {
        const int j = j0 + q, lab = q < jn ? labs[q] : kAssocDiscard;  // (past the end: no label, no atomic)
        // a wave's particles nearly always agree: one atomic per wave and distinct label, and only to LOWER the word (it is soon at its
        // final value).  One atomic per particle was 10^5 atomics on one address per observation: 1.35 ms of a 1.5 ms step at 10^5
        // particles on example_webmap (profiles/particle_association_r06.txt)
        unsigned long long todo = __ballot(lab >= 0);
        while (todo) {
            const int src = __ffsll((long long) todo) - 1;
            const int lab0 = __builtin_amdgcn_readlane(lab, src);
            // (a plain look: a stale word is only ever too HIGH and costs one atomic more, at most one per wave and observation)
            if ((int) (threadIdx.x & (kWave - 1)) == src && first[lab0] > j) atomicMin(first + lab0, j);
            todo &= ~__ballot(lab == lab0);
        }
        const unsigned long long nw = __ballot(lab == kAssocNew);
        if (nw && (threadIdx.x & (kWave - 1)) == (int) __ffsll((long long) __ballot(true)) - 1) atomicAdd(news + j, (int) __popcll(nw));
    }
}
// resolve: labels -> PerParticle::obs / wf / any.  uidx[l] = packet entry of landmark slot l (-1: not in the packet), newk[j] = new
// slot (entry m + newk[j]) opened for observation j (-1: none).  One observation per landmark and particle (the first to name it; a
// scan sees a landmark once); an observation the particle does not use -- discarded between the gates, a second claim on a
// landmark, called new without a slot being opened for it, or opening one -- counts as unexplained and costs the factor p_new
// (FastSLAM's constant likelihood of a new feature: without it a particle that ignores an observation would outweigh one that
// explains it).
// (The walk is sequential per particle -- the first claim on a landmark wins -- but nothing in it needs to wait for memory: the labels
// and their packet entries come eight observations at a time, all in flight together, and "has this particle claimed entry k yet"
// is a bit in LDS (one column of words per thread) instead of a read of the obs row it has just initialised; contexts whose packet
// has more entries than the LDS holds bits for -- lds_words = 0 -- read the row.  Before: 865 x two dependent trips, 0.92 ms.)
constexpr int kResolveBatch = 8;
__global__ void __launch_bounds__(kBlock) pp_resolve_kernel(const int32_t *__restrict__ labels, int n, int nz, size_t S, const int32_t *__restrict__ uidx,
                                                             const int32_t *__restrict__ newk, int m, int nn, float p_new, int logw, int lds_words,
                                                             int16_t *__restrict__ obs, float *__restrict__ wf, uint8_t *__restrict__ any) {
    extern __shared__ uint32_t sh_claim[];  // [lds_words][kBlock]
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int) S) return;
    for (int k = 0; k < m + nn; k++) obs[(size_t) k * S + i] = (int16_t) -1;
    for (int w = 0; w < lds_words; w++) sh_claim[w * kBlock + threadIdx.x] = 0u;
    if (i >= n) {
        wf[i] = logw ? 0.0f : 1.0f;
        any[i] = 0;
        return;
    }
    int unexplained = 0, flags = 0;
    for (int j0 = 0; j0 < nz; j0 += kResolveBatch) {
        int labs[kResolveBatch], ks[kResolveBatch];
#pragma unroll
        for (int q = 0; q < kResolveBatch; q++) labs[q] = labels[(size_t) min(j0 + q, nz - 1) * S + i];
#pragma unroll
        for (int q = 0; q < kResolveBatch; q++) ks[q] = labs[q] >= 0 ? uidx[labs[q]] : (labs[q] == kAssocNew ? newk[min(j0 + q, nz - 1)] : -1);
#pragma unroll
        for (int q = 0; q < kResolveBatch; q++) {
            const int j = j0 + q;
            if (j >= nz) break;
            const int lab = labs[q], k = ks[q];
            if (lab >= 0) {
                bool fresh = false;
                if (k >= 0) {
                    if (lds_words) {
                        uint32_t &word = sh_claim[(k >> 5) * kBlock + threadIdx.x];
                        fresh = !((word >> (k & 31)) & 1u);
                        word |= 1u << (k & 31);
                    } else {
                        fresh = obs[(size_t) k * S + i] < 0;
                    }
                }
                if (fresh) {
                    obs[(size_t) k * S + i] = (int16_t) j;
                    flags |= 1;
                } else {
                    unexplained++;
                }
            } else {
                unexplained++;
                if (k >= 0) {  // (called new, and a slot was opened for this observation)
                    obs[(size_t) (m + k) * S + i] = (int16_t) j;
                    flags |= 2;
                }
            }
        }
    }
    // (repeated multiplication, not powf: the same bits in both builds and on the host's restatement)
    float f = logw ? 0.0f : 1.0f;
    const float lp = logw ? logf(p_new) : 0.0f;
    for (int q = 0; q < unexplained; q++) f = logw ? f + lp : f * p_new;
    wf[i] = f;
    any[i] = (uint8_t) flags;
}

// holders: how many particles of the (plain) set hold landmark slot l, l < nf -- a slot nobody holds any more (its hypotheses died
// in a resample) is dead: the host takes it out of the association and opens it again for a later landmark
// (ids: the slots to count -- the host lists the PARTIAL ones: a slot every particle opened is held by every descendant for good --
// or null: slots 0 .. nf - 1)
__global__ void __launch_bounds__(kBlock) pp_holders_kernel(Buffers B, int nf, const int32_t *__restrict__ ids, int32_t *__restrict__ holders) {
    // per block: the waves' counts meet in LDS, a chunk of landmarks at a time; one global atomic per block and landmark (one per wave
    // and landmark was 55 000 atomics on 35 addresses at 10^5 particles on example_webmap: 0.45 ms)
    constexpr int kChunk = 1024;
    __shared__ int32_t sh[kChunk];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    const bool on = i < B.n;
    const int cur = B.ctrl->live[B.slot];
    const size_t S = (size_t) B.ncap;
    for (int l0 = 0; l0 < nf; l0 += kChunk) {
        const int ln = min(kChunk, nf - l0);
        for (int t = threadIdx.x; t < ln; t += kBlock) sh[t] = 0;
        __syncthreads();
        for (int l = l0; l < l0 + ln; l++) {
            const int slot = ids ? ids[l] : l;
            bool has = false;
            if (on) {
                float4 la;
                float lb;
                read_through_genealogy(B, B.lmk_live, cur, S, slot, i, la, lb);
                has = la.x == la.x;
            }
            const unsigned long long hm = __ballot(has);
            if (hm && (threadIdx.x & (kWave - 1)) == (int) __ffsll((long long) __ballot(true)) - 1) atomicAdd(sh + (l - l0), (int) __popcll(hm));
        }
        __syncthreads();
        for (int t = threadIdx.x; t < ln; t += kBlock)
            if (sh[t]) atomicAdd(holders + (ids ? ids[l0 + t] : l0 + t), sh[t]);
        __syncthreads();
    }
}

static void launch_pp_census(hipStream_t st, const int32_t *labels, int n, int nz, int ncap, int32_t *first, int32_t *news) {
    hipLaunchKernelGGL(pp_census_kernel, dim3((n + kBlock - 1) / kBlock, (nz + kCensusObs - 1) / kCensusObs), dim3(kBlock), 0, st, labels, n, nz, (size_t) ncap, first, news);
}
static void launch_pp_resolve(hipStream_t st, const int32_t *labels, int n, int nz, int ncap, const int32_t *uidx, const int32_t *newk, int m, int nn,
                              float p_new, int logw, int16_t *obs, float *wf, uint8_t *any) {
    // claimed-entry bits in LDS while the packet's re-observed entries fit 48 KB of them (m <= 1 536: every step of the 10 000-landmark map)
    const int words = (m + 31) / 32, lds_words = words * kBlock * 4 <= 48 * 1024 ? words : 0;
    hipLaunchKernelGGL(pp_resolve_kernel, dim3(ncap / kBlock), dim3(kBlock), (size_t) lds_words * kBlock * sizeof(uint32_t), st, labels, n, nz, (size_t) ncap, uidx, newk, m,
                       nn, p_new, logw, lds_words, obs, wf, any);
}
static void launch_pp_holders(hipStream_t st, const Buffers &B, int count, const int32_t *ids, int32_t *holders) {
    hipLaunchKernelGGL(pp_holders_kernel, dim3(B.ncap / kBlock), dim3(kBlock), 0, st, B, count, ids, holders);
}

static const KernelTable kTable = {launch_update, launch_update_particle, launch_update_persist, launch_resample, launch_resample_ref, launch_scan, launch_gather, launch_flatten, launch_identity, launch_decompact, launch_finish, launch_predict, launch_estimate, launch_jacobians, launch_kat, launch_observe, launch_observe_book, launch_associate,
                                   launch_shard_plan, launch_shard_pack, launch_shard_unpack, launch_shard_finish, launch_dist_gather, launch_dist_flags, launch_peek, launch_lmk_box, launch_assoc_grid, launch_assoc_lists, launch_vote_compact,
                                   launch_associate_grid, launch_jacobians_multi, launch_pp_census, launch_pp_resolve, launch_pp_holders};

}  // namespace SLAM_KNS

namespace slamgpu {
#define SLAM_CAT2(a, b) a##b
#define SLAM_CAT(a, b) SLAM_CAT2(a, b)
const KernelTable *SLAM_CAT(kernels_, SLAM_TABLE)() { return &SLAM_KNS::kTable; }
}  // namespace slamgpu
