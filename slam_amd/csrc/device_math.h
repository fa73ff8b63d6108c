// Per-particle FastSLAM arithmetic for gfx950, one particle per work-item, everything in registers.
// Fixed-size 2x2 / 3x3 code following the reference's float32 operation order (file:line cited per
// function, relative to the reference tree) so that the only deviations from the CPU result are
//   (1) device libm (atan2f/sinf/cosf/expf/logf are 1-2 ulp routines, not glibc's),
//   (2) gaussEvaluate solved by forward substitution instead of the reference's JacobiSVD pseudo-inverse,
//   (3) covariances stored symmetric-packed (Pv: 6 floats, Pf: 3 floats) in HBM.
// This header is compiled twice: namespace slam_strict with -ffp-contract=off (no FMA, IEEE divide and
// sqrt) and namespace slam_fast with contraction allowed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef SLAM_KNS
#error "define SLAM_KNS (slam_strict or slam_fast)"
#endif

namespace SLAM_KNS {

#define SLAM_DEV __device__ __forceinline__

constexpr double kPi = 3.14159265358979323846;

// Division / reciprocal / square root.  strict: IEEE-rounded (what the reference's SSE2 code does; on gfx950 each
// costs a 10-12 instruction v_div_scale / v_rcp / fma / v_div_fmas / v_div_fixup sequence, about a third of the
// update kernel's instructions).  fast (-DSLAM_FAST_MATH): the 1-ulp hardware v_rcp_f32 / v_sqrt_f32.
#ifdef SLAM_FAST_MATH
SLAM_DEV float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
SLAM_DEV float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
SLAM_DEV float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
#else
SLAM_DEV float frcp(float x) { return 1.0f / x; }
SLAM_DEV float fdiv(float a, float b) { return a / b; }
SLAM_DEV float fsqrt(float x) { return sqrtf(x); }
#endif

// core.cpp:460-477 — double constants against a float argument.
SLAM_DEV float trig_offset(float ang) {
    if (((double) ang < -2 * kPi) || ((double) ang > 2 * kPi)) {
        int n = (int) floor((double) ang / (2 * kPi));
        ang = (float) ((double) ang - n * (2 * kPi));
    }
    if ((double) ang > kPi) ang = (float) ((double) ang - (2 * kPi));
    if ((double) ang < -kPi) ang = (float) ((double) ang + (2 * kPi));
    return ang;
}

struct Jac {
    float zp0, zp1;
    float hv00, hv01, hv10, hv11;  // Hv = [[hv00, hv01, 0], [hv10, hv11, -1]]
    float hf00, hf01, hf10, hf11;  // Hf
    float s00, s01, s10, s11;      // Sf = Hf Pf Hf^T + R
};

// computeJacobians, one feature (core.cpp:682-704).  Pf symmetric-packed: p00, p10, p11.
SLAM_DEV Jac jacobian(float x, float y, float th, float fx, float fy, float p00, float p10, float p11, float r00,
                      float r01, float r10, float r11) {
    Jac j;
    float dx = fx - x;
    float dy = fy - y;
    float d2 = (float) ((double) dx * (double) dx + (double) dy * (double) dy);  // pow(dx,2)+pow(dy,2) in double (:685)
    float d = fsqrt(d2);
    j.zp0 = d;
    j.zp1 = trig_offset(atan2f(dy, dx) - th);
    // Hv's entries are Hf's negated (core.cpp:690-697 writes eight quotients); IEEE division is sign-symmetric, (-a) / b ==
    // -(a / b) bit for bit, so four quotients are computed (the strict build's division is an 11-instruction sequence)
    j.hf00 = fdiv(dx, d);
    j.hf01 = fdiv(dy, d);
    j.hf10 = fdiv(-dy, d2);
    j.hf11 = fdiv(dx, d2);
    j.hv00 = -j.hf00;
    j.hv01 = -j.hf01;
    j.hv10 = -j.hf10;
    j.hv11 = -j.hf11;
    // T = Hf * Pf ; Sf = T * Hf^T + R  (k-ascending sums, GEMM order)
    float t00 = j.hf00 * p00 + j.hf01 * p10;
    float t01 = j.hf00 * p10 + j.hf01 * p11;
    float t10 = j.hf10 * p00 + j.hf11 * p10;
    float t11 = j.hf10 * p10 + j.hf11 * p11;
    j.s00 = (t00 * j.hf00 + t01 * j.hf01) + r00;
    j.s01 = (t00 * j.hf10 + t01 * j.hf11) + r01;
    j.s10 = (t10 * j.hf00 + t11 * j.hf01) + r10;
    j.s11 = (t10 * j.hf10 + t11 * j.hf11) + r11;
    return j;
}

// A.inverse() of a dynamic 2x2 = PartialPivLU (Eigen LU/PartialPivLU.h:239-283 + the two triangular solves).
SLAM_DEV void inverse2(float a00, float a01, float a10, float a11, float &x00, float &x01, float &x10, float &x11) {
    bool swap = fabsf(a10) > fabsf(a00);
    float u00 = swap ? a10 : a00, u01 = swap ? a11 : a01;
    float l10 = swap ? a00 : a10, r11 = swap ? a01 : a11;
    if (u00 != 0.0f) l10 = l10 * frcp(u00);
    float u11 = r11 - l10 * u01;
    // P * I
    float b00 = swap ? 0.0f : 1.0f, b01 = swap ? 1.0f : 0.0f;
    float b10 = swap ? 1.0f : 0.0f, b11 = swap ? 0.0f : 1.0f;
    // unit-lower solve
    b10 -= b00 * l10;
    b11 -= b01 * l10;
    // upper solve (column-major order: row 1 first, then eliminate into row 0)
    float a = frcp(u11);
    b10 *= a;
    b11 *= a;
    b00 -= b10 * u01;
    b01 -= b11 * u01;
    a = frcp(u00);
    x00 = b00 * a;
    x01 = b01 * a;
    x10 = b10;
    x11 = b11;
}

SLAM_DEV float determinant2(float a00, float a01, float a10, float a11) {
    bool swap = fabsf(a10) > fabsf(a00);
    float u00 = swap ? a10 : a00, u01 = swap ? a11 : a01;
    float l10 = swap ? a00 : a10, r11 = swap ? a01 : a11;
    if (u00 != 0.0f) l10 = l10 * frcp(u00);
    float u11 = r11 - l10 * u01;
    return (swap ? -1.0f : 1.0f) * (u00 * u11);
}

// Lower Cholesky of a symmetric 3x3 given by its lower triangle (Eigen Cholesky/LLT.h:260-287).
// On a non-positive pivot at step k the remaining columns keep the input values, as Eigen leaves them.
struct L3 {
    float l00, l10, l11, l20, l21, l22;
    // 1 / l00 and 1 / l11 as the factorisation computed them (valid when it got that far: r0 != 0 / r1 != 0); the solve below
    // needs the same two quotients again
    float r0, r1;
};

SLAM_DEV L3 llt3(float a00, float a10, float a11, float a20, float a21, float a22) {
    L3 L = {a00, a10, a11, a20, a21, a22, 0.0f, 0.0f};
    float x = a00;
    if (x <= 0.0f) return L;
    x = fsqrt(x);
    L.l00 = x;
    float r = frcp(x);
    L.r0 = r;
    L.l10 = a10 * r;
    L.l20 = a20 * r;
    x = a11 - L.l10 * L.l10;
    if (x <= 0.0f) return L;
    x = fsqrt(x);
    L.l11 = x;
    r = frcp(x);
    L.r1 = r;
    L.l21 = (a21 + L.l20 * (-1.0f * L.l10)) * r;
    x = a22 - (L.l20 * L.l20 + L.l21 * L.l21);
    if (x <= 0.0f) return L;
    L.l22 = fsqrt(x);
    return L;
}

struct L2 {
    float l00, l10, l11;
};

SLAM_DEV L2 llt2(float a00, float a10, float a11) {
    L2 L = {a00, a10, a11};
    float x = a00;
    if (x <= 0.0f) return L;
    x = fsqrt(x);
    L.l00 = x;
    L.l10 = a10 * frcp(x);
    x = a11 - L.l10 * L.l10;
    if (x <= 0.0f) return L;
    L.l11 = fsqrt(x);
    return L;
}

// A.llt().solve(Identity) for 3x3 (TriangularSolverMatrix.h:109-137): column-oriented forward
// substitution with reciprocal diagonals, then row-oriented back substitution.  X row-major, full.
SLAM_DEV void llt_solve_identity3(const L3 &L, float X[9]) {
    // (a reciprocal is never 0 for a finite positive pivot: r0 / r1 == 0 means the factorisation stopped before it)
    float a0 = L.r0 != 0.0f ? L.r0 : frcp(L.l00), a1 = L.r1 != 0.0f ? L.r1 : frcp(L.l11), a2 = frcp(L.l22);
    // Y = L^-1 I
    float y00 = a0;
    float y10 = 0.0f - y00 * L.l10;
    float y20 = 0.0f - y00 * L.l20;
    y10 = y10 * a1;
    y20 = y20 - y10 * L.l21;
    y20 = y20 * a2;
    float y11 = a1;
    float y21 = 0.0f - y11 * L.l21;
    y21 = y21 * a2;
    float y22 = a2;
    // X = U^-1 Y with U = L^T; row 2, then 1, then 0.  y01 = y02 = y12 = 0.
    float x20 = (y20 - 0.0f) * a2, x21 = (y21 - 0.0f) * a2, x22 = (y22 - 0.0f) * a2;
    float x10 = (y10 - L.l21 * x20) * a1, x11 = (y11 - L.l21 * x21) * a1, x12 = (0.0f - L.l21 * x22) * a1;
    float x00 = (y00 - (L.l10 * x10 + L.l20 * x20)) * a0;
    float x01 = (0.0f - (L.l10 * x11 + L.l20 * x21)) * a0;
    float x02 = (0.0f - (L.l10 * x12 + L.l20 * x22)) * a0;
    X[0] = x00; X[1] = x01; X[2] = x02;
    X[3] = x10; X[4] = x11; X[5] = x12;
    X[6] = x20; X[7] = x21; X[8] = x22;
}

// gaussEvaluate, logflag = 0 (fastslam2.cpp:127-163).  The reference solves Sc*nin = v with a JacobiSVD
// pseudo-inverse; Sc is lower triangular so forward substitution gives the same nin to rounding.
// C = (2*pi)^(D/2) * prod(diag) with INTEGER D/2 => (2*pi)^1 for D = 2 and D = 3 (:152).
SLAM_DEV float gauss2(float v0, float v1, float s00, float s10, float s11) {
    L2 L = llt2(s00, s10, s11);
    float n0 = fdiv(v0, L.l00);
    float n1 = fdiv(v1 - L.l10 * n0, L.l11);
    float E = n0 * n0;
    E += n1 * n1;
    E = -0.5f * E;
    float prod = (1.0f * L.l00) * L.l11;
    float C = (float) ((2 * kPi) * (double) prod);
    return fdiv(expf(E), C);
}

SLAM_DEV float gauss3(float v0, float v1, float v2, float a00, float a10, float a11, float a20, float a21, float a22) {
    L3 L = llt3(a00, a10, a11, a20, a21, a22);
    float n0 = fdiv(v0, L.l00);
    float n1 = fdiv(v1 - L.l10 * n0, L.l11);
    float n2 = fdiv(v2 - (L.l20 * n0 + L.l21 * n1), L.l22);
    float E = n0 * n0;
    E += n1 * n1;
    E += n2 * n2;
    E = -0.5f * E;
    float prod = ((1.0f * L.l00) * L.l11) * L.l22;
    float C = (float) ((2 * kPi) * (double) prod);
    return fdiv(expf(E), C);
}

// gaussEvaluate with logflag = 1 (fastslam2.cpp:154-160): E - (0.5 * D * log(2 pi) + sum log diag(Sc)).  The reference
// never calls it; log-weight contexts (slamgpu_config.log_weights) accumulate it instead of multiplying the
// logflag = 0 value, whose float32 product overflows beyond ~20 re-observed landmarks per step.  (NB the log form uses
// the true D/2, the linear form integer D/2: for D = 3 they differ by the constant factor sqrt(2 pi), which cancels in
// prior / proposal.)
SLAM_DEV float gauss2_log(float v0, float v1, float s00, float s10, float s11) {
    L2 L = llt2(s00, s10, s11);
    float n0 = fdiv(v0, L.l00);
    float n1 = fdiv(v1 - L.l10 * n0, L.l11);
    float E = n0 * n0;
    E += n1 * n1;
    E = -0.5f * E;
    float sum = logf(L.l00) + logf(L.l11);
    float C = (float) (0.5 * 2 * 1.8378770664093453 + (double) sum);
    return E - C;
}

SLAM_DEV float gauss3_log(float v0, float v1, float v2, float a00, float a10, float a11, float a20, float a21, float a22) {
    L3 L = llt3(a00, a10, a11, a20, a21, a22);
    float n0 = fdiv(v0, L.l00);
    float n1 = fdiv(v1 - L.l10 * n0, L.l11);
    float n2 = fdiv(v2 - (L.l20 * n0 + L.l21 * n1), L.l22);
    float E = n0 * n0;
    E += n1 * n1;
    E += n2 * n2;
    E = -0.5f * E;
    float sum = (logf(L.l00) + logf(L.l11)) + logf(L.l22);
    float C = (float) (0.5 * 3 * 1.8378770664093453 + (double) sum);
    return E - C;
}

// choleskyUpdate for a 2x2 landmark (core.cpp:275-291).  P symmetric-packed in/out (p00,p10,p11).
SLAM_DEV void cholesky_update2(float &fx, float &fy, float &p00, float &p10, float &p11, float v0, float v1, float r00,
                               float r01, float r10, float r11, float h00, float h01, float h10, float h11) {
    float p01 = p10;
    // PHt = P * H^T
    float a00 = p00 * h00 + p01 * h01, a01 = p00 * h10 + p01 * h11;
    float a10 = p10 * h00 + p11 * h01, a11 = p10 * h10 + p11 * h11;
    // S = H * PHt + R
    float s00 = (h00 * a00 + h01 * a10) + r00, s01 = (h00 * a01 + h01 * a11) + r01;
    float s10 = (h10 * a00 + h11 * a10) + r10, s11 = (h10 * a01 + h11 * a11) + r11;
    // S = (S + S^T) * 0.5 evaluated in place: only the lower triangle feeds the LLT (see oracle note)
    float t00 = (s00 + s00) * 0.5f, t10 = (s10 + s01) * 0.5f, t11 = (s11 + s11) * 0.5f;
    L2 L = llt2(t00, t10, t11);
    // SChol = U = L^T ; SCholInv = U.inverse() (PartialPivLU of an upper-triangular matrix: no swap)
    float i00, i01, i10, i11;
    inverse2(L.l00, L.l10, 0.0f, L.l11, i00, i01, i10, i11);
    // W1 = PHt * SCholInv ; W = W1 * SCholInv^T
    float w100 = a00 * i00 + a01 * i10, w101 = a00 * i01 + a01 * i11;
    float w110 = a10 * i00 + a11 * i10, w111 = a10 * i01 + a11 * i11;
    float w00 = w100 * i00 + w101 * i01, w01 = w100 * i10 + w101 * i11;
    float w10 = w110 * i00 + w111 * i01, w11 = w110 * i10 + w111 * i11;
    fx = fx + (w00 * v0 + w01 * v1);
    fy = fy + (w10 * v0 + w11 * v1);
    p00 = p00 - (w100 * w100 + w101 * w101);
    p10 = p10 - (w110 * w100 + w111 * w101);
    p11 = p11 - (w110 * w110 + w111 * w111);
}

// addFeature for one new observation (core.cpp:488-501), R general 2x2.
SLAM_DEV void add_feature(float x, float y, float th, float r, float b, float r00, float r01, float r10, float r11,
                          float &fx, float &fy, float &p00, float &p10, float &p11) {
    float s = sinf(th + b);
    float c = cosf(th + b);
    fx = x + r * c;
    fy = y + r * s;
    float g00 = c, g01 = -r * s, g10 = s, g11 = r * c;
    float t00 = g00 * r00 + g01 * r10, t01 = g00 * r01 + g01 * r11;
    float t10 = g10 * r00 + g11 * r10, t11 = g10 * r01 + g11 * r11;
    p00 = t00 * g00 + t01 * g01;
    p10 = t10 * g00 + t11 * g01;
    p11 = t10 * g10 + t11 * g11;
}

// multivariateGauss(x, P, 1) for D = 3 (core.cpp:452-458): L*g + x with L = P.llt().matrixL().
SLAM_DEV void mvgauss3(float &x0, float &x1, float &x2, const L3 &L, float g0, float g1, float g2) {
    float s0 = (L.l00 * g0 + 0.0f * g1) + 0.0f * g2;
    float s1 = (L.l10 * g0 + L.l11 * g1) + 0.0f * g2;
    float s2 = (L.l20 * g0 + L.l21 * g1) + L.l22 * g2;
    x0 = s0 + x0;
    x1 = s1 + x1;
    x2 = s2 + x2;
}

// ---- wave64 prefix sums / reductions on the DPP data path ---------------------------------------------------------
// `__shfl_up` / `__shfl_xor` compile to ds_bpermute_b32 (an LDS-crossbar round trip, ~100 cycles of dependent latency
// each, two per double): the six-step scans and reductions of the update kernel spent ~1.5 us per launch in them
// (profiles/update_kernel_levels_r02_before_*.txt: levels 2 and 9).  DPP moves are VALU operands (~8 cycles).  gfx9
// wave64 scan: row_shr 1, 2, 4, 8 inside each row of 16 lanes, then row_bcast:15 into rows 1 and 3 and row_bcast:31
// into rows 2 and 3; lanes without a source read 0 (old = 0, bound_ctrl off), which is the identity of the sum.
// The association is fixed by the lane layout, so every block, kernel and shard computes bit-identical sums.
template <int CTRL, int ROW_MASK>
SLAM_DEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false); }
template <int CTRL, int ROW_MASK>
SLAM_DEV float dpp_f(float v) { return __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(v))); }
template <int CTRL, int ROW_MASK>
SLAM_DEV double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL, ROW_MASK>((int) (b & 0xffffffffll)), hi = dpp_i<CTRL, ROW_MASK>((int) (b >> 32));
    return __longlong_as_double(((long long) hi << 32) | (unsigned int) lo);
}

SLAM_DEV float wave_scan_f(float v) {  // inclusive prefix sum over the 64 lanes
    v += dpp_f<0x111, 0xf>(v);
    v += dpp_f<0x112, 0xf>(v);
    v += dpp_f<0x114, 0xf>(v);
    v += dpp_f<0x118, 0xf>(v);
    v += dpp_f<0x142, 0xa>(v);
    v += dpp_f<0x143, 0xc>(v);
    return v;
}

SLAM_DEV double wave_scan_d(double v) {
    v += dpp_d<0x111, 0xf>(v);
    v += dpp_d<0x112, 0xf>(v);
    v += dpp_d<0x114, 0xf>(v);
    v += dpp_d<0x118, 0xf>(v);
    v += dpp_d<0x142, 0xa>(v);
    v += dpp_d<0x143, 0xc>(v);
    return v;
}

// lane 63's value in every lane (uniform)
SLAM_DEV float wave_last_f(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
SLAM_DEV double wave_last_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int) (b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int) (b >> 32), 63);
    return __longlong_as_double(((long long) hi << 32) | (unsigned int) lo);
}
// minimum / maximum over the 64 lanes (DPP scan, result read from lane 63): six cross-lane moves inside the VALU instead of
// six LDS-crossbar shuffles per value
template <int CTRL, int ROW_MASK>
SLAM_DEV int dpp_i_or(int v, int otherwise) { return __builtin_amdgcn_update_dpp(otherwise, v, CTRL, ROW_MASK, 0xf, false); }
SLAM_DEV int wave_min_i(int v) {
    constexpr int id = 0x7fffffff;
    v = min(v, dpp_i_or<0x111, 0xf>(v, id));
    v = min(v, dpp_i_or<0x112, 0xf>(v, id));
    v = min(v, dpp_i_or<0x114, 0xf>(v, id));
    v = min(v, dpp_i_or<0x118, 0xf>(v, id));
    v = min(v, dpp_i_or<0x142, 0xa>(v, id));
    v = min(v, dpp_i_or<0x143, 0xc>(v, id));
    return __builtin_amdgcn_readlane(v, 63);
}
SLAM_DEV int wave_max_i(int v) {
    constexpr int id = (int) 0x80000000;
    v = max(v, dpp_i_or<0x111, 0xf>(v, id));
    v = max(v, dpp_i_or<0x112, 0xf>(v, id));
    v = max(v, dpp_i_or<0x114, 0xf>(v, id));
    v = max(v, dpp_i_or<0x118, 0xf>(v, id));
    v = max(v, dpp_i_or<0x142, 0xa>(v, id));
    v = max(v, dpp_i_or<0x143, 0xc>(v, id));
    return __builtin_amdgcn_readlane(v, 63);
}
// Loads of what ANOTHER workgroup of the running launch has stored (the persistent step loop): BYP = true reads past this CU's
// vector cache (agent-scope relaxed atomic loads = global_load ... sc1, served by the XCD's L2 where the producer's drained
// stores are; 16-byte values as two 8-byte halves: these are single latency-bound requests), so that no cache invalidate -- an
// acquire fence is 1.2-1.7 us per workgroup and iteration here -- stands between the workgroups' meeting and the first load.
// BYP = false: a plain load.
template <bool BYP>
SLAM_DEV uint32_t ldg_u32(const void *p) {
    if constexpr (BYP) return __hip_atomic_load(reinterpret_cast<const uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *reinterpret_cast<const uint32_t *>(p);
}
template <bool BYP>
SLAM_DEV unsigned long long ldg_u64(const void *p) {
    if constexpr (BYP) return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *reinterpret_cast<const unsigned long long *>(p);
}
template <bool BYP>
SLAM_DEV float ldg(const float *p) {
    if constexpr (BYP) return __uint_as_float(ldg_u32<true>(p));
    else return *p;
}
template <bool BYP>
SLAM_DEV int32_t ldg(const int32_t *p) {
    if constexpr (BYP) return (int32_t) ldg_u32<true>(p);
    else return *p;
}
template <bool BYP>
SLAM_DEV float2 ldg(const float2 *p) {
    if constexpr (BYP) {
        const unsigned long long a = ldg_u64<true>(p);
        return make_float2(__uint_as_float((uint32_t) a), __uint_as_float((uint32_t) (a >> 32)));
    } else {
        return *p;
    }
}
template <bool BYP>
SLAM_DEV float4 ldg(const float4 *p) {
    if constexpr (BYP) {
        const unsigned long long a = ldg_u64<true>(p), b = ldg_u64<true>(reinterpret_cast<const char *>(p) + 8);
        return make_float4(__uint_as_float((uint32_t) a), __uint_as_float((uint32_t) (a >> 32)), __uint_as_float((uint32_t) b),
                           __uint_as_float((uint32_t) (b >> 32)));
    } else {
        return *p;
    }
}
template <bool BYP>
SLAM_DEV int4 ldg(const int4 *p) {
    if constexpr (BYP) {
        const unsigned long long a = ldg_u64<true>(p), b = ldg_u64<true>(reinterpret_cast<const char *>(p) + 8);
        return make_int4((int) (uint32_t) a, (int) (uint32_t) (a >> 32), (int) (uint32_t) b, (int) (uint32_t) (b >> 32));
    } else {
        return *p;
    }
}

template <int CTRL, int ROW_MASK>
SLAM_DEV float dpp_f_or(float v, float otherwise) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(otherwise), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
SLAM_DEV float wave_max_f(float v) {  // maximum over the 64 lanes (fmaxf semantics), uniform
    const float id = -INFINITY;
    v = fmaxf(v, dpp_f_or<0x111, 0xf>(v, id));
    v = fmaxf(v, dpp_f_or<0x112, 0xf>(v, id));
    v = fmaxf(v, dpp_f_or<0x114, 0xf>(v, id));
    v = fmaxf(v, dpp_f_or<0x118, 0xf>(v, id));
    v = fmaxf(v, dpp_f_or<0x142, 0xa>(v, id));
    v = fmaxf(v, dpp_f_or<0x143, 0xc>(v, id));
    return wave_last_f(v);
}
SLAM_DEV float wave_sum_f(float v) { return wave_last_f(wave_scan_f(v)); }
SLAM_DEV double wave_sum_d(double v) { return wave_last_d(wave_scan_d(v)); }

// ---- Philox4x32-10, identical to oracle/slam_oracle.c:orc_philox4x32 --------------------------------
struct U4 {
    uint32_t x, y, z, w;
};

SLAM_DEV U4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t) 0xD2511F53u * c0;
        uint64_t p1 = (uint64_t) 0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t) (p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t) p1;
        uint32_t n2 = (uint32_t) (p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t) p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

// W counters that differ in c1 only (the predict steps of one particle), their rounds interleaved: W x 2 independent multiply
// chains in one basic block.  A Philox round is two v_mad_u64_u32 (~10 cycles of issue each) feeding each other across
// the halves: one generator alone runs at ~8 cycles per instruction, four together at ~6.5 (tools/microbench/valu_latency.hip).
// Same integers as philox4x32() W times.
template <int W>
SLAM_DEV void philox4x32_n(U4 (&out)[W], uint32_t c0, const uint32_t (&c1)[W], uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    uint32_t a0[W], a1[W], a2[W], a3[W];
#pragma unroll
    for (int q = 0; q < W; q++) {
        a0[q] = c0;
        a1[q] = c1[q];
        a2[q] = c2;
        a3[q] = c3;
    }
#pragma unroll
    for (int r = 0; r < 10; r++) {
#pragma unroll
        for (int q = 0; q < W; q++) {
            const uint64_t p0 = (uint64_t) 0xD2511F53u * a0[q];
            const uint64_t p1 = (uint64_t) 0xCD9E8D57u * a2[q];
            const uint32_t n0 = (uint32_t) (p1 >> 32) ^ a1[q] ^ k0;
            const uint32_t n2 = (uint32_t) (p0 >> 32) ^ a3[q] ^ k1;
            a1[q] = (uint32_t) p1;
            a3[q] = (uint32_t) p0;
            a0[q] = n0;
            a2[q] = n2;
        }
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int q = 0; q < W; q++) out[q] = U4{a0[q], a1[q], a2[q], a3[q]};
}

SLAM_DEV float u01(uint32_t x) { return ((float) (x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// Same pairing as nRandMat::randn(3,1) (core.cpp:401-416): (u0,u1) -> g0 (sin), g1 (cos); (u2,u3) -> g2 (sin)
SLAM_DEV void box_muller3(U4 r, float &g0, float &g1, float &g2) {
    float amp = sqrtf(-2.0f * logf(u01(r.x)));
    float ang = 6.28318530717958647692f * u01(r.y);
    g0 = amp * sinf(ang);
    g1 = amp * cosf(ang);
    amp = sqrtf(-2.0f * logf(u01(r.z)));
    ang = 6.28318530717958647692f * u01(r.w);
    g2 = amp * sinf(ang);
}

#ifdef SLAM_FAST_MATH
// ---------------------------------------------------------------------------------------------------
// Fast build only: the same FastSLAM2 step, algebraically restructured for the VALU (the update kernel is
// instruction-issue bound, not HBM bound: profiles/rocprof_sq_counters_r01.txt).  Same float32 storage, same
// random draws, same sample mapping (chol(P) * g), but
//   * the proposal is refined in covariance (Kalman-gain) form, K = P Hv^T (Hv P Hv^T + Sf)^-1 -- by the matrix
//     inversion lemma the same Pv = (Hv^T Sf^-1 Hv + Pv^-1)^-1 and xv += Pv Hv^T Sf^-1 v of fastslam2.cpp:335-345
//     without the two 3x3 LLT inversions per landmark (and without inverting the near-singular predicted Pv),
//   * covariances stay symmetric-packed in registers,
//   * 2x2 solves are closed-form (one v_rcp_f32), the feature update reuses Sf (core.cpp:275-291 recomputes it),
//   * sin/cos/atan2 are branch-free 1.5-ulp polynomials (arguments are bounded angles: no Payne-Hanek path),
//   * the k pending predicts are applied as ONE composite step (see predict_composite).
// ---------------------------------------------------------------------------------------------------
SLAM_DEV float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// sin and cos of a bounded angle (|x| < ~100): 2-constant Cody-Waite reduction by pi/2 + Cephes minimax polynomials
// on [-pi/4, pi/4]; max error 1.5 ulp (tools/check_fast_math.py)
SLAM_DEV void sincos_cw(float x, float &sn, float &cs) {
    const float q = __builtin_rintf(x * 0.63661977236758134f);
    float r = ffma(q, -1.57079637050628662f, x);
    r = ffma(q, 4.37113900018624283e-8f, r);
    const float z = r * r;
    float ps = ffma(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = ffma(ps, z, -1.6666654611e-1f);
    const float s = ffma(r * z, ps, r);
    float pc = ffma(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = ffma(pc, z, 4.166664568298827e-2f);
    const float c = ffma(z * z, pc, ffma(z, -0.5f, 1.0f));
    const int n = (int) q;
    const float a = (n & 1) ? c : s;
    const float b = (n & 1) ? s : c;
    sn = __uint_as_float(__float_as_uint(a) ^ ((uint32_t) (n & 2) << 30));
    cs = __uint_as_float(__float_as_uint(b) ^ ((uint32_t) ((n + 1) & 2) << 30));
}

// atan2 via min/max ratio + degree-8 minimax polynomial in t^2 on [0,1] (fit: tools/check_fast_math.py, 1.1 ulp)
SLAM_DEV float atan2_poly(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float z = t * t;
    float p = 0.0029206702020019293f;
    p = ffma(p, z, -0.016367841511964798f);
    p = ffma(p, z, 0.04321172460913658f);
    p = ffma(p, z, -0.0755220279097557f);
    p = ffma(p, z, 0.10665999352931976f);
    p = ffma(p, z, -0.14211054146289825f);
    p = ffma(p, z, 0.19993773102760315f);
    p = ffma(p, z, -0.33333152532577515f);
    float r = ffma(t * z, p, t);
    if (ay > ax) r = 1.57079637050628662f - r;
    if (x < 0.0f) r = 3.14159274101257324f - r;
    if (mx == 0.0f) r = 0.0f;
    return __builtin_copysignf(r, y);
}

// pi_to_pi of a bounded angle: one rounding instead of trig_offset's compare/branch ladder (core.cpp:460-477)
SLAM_DEV float wrap_pi(float a) {
    const float n = __builtin_rintf(a * 0.15915494309189535f);
    a = ffma(n, -6.28318548202514648f, a);
    return ffma(n, 1.74845553146951715e-7f, a);
}

struct Sym3 {
    float p00, p10, p11, p20, p21, p22;
};

// Lower Cholesky factor plus the reciprocals of its diagonal (Eigen LLT's keep-input-on-bad-pivot behaviour is kept)
struct L3r {
    float l00, l10, l11, l20, l21, l22, r0, r1, r2;
};

SLAM_DEV L3r llt3r(const Sym3 &A) {
    L3r L = {A.p00, A.p10, A.p11, A.p20, A.p21, A.p22, 1.0f, 1.0f, 1.0f};
    if (A.p00 > 0.0f) {
        L.r0 = __builtin_amdgcn_rsqf(A.p00);
        L.l00 = A.p00 * L.r0;
        L.l10 = A.p10 * L.r0;
        L.l20 = A.p20 * L.r0;
        const float x = ffma(-L.l10, L.l10, A.p11);
        if (x > 0.0f) {
            L.r1 = __builtin_amdgcn_rsqf(x);
            L.l11 = x * L.r1;
            L.l21 = ffma(-L.l20, L.l10, A.p21) * L.r1;
            const float y = ffma(-L.l21, L.l21, ffma(-L.l20, L.l20, A.p22));
            if (y > 0.0f) {
                L.r2 = __builtin_amdgcn_rsqf(y);
                L.l22 = y * L.r2;
            } else {
                L.r2 = __builtin_amdgcn_rcpf(L.l22);
            }
        } else {
            L.r1 = __builtin_amdgcn_rcpf(L.l11);
            L.r2 = __builtin_amdgcn_rcpf(L.l22);
        }
    } else {
        L.r0 = __builtin_amdgcn_rcpf(L.l00);
        L.r1 = __builtin_amdgcn_rcpf(L.l11);
        L.r2 = __builtin_amdgcn_rcpf(L.l22);
    }
    return L;
}

// -0.5 * |L^-1 v|^2 (the exponent of gaussEvaluate, fastslam2.cpp:127-163)
SLAM_DEV float gauss3_exponent(const L3r &L, float v0, float v1, float v2) {
    const float n0 = v0 * L.r0;
    const float n1 = ffma(-L.l10, n0, v1) * L.r1;
    const float n2 = ffma(-L.l21, n1, ffma(-L.l20, n0, v2)) * L.r2;
    return -0.5f * ffma(n2, n2, ffma(n1, n1, n0 * n0));
}

// computeJacobians for one feature (core.cpp:682-704) in the form the restructured update consumes:
// predicted observation, Hf (Hv = [-Hf | (0,-1)^T]) and the symmetric Sf = Hf Pf Hf^T + R
struct Obs2 {
    float zp0, zp1;
    float hf00, hf01, hf10, hf11;
    float s00, s10, s11;
};

SLAM_DEV Obs2 observe2(float x, float y, float th, float fx, float fy, float f00, float f10, float f11, float r00, float r10,
                       float r11) {
    Obs2 o;
    const float dx = fx - x, dy = fy - y;
    const float d2 = ffma(dx, dx, dy * dy);
    const float rd = __builtin_amdgcn_rsqf(d2);
    const float rd2 = rd * rd;
    o.zp0 = d2 * rd;
    o.zp1 = atan2_poly(dy, dx) - th;
    o.hf00 = dx * rd;
    o.hf01 = dy * rd;
    o.hf10 = -dy * rd2;
    o.hf11 = dx * rd2;
    const float t00 = ffma(o.hf00, f00, o.hf01 * f10), t01 = ffma(o.hf00, f10, o.hf01 * f11);
    const float t10 = ffma(o.hf10, f00, o.hf11 * f10), t11 = ffma(o.hf10, f10, o.hf11 * f11);
    o.s00 = ffma(t00, o.hf00, ffma(t01, o.hf01, r00));
    o.s10 = ffma(t10, o.hf00, ffma(t11, o.hf01, r10));
    o.s11 = ffma(t10, o.hf10, ffma(t11, o.hf11, r11));
    return o;
}

// One refinement of the proposal N(xv, P) by a re-observed feature (fastslam2.cpp:320-349) in Kalman-gain form
SLAM_DEV void proposal_update(float &x, float &y, float &th, Sym3 &P, const Obs2 &o, float v0, float v1) {
    const float a0 = -o.hf00, a1 = -o.hf01, b0 = -o.hf10, b1 = -o.hf11;  // Hv rows: (a0 a1 0), (b0 b1 -1)
    // C = P Hv^T (3x2)
    const float c00 = ffma(P.p00, a0, P.p10 * a1), c01 = ffma(P.p00, b0, ffma(P.p10, b1, -P.p20));
    const float c10 = ffma(P.p10, a0, P.p11 * a1), c11 = ffma(P.p10, b0, ffma(P.p11, b1, -P.p21));
    const float c20 = ffma(P.p20, a0, P.p21 * a1), c21 = ffma(P.p20, b0, ffma(P.p21, b1, -P.p22));
    // S = Hv C + Sf (symmetric)
    const float s00 = ffma(a0, c00, ffma(a1, c10, o.s00));
    const float s10 = ffma(b0, c00, ffma(b1, c10, o.s10 - c20));
    const float s11 = ffma(b0, c01, ffma(b1, c11, o.s11 - c21));
    const float rdet = __builtin_amdgcn_rcpf(ffma(s00, s11, -s10 * s10));
    const float i00 = s11 * rdet, i10 = -s10 * rdet, i11 = s00 * rdet;
    // K = C S^-1
    const float k00 = ffma(c00, i00, c01 * i10), k01 = ffma(c00, i10, c01 * i11);
    const float k10 = ffma(c10, i00, c11 * i10), k11 = ffma(c10, i10, c11 * i11);
    const float k20 = ffma(c20, i00, c21 * i10), k21 = ffma(c20, i10, c21 * i11);
    x = ffma(k00, v0, ffma(k01, v1, x));
    y = ffma(k10, v0, ffma(k11, v1, y));
    th = ffma(k20, v0, ffma(k21, v1, th));
    // P -= K C^T
    P.p00 = ffma(-k00, c00, ffma(-k01, c01, P.p00));
    P.p10 = ffma(-k10, c00, ffma(-k11, c01, P.p10));
    P.p11 = ffma(-k10, c10, ffma(-k11, c11, P.p11));
    P.p20 = ffma(-k20, c00, ffma(-k21, c01, P.p20));
    P.p21 = ffma(-k20, c10, ffma(-k21, c11, P.p21));
    P.p22 = ffma(-k20, c20, ffma(-k21, c21, P.p22));
}

// likelihoodGivenXv term (fastslam2.cpp:370-400) + featureUpdate (core.cpp:132-175, :275-291) of one feature at the
// sampled pose; both use the same Sf.  Returns the parts of gaussEvaluate(v, Sf) = exp(-v^T Sf^-1 v / 2) / (2 pi sqrt(det Sf)).
// gaussEvaluate(v, S), D = 2 (fastslam2.cpp:127-163), closed form: exp(-v^T S^-1 v / 2) / (2 pi sqrt(det S)); also hands
// back S^-1 and S^-1 v for the feature update that follows
struct Gauss2 {
    float i00, i10, i11, u0, u1, E, norm;  // lik = exp(E) * norm
};

SLAM_DEV Gauss2 gauss2_parts(float s00, float s10, float s11, float v0, float v1) {
    Gauss2 g;
    const float det = ffma(s00, s11, -s10 * s10);
    const float rdet = __builtin_amdgcn_rcpf(det);
    g.i00 = s11 * rdet;
    g.i10 = -s10 * rdet;
    g.i11 = s00 * rdet;
    g.u0 = ffma(g.i00, v0, g.i10 * v1);
    g.u1 = ffma(g.i10, v0, g.i11 * v1);  // S^-1 v
    g.E = -0.5f * ffma(v0, g.u0, v1 * g.u1);
    g.norm = 0.15915494309189535f * __builtin_amdgcn_rsqf(det);
    return g;
}

SLAM_DEV Gauss2 feature_update2(float &fx, float &fy, float &f00, float &f10, float &f11, const Obs2 &o, float v0, float v1) {
    const Gauss2 g = gauss2_parts(o.s00, o.s10, o.s11, v0, v1);
    const float i00 = g.i00, i10 = g.i10, i11 = g.i11, u0 = g.u0, u1 = g.u1;
    // C = Pf Hf^T ; W = C Sf^-1 ; xf += W v = C (Sf^-1 v) ; Pf -= W C^T
    const float c00 = ffma(f00, o.hf00, f10 * o.hf01), c01 = ffma(f00, o.hf10, f10 * o.hf11);
    const float c10 = ffma(f10, o.hf00, f11 * o.hf01), c11 = ffma(f10, o.hf10, f11 * o.hf11);
    fx = ffma(c00, u0, ffma(c01, u1, fx));
    fy = ffma(c10, u0, ffma(c11, u1, fy));
    const float w00 = ffma(c00, i00, c01 * i10), w01 = ffma(c00, i10, c01 * i11);
    const float w10 = ffma(c10, i00, c11 * i10), w11 = ffma(c10, i10, c11 * i11);
    f00 = ffma(-w00, c00, ffma(-w01, c01, f00));
    f10 = ffma(-w10, c00, ffma(-w11, c01, f10));
    f11 = ffma(-w10, c10, ffma(-w11, c11, f11));
    return g;  // likelihood = exp(g.E) * g.norm
}

// addFeature (core.cpp:488-501) with the polynomial sincos
SLAM_DEV void add_feature_fast(float x, float y, float th, float r, float b, float r00, float r01, float r10, float r11,
                               float &fx, float &fy, float &p00, float &p10, float &p11) {
    float s, c;
    sincos_cw(th + b, s, c);
    fx = ffma(r, c, x);
    fy = ffma(r, s, y);
    const float g00 = c, g01 = -r * s, g10 = s, g11 = r * c;
    const float t00 = ffma(g00, r00, g01 * r10), t01 = ffma(g00, r01, g01 * r11);
    const float t10 = ffma(g10, r00, g11 * r10), t11 = ffma(g10, r01, g11 * r11);
    p00 = ffma(t00, g00, t01 * g01);
    p10 = ffma(t10, g00, t11 * g01);
    p11 = ffma(t10, g10, t11 * g11);
}

// Box-Muller with the hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in
// revolutions, which is exactly u).  Same pairing as box_muller3.
SLAM_DEV void box_muller3_fast(U4 r, float &g0, float &g1, float &g2) {
    const float k = -1.38629436111989062f;  // -2 ln 2
    float amp = __builtin_amdgcn_sqrtf(k * __builtin_amdgcn_logf(u01(r.x)));
    float u = u01(r.y);
    g0 = amp * __builtin_amdgcn_sinf(u);
    g1 = amp * __builtin_amdgcn_cosf(u);
    amp = __builtin_amdgcn_sqrtf(k * __builtin_amdgcn_logf(u01(r.z)));
    g2 = amp * __builtin_amdgcn_sinf(u01(r.w));
}
#endif  // SLAM_FAST_MATH

}  // namespace SLAM_KNS
