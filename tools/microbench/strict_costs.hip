// Static VALU instruction counts of the strict build's building blocks (round 6): every function of
// device_math.h the FastSLAM 2 step calls per landmark / per predict, alone in a kernel (inputs loaded, outputs stored), compiled
// with the strict build's flags to gfx950 assembly; tools/strict_costs.py counts the instructions per kernel and prints the table
// committed as profiles/strict_instruction_table_r06.txt.  Never run: only compiled (hipcc -S).
#define SLAM_KNS slam_strict
#include "../../slam_amd/csrc/device_math.h"
using namespace slam_strict;

#define K(name) extern "C" __global__ void k_##name(const float *__restrict__ in, float *__restrict__ out)
#define IN(k) in[(size_t) (k) * 4096 + threadIdx.x + blockIdx.x * 256]
#define OUT(k) out[(size_t) (k) * 4096 + threadIdx.x + blockIdx.x * 256]

K(baseline_16_in_8_out) {
    float a = 0;
    for (int k = 0; k < 16; k++) a += IN(k);   // (not counted: see the table's first row)
    for (int k = 0; k < 8; k++) OUT(k) = a;
}
K(ieee_div) { OUT(0) = fdiv(IN(0), IN(1)); }
K(ieee_rcp) { OUT(0) = frcp(IN(0)); }
K(ieee_sqrt) { OUT(0) = fsqrt(IN(0)); }
K(mul_add_chain_8) {   // eight dependent mul + add pairs: what "one multiply-add of the algebra" costs without contraction
    float a = IN(0);
    for (int k = 1; k <= 8; k++) a = a * IN(k) + IN(k + 8);
    OUT(0) = a;
}
K(trig_offset) { OUT(0) = trig_offset(IN(0)); }
K(atan2f) { OUT(0) = atan2f(IN(0), IN(1)); }
K(sincosf) {
    float s, c;
    sincosf(IN(0), &s, &c);
    OUT(0) = s;
    OUT(1) = c;
}
K(expf) { OUT(0) = expf(IN(0)); }
K(jacobian) {
    const Jac j = jacobian(IN(0), IN(1), IN(2), IN(3), IN(4), IN(5), IN(6), IN(7), IN(8), IN(9), IN(10), IN(11));
    OUT(0) = j.zp0; OUT(1) = j.zp1; OUT(2) = j.hf00; OUT(3) = j.hf01; OUT(4) = j.hf10; OUT(5) = j.hf11;
    OUT(6) = j.s00; OUT(7) = j.s01; OUT(8) = j.s10; OUT(9) = j.s11;
}
K(jacobian_algebra_only) {   // T = Hf Pf, S = T Hf^T + R: the part of jacobian() that is plain products and sums
    const float hf00 = IN(0), hf01 = IN(1), hf10 = IN(2), hf11 = IN(3), p00 = IN(4), p10 = IN(5), p11 = IN(6);
    float t00 = hf00 * p00 + hf01 * p10, t01 = hf00 * p10 + hf01 * p11, t10 = hf10 * p00 + hf11 * p10, t11 = hf10 * p10 + hf11 * p11;
    OUT(0) = (t00 * hf00 + t01 * hf01) + IN(7);
    OUT(1) = (t00 * hf10 + t01 * hf11) + IN(8);
    OUT(2) = (t10 * hf00 + t11 * hf01) + IN(9);
    OUT(3) = (t10 * hf10 + t11 * hf11) + IN(10);
}
K(inverse2) {
    float a, b, c, d;
    inverse2(IN(0), IN(1), IN(2), IN(3), a, b, c, d);
    OUT(0) = a; OUT(1) = b; OUT(2) = c; OUT(3) = d;
}
K(llt3_and_solve_identity) {
    float X[9];
    llt_solve_identity3(llt3(IN(0), IN(1), IN(2), IN(3), IN(4), IN(5)), X);
    for (int k = 0; k < 9; k++) OUT(k) = X[k];
}
K(gauss2) { OUT(0) = gauss2(IN(0), IN(1), IN(2), IN(3), IN(4)); }
K(gauss3) { OUT(0) = gauss3(IN(0), IN(1), IN(2), IN(3), IN(4), IN(5), IN(6), IN(7), IN(8)); }
K(cholesky_update2) {
    float fx = IN(0), fy = IN(1), p00 = IN(2), p10 = IN(3), p11 = IN(4);
    cholesky_update2(fx, fy, p00, p10, p11, IN(5), IN(6), IN(7), IN(8), IN(9), IN(10), IN(11), IN(12), IN(13), IN(14));
    OUT(0) = fx; OUT(1) = fy; OUT(2) = p00; OUT(3) = p10; OUT(4) = p11;
}
K(mvgauss3) {
    float x = IN(0), y = IN(1), t = IN(2);
    mvgauss3(x, y, t, llt3(IN(3), IN(4), IN(5), IN(6), IN(7), IN(8)), IN(9), IN(10), IN(11));
    OUT(0) = x; OUT(1) = y; OUT(2) = t;
}
// the proposal refinement of ONE landmark (update_step.inl, strict branch: fastslam2.cpp:315-350) as the step body writes it
K(first_pass_one_landmark) {
    float x = IN(0), y = IN(1), th = IN(2);
    float P[9];
    for (int k = 0; k < 9; k++) P[k] = IN(3 + k);
    Jac j = jacobian(x, y, th, IN(12), IN(13), IN(14), IN(15), IN(16), IN(17), IN(18), IN(19), IN(20));
    float s00, s01, s10, s11;
    inverse2(j.s00, j.s01, j.s10, j.s11, s00, s01, s10, s11);
    const float v0 = IN(21) - j.zp0, v1 = trig_offset(IN(22) - j.zp1);
    float Pinv[9];
    llt_solve_identity3(llt3(P[0], P[3], P[4], P[6], P[7], P[8]), Pinv);
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
    llt_solve_identity3(llt3(P[0], P[3], P[4], P[6], P[7], P[8]), P);
    float c[3];
    for (int r = 0; r < 3; r++) {
        const float a0 = P[3 * r] * j.hv00 + P[3 * r + 1] * j.hv01;
        const float a1 = (P[3 * r] * j.hv10 + P[3 * r + 1] * j.hv11) + P[3 * r + 2] * -1.0f;
        const float b0 = a0 * s00 + a1 * s10, b1 = a0 * s01 + a1 * s11;
        c[r] = b0 * v0 + b1 * v1;
    }
    OUT(0) = x + c[0]; OUT(1) = y + c[1]; OUT(2) = th + c[2];
    for (int k = 0; k < 9; k++) OUT(3 + k) = P[k];
}
// ... and its algebra alone (the products and sums between the solves): what packing can touch
K(first_pass_algebra_only) {
    const float hv00 = IN(0), hv01 = IN(1), hv10 = IN(2), hv11 = IN(3), s00 = IN(4), s01 = IN(5), s10 = IN(6), s11 = IN(7), v0 = IN(8), v1 = IN(9);
    float P[9], Pinv[9];
    for (int k = 0; k < 9; k++) Pinv[k] = IN(10 + k);
    const float t00 = hv00 * s00 + hv10 * s10, t01 = hv00 * s01 + hv10 * s11;
    const float t10 = hv01 * s00 + hv11 * s10, t11 = hv01 * s01 + hv11 * s11;
    const float t20 = -s10, t21 = -s11;
    P[0] = (t00 * hv00 + t01 * hv10) + Pinv[0];
    P[1] = (t00 * hv01 + t01 * hv11) + Pinv[1];
    P[2] = (-t01) + Pinv[2];
    P[3] = (t10 * hv00 + t11 * hv10) + Pinv[3];
    P[4] = (t10 * hv01 + t11 * hv11) + Pinv[4];
    P[5] = (-t11) + Pinv[5];
    P[6] = (t20 * hv00 + t21 * hv10) + Pinv[6];
    P[7] = (t20 * hv01 + t21 * hv11) + Pinv[7];
    P[8] = (-t21) + Pinv[8];
    float c[3];
    for (int r = 0; r < 3; r++) {
        const float a0 = P[3 * r] * hv00 + P[3 * r + 1] * hv01;
        const float a1 = (P[3 * r] * hv10 + P[3 * r + 1] * hv11) + P[3 * r + 2] * -1.0f;
        const float b0 = a0 * s00 + a1 * s10, b1 = a0 * s01 + a1 * s11;
        c[r] = b0 * v0 + b1 * v1;
    }
    for (int k = 0; k < 3; k++) OUT(k) = c[k];
    for (int k = 0; k < 9; k++) OUT(3 + k) = P[k];
}
// the likelihood / feature-update pass of ONE landmark (fastslam2.cpp:370-400, core.cpp:132-175, 275-291)
K(second_pass_one_landmark) {
    float4 la = make_float4(IN(3), IN(4), IN(5), IN(6));
    float lb = IN(7);
    Jac j = jacobian(IN(0), IN(1), IN(2), la.x, la.y, la.z, la.w, lb, IN(8), IN(9), IN(10), IN(11));
    const float v0 = IN(12) - j.zp0, v1 = trig_offset(IN(13) - j.zp1);
    OUT(0) = gauss2(v0, v1, j.s00, j.s10, j.s11);
    cholesky_update2(la.x, la.y, la.z, la.w, lb, v0, v1, IN(8), IN(9), IN(10), IN(11), j.hf00, j.hf01, j.hf10, j.hf11);
    OUT(1) = la.x; OUT(2) = la.y; OUT(3) = la.z; OUT(4) = la.w; OUT(5) = lb;
}
// one predictState of FastSLAM 2 without noise and heading observation (kernels.hip: predict_steps, fastslam2.cpp:70-105)
K(predict_one_step) {
    float x = IN(0), y = IN(1), th = IN(2);
    float P[9];
    for (int k = 0; k < 9; k++) P[k] = IN(3 + k);
    const float V = IN(12), G = IN(13), dt = IN(14), wb = IN(15), sinG = IN(16), cosG = IN(17), sinGw = IN(18);
    const float Q00 = IN(19), Q01 = IN(20), Q10 = IN(21), Q11 = IN(22);
    float sn, cs;
    sincosf(G + th, &sn, &cs);
    const float gv02 = -V * dt * sn, gv12 = V * dt * cs, gu00 = dt * cs, gu01 = -V * dt * sn, gu10 = dt * sn, gu11 = V * dt * cs;
    const float gu20 = dt * sinG / wb, gu21 = V * dt * cosG / wb;
    float T[9], N9[9];
    for (int c = 0; c < 3; c++) {
        T[c] = P[c] + gv02 * P[6 + c];
        T[3 + c] = P[3 + c] + gv12 * P[6 + c];
        T[6 + c] = P[6 + c];
    }
    for (int r = 0; r < 3; r++) {
        N9[3 * r] = T[3 * r] + T[3 * r + 2] * gv02;
        N9[3 * r + 1] = T[3 * r + 1] + T[3 * r + 2] * gv12;
        N9[3 * r + 2] = T[3 * r + 2];
    }
    const float u00 = gu00 * Q00 + gu01 * Q10, u01 = gu00 * Q01 + gu01 * Q11, u10 = gu10 * Q00 + gu11 * Q10, u11 = gu10 * Q01 + gu11 * Q11;
    const float u20 = gu20 * Q00 + gu21 * Q10, u21 = gu20 * Q01 + gu21 * Q11;
    OUT(3) = N9[0] + (u00 * gu00 + u01 * gu01); OUT(4) = N9[1] + (u00 * gu10 + u01 * gu11); OUT(5) = N9[2] + (u00 * gu20 + u01 * gu21);
    OUT(6) = N9[3] + (u10 * gu00 + u11 * gu01); OUT(7) = N9[4] + (u10 * gu10 + u11 * gu11); OUT(8) = N9[5] + (u10 * gu20 + u11 * gu21);
    OUT(9) = N9[6] + (u20 * gu00 + u21 * gu01); OUT(10) = N9[7] + (u20 * gu10 + u21 * gu11); OUT(11) = N9[8] + (u20 * gu20 + u21 * gu21);
    OUT(0) = x + V * dt * cs; OUT(1) = y + V * dt * sn; OUT(2) = trig_offset(th + V * dt * sinGw);
}
