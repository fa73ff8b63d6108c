/* TEST INFRASTRUCTURE — see slam_oracle.h.  Plain-C restatement of the reference algorithm,
 * float32, following the reference's (and vendored Eigen 3.1.3's) operation order so that it is
 * bit-comparable with the reference objects built with g++ -O2 -msse (x86-64 SSE2 scalar math).
 * Build with -ffp-contract=off and without -ffast-math (oracle/Makefile).
 *
 * Eigen evaluation-order facts this file relies on (all sizes here are Dynamic => "Large"):
 *  - MatrixXf*MatrixXf is a GEMM into a zeroed temporary; every coefficient is the k-ascending
 *    sequential sum  ((0 + a_i0*b_0j) + a_i1*b_1j) + ...  (GeneralBlockPanelKernel.h madd chain).
 *  - MatrixXf*VectorXf is a column-major GEMV: res_i += a_ik * (1*x_k), k ascending
 *    (GeneralMatrixVector.h:229-250 for < 4 columns, aligned operands => skipColumns = 0).
 *  - A.llt(): unblocked lower Cholesky (LLT.h:260-287): x = a_kk - |L_k,0..k-1|^2; sqrt; the
 *    sub-diagonal column is *multiplied by 1/x*.
 *  - llt().solve(I): column-oriented forward substitution with reciprocal diagonals, then
 *    row-oriented back substitution (TriangularSolverMatrix.h:109-137, one small panel).
 *  - A.inverse() / A.determinant() for dynamic sizes go through PartialPivLU
 *    (Inverse.h:22-28, Determinant.h) : unblocked_lu (PartialPivLU.h:239-283) then
 *    UnitLower / Upper column-major triangular solves of the permuted identity.
 *  - VectorXf::sum() is the 2-packet-unrolled SSE reduction of Redux.h:200-240 with the SSE2
 *    predux (a0+a2)+(a1+a3) (arch/SSE/PacketMath.h:406-410).
 *  - gaussEvaluate solves with JacobiSVD (SVD/JacobiSVD.h:704-790, Jacobi/Jacobi.h:80-108).
 */
#include "slam_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ============================================================================================
 * small dense helpers (row-major storage, Eigen operation order)
 * ========================================================================================== */

/* C(m x n) = A(m x k) * B(k x n); transA/transB read the stored matrix transposed. */
static void gemm(int m, int n, int k, const float *A, int lda, int transA, const float *B, int ldb, int transB,
                 float *C, int ldc) {
    float tmp[16 * 16];
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) {
            float acc = 0.0f;
            for (int l = 0; l < k; l++) {
                float a = transA ? A[l * lda + i] : A[i * lda + l];
                float b = transB ? B[j * ldb + l] : B[l * ldb + j];
                acc = acc + a * b;
            }
            tmp[i * n + j] = acc;
        }
    for (int i = 0; i < m; i++)
        for (int j = 0; j < n; j++) C[i * ldc + j] = tmp[i * n + j];
}

/* y(m) = A(m x k) * x(k)  (column-major GEMV order == k-ascending sequential per row) */
static void gemv(int m, int k, const float *A, int lda, const float *x, float *y) {
    float tmp[16];
    for (int i = 0; i < m; i++) {
        float acc = 0.0f;
        for (int l = 0; l < k; l++) acc = acc + A[i * lda + l] * x[l];
        tmp[i] = acc;
    }
    for (int i = 0; i < m; i++) y[i] = tmp[i];
}

int orc_llt_lower(int n, const float *A, float *L) {
    /* LLT.h:260-287.  Works in place on a full copy; on failure at step k the remaining columns keep the
     * input values (Eigen returns NumericalIssue but callers here never look at it). */
    float M[9];
    for (int i = 0; i < n * n; i++) M[i] = A[i];
    int info = -1;
    for (int k = 0; k < n; k++) {
        float x = M[k * n + k];
        if (k > 0) {
            float sq = M[k * n + 0] * M[k * n + 0];
            for (int j = 1; j < k; j++) sq = sq + M[k * n + j] * M[k * n + j];
            x = x - sq;
        }
        if (x <= 0.0f) {
            info = k;
            break;
        }
        x = sqrtf(x);
        M[k * n + k] = x;
        int rs = n - k - 1;
        if (k > 0 && rs > 0) {
            /* A21 -= A20 * A10^T : GEMV with alpha=-1 folded into the vector operand */
            for (int i = k + 1; i < n; i++)
                for (int j = 0; j < k; j++) M[i * n + k] = M[i * n + k] + M[i * n + j] * (-1.0f * M[k * n + j]);
        }
        if (rs > 0) {
            float r = 1.0f / x;
            for (int i = k + 1; i < n; i++) M[i * n + k] = M[i * n + k] * r;
        }
    }
    /* matrixL(): lower triangular view, strict upper part reads as zero */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) L[i * n + j] = (j <= i) ? M[i * n + j] : 0.0f;
    return info;
}

/* in-place solves used by llt().solve and PartialPivLU::solve; B is n x nc row-major */
static void solve_lower_colmajor(int n, const float *T, int unit, float *B, int nc) {
    /* TriangularSolverMatrix.h:109-137, ColMajor branch, IsLower */
    for (int k = 0; k < n; k++) {
        float a = unit ? 1.0f : 1.0f / T[k * n + k];
        for (int j = 0; j < nc; j++) {
            float b = (B[k * nc + j] *= a);
            for (int i = k + 1; i < n; i++) B[i * nc + j] -= b * T[i * n + k];
        }
    }
}

static void solve_upper_colmajor(int n, const float *T, float *B, int nc) {
    /* ColMajor branch, !IsLower: i runs n-1 .. 0, eliminates the rows above */
    for (int k = 0; k < n; k++) {
        int i = n - k - 1;
        float a = 1.0f / T[i * n + i];
        for (int j = 0; j < nc; j++) {
            float b = (B[i * nc + j] *= a);
            for (int r = 0; r < i; r++) B[r * nc + j] -= b * T[r * n + i];
        }
    }
}

static void solve_upper_rowmajor(int n, const float *U, float *B, int nc) {
    /* RowMajor branch (matrixU() = adjoint of the col-major L), !IsLower */
    for (int k = 0; k < n; k++) {
        int i = n - k - 1;
        float a = 1.0f / U[i * n + i];
        for (int j = 0; j < nc; j++) {
            float b = 0.0f;
            for (int c = 0; c < k; c++) b += U[i * n + (i + 1 + c)] * B[(i + 1 + c) * nc + j];
            B[i * nc + j] = (B[i * nc + j] - b) * a;
        }
    }
}

void orc_llt_solve_identity(int n, const float *A, float *X) {
    float L[9], U[9];
    orc_llt_lower(n, A, L);
    /* NB: on LLT failure Eigen's m_matrix keeps the partially factored data; its strictly-upper part is
     * never read by matrixL()/matrixU(), so using L (zeros above) is the same computation. */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            U[i * n + j] = L[j * n + i];
            X[i * n + j] = (i == j) ? 1.0f : 0.0f;
        }
    solve_lower_colmajor(n, L, 0, X, n);
    solve_upper_rowmajor(n, U, X, n);
}

static int lu_factor(int n, float *M, int *perm, int *det_p) {
    /* PartialPivLU.h:239-283 unblocked_lu */
    int nt = 0;
    for (int k = 0; k < n; k++) {
        int piv = k;
        float big = fabsf(M[k * n + k]);
        for (int i = k + 1; i < n; i++) {
            float v = fabsf(M[i * n + k]);
            if (v > big) {
                big = v;
                piv = i;
            }
        }
        perm[k] = piv;
        if (big != 0.0f) {
            if (piv != k) {
                for (int j = 0; j < n; j++) {
                    float t = M[k * n + j];
                    M[k * n + j] = M[piv * n + j];
                    M[piv * n + j] = t;
                }
                nt++;
            }
            /* "col /= pivot" is a multiply by the reciprocal in Eigen 3.1 (Core/SelfCwiseBinaryOp.h operator/=) */
            float rp = 1.0f / M[k * n + k];
            for (int i = k + 1; i < n; i++) M[i * n + k] = M[i * n + k] * rp;
        }
        if (k < n - 1)
            for (int i = k + 1; i < n; i++)
                for (int j = k + 1; j < n; j++) M[i * n + j] -= M[i * n + k] * M[k * n + j];
    }
    *det_p = (nt % 2) ? -1 : 1;
    return 0;
}

void orc_lu_inverse(int n, const float *A, float *X) {
    float M[9];
    int perm[3], det_p;
    for (int i = 0; i < n * n; i++) M[i] = A[i];
    lu_factor(n, M, perm, &det_p);
    /* dst = P * I : apply the row transpositions in order to the identity */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) X[i * n + j] = (i == j) ? 1.0f : 0.0f;
    for (int k = 0; k < n; k++)
        if (perm[k] != k)
            for (int j = 0; j < n; j++) {
                float t = X[k * n + j];
                X[k * n + j] = X[perm[k] * n + j];
                X[perm[k] * n + j] = t;
            }
    solve_lower_colmajor(n, M, 1, X, n);
    solve_upper_colmajor(n, M, X, n);
}

float orc_lu_determinant(int n, const float *A) {
    float M[9];
    int perm[3], det_p;
    for (int i = 0; i < n * n; i++) M[i] = A[i];
    lu_factor(n, M, perm, &det_p);
    float prod = M[0];
    for (int i = 1; i < n; i++) prod = prod * M[i * n + i];
    return (float) det_p * prod;
}

/* ---- JacobiSVD (SVD/JacobiSVD.h:704-790) for real square n<=3, then the pseudo-inverse solve ---- */
typedef struct {
    float c, s;
} rot_t;

static int make_jacobi(rot_t *r, float x, float y, float z) {
    /* Jacobi/Jacobi.h:80-108 */
    if (y == 0.0f) {
        r->c = 1.0f;
        r->s = 0.0f;
        return 0;
    }
    float tau = (x - z) / (2.0f * fabsf(y));
    float w = sqrtf(tau * tau + 1.0f);
    float t;
    if (tau > 0.0f)
        t = 1.0f / (tau + w);
    else
        t = 1.0f / (tau - w);
    float sign_t = t > 0.0f ? 1.0f : -1.0f;
    float n = 1.0f / sqrtf(t * t + 1.0f);
    r->s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
    r->c = n;
    return 1;
}

/* apply_rotation_in_the_plane(x, y, j): x' = c x + s y ; y' = -s x + c y */
static void rot_apply(float *x, int incx, float *y, int incy, int n, rot_t j) {
    for (int i = 0; i < n; i++) {
        float xi = x[i * incx], yi = y[i * incy];
        x[i * incx] = j.c * xi + j.s * yi;
        y[i * incy] = -j.s * xi + j.c * yi;
    }
}

static void svd_solve(int n, const float *A, const float *b, float *x) {
    float W[9], U[9], V[9], sv[3];
    for (int i = 0; i < n * n; i++) {
        W[i] = A[i];
        U[i] = V[i] = 0.0f;
    }
    for (int i = 0; i < n; i++) U[i * n + i] = V[i * n + i] = 1.0f;
    const float precision = 2.0f * 1.1920929e-07f;
    const float considerAsZero = 2.0f * 1.40129846e-45f;
    int finished = 0;
    while (!finished) {
        finished = 1;
        for (int p = 1; p < n; p++)
            for (int q = 0; q < p; q++) {
                float mx = fmaxf(fabsf(W[p * n + p]), fabsf(W[q * n + q]));
                float threshold = fmaxf(considerAsZero, precision * mx);
                float off = fmaxf(fabsf(W[p * n + q]), fabsf(W[q * n + p]));
                if (off > threshold) {
                    finished = 0;
                    /* real_2x2_jacobi_svd (JacobiSVD.h:395-421) */
                    float m00 = W[p * n + p], m01 = W[p * n + q], m10 = W[q * n + p], m11 = W[q * n + q];
                    rot_t rot1, jr, jl;
                    float t = m00 + m11;
                    float d = m10 - m01;
                    if (t == 0.0f) {
                        rot1.c = 0.0f;
                        rot1.s = d > 0.0f ? 1.0f : -1.0f;
                    } else {
                        float u = d / t;
                        rot1.c = 1.0f / sqrtf(1.0f + u * u);
                        rot1.s = rot1.c * u;
                    }
                    /* m.applyOnTheLeft(0,1,rot1) */
                    {
                        float r0[2] = {m00, m01}, r1[2] = {m10, m11};
                        rot_apply(r0, 1, r1, 1, 2, rot1);
                        m00 = r0[0];
                        m01 = r0[1];
                        m10 = r1[0];
                        m11 = r1[1];
                    }
                    make_jacobi(&jr, m00, m01, m11);
                    /* j_left = rot1 * j_right.transpose()  (Jacobi.h:52-56) */
                    {
                        rot_t o = {jr.c, -jr.s};
                        jl.c = rot1.c * o.c - rot1.s * o.s;
                        jl.s = rot1.c * o.s + rot1.s * o.c;
                    }
                    /* workMatrix.applyOnTheLeft(p,q,j_left): rows p,q */
                    rot_apply(&W[p * n], 1, &W[q * n], 1, n, jl);
                    /* U.applyOnTheRight(p,q,j_left.transpose()): columns p,q rotated by (j^T)^T = j */
                    {
                        rot_t jt = {jl.c, -jl.s};
                        rot_t jtt = {jt.c, -jt.s};
                        rot_apply(&U[p], n, &U[q], n, n, jtt);
                    }
                    /* workMatrix.applyOnTheRight(p,q,j_right): columns rotated by j_right^T */
                    {
                        rot_t jt = {jr.c, -jr.s};
                        rot_apply(&W[p], n, &W[q], n, n, jt);
                        rot_apply(&V[p], n, &V[q], n, n, jt);
                    }
                }
            }
    }
    for (int i = 0; i < n; i++) {
        float a = fabsf(W[i * n + i]);
        sv[i] = a;
        if (a != 0.0f) {
            float f = W[i * n + i] / a;
            for (int r = 0; r < n; r++) U[r * n + i] *= f;
        }
    }
    int nonzero = n;
    for (int i = 0; i < n; i++) {
        int pos = 0;
        float best = sv[i];
        for (int j = i + 1; j < n; j++)
            if (sv[j] > best) {
                best = sv[j];
                pos = j - i;
            }
        if (best == 0.0f) {
            nonzero = i;
            break;
        }
        if (pos) {
            pos += i;
            float t = sv[i];
            sv[i] = sv[pos];
            sv[pos] = t;
            for (int r = 0; r < n; r++) {
                t = U[r * n + pos];
                U[r * n + pos] = U[r * n + i];
                U[r * n + i] = t;
                t = V[r * n + pos];
                V[r * n + pos] = V[r * n + i];
                V[r * n + i] = t;
            }
        }
    }
    /* dst = V * diag(inv) * U^T * rhs, evaluated as ((V*diag) * U^T) * rhs */
    float inv[3], VD[9], PI[9];
    for (int i = 0; i < n; i++) inv[i] = (i < nonzero) ? 1.0f / sv[i] : 0.0f;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) VD[i * n + j] = V[i * n + j] * inv[j];
    gemm(n, n, n, VD, n, 0, U, n, 1, PI, n);
    gemv(n, n, PI, n, b, x);
}

/* ============================================================================================
 * scalar pieces
 * ========================================================================================== */

float orc_trig_offset(float ang) {
    /* core.cpp:460-477 — double constants against a float argument */
    int n;
    if ((ang < -2 * M_PI) || (ang > 2 * M_PI)) {
        n = (int) floor(ang / (2 * M_PI));
        ang = (float) (ang - n * (2 * M_PI));
    }
    if (ang > M_PI) ang = (float) (ang - (2 * M_PI));
    if (ang < -M_PI) ang = (float) (ang + (2 * M_PI));
    return ang;
}

/* one feature of computeJacobians (core.cpp:682-704) */
static void jac1(const float *xv, const float *R, const float *xf, const float *Pf, float *zp, float *Hv, float *Hf,
                 float *Sf) {
    float dx = xf[0] - xv[0];
    float dy = xf[1] - xv[1];
    float d2 = (float) (pow((double) dx, 2) + pow((double) dy, 2));
    float d = sqrtf(d2);
    zp[0] = d;
    zp[1] = orc_trig_offset(atan2f(dy, dx) - xv[2]);
    Hv[0] = -dx / d;
    Hv[1] = -dy / d;
    Hv[2] = 0;
    Hv[3] = dy / d2;
    Hv[4] = -dx / d2;
    Hv[5] = -1;
    Hf[0] = dx / d;
    Hf[1] = dy / d;
    Hf[2] = -dy / d2;
    Hf[3] = dx / d2;
    float T[4];
    gemm(2, 2, 2, Hf, 2, 0, Pf, 2, 0, T, 2);
    gemm(2, 2, 2, T, 2, 0, Hf, 2, 1, Sf, 2);
    for (int i = 0; i < 4; i++) Sf[i] = Sf[i] + R[i];
}

void orc_compute_jacobians(const float *xv, const float *R4, const float *xf, const float *Pf4, int n, float *zp,
                           float *Hv6, float *Hf4, float *Sf4) {
    for (int i = 0; i < n; i++) jac1(xv, R4, xf + 2 * i, Pf4 + 4 * i, zp + 2 * i, Hv6 + 6 * i, Hf4 + 4 * i, Sf4 + 4 * i);
}

float orc_gauss_evaluate(const float *v, const float *S, int D, int logflag) {
    /* fastslam2.cpp:127-163 */
    float Sc[9], nin[3];
    orc_llt_lower(D, S, Sc);
    svd_solve(D, Sc, v, nin);
    float E = 0;
    for (int s = 0; s < D; s++) {
        nin[s] = (float) pow((double) nin[s], 2);
        E += nin[s];
    }
    E = (float) (-0.5 * E);
    float C, w;
    if (logflag != 1) {
        float prod = 1;
        for (int i = 0; i < D; i++) prod = prod * Sc[i * D + i];
        C = (float) (pow((2 * M_PI), (D / 2)) * prod); /* integer D/2, as upstream */
        w = expf(E) / C;
    } else {
        float sum = 0;
        for (int i = 0; i < D; i++) sum += logf(Sc[i * D + i]);
        C = (float) (0.5 * D * log(2 * M_PI) + sum);
        w = E - C;
    }
    return w;
}

void orc_cholesky_update2(float *x, float *P4, const float *v, const float *R4, const float *H4) {
    /* core.cpp:275-291, 2x2 */
    float PHt[4], S[4], St[4], L[4], U[4], Ui[4], W1[4], W[4], Wv[2], WW[4];
    gemm(2, 2, 2, P4, 2, 0, H4, 2, 1, PHt, 2);
    gemm(2, 2, 2, H4, 2, 0, PHt, 2, 0, S, 2);
    for (int i = 0; i < 4; i++) S[i] = S[i] + R4[i];
    St[0] = (S[0] + S[0]) * 0.5f;
    St[1] = (S[1] + S[2]) * 0.5f;
    St[2] = (S[2] + S[1]) * 0.5f;
    St[3] = (S[3] + S[3]) * 0.5f;
    orc_llt_lower(2, St, L);
    U[0] = L[0];
    U[1] = L[2];
    U[2] = 0.0f;
    U[3] = L[3];
    orc_lu_inverse(2, U, Ui);
    gemm(2, 2, 2, PHt, 2, 0, Ui, 2, 0, W1, 2);
    gemm(2, 2, 2, W1, 2, 0, Ui, 2, 1, W, 2);
    gemv(2, 2, W, 2, v, Wv);
    x[0] = x[0] + Wv[0];
    x[1] = x[1] + Wv[1];
    gemm(2, 2, 2, W1, 2, 0, W1, 2, 1, WW, 2);
    for (int i = 0; i < 4; i++) P4[i] = P4[i] - WW[i];
}

void orc_observe_heading(float *xv, float *Pv9, float phi, float sigmaPhi) {
    /* fastslam2.cpp:113-125 -> core.cpp:294-317 with H = [0 0 1] */
    const float H[3] = {0, 0, 1};
    float v = orc_trig_offset(phi - xv[2]);
    float R = (float) pow((double) sigmaPhi, 2);
    float PHt[3];
    gemv(3, 3, Pv9, 3, H, PHt);
    float S = 0.0f;
    for (int k = 0; k < 3; k++) S = S + H[k] * PHt[k];
    S = S + 1.0f * R;
    float Si = 1.0f * (1.0f / S); /* 1x1 PartialPivLU inverse */
    float W[3];
    for (int i = 0; i < 3; i++) W[i] = PHt[i] * Si;
    for (int i = 0; i < 3; i++) xv[i] = xv[i] + W[i] * v;
    float C[9], CP[9], CPC[9], WRW[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C[i * 3 + j] = ((i == j) ? 1.0f : 0.0f) - H[j] * W[i];
    gemm(3, 3, 3, C, 3, 0, Pv9, 3, 0, CP, 3);
    gemm(3, 3, 3, CP, 3, 0, C, 3, 1, CPC, 3);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) WRW[i * 3 + j] = W[j] * (W[i] * R);
    float eps = (float) (2.2204 * pow(10.0, -16));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float p = CPC[i * 3 + j] + WRW[i * 3 + j];
            Pv9[i * 3 + j] = p + ((i == j) ? 1.0f : 0.0f) * eps;
        }
}

void orc_add_feature(const float *xv, const float *zn, int n, const float *R4, float *xf, float *Pf4) {
    /* core.cpp:479-509 */
    for (int i = 0; i < n; i++) {
        float r = zn[2 * i], b = zn[2 * i + 1];
        float s = sinf(xv[2] + b);
        float c = cosf(xv[2] + b);
        xf[2 * i] = xv[0] + r * c;
        xf[2 * i + 1] = xv[1] + r * s;
        float Gz[4] = {c, -r * s, s, r * c};
        float T[4];
        gemm(2, 2, 2, Gz, 2, 0, R4, 2, 0, T, 2);
        gemm(2, 2, 2, T, 2, 0, Gz, 2, 1, Pf4 + 4 * i, 2);
    }
}

void orc_multivariate_gauss(const float *x, const float *P, int D, const float *g, float *out) {
    /* core.cpp:452-458 : S = P.llt().matrixL(); S*X + x */
    float L[9], Sx[3];
    orc_llt_lower(D, P, L);
    gemv(D, D, L, D, g, Sx);
    for (int i = 0; i < D; i++) out[i] = Sx[i] + x[i];
}

void orc_fs2_predict_state(float *xv, float *Pv9, float V, float G, const float *Q4, float wheelBase, float dt,
                           const float *noise2) {
    /* fastslam2.cpp:70-105 */
    float phi = xv[2];
    float Gv[9] = {1, 0, -V * dt * sinf(G + phi), 0, 1, V * dt * cosf(G + phi), 0, 0, 1};
    float Gu[6] = {dt * cosf(G + phi),     -V * dt * sinf(G + phi), dt * sinf(G + phi),
                   V * dt * cosf(G + phi), dt * sinf(G) / wheelBase, V * dt * cosf(G) / wheelBase};
    float T[9], A[9], T2[6], B[9];
    gemm(3, 3, 3, Gv, 3, 0, Pv9, 3, 0, T, 3);
    gemm(3, 3, 3, T, 3, 0, Gv, 3, 1, A, 3);
    gemm(3, 2, 2, Gu, 2, 0, Q4, 2, 0, T2, 2);
    gemm(3, 3, 2, T2, 2, 0, Gu, 2, 1, B, 3);
    for (int i = 0; i < 9; i++) Pv9[i] = A[i] + B[i];
    if (noise2) {
        float Am[2] = {V, G}, VG[2];
        orc_multivariate_gauss(Am, Q4, 2, noise2, VG);
        V = VG[0];
        G = VG[1];
    }
    float x0 = xv[0] + V * dt * cosf(G + xv[2]);
    float x1 = xv[1] + V * dt * sinf(G + xv[2]);
    float x2 = orc_trig_offset(xv[2] + V * dt * sinf(G / wheelBase)); /* sin(G/wheelBase): upstream quirk */
    xv[0] = x0;
    xv[1] = x1;
    xv[2] = x2;
}

void orc_fs1_predict_state(float *xv, float V, float G, const float *Q4, float wheelBase, float dt,
                           const float *noise2) {
    /* fastslam1.cpp:37-54 */
    if (noise2) {
        float Am[2] = {V, G}, VG[2];
        orc_multivariate_gauss(Am, Q4, 2, noise2, VG);
        V = VG[0];
        G = VG[1];
    }
    float x0 = xv[0] + V * dt * cosf(G + xv[2]);
    float x1 = xv[1] + V * dt * sinf(G + xv[2]);
    float x2 = orc_trig_offset(xv[2] + V * dt * sinf(G / wheelBase));
    xv[0] = x0;
    xv[1] = x1;
    xv[2] = x2;
}

/* logw: the LOG of the same product, each factor entering as q - log(den) (log-weight extension, see orc_particles) */
static double fs1_compute_weight(const float *xv, const float *xf, const float *Pf4, const float *zf, const int *idf,
                                 int m, const float *R4, int logw) {
    /* fastslam1.cpp:91-118 */
    float w = 1.0f;
    double dl = 0.0;
    for (int i = 0; i < m; i++) {
        float zp[2], Hv[6], Hf[4], S[4], Si[4], v[2];
        jac1(xv, R4, xf + 2 * idf[i], Pf4 + 4 * idf[i], zp, Hv, Hf, S);
        v[0] = zf[2 * i] - zp[0];
        v[1] = orc_trig_offset(zf[2 * i + 1] - zp[1]);
        float den = (float) (2 * M_PI * sqrtf(orc_lu_determinant(2, S)));
        orc_lu_inverse(2, S, Si);
        /* (-0.5 v^T) * Sinv : row-vector GEMV, alpha = -0.5 applied to each finished dot product */
        float t0 = -0.5f * ((0.0f + v[0] * Si[0]) + v[1] * Si[2]);
        float t1 = -0.5f * ((0.0f + v[0] * Si[1]) + v[1] * Si[3]);
        float q = t0 * v[0] + t1 * v[1];
        if (logw) {
            dl += (double) (q - logf(den));
        } else {
            float num = expf(q);
            w = w * num / den;
        }
    }
    return logw ? dl : (double) w;
}

float orc_fs1_compute_weight(const float *xv, const float *xf, const float *Pf4, const float *zf, const int *idf,
                             int m, const float *R4) {
    return (float) fs1_compute_weight(xv, xf, Pf4, zf, idf, m, R4, 0);
}

void orc_feature_update(const float *xv, float *xf, float *Pf4, const float *zf, const int *idf, int m,
                        const float *R4) {
    /* core.cpp:132-175 : all Jacobians first (batch), then the per-feature Cholesky updates */
    for (int i = 0; i < m; i++) {
        float zp[2], Hv[6], Hf[4], Sf[4], v[2];
        float *x = xf + 2 * idf[i], *P = Pf4 + 4 * idf[i];
        jac1(xv, R4, x, P, zp, Hv, Hf, Sf);
        v[0] = zf[2 * i] - zp[0];
        v[1] = orc_trig_offset(zf[2 * i + 1] - zp[1]);
        orc_cholesky_update2(x, P, v, R4, Hf);
    }
}

/* logw: *pw is a LOG-weight and every factor enters through gaussEvaluate's own logflag = 1 branch
 * (fastslam2.cpp:154-160); the sum of the m log-likelihoods is kept in double */
static void fs2_sample_proposal(float *pxv, float *pPv, float *pw, const float *xf, const float *Pf4, const float *zf,
                                const int *idf, int m, const float *R4, const float *g3, int logw) {
    /* fastslam2.cpp:290-368 */
    float xv[3], Pv[9], xv0[3], Pv0[9];
    memcpy(xv, pxv, sizeof xv);
    memcpy(Pv, pPv, sizeof Pv);
    memcpy(xv0, xv, sizeof xv);
    memcpy(Pv0, Pv, sizeof Pv);
    for (int i = 0; i < m; i++) {
        float zp[2], Hv[6], Hf[4], Sf[4], Sfi[4], vi[2];
        /* Jacobians at the particle's *current* pose = running proposal mean (:320,:348) */
        jac1(pxv, R4, xf + 2 * idf[i], Pf4 + 4 * idf[i], zp, Hv, Hf, Sf);
        orc_lu_inverse(2, Sf, Sfi);
        vi[0] = zf[2 * i] - zp[0];
        vi[1] = orc_trig_offset(zf[2 * i + 1] - zp[1]);
        float Pinv[9], T1[6], T2[9];
        orc_llt_solve_identity(3, Pv, Pinv);
        gemm(3, 2, 2, Hv, 3, 1, Sfi, 2, 0, T1, 2);
        gemm(3, 3, 2, T1, 2, 0, Hv, 3, 0, T2, 3);
        for (int k = 0; k < 9; k++) Pv[k] = T2[k] + Pinv[k];
        orc_llt_solve_identity(3, Pv, Pv);
        float A[6], B[6], c[3];
        gemm(3, 2, 3, Pv, 3, 0, Hv, 3, 1, A, 2);
        gemm(3, 2, 2, A, 2, 0, Sfi, 2, 0, B, 2);
        gemv(3, 2, B, 2, vi, c);
        for (int k = 0; k < 3; k++) xv[k] = xv[k] + c[k];
        memcpy(pxv, xv, sizeof xv);
        memcpy(pPv, Pv, sizeof Pv);
    }
    float xvs[3];
    orc_multivariate_gauss(xv, Pv, 3, g3, xvs);
    memcpy(pxv, xvs, sizeof xvs);
    memset(pPv, 0, 9 * sizeof(float));
    float v1[3], v2[3];
    for (int k = 0; k < 3; k++) {
        v1[k] = xv0[k] - xvs[k];
        v2[k] = xv[k] - xvs[k];
    }
    v1[2] = orc_trig_offset(v1[2]);
    v2[2] = orc_trig_offset(v2[2]);
    /* likelihoodGivenXv (fastslam2.cpp:370-400) at the sampled pose */
    float lik = 1;
    double dl = 0.0;
    for (int i = 0; i < m; i++) {
        float zp[2], Hv[6], Hf[4], Sf[4], v[2];
        jac1(pxv, R4, xf + 2 * idf[i], Pf4 + 4 * idf[i], zp, Hv, Hf, Sf);
        v[0] = zf[2 * i] - zp[0];
        v[1] = zf[2 * i + 1] - zp[1];
        v[1] = orc_trig_offset(v[1]);
        if (logw) dl += (double) orc_gauss_evaluate(v, Sf, 2, 1);
        else lik = lik * orc_gauss_evaluate(v, Sf, 2, 0);
    }
    if (logw) {
        float prior = orc_gauss_evaluate(v1, Pv0, 3, 1);
        float proposal = orc_gauss_evaluate(v2, Pv, 3, 1);
        *pw = (float) (((double) *pw + dl) + ((double) prior - (double) proposal));
    } else {
        float prior = orc_gauss_evaluate(v1, Pv0, 3, 0);
        float proposal = orc_gauss_evaluate(v2, Pv, 3, 0);
        *pw = *pw * lik * prior / proposal;
    }
}

void orc_fs2_sample_proposal(float *pxv, float *pPv, float *pw, const float *xf, const float *Pf4, const float *zf,
                             const int *idf, int m, const float *R4, const float *g3) {
    fs2_sample_proposal(pxv, pPv, pw, xf, Pf4, zf, idf, m, R4, g3, 0);
}

/* ============================================================================================
 * libc rand() tape, reference draw order
 * ========================================================================================== */

void orc_srand(unsigned seed) { srand(seed); }

void orc_randn(int m, int n, float *out) {
    /* core.cpp:383-419 */
    int urows = m * n + 1;
    float *u = (float *) malloc(sizeof(float) * (size_t) urows);
    for (int r = 0; r < urows; r++) u[r] = (float) (rand() * 1.0 / RAND_MAX);
    float square, amp = 0, angle = 0;
    for (int k = 0; k < m * n; k++) {
        if (k % 2 == 0) {
            square = (float) (-2. * logf(u[k]));
            if (square < 0.) square = 0.;
            amp = sqrtf(square);
            angle = (float) (2. * M_PI * u[k + 1]);
            out[k] = amp * sinf(angle);
        } else {
            out[k] = amp * cosf(angle);
        }
    }
    free(u);
}

int orc_stratified_random(int N, float *sel) {
    /* core.cpp:751-769 */
    float k = (float) (1.0 / (float) N);
    float temp = k / 2;
    int cnt = 0;
    while (temp < (1 - k / 2)) {
        if (cnt < N) sel[cnt] = temp;
        cnt++;
        temp = temp + k;
    }
    if (cnt == N) {
        for (int i = 0; i < N; i++) {
            double u = rand() / (double) RAND_MAX;
            sel[i] = (float) (sel[i] + u * k - (k / 2));
        }
    } else {
        /* the reference asserts here (core.cpp:762); the build's definition for such N */
        for (int i = 0; i < N; i++) {
            double u = rand() / (double) RAND_MAX;
            sel[i] = (float) (((double) i + u) / (double) N);
        }
    }
    return cnt;
}

float orc_eigen_sum(const float *v, int n) {
    /* Redux.h:200-240 (aligned start 0) + SSE2 predux */
    int aligned2 = (n / 8) * 8, aligned = (n / 4) * 4;
    float res;
    if (aligned) {
        float p0[4] = {v[0], v[1], v[2], v[3]};
        if (aligned > 4) {
            float p1[4] = {v[4], v[5], v[6], v[7]};
            for (int i = 8; i < aligned2; i += 8)
                for (int l = 0; l < 4; l++) {
                    p0[l] = p0[l] + v[i + l];
                    p1[l] = p1[l] + v[i + 4 + l];
                }
            for (int l = 0; l < 4; l++) p0[l] = p0[l] + p1[l];
            if (aligned > aligned2)
                for (int l = 0; l < 4; l++) p0[l] = p0[l] + v[aligned2 + l];
        }
        res = (p0[0] + p0[2]) + (p0[1] + p0[3]);
        for (int i = aligned; i < n; i++) res = res + v[i];
    } else {
        res = v[0];
        for (int i = 1; i < n; i++) res = res + v[i];
    }
    return res;
}

void orc_stratified_resample(const float *win, int N, const float *sel, int *keep, float *neff) {
    /* core.cpp:780-807 (+ cumulativeSum :813-824, whose restart-from-zero prefix sums are the same float
     * operation sequence as a running sum) */
    float *w = (float *) malloc(sizeof(float) * (size_t) N);
    float *sq = (float *) malloc(sizeof(float) * (size_t) N);
    float wSum = orc_eigen_sum(win, N);
    for (int i = 0; i < N; i++) {
        w[i] = win[i] / wSum;
        sq[i] = (float) pow((double) w[i], 2);
    }
    *neff = 1 / orc_eigen_sum(sq, N);
    for (int i = 0; i < N; i++) keep[i] = -1;
    float run = 0;
    for (int i = 0; i < N; i++) {
        run += w[i];
        w[i] = run;
    }
    int ctr = 0;
    for (int i = 0; i < N; i++)
        while ((ctr < N) && (sel[ctr] < w[i])) {
            keep[ctr] = i;
            ctr++;
        }
    free(w);
    free(sq);
}

/* ============================================================================================
 * Philox4x32-10 — the build's counter-based RNG (throughput mode), mirrored bit-for-bit on device
 * ========================================================================================== */

void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t) 0xD2511F53u * c0;
        uint64_t p1 = (uint64_t) 0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t) (p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t) p1;
        uint32_t n2 = (uint32_t) (p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t) p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

static inline float u01(uint32_t x) { return ((float) (x >> 8) + 0.5f) * (1.0f / 16777216.0f); } /* (0,1) */

static void box_muller3(const uint32_t r[4], float g[3]) {
    /* same pairing as nRandMat::randn(3,1): (u0,u1) -> g0 (sin), g1 (cos); (u2,u3) -> g2 (sin) */
    float amp = sqrtf(-2.0f * logf(u01(r[0])));
    float ang = 6.28318530717958647692f * u01(r[1]);
    g[0] = amp * sinf(ang);
    g[1] = amp * cosf(ang);
    amp = sqrtf(-2.0f * logf(u01(r[2])));
    ang = 6.28318530717958647692f * u01(r[3]);
    g[2] = amp * sinf(ang);
}

void orc_philox_update_tape(uint64_t seed, uint32_t step, int first, int count, int Ntotal, float *normals,
                            float *sel) {
    uint32_t k0 = (uint32_t) seed, k1 = (uint32_t) (seed >> 32);
    for (int i = 0; i < count; i++) {
        uint32_t r[4];
        uint32_t gid = (uint32_t) (first + i);
        if (normals) {
            orc_philox4x32(gid, step, 0u, 0u, k0, k1, r);
            box_muller3(r, normals + 3 * i);
        }
        if (sel) {
            orc_philox4x32(gid, step, 1u, 0u, k0, k1, r);
            double u = ((double) (r[0] >> 8) + 0.5) * (1.0 / 16777216.0);
            sel[i] = (float) (((double) gid + u) / (double) Ntotal);
        }
    }
}

void orc_philox_predict_tape(uint64_t seed, uint32_t step, int first, int count, float *normals2) {
    uint32_t k0 = (uint32_t) seed, k1 = (uint32_t) (seed >> 32);
    for (int i = 0; i < count; i++) {
        uint32_t r[4];
        float g[3];
        orc_philox4x32((uint32_t) (first + i), step, 2u, 0u, k0, k1, r);
        box_muller3(r, g);
        normals2[2 * i] = g[0];
        normals2[2 * i + 1] = g[1];
    }
}

/* ============================================================================================
 * particle set
 * ========================================================================================== */

struct orc_particles {
    int N, cap, nf;
    int logw; /* log-weight extension (orc_particles_set_log_weights): w[] holds log-weights */
    float *w, *xv, *Pv, *xf, *Pf; /* particle-major; xf/Pf strided by cap */
};

orc_particles *orc_particles_create(int N, int cap_nf) {
    orc_particles *p = (orc_particles *) calloc(1, sizeof *p);
    p->N = N;
    p->cap = cap_nf > 0 ? cap_nf : 1;
    p->nf = 0;
    p->w = (float *) calloc((size_t) N, sizeof(float));
    p->xv = (float *) calloc((size_t) N * 3, sizeof(float));
    p->Pv = (float *) calloc((size_t) N * 9, sizeof(float));
    p->xf = (float *) calloc((size_t) N * p->cap * 2, sizeof(float));
    p->Pf = (float *) calloc((size_t) N * p->cap * 4, sizeof(float));
    /* ParticleSLAMWrapper.cpp:14-25: Particle() then w = 1/N */
    float uw = (float) (1.0 / (float) N);
    for (int i = 0; i < N; i++) p->w[i] = uw;
    return p;
}

void orc_particles_destroy(orc_particles *p) {
    if (!p) return;
    free(p->w);
    free(p->xv);
    free(p->Pv);
    free(p->xf);
    free(p->Pf);
    free(p);
}

/* LOG-WEIGHT EXTENSION (not in the reference; mirrors slamgpu_config.log_weights).  The reference's weight is a float32
 * product of one gaussEvaluate(.., logflag = 0) per re-observed landmark, ~90 each at the bundled noise levels: it
 * overflows to inf beyond ~20 landmarks per step.  With the flag on, w[] holds log-weights: every factor enters through
 * the reference's own logflag = 1 branch of gaussEvaluate (fastslam2.cpp:154-160; FastSLAM1: q - log(den)), and
 * resampleParticles works on exp(l - max l).  Pinned to the reference where the reference can go: gaussEvaluate(logflag=1)
 * against the reference objects (tests/golden/kat_log.npz) and exp(log-weights) against the linear weights for small m
 * (tests/test_oracle_golden.py). */
void orc_particles_set_log_weights(orc_particles *p, int on) {
    on = on ? 1 : 0;
    if (on == p->logw) return;
    for (int i = 0; i < p->N; i++) p->w[i] = on ? logf(p->w[i]) : expf(p->w[i]);
    p->logw = on;
}
int orc_particles_log_weights(const orc_particles *p) { return p->logw; }

int orc_particles_n(const orc_particles *p) { return p->N; }
int orc_particles_nf(const orc_particles *p) { return p->nf; }

void orc_particles_get(const orc_particles *p, float *xv, float *Pv9, float *w, float *xf, float *Pf4) {
    if (xv) memcpy(xv, p->xv, sizeof(float) * 3 * (size_t) p->N);
    if (Pv9) memcpy(Pv9, p->Pv, sizeof(float) * 9 * (size_t) p->N);
    if (w) memcpy(w, p->w, sizeof(float) * (size_t) p->N);
    for (int i = 0; i < p->N; i++) {
        if (xf) memcpy(xf + (size_t) i * 2 * p->nf, p->xf + (size_t) i * 2 * p->cap, sizeof(float) * 2 * (size_t) p->nf);
        if (Pf4) memcpy(Pf4 + (size_t) i * 4 * p->nf, p->Pf + (size_t) i * 4 * p->cap, sizeof(float) * 4 * (size_t) p->nf);
    }
}

void orc_particles_set(orc_particles *p, int nf, const float *xv, const float *Pv9, const float *w, const float *xf,
                       const float *Pf4) {
    if (nf > p->cap) {
        fprintf(stderr, "orc_particles_set: nf %d > cap %d\n", nf, p->cap);
        abort();
    }
    p->nf = nf;
    if (xv) memcpy(p->xv, xv, sizeof(float) * 3 * (size_t) p->N);
    if (Pv9) memcpy(p->Pv, Pv9, sizeof(float) * 9 * (size_t) p->N);
    if (w) memcpy(p->w, w, sizeof(float) * (size_t) p->N);
    for (int i = 0; i < p->N; i++) {
        if (xf) memcpy(p->xf + (size_t) i * 2 * p->cap, xf + (size_t) i * 2 * nf, sizeof(float) * 2 * (size_t) nf);
        if (Pf4) memcpy(p->Pf + (size_t) i * 4 * p->cap, Pf4 + (size_t) i * 4 * nf, sizeof(float) * 4 * (size_t) nf);
    }
}

void orc_estimate(const orc_particles *p, double *xyt) {
    /* ParticleSLAMWrapper.cpp:56-77 */
    double x = 0, y = 0, t = 0, wMax = -1e30;
    for (int i = 0; i < p->N; i++) {
        if (p->w[i] > wMax) {
            wMax = p->w[i];
            t = p->xv[3 * i + 2];
        }
        x += p->xv[3 * i];
        y += p->xv[3 * i + 1];
    }
    xyt[0] = x / p->N;
    xyt[1] = y / p->N;
    xyt[2] = t;
}

/* Host-core baseline only (bench.py cpu_baseline): the per-particle loops below are independent per particle, so they
 * can be spread over threads without changing a single result bit.  Default 1 thread (the scalar port, and what every
 * test uses); orc_set_threads(n) enables n OpenMP threads when the library was built with -fopenmp. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int orc_get_threads(void) {
#ifdef _OPENMP
    return g_threads;
#else
    return 1;
#endif
}

void orc_predict(orc_particles *p, const orc_algo *a, float V, float G, const float *Q4, float dt, float phi_true,
                 const float *noise2) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (int i = 0; i < p->N; i++) {
        const float *nz = (a->add_predict_noise && noise2) ? noise2 + 2 * i : NULL;
        if (a->method == 2)
            orc_fs2_predict_state(p->xv + 3 * i, p->Pv + 9 * i, V, G, Q4, a->wheel_base, dt, nz);
        else
            orc_fs1_predict_state(p->xv + 3 * i, V, G, Q4, a->wheel_base, dt, nz);
        if (a->use_heading) orc_observe_heading(p->xv + 3 * i, p->Pv + 9 * i, phi_true, a->sigma_phi);
    }
}

void orc_update_local(orc_particles *p, const orc_algo *a, const float *zf, const int *idf, int m, const float *zn, int n,
                      const float *R4, const float *normals) {
    /* the per-particle loop of FastSLAM{1,2}::update, without resampleParticles */
    int N = p->N;
    if (p->nf + n > p->cap) {
        fprintf(stderr, "orc_update: landmark capacity exceeded (%d + %d > %d)\n", p->nf, n, p->cap);
        abort();
    }
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
    for (int i = 0; i < N; i++) {
        float *xv = p->xv + 3 * i, *Pv = p->Pv + 9 * i, *xf = p->xf + (size_t) i * 2 * p->cap,
              *Pf = p->Pf + (size_t) i * 4 * p->cap;
        if (a->method == 2) {
            /* fastslam2.cpp:26-45 */
            if (m > 0) {
                fs2_sample_proposal(xv, Pv, &p->w[i], xf, Pf, zf, idf, m, R4, normals + 3 * i, p->logw);
                orc_feature_update(xv, xf, Pf, zf, idf, m, R4);
            }
            if (n > 0) {
                if (m == 0) {
                    float xs[3];
                    orc_multivariate_gauss(xv, Pv, 3, normals + 3 * i, xs);
                    memcpy(xv, xs, sizeof xs);
                    memset(Pv, 0, 9 * sizeof(float));
                }
                orc_add_feature(xv, zn, n, R4, xf + 2 * p->nf, Pf + 4 * p->nf);
            }
        } else {
            /* fastslam1.cpp:21-32 */
            if (m > 0) {
                double w = fs1_compute_weight(xv, xf, Pf, zf, idf, m, R4, p->logw);
                p->w[i] = p->logw ? (float) ((double) p->w[i] + w) : p->w[i] * (float) w;
                orc_feature_update(xv, xf, Pf, zf, idf, m, R4);
            }
            if (n > 0) orc_add_feature(xv, zn, n, R4, xf + 2 * p->nf, Pf + 4 * p->nf);
        }
    }
    p->nf += n;
}

/* resampleParticles (core.cpp:718-749) on a set whose per-particle update has run: normalise, Neff, the reference's own
 * decision and stratified ancestors.  forced_did < 0: apply them (the reference).  forced_did = 0 / 1: apply THAT decision
 * and, for 1, the ancestors in forced_keep instead -- tests drive the oracle with the GPU's genealogy this way, so that a
 * free-running comparison does not decorrelate at the first stratum that lands on the other side of a cumulative-sum
 * boundary (tests/test_gpu_freerun.py).  keep_out / neff_out / resampled_out always report the oracle's OWN plan. */
void orc_resample_forced(orc_particles *p, const orc_algo *a, const float *sel, int forced_did, const int *forced_keep,
                         int *keep_out, float *neff_out, int *resampled_out) {
    int N = p->N;
    float *w = (float *) malloc(sizeof(float) * (size_t) N);
    int *keep = (int *) malloc(sizeof(int) * (size_t) N);
    memcpy(w, p->w, sizeof(float) * (size_t) N);
    if (p->logw) {
        /* log-weights: resampleParticles on exp(l - M), M = max l; normalised log-weight = l - (M + log sum exp(l - M)) */
        double M = -INFINITY;
        for (int i = 0; i < N; i++)
            if ((double) p->w[i] > M) M = (double) p->w[i];
        for (int i = 0; i < N; i++) w[i] = (p->w[i] == -INFINITY) ? 0.0f : (float) exp((double) p->w[i] - M);
        double wsd = 0;
        for (int i = 0; i < N; i++) wsd += (double) w[i];
        for (int i = 0; i < N; i++) p->w[i] = p->w[i] - (float) (M + log(wsd));
    } else {
        float ws = orc_eigen_sum(w, N);
        for (int i = 0; i < N; i++) p->w[i] = w[i] / ws;
    }
    float nEff = 0;
    orc_stratified_resample(w, N, sel, keep, &nEff);
    const int own = (a->resample && (nEff < (float) a->n_effective)) ? 1 : 0;
    const int did = forced_did < 0 ? own : (forced_did ? 1 : 0);
    const int *use = (forced_did > 0 && forced_keep) ? forced_keep : keep;
    if (did) {
        float *oxv = (float *) malloc(sizeof(float) * 3 * (size_t) N);
        float *oPv = (float *) malloc(sizeof(float) * 9 * (size_t) N);
        float *oxf = (float *) malloc(sizeof(float) * 2 * (size_t) N * p->cap);
        float *oPf = (float *) malloc(sizeof(float) * 4 * (size_t) N * p->cap);
        memcpy(oxv, p->xv, sizeof(float) * 3 * (size_t) N);
        memcpy(oPv, p->Pv, sizeof(float) * 9 * (size_t) N);
        memcpy(oxf, p->xf, sizeof(float) * 2 * (size_t) N * p->cap);
        memcpy(oPf, p->Pf, sizeof(float) * 4 * (size_t) N * p->cap);
        const size_t nfl = (size_t) p->nf; /* landmarks beyond nf are never read before they are written */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
#endif
        for (int i = 0; i < N; i++) {
            int k = use[i];
            if (k < 0 || k >= N) k = N - 1; /* unfilled keep[] is UB upstream (core.cpp:793,741); clamp */
            memcpy(p->xv + 3 * i, oxv + 3 * k, 3 * sizeof(float));
            memcpy(p->Pv + 9 * i, oPv + 9 * k, 9 * sizeof(float));
            memcpy(p->xf + (size_t) i * 2 * p->cap, oxf + (size_t) k * 2 * p->cap, sizeof(float) * 2 * nfl);
            memcpy(p->Pf + (size_t) i * 4 * p->cap, oPf + (size_t) k * 4 * p->cap, sizeof(float) * 4 * nfl);
        }
        float nw = 1.0f / (float) N;
        if (p->logw) nw = logf(nw);
        for (int i = 0; i < N; i++) p->w[i] = nw;
        free(oxv);
        free(oPv);
        free(oxf);
        free(oPf);
    }
    if (keep_out) memcpy(keep_out, keep, sizeof(int) * (size_t) N);
    if (neff_out) *neff_out = nEff;
    if (resampled_out) *resampled_out = forced_did < 0 ? did : own;
    free(w);
    free(keep);
}

void orc_update(orc_particles *p, const orc_algo *a, const float *zf, const int *idf, int m, const float *zn, int n,
                const float *R4, const float *normals, const float *sel, int *keep_out, float *neff_out,
                int *resampled_out) {
    orc_update_local(p, a, zf, idf, m, zn, n, R4, normals);
    orc_resample_forced(p, a, sel, -1, NULL, keep_out, neff_out, resampled_out);
}

/* ============================================================================================
 * host front end: ini/args, map reader, vehicle + sensor simulator, known data association
 * ========================================================================================== */

typedef struct {
    float V, MAXG, RATEG, WHEELBASE, DT_CONTROLS, sigmaV, sigmaG, MAX_RANGE, DT_OBSERVE, sigmaR, sigmaB, sigmaT;
    float GATE_REJECT, GATE_AUGMENT, AT_WAYPOINT;
    int NUMBER_LOOPS, NPARTICLES, NEFFECTIVE;
    int SWITCH_CONTROL_NOISE, SWITCH_SENSOR_NOISE, SWITCH_INFLATE_NOISE, SWITCH_PREDICT_NOISE, SWITCH_SAMPLE_PROPOSAL,
        SWITCH_HEADING_KNOWN, SWITCH_RESAMPLE, SWITCH_PROFILE, SWITCH_SEED_RANDOM, SWITCH_ASSOCIATION_KNOWN,
        SWITCH_BATCH_UPDATE, SWITCH_USE_IEKF;
    char method[32], map[512];
} conf_t;

#define MAXKV 128
typedef struct {
    char k[MAXKV][64], v[MAXKV][256];
    int n;
} kv_t;

static char *trim(char *s) {
    while (*s && isspace((unsigned char) *s)) s++;
    char *e = s + strlen(s);
    while (e > s && isspace((unsigned char) e[-1])) *--e = 0;
    return s;
}

static void kv_set(kv_t *kv, const char *k, const char *v) {
    /* later entries override earlier ones on lookup (utils.cpp set_s appends/replaces) */
    for (int i = 0; i < kv->n; i++)
        if (strcmp(kv->k[i], k) == 0) {
            snprintf(kv->v[i], sizeof kv->v[i], "%s", v);
            return;
        }
    if (kv->n < MAXKV) {
        snprintf(kv->k[kv->n], sizeof kv->k[0], "%s", k);
        snprintf(kv->v[kv->n], sizeof kv->v[0], "%s", v);
        kv->n++;
    }
}

static const char *kv_get(const kv_t *kv, const char *k) {
    for (int i = 0; i < kv->n; i++)
        if (strcmp(kv->k[i], k) == 0) return kv->v[i];
    return NULL;
}

static void kv_load_ini(kv_t *kv, const char *path) {
    /* utils.cpp:504-565: "key = value", '#' or ':' comment lines */
    FILE *fp = fopen(path, "rt");
    if (!fp) return;
    char buf[4096];
    while (fgets(buf, sizeof buf, fp)) {
        char *b = trim(buf);
        if (!*b || *b == '#' || *b == ':') continue;
        char *eq = strchr(b, '=');
        if (!eq) continue;
        *eq = 0;
        kv_set(kv, trim(b), trim(eq + 1));
    }
    fclose(fp);
}

static void cf(const kv_t *kv, const char *k, float *v) {
    const char *s = kv_get(kv, k);
    if (s) *v = (float) atof(s);
}
static void ci(const kv_t *kv, const char *k, int *v) {
    const char *s = kv_get(kv, k);
    if (s) *v = atoi(s);
}

static void conf_parse(conf_t *c, const kv_t *kv) {
    /* core.cpp:971-1073 defaults then overrides */
    c->V = 3.0;
    c->MAXG = (float) (30 * M_PI / 180);
    c->RATEG = (float) (20 * M_PI / 180);
    c->WHEELBASE = 4;
    c->DT_CONTROLS = 0.025;
    c->sigmaV = 0.3;
    c->sigmaG = (float) (3.0 * M_PI / 180);
    c->MAX_RANGE = 30.0;
    c->DT_OBSERVE = 8 * c->DT_CONTROLS;
    c->sigmaR = 0.1;
    c->sigmaB = (float) (1.0 * M_PI / 180);
    c->sigmaT = (float) (1.0 * M_PI / 180);
    c->GATE_REJECT = 4.0;
    c->GATE_AUGMENT = 25.0;
    c->AT_WAYPOINT = 1.0;
    c->NUMBER_LOOPS = 2;
    c->NPARTICLES = 100;
    c->NEFFECTIVE = (int) (0.75 * c->NPARTICLES);
    c->SWITCH_CONTROL_NOISE = 1;
    c->SWITCH_SENSOR_NOISE = 1;
    c->SWITCH_INFLATE_NOISE = 0;
    c->SWITCH_PREDICT_NOISE = 0;
    c->SWITCH_SAMPLE_PROPOSAL = 1;
    c->SWITCH_HEADING_KNOWN = 1;
    c->SWITCH_RESAMPLE = 1;
    c->SWITCH_PROFILE = 1;
    c->SWITCH_SEED_RANDOM = 0;
    c->SWITCH_ASSOCIATION_KNOWN = 0;
    c->SWITCH_BATCH_UPDATE = 1;
    c->SWITCH_USE_IEKF = 0;
    cf(kv, "Vtrue", &c->V);
    cf(kv, "MAXG", &c->MAXG);
    cf(kv, "RATEG", &c->RATEG);
    cf(kv, "WHEELBASE", &c->WHEELBASE);
    cf(kv, "DT_CONTROLS", &c->DT_CONTROLS);
    cf(kv, "sigmaV", &c->sigmaV);
    cf(kv, "sigmaG", &c->sigmaG);
    cf(kv, "MAX_RANGE", &c->MAX_RANGE);
    cf(kv, "DT_OBSERVE", &c->DT_OBSERVE);
    cf(kv, "sigmaR", &c->sigmaR);
    cf(kv, "sigmaB", &c->sigmaB);
    cf(kv, "sigmaT", &c->sigmaT);
    cf(kv, "GATE_REJECT", &c->GATE_REJECT);
    cf(kv, "GATE_AUGMENT", &c->GATE_AUGMENT);
    cf(kv, "AT_WAYPOINT", &c->AT_WAYPOINT);
    ci(kv, "NUMBER_LOOPS", &c->NUMBER_LOOPS);
    ci(kv, "NPARTICLES", &c->NPARTICLES);
    ci(kv, "NEFFECTIVE", &c->NEFFECTIVE);
    ci(kv, "SWITCH_CONTROL_NOISE", &c->SWITCH_CONTROL_NOISE);
    ci(kv, "SWITCH_SENSOR_NOISE", &c->SWITCH_SENSOR_NOISE);
    ci(kv, "SWITCH_INFLATE_NOISE", &c->SWITCH_INFLATE_NOISE);
    ci(kv, "SWITCH_PREDICT_NOISE", &c->SWITCH_PREDICT_NOISE);
    ci(kv, "SWITCH_SAMPLE_PROPOSAL", &c->SWITCH_SAMPLE_PROPOSAL);
    ci(kv, "SWITCH_HEADING_KNOWN", &c->SWITCH_HEADING_KNOWN);
    ci(kv, "SWITCH_RESAMPLE", &c->SWITCH_RESAMPLE);
    ci(kv, "SWITCH_PROFILE", &c->SWITCH_PROFILE);
    ci(kv, "SWITCH_SEED_RANDOM", &c->SWITCH_SEED_RANDOM);
    ci(kv, "SWITCH_ASSOCIATION_KNOWN", &c->SWITCH_ASSOCIATION_KNOWN);
    ci(kv, "SWITCH_BATCH_UPDATE", &c->SWITCH_BATCH_UPDATE);
    ci(kv, "SWITCH_USE_IEKF", &c->SWITCH_USE_IEKF);
}

void orc_free(void *p) { free(p); }

int orc_read_map(const char *path, float **lm_out, int *nlm, float **wp_out, int *nwp) {
    /* core.cpp:855-962: "lm <rows> <cols>" / "wp <rows> <cols>" then one column per line; '#' comments.
     * Output arrays are 2 x n row-major (row 0 = x, row 1 = y). */
    FILE *fp = fopen(path, "rt");
    if (!fp) return -1;
    char buf[4096];
    float *lm = NULL, *wp = NULL;
    int lm_rows = 0, lm_cols = 0, wp_rows = 0, wp_cols = 0;
    while (fgets(buf, sizeof buf, fp)) {
        char *b = trim(buf);
        if (!*b || *b == '#') continue;
        char tag[16];
        float a = 0, c = 0;
        if (sscanf(b, "%15s %f %f", tag, &a, &c) != 3) {
            fclose(fp);
            free(lm);
            free(wp);
            return -2;
        }
        int rows = (int) a, cols = (int) c;
        float *dst = (float *) calloc((size_t) (rows > 2 ? rows : 2) * (size_t) cols, sizeof(float));
        int is_lm = strcmp(tag, "lm") == 0, is_wp = strcmp(tag, "wp") == 0;
        if (!is_lm && !is_wp) {
            fclose(fp);
            free(dst);
            free(lm);
            free(wp);
            return -3;
        }
        /* NB upstream reads lm_rows values per waypoint line too (core.cpp:950) */
        int nread = is_lm ? rows : lm_rows;
        for (int col = 0; col < cols; col++) {
            if (!fgets(buf, sizeof buf, fp)) {
                fclose(fp);
                free(dst);
                free(lm);
                free(wp);
                return -4;
            }
            char *q = buf;
            for (int r = 0; r < nread && r < rows; r++) {
                char *end;
                float val = strtof(q, &end);
                if (end == q) {
                    fclose(fp);
                    free(dst);
                    free(lm);
                    free(wp);
                    return -5;
                }
                dst[r * cols + col] = val;
                q = end;
            }
        }
        if (is_lm) {
            free(lm);
            lm = dst;
            lm_rows = rows;
            lm_cols = cols;
        } else {
            free(wp);
            wp = dst;
            wp_rows = rows;
            wp_cols = cols;
        }
    }
    fclose(fp);
    (void) wp_rows;
    *lm_out = lm;
    *nlm = lm_cols;
    *wp_out = wp;
    *nwp = wp_cols;
    return 0;
}

struct orc_sim {
    conf_t conf;
    orc_algo algo;
    int method;
    float *lm, *wp;
    int nlm, nwp;
    float Q[4], R[4], Qe[4], Re[4];
    float Vtrue, Gtrue, Vnoisy, Gnoisy, dt, dtSum;
    int nLoop, iwp;
    float xTrue[3];
    float *table; /* dataAssociationTable (float, -1 = unseen) */
    orc_particles *P;
    /* last observation */
    float *z, *zf, *zn;
    int *vis, *idf;
    int nz, m, n;
    /* tape of last update */
    float *normals, *sel, *noise2;
    float last_neff;
    int last_resampled;
    int rng_mode;
    uint64_t seed;
    uint32_t obs_step, ctl_step;
    /* EKF */
    float *ex, *eP;
    int edim, ecap;
    int *etable;
};

static void update_steering(orc_sim *s) {
    /* core.cpp:41-78 */
    conf_t *c = &s->conf;
    double cw0 = s->wp[0 * s->nwp + s->iwp], cw1 = s->wp[1 * s->nwp + s->iwp];
    float d2 = (float) (pow(cw0 - s->xTrue[0], 2) + pow(cw1 - s->xTrue[1], 2));
    if (d2 < c->AT_WAYPOINT * c->AT_WAYPOINT) {
        s->iwp++;
        if (s->iwp >= s->nwp) {
            s->iwp = -1;
            return;
        }
        cw0 = s->wp[0 * s->nwp + s->iwp];
        cw1 = s->wp[1 * s->nwp + s->iwp];
    }
    float deltaG = (float) (atan2(cw1 - s->xTrue[1], cw0 - s->xTrue[0]) - s->xTrue[2] - s->Gtrue);
    deltaG = orc_trig_offset(deltaG);
    float maxDelta = c->RATEG * s->dt;
    if (fabsf(deltaG) > maxDelta) {
        int sign = (deltaG > 0) ? 1 : ((deltaG < 0) ? -1 : 0);
        deltaG = sign * maxDelta;
    }
    s->Gtrue = s->Gtrue + deltaG;
    if (fabsf(s->Gtrue) > c->MAXG) {
        int sign2 = (s->Gtrue > 0) ? 1 : ((s->Gtrue < 0) ? -1 : 0);
        s->Gtrue = sign2 * c->MAXG;
    }
}

static void predict_true(orc_sim *s) {
    /* core.cpp:35-39 */
    float *x = s->xTrue, V = s->Vtrue, G = s->Gtrue, dt = s->dt;
    x[0] = x[0] + V * dt * cosf(G + x[2]);
    x[1] = x[1] + V * dt * sinf(G + x[2]);
    x[2] = orc_trig_offset(x[2] + V * dt * sinf(G) / s->conf.WHEELBASE);
}

static void observe(orc_sim *s) {
    /* getObservations (core.cpp:185-273) + addObservationNoise (:438-449) */
    const float *x = s->xTrue;
    float range = s->conf.MAX_RANGE;
    float phi = x[2];
    s->nz = 0;
    for (int j = 0; j < s->nlm; j++) {
        float dx = s->lm[j] - x[0];
        float dy = s->lm[s->nlm + j] - x[1];
        if ((fabsf(dx) < range) && (fabsf(dy) < range) && ((dx * cosf(phi) + dy * sinf(phi)) > 0.0) &&
            ((pow((double) dx, 2) + pow((double) dy, 2)) < pow((double) range, 2))) {
            s->vis[s->nz] = j;
            s->z[2 * s->nz] = (float) sqrt(pow((double) dx, 2) + pow((double) dy, 2));
            s->z[2 * s->nz + 1] = atan2f(dy, dx) - phi;
            s->nz++;
        }
    }
    if (s->conf.SWITCH_SENSOR_NOISE && s->nz > 0) {
        float *r1 = (float *) malloc(sizeof(float) * (size_t) s->nz);
        float *r2 = (float *) malloc(sizeof(float) * (size_t) s->nz);
        orc_randn(1, s->nz, r1);
        orc_randn(1, s->nz, r2);
        for (int c = 0; c < s->nz; c++) {
            s->z[2 * c] = s->z[2 * c] + r1[c] * sqrtf(s->R[0]);
            s->z[2 * c + 1] = s->z[2 * c + 1] + r2[c] * sqrtf(s->R[3]);
        }
        free(r1);
        free(r2);
    }
}

static void associate_known(orc_sim *s, int Nf) {
    /* core.cpp:91-120 */
    s->m = s->n = 0;
    int nnew = 0;
    int *idn = (int *) malloc(sizeof(int) * (size_t) (s->nz + 1));
    for (int i = 0; i < s->nz; i++) {
        int ii = s->vis[i];
        if (s->table[ii] == -1) {
            s->zn[2 * s->n] = s->z[2 * i];
            s->zn[2 * s->n + 1] = s->z[2 * i + 1];
            s->n++;
            idn[nnew++] = ii;
        } else {
            s->zf[2 * s->m] = s->z[2 * i];
            s->zf[2 * s->m + 1] = s->z[2 * i + 1];
            s->idf[s->m] = (int) s->table[ii];
            s->m++;
        }
    }
    for (int i = 0; i < nnew; i++) s->table[idn[i]] = (float) (Nf + i);
    free(idn);
}

static void orc_ekf_step(orc_sim *s, int observe_flag);

orc_sim *orc_sim_create(int argc, char **argv) {
    orc_sim *s = (orc_sim *) calloc(1, sizeof *s);
    kv_t *kv = (kv_t *) calloc(1, sizeof *kv);
    const char *map = "example_webmap.mat";
    for (int i = 1; i + 1 < argc; i++)
        if (strcmp(argv[i], "-m") == 0) map = argv[i + 1];
    char ini[600];
    snprintf(ini, sizeof ini, "%s", map);
    char *dot = strrchr(ini, '.');
    if (dot) *dot = 0;
    strncat(ini, ".ini", sizeof ini - strlen(ini) - 1);
    kv_load_ini(kv, ini);
    for (int i = 1; i < argc; i++) /* utils.cpp:1032-1046 */
        if (argv[i][0] == '-' && i + 1 < argc) {
            kv_set(kv, argv[i] + 1, argv[i + 1]);
            i++;
        }
    conf_parse(&s->conf, kv);
    const char *method = kv_get(kv, "method");
    s->method = (method && strcmp(method, "FASTSLAM1") == 0) ? 1 : ((method && strcmp(method, "FASTSLAM2") == 0) ? 2 : 0);
    free(kv);
    if (orc_read_map(map, &s->lm, &s->nlm, &s->wp, &s->nwp) != 0) {
        fprintf(stderr, "orc_sim_create: cannot read map %s\n", map);
        free(s);
        return NULL;
    }
    conf_t *c = &s->conf;
    /* slamwrapper.cpp:19-53 */
    s->Vtrue = c->V;
    s->Gtrue = 0;
    s->Q[0] = (float) pow((double) c->sigmaV, 2);
    s->Q[3] = (float) pow((double) c->sigmaG, 2);
    s->R[0] = (float) pow((double) c->sigmaR, 2);
    s->R[3] = (float) pow((double) c->sigmaB, 2);
    if (c->SWITCH_INFLATE_NOISE == 1) {
        for (int i = 0; i < 4; i++) {
            s->Q[i] = 2 * s->Q[i];
            s->R[i] = 2 * s->R[i];
        }
        /* upstream leaves Qe/Re unset in this branch (slamwrapper.cpp:31-37); we keep them zero */
    } else {
        memcpy(s->Qe, s->Q, sizeof s->Q);
        memcpy(s->Re, s->R, sizeof s->R);
    }
    s->nLoop = c->NUMBER_LOOPS;
    s->dt = c->DT_CONTROLS;
    s->iwp = 0;
    if (c->SWITCH_SEED_RANDOM != 0) srand((unsigned) c->SWITCH_SEED_RANDOM);
    s->algo.method = s->method;
    s->algo.use_heading = c->SWITCH_HEADING_KNOWN == 1;
    s->algo.add_predict_noise = (s->method == 1) ? 1 : (c->SWITCH_PREDICT_NOISE == 1);
    s->algo.resample = c->SWITCH_RESAMPLE == 1;
    s->algo.n_effective = c->NEFFECTIVE;
    s->algo.wheel_base = c->WHEELBASE;
    s->algo.sigma_phi = c->sigmaT;
    s->table = (float *) malloc(sizeof(float) * (size_t) s->nlm);
    for (int i = 0; i < s->nlm; i++) s->table[i] = -1;
    s->z = (float *) calloc((size_t) s->nlm * 2 + 2, sizeof(float));
    s->zf = (float *) calloc((size_t) s->nlm * 2 + 2, sizeof(float));
    s->zn = (float *) calloc((size_t) s->nlm * 2 + 2, sizeof(float));
    s->vis = (int *) calloc((size_t) s->nlm + 1, sizeof(int));
    s->idf = (int *) calloc((size_t) s->nlm + 1, sizeof(int));
    if (s->method != 0) {
        int N = c->NPARTICLES;
        s->P = orc_particles_create(N, s->nlm);
        s->normals = (float *) calloc((size_t) N * 3, sizeof(float));
        s->sel = (float *) calloc((size_t) N, sizeof(float));
        s->noise2 = (float *) calloc((size_t) N * 2, sizeof(float));
    }
    s->seed = (uint64_t) c->SWITCH_SEED_RANDOM;
    return s;
}

void orc_sim_destroy(orc_sim *s) {
    if (!s) return;
    free(s->lm);
    free(s->wp);
    free(s->table);
    free(s->z);
    free(s->zf);
    free(s->zn);
    free(s->vis);
    free(s->idf);
    free(s->normals);
    free(s->sel);
    free(s->noise2);
    free(s->ex);
    free(s->eP);
    free(s->etable);
    orc_particles_destroy(s->P);
    free(s);
}

void orc_sim_set_rng(orc_sim *s, int rng_mode, uint64_t seed) {
    s->rng_mode = rng_mode;
    s->seed = seed;
}

int orc_sim_control(orc_sim *s) {
    conf_t *c = &s->conf;
    /* slamwrapper.cpp:174-238 */
    if (s->iwp == -1) return -1;
    update_steering(s);
    if (s->iwp == -1 && s->nLoop > 1) {
        s->iwp = 0;
        s->nLoop--;
    }
    if (s->iwp == -1 && s->nLoop == 1) return -1;
    predict_true(s);
    if (c->SWITCH_CONTROL_NOISE) {
        /* addControlNoise (core.cpp:24-32) */
        float g[2], A[2] = {s->Vtrue, s->Gtrue}, C[2];
        orc_randn(2, 1, g);
        orc_multivariate_gauss(A, s->Q, 2, g, C);
        s->Vnoisy = C[0];
        s->Gnoisy = C[1];
    }
    s->ctl_step++;
    if (s->method != 0) {
        int N = s->P->N;
        const float *nz = NULL;
        if (s->algo.add_predict_noise) {
            if (s->rng_mode == 0)
                for (int i = 0; i < N; i++) orc_randn(2, 1, s->noise2 + 2 * i); /* 3 rand() each */
            else
                orc_philox_predict_tape(s->seed, s->ctl_step, 0, N, s->noise2);
            nz = s->noise2;
        }
        orc_predict(s->P, &s->algo, s->Vnoisy, s->Gnoisy, s->Qe, s->dt, s->xTrue[2], nz);
    }
    s->dtSum += s->dt;
    if (s->dtSum >= c->DT_OBSERVE) {
        s->dtSum = 0;
        return 1;
    }
    return 0;
}

/* first half of orc_sim_observe: sensor, association, the step's random tape and the per-particle update -- everything
 * but resampleParticles */
void orc_sim_observe_local(orc_sim *s) {
    observe(s);
    if (s->method != 0) {
        int N = s->P->N;
        associate_known(s, s->P->nf);
        s->obs_step++;
        int need_normals = (s->method == 2) && (s->m > 0 || s->n > 0);
        if (s->rng_mode == 0) {
            /* reference draw order: 4 rand() per particle inside the particle loop, then N for the strata */
            if (need_normals)
                for (int i = 0; i < N; i++) orc_randn(3, 1, s->normals + 3 * i);
            orc_stratified_random(N, s->sel);
        } else {
            orc_philox_update_tape(s->seed, s->obs_step, 0, N, N, need_normals ? s->normals : NULL, s->sel);
        }
        orc_update_local(s->P, &s->algo, s->zf, s->idf, s->m, s->zn, s->n, s->Re, s->normals);
    }
}

/* second half: resampleParticles; forced_did / forced_keep as in orc_resample_forced; own_keep (optional) receives the
 * oracle's own ancestors.  last_resample reports the oracle's OWN Neff and decision. */
void orc_sim_resample(orc_sim *s, int forced_did, const int *forced_keep, int *own_keep) {
    if (s->method != 0)
        orc_resample_forced(s->P, &s->algo, s->sel, forced_did, forced_keep, own_keep, &s->last_neff, &s->last_resampled);
}

void orc_sim_observe(orc_sim *s) {
    orc_sim_observe_local(s);
    orc_sim_resample(s, -1, NULL, NULL);
}

int orc_sim_step(orc_sim *s) {
    int r = orc_sim_control(s);
    if (r < 0) return r;
    if (r == 1) orc_sim_observe(s);
    if (s->method == 0) orc_ekf_step(s, r == 1);
    return r;
}

orc_particles *orc_sim_particles(orc_sim *s) { return s->P; }
int orc_sim_nlandmarks(const orc_sim *s) { return s->nlm; }
const orc_algo *orc_sim_algo(const orc_sim *s) { return &s->algo; }

void orc_sim_true(const orc_sim *s, float *x3, float *VnGn) {
    memcpy(x3, s->xTrue, 3 * sizeof(float));
    if (VnGn) {
        VnGn[0] = s->Vnoisy;
        VnGn[1] = s->Gnoisy;
    }
}

int orc_sim_last_obs(const orc_sim *s, float *zf, int *idf, float *zn, int *n_out, float *z, int *vis, int *nz_out) {
    if (zf) memcpy(zf, s->zf, sizeof(float) * 2 * (size_t) s->m);
    if (idf) memcpy(idf, s->idf, sizeof(int) * (size_t) s->m);
    if (zn) memcpy(zn, s->zn, sizeof(float) * 2 * (size_t) s->n);
    if (z) memcpy(z, s->z, sizeof(float) * 2 * (size_t) s->nz);
    if (vis) memcpy(vis, s->vis, sizeof(int) * (size_t) s->nz);
    if (n_out) *n_out = s->n;
    if (nz_out) *nz_out = s->nz;
    return s->m;
}

void orc_sim_last_resample(const orc_sim *s, float *neff, int *resampled) {
    if (neff) *neff = s->last_neff;
    if (resampled) *resampled = s->last_resampled;
}

void orc_sim_last_tape(const orc_sim *s, float *normals, float *sel) {
    int N = s->P->N;
    if (normals) memcpy(normals, s->normals, sizeof(float) * 3 * (size_t) N);
    if (sel) memcpy(sel, s->sel, sizeof(float) * (size_t) N);
}

void orc_sim_last_noise2(const orc_sim *s, float *noise2) {
    if (s->P && noise2) memcpy(noise2, s->noise2, sizeof(float) * 2 * (size_t) s->P->N);
}

void orc_sim_noise(const orc_sim *s, float *Q4, float *R4, float *dt) {
    if (Q4) memcpy(Q4, s->Qe, sizeof s->Qe);
    if (R4) memcpy(R4, s->Re, sizeof s->Re);
    if (dt) *dt = s->dt;
}

static void orc_ekf_step(orc_sim *s, int observe_flag) {
    (void) s;
    (void) observe_flag; /* TODO(config 1): EKF restatement */
}

int orc_sim_ekf_state(const orc_sim *s, float *x, float *P, int cap) {
    (void) s;
    (void) x;
    (void) P;
    (void) cap;
    return 0;
}
