#include "ekfslam.h"

#include <cmath>
#include <cstring>

namespace slamhost {

namespace {

// C = A * B (k-ascending accumulation from zero, like a GEMM)
Mat mul(const Mat &A, const Mat &B) {
    Mat C(A.r, B.c);
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < B.c; j++) {
            float acc = 0.0f;
            for (int k = 0; k < A.c; k++) acc = acc + A(i, k) * B(k, j);
            C(i, j) = acc;
        }
    return C;
}

Mat mulT(const Mat &A, const Mat &B) {  // A * B^T
    Mat C(A.r, B.r);
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < B.r; j++) {
            float acc = 0.0f;
            for (int k = 0; k < A.c; k++) acc = acc + A(i, k) * B(j, k);
            C(i, j) = acc;
        }
    return C;
}

// lower Cholesky (Eigen LLT order: reciprocal-scaled columns); returns false on a non-positive pivot
bool llt_lower(Mat &M) {
    const int n = M.r;
    for (int k = 0; k < n; k++) {
        float x = M(k, k);
        if (k > 0) {
            float sq = M(k, 0) * M(k, 0);
            for (int j = 1; j < k; j++) sq = sq + M(k, j) * M(k, j);
            x = x - sq;
        }
        if (x <= 0.0f) return false;
        x = std::sqrt(x);
        M(k, k) = x;
        for (int i = k + 1; i < n; i++) {
            float v = M(i, k);
            for (int j = 0; j < k; j++) v = v + M(i, j) * (-1.0f * M(k, j));
            M(i, k) = v;
        }
        const float r = 1.0f / x;
        for (int i = k + 1; i < n; i++) M(i, k) = M(i, k) * r;
    }
    return true;
}

// inverse of an upper-triangular matrix by back substitution (what PartialPivLU reduces to: no pivoting needed,
// |u_kk| is the only candidate below the diagonal being zero)
Mat upper_inverse(const Mat &U) {
    const int n = U.r;
    Mat X(n, n);
    for (int i = 0; i < n; i++) X(i, i) = 1.0f;
    for (int k = 0; k < n; k++) {
        const int i = n - k - 1;
        const float a = 1.0f / U(i, i);
        for (int j = 0; j < n; j++) {
            const float b = (X(i, j) *= a);
            for (int r = 0; r < i; r++) X(r, j) -= b * U(r, i);
        }
    }
    return X;
}

// 2x2 inverse / determinant through partial-pivot LU (dynamic-size Eigen path)
void inverse2(const float S[4], float X[4], float *det) {
    const bool swap = std::fabs(S[2]) > std::fabs(S[0]);
    const float u00 = swap ? S[2] : S[0], u01 = swap ? S[3] : S[1];
    float l10 = swap ? S[0] : S[2];
    const float r11 = swap ? S[1] : S[3];
    if (u00 != 0.0f) l10 = l10 * (1.0f / u00);
    const float u11 = r11 - l10 * u01;
    float b00 = swap ? 0.0f : 1.0f, b01 = swap ? 1.0f : 0.0f, b10 = swap ? 1.0f : 0.0f, b11 = swap ? 0.0f : 1.0f;
    b10 -= b00 * l10;
    b11 -= b01 * l10;
    float a = 1.0f / u11;
    b10 *= a;
    b11 *= a;
    b00 -= b10 * u01;
    b01 -= b11 * u01;
    a = 1.0f / u00;
    X[0] = b00 * a;
    X[1] = b01 * a;
    X[2] = b10;
    X[3] = b11;
    if (det) *det = (swap ? -1.0f : 1.0f) * (u00 * u11);
}

}  // namespace

EkfSlam::EkfSlam() : x(3, 0.0f), P(3, 3) {}

void EkfSlam::sim(float Vn, float Gn, const float Qe[4], float dt, float phi, const std::vector<float> &z,
                  const std::vector<int32_t> &ids, const float Re[4], bool observe, const float R[4],
                  std::vector<float> &table) {
    predict(Vn, Gn, Qe, dt);
    if (useHeading) observe_heading(phi);
    if (observe) {
        std::vector<float> zf, zn;
        std::vector<int> idf;
        if (associationKnown)
            associate_known(z, ids, zf, idf, zn, table);
        else
            associate(z, Re, zf, idf, zn);
        if (enableBatchUpdate) batch_update(zf, R, idf);
        augment(zn, Re);
    }
}

void EkfSlam::predict(float V, float G, const float Q[4], float dt) {
    const int m = P.r;
    const float s = std::sin(G + x[2]), c = std::cos(G + x[2]);
    const float vts = V * dt * s, vtc = V * dt * c;
    Mat Gv(3, 3), Gu(3, 2), Qm(2, 2);
    Gv(0, 0) = 1; Gv(0, 2) = -vts; Gv(1, 1) = 1; Gv(1, 2) = vtc; Gv(2, 2) = 1;
    Gu(0, 0) = dt * c; Gu(0, 1) = -vts; Gu(1, 0) = dt * s; Gu(1, 1) = vtc;
    Gu(2, 0) = dt * std::sin(G) / wheelBase; Gu(2, 1) = V * dt * std::cos(G) / wheelBase;
    for (int i = 0; i < 4; i++) Qm.a[i] = Q[i];
    Mat Pvv(3, 3);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Pvv(i, j) = P(i, j);
    Mat A = mulT(mul(Gv, Pvv), Gv), B = mulT(mul(Gu, Qm), Gu);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) P(i, j) = A(i, j) + B(i, j);
    if (m > 3) {
        Mat Pvm(3, m - 3);
        for (int i = 0; i < 3; i++)
            for (int j = 3; j < m; j++) Pvm(i, j - 3) = P(i, j);
        Mat N = mul(Gv, Pvm);
        for (int i = 0; i < 3; i++)
            for (int j = 3; j < m; j++) {
                P(i, j) = N(i, j - 3);
                P(j, i) = N(i, j - 3);
            }
    }
    x[0] = x[0] + vtc;
    x[1] = x[1] + vts;
    x[2] = trig_offset(x[2] + V * dt * std::sin(G) / wheelBase);
}

void EkfSlam::observe_heading(float phi) {
    // josephUpdate (core.cpp:294-317) with H = e_2^T on the whole state
    const int n = P.r;
    const float v = trig_offset(phi - x[2]);
    const float R = (float) std::pow((double) sigmaPhi, 2);
    std::vector<float> PHt(n), W(n);
    for (int i = 0; i < n; i++) PHt[i] = P(i, 2);
    const float S = PHt[2] + R;
    const float Si = 1.0f / S;
    for (int i = 0; i < n; i++) W[i] = PHt[i] * Si;
    for (int i = 0; i < n; i++) x[i] = x[i] + W[i] * v;
    // C = I - W H : differs from I only in column 2.  P = C P C^T + W R W^T + eps I
    Mat C(n, n);
    for (int i = 0; i < n; i++) {
        C(i, i) = 1.0f;
        C(i, 2) = ((i == 2) ? 1.0f : 0.0f) - W[i];
    }
    Mat CPC = mulT(mul(C, P), C);
    const float eps = (float) (2.2204 * std::pow(10.0, -16));
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            const float p = CPC(i, j) + W[j] * (W[i] * R);
            P(i, j) = p + ((i == j) ? 1.0f : 0.0f) * eps;
        }
}

// H restricted to its five non-zero columns {0,1,2,fpos,fpos+1}: H5 = [row0(5), row1(5)]
void EkfSlam::observe_model(int idf, float zp[2], float H5[10]) const {
    const int fpos = 3 + idf * 2;
    const float dx = x[fpos] - x[0], dy = x[fpos + 1] - x[1];
    const float d2 = dx * dx + dy * dy;
    const float d = sqrtf(d2);
    const float xd = dx / d, yd = dy / d, xd2 = dx / d2, yd2 = dy / d2;
    zp[0] = d;
    zp[1] = std::atan2(dy, dx) - x[2];
    H5[0] = -xd; H5[1] = -yd; H5[2] = 0; H5[3] = xd; H5[4] = yd;
    H5[5] = yd2; H5[6] = -xd2; H5[7] = -1; H5[8] = -yd2; H5[9] = xd2;
}

void EkfSlam::associate(const std::vector<float> &z, const float R[4], std::vector<float> &zf, std::vector<int> &idf,
                        std::vector<float> &zn) const {
    const int Nf = num_features();
    const int nz = (int) z.size() / 2;
    for (int i = 0; i < nz; i++) {
        long jbest = -1;
        float nbest = 1e60, outer = 1e60;  // float(1e60) = +inf, as upstream
        for (int j = 0; j < Nf; j++) {
            float zp[2], H5[10];
            observe_model(j, zp, H5);
            const float v0 = z[2 * i] - zp[0];
            const float v1 = trig_offset(z[2 * i + 1] - zp[1]);
            const int cols[5] = {0, 1, 2, 3 + 2 * j, 4 + 2 * j};
            // S = H P H^T + R over the 5 non-zero columns (the zero columns contribute exact zeros upstream)
            float HP[2][5];
            for (int r = 0; r < 2; r++)
                for (int c = 0; c < 5; c++) {
                    float acc = 0.0f;
                    for (int k = 0; k < 5; k++) acc = acc + H5[5 * r + k] * P(cols[k], cols[c]);
                    HP[r][c] = acc;
                }
            float S[4];
            for (int r = 0; r < 2; r++)
                for (int c = 0; c < 2; c++) {
                    float acc = 0.0f;
                    for (int k = 0; k < 5; k++) acc = acc + HP[r][k] * H5[5 * c + k];
                    S[2 * r + c] = acc + R[2 * r + c];
                }
            float Si[4], det;
            inverse2(S, Si, &det);
            const float t0 = v0 * Si[0] + v1 * Si[2], t1 = v0 * Si[1] + v1 * Si[3];
            const float nis = t0 * v0 + t1 * v1;
            const float nd = nis + std::log(det);
            if (nis < gateReject && nd < nbest) {
                nbest = nd;
                jbest = j;
            } else if (nis < outer) {
                outer = nis;
            }
        }
        if (jbest > -1) {
            zf.push_back(z[2 * i]);
            zf.push_back(z[2 * i + 1]);
            idf.push_back((int) jbest);
        } else if (outer > gateAugment) {
            zn.push_back(z[2 * i]);
            zn.push_back(z[2 * i + 1]);
        }
    }
}

void EkfSlam::associate_known(const std::vector<float> &z, const std::vector<int32_t> &ids, std::vector<float> &zf,
                              std::vector<int> &idf, std::vector<float> &zn, std::vector<float> &table) const {
    // upstream indexes z by the FULL id list and passes the table by value (ekfslam.cpp:29, ekfslamwrapper.cpp:81-84):
    // out of bounds / never persisted.  Here: the visible ids, and a persistent table.
    std::vector<int> idn;
    for (size_t i = 0; i < ids.size(); i++) {
        const int ii = ids[i];
        if (table[ii] == -1) {
            zn.push_back(z[2 * i]);
            zn.push_back(z[2 * i + 1]);
            idn.push_back(ii);
        } else {
            zf.push_back(z[2 * i]);
            zf.push_back(z[2 * i + 1]);
            idf.push_back((int) table[ii]);
        }
    }
    const int Nf = num_features();
    for (size_t i = 0; i < idn.size(); i++) table[idn[i]] = (float) (Nf + (int) i);
}

void EkfSlam::batch_update(const std::vector<float> &zf, const float R[4], const std::vector<int> &idf) {
    const int lenz = (int) idf.size(), lenx = (int) x.size();
    if (lenz == 0) return;  // upstream runs choleskyUpdate on 0-row matrices: a no-op
    Mat H(2 * lenz, lenx), RR(2 * lenz, 2 * lenz);
    std::vector<float> v(2 * lenz);
    for (int i = 0; i < lenz; i++) {
        float zp[2], H5[10];
        observe_model(idf[i], zp, H5);
        const int cols[5] = {0, 1, 2, 3 + 2 * idf[i], 4 + 2 * idf[i]};
        for (int k = 0; k < 5; k++) {
            H(2 * i, cols[k]) = H5[k];
            H(2 * i + 1, cols[k]) = H5[5 + k];
        }
        v[2 * i] = zf[2 * i] - zp[0];
        v[2 * i + 1] = trig_offset(zf[2 * i + 1] - zp[1]);
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 2; c++) RR(2 * i + r, 2 * i + c) = R[2 * r + c];
    }
    // choleskyUpdate (core.cpp:275-291)
    Mat PHt = mulT(P, H);
    Mat S = mul(H, PHt);
    const int n = S.r;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) S(i, j) = S(i, j) + RR(i, j);
    // S = (S + S^T) * 0.5 evaluated in place, column by column: the lower triangle gets the true average
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) S(i, j) = (S(i, j) + S(j, i)) * 0.5f;
    Mat L = S;
    llt_lower(L);
    Mat U(n, n);
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) U(i, j) = L(j, i);
    Mat Ui = upper_inverse(U);
    Mat W1 = mul(PHt, Ui);
    Mat W = mulT(W1, Ui);
    for (int i = 0; i < lenx; i++) {
        float acc = 0.0f;
        for (int k = 0; k < n; k++) acc = acc + W(i, k) * v[k];
        x[i] = x[i] + acc;
    }
    Mat WW = mulT(W1, W1);
    for (int i = 0; i < lenx; i++)
        for (int j = 0; j < lenx; j++) P(i, j) = P(i, j) - WW(i, j);
}

void EkfSlam::augment(const std::vector<float> &zn, const float Re[4]) {
    for (size_t q = 0; q + 1 < zn.size(); q += 2) {
        const int len = (int) x.size();
        const float r = zn[q], b = zn[q + 1];
        const float s = std::sin(x[2] + b), c = std::cos(x[2] + b);
        x.push_back(x[0] + r * c);
        x.push_back(x[1] + r * s);
        Mat Gv(2, 3), Gz(2, 2), Rm(2, 2);
        Gv(0, 0) = 1; Gv(0, 2) = -r * s; Gv(1, 1) = 1; Gv(1, 2) = r * c;
        Gz(0, 0) = c; Gz(0, 1) = -r * s; Gz(1, 0) = s; Gz(1, 1) = r * c;
        for (int i = 0; i < 4; i++) Rm.a[i] = Re[i];
        Mat Pn(len + 2, len + 2);
        for (int i = 0; i < len; i++)
            for (int j = 0; j < len; j++) Pn(i, j) = P(i, j);
        Mat Pvv(3, 3);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Pvv(i, j) = P(i, j);
        Mat A = mulT(mul(Gv, Pvv), Gv), B = mulT(mul(Gz, Rm), Gz);
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++) Pn(len + i, len + j) = A(i, j) + B(i, j);
        Mat C = mul(Gv, Pvv);  // vehicle to feature cross-correlation
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 3; j++) {
                Pn(len + i, j) = C(i, j);
                Pn(j, len + i) = C(i, j);
            }
        if (len > 3) {
            Mat Pvm(3, len - 3);
            for (int i = 0; i < 3; i++)
                for (int j = 3; j < len; j++) Pvm(i, j - 3) = P(i, j);
            Mat D = mul(Gv, Pvm);
            for (int i = 0; i < 2; i++)
                for (int j = 3; j < len; j++) {
                    Pn(len + i, j) = D(i, j - 3);
                    Pn(j, len + i) = D(i, j - 3);
                }
        }
        P = Pn;
    }
}

}  // namespace slamhost
