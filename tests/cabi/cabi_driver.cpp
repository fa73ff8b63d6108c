// TEST HARNESS for the two reference-side bindings of INTEGRATION.md (tests/cabi/accel_shim.h, fastslam2gpu_adapter.h):
// plays the part of the reference's callers, with two small stand-ins for the Eigen types whose accessor syntax the
// bindings rely on (column-major storage, v(i) / M(i) linear / M(i, j)), and exports plain C entry points for
// tests/test_gpu_cabi.py.  Built by tests/cabi/Makefile with g++ against include/slamgpu.h + libslamgpu / libslamhost.
#include <cstring>
#include <memory>
#include <string>

#include "accel_shim.h"
#include "fastslam2gpu_adapter.h"
#include <slamhost.h>

namespace {
struct Vec {                       // Eigen::VectorXf's syntax
    std::vector<float> d;
    explicit Vec(int n = 0) : d((size_t) n) {}
    float &operator()(int i) { return d[(size_t) i]; }
    int size() const { return (int) d.size(); }
};
struct Mat {                       // Eigen::MatrixXf's syntax; storage column-major, as Eigen's default
    int r, c;
    std::vector<float> d;
    Mat(int r_ = 0, int c_ = 0) : r(r_), c(c_), d((size_t) r_ * c_) {}
    float &operator()(int i) { return d[(size_t) i]; }                    // linear index = column-major (core.cpp:601, :614)
    float &operator()(int i, int j) { return d[(size_t) j * r + i]; }
};
struct ParticleStub {              // what computeJacobians reads from a Particle (Particle.h:20-29)
    Vec xv_{3};
    std::vector<Vec> xs;
    std::vector<Mat> ps;
    Vec xv() { return xv_; }
    std::vector<Vec> landmarkXs() { return xs; }
    std::vector<Mat> landmarkPs() { return ps; }
};
thread_local std::string g_err;

// The accelerator branch of computeJacobians, as the reference's caller performs it (core.cpp:586-664): write xv, R and the
// idf-selected landmarks into the window in Eigen's linear order, setN, start, spin on isDone, read 16 floats per feature.
void compute_jacobians_via_window(AcceleratorHandler *acc, ParticleStub &p, std::vector<int> &idf, Mat &R, std::vector<Vec> *zp,
                                  std::vector<Mat> *Hv, std::vector<Mat> *Hf, std::vector<Mat> *Sf) {
    Vec xv = p.xv();
    float *win = (float *) acc->getMemoryPointer();
    unsigned wr = 0;
    const uint32_t n = (uint32_t) idf.size();
    for (int i = 0; i < 3; i++) win[wr++] = xv(i);
    for (int i = 0; i < 4; i++) win[wr++] = R(i);
    std::vector<Vec> xs = p.landmarkXs();
    std::vector<Mat> ps = p.landmarkPs();
    for (uint32_t i = 0; i < n; i++) {
        for (int j = 0; j < 2; j++) win[wr++] = xs[(size_t) idf[i]](j);
        for (int j = 0; j < 4; j++) win[wr++] = ps[(size_t) idf[i]](j);
    }
    acc->setN(n);
    acc->start();
    while (!acc->isDone()) {
    }
    unsigned rd = 3 + 4 + (2 + 4) * n;
    for (uint32_t i = 0; i < n; i++) {
        Vec z(2);
        Mat hf(2, 2), hv(2, 3), sf(2, 2);
        z(0) = win[rd++];
        z(1) = win[rd++];
        for (int a = 0; a < 2; a++)          // (the comma initialiser of core.cpp:635-650 fills row by row)
            for (int b = 0; b < 2; b++) hf(a, b) = win[rd++];
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 3; b++) hv(a, b) = win[rd++];
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) sf(a, b) = win[rd++];
        zp->push_back(z);
        Hv->push_back(hv);
        Hf->push_back(hf);
        Sf->push_back(sf);
    }
}

typedef FastSLAMGpuT<ParticleStub, Vec, Mat> FastSLAMGpu;
struct Session {
    FastSLAMGpu algo;
    std::vector<ParticleStub> particles;   // the wrapper's vector<Particle>: handed through, never touched
};
}  // namespace

extern "C" {
const char *cabi_last_error() { return g_err.c_str(); }

// xv[3]; R[4], xf[n][2], Pf[n][4] ROW-major on this C boundary (converted into the column-major stand-ins here, the way the
// reference's data lives in Eigen objects); idf[k] selects landmarks; outputs row-major: zp[k][2], Hv[k][6], Hf[k][4], Sf[k][4]
int cabi_compute_jacobians(const float *xv, const float *R4, const float *xf, const float *Pf4, int nf, const int *idf, int k, float *zp,
                           float *Hv, float *Hf, float *Sf) {
    try {
        static AcceleratorHandler *acc = new AcceleratorHandler();   // the global of SLAMBackendApplication.cpp:11-24
        ParticleStub p;
        for (int i = 0; i < 3; i++) p.xv_(i) = xv[i];
        for (int j = 0; j < nf; j++) {
            Vec x(2);
            Mat P(2, 2);
            x(0) = xf[2 * j];
            x(1) = xf[2 * j + 1];
            for (int a = 0; a < 2; a++)
                for (int b = 0; b < 2; b++) P(a, b) = Pf4[4 * j + 2 * a + b];
            p.xs.push_back(x);
            p.ps.push_back(P);
        }
        Mat R(2, 2);
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) R(a, b) = R4[2 * a + b];
        std::vector<int> ids(idf, idf + k);
        std::vector<Vec> z;
        std::vector<Mat> hv, hf, sf;
        compute_jacobians_via_window(acc, p, ids, R, &z, &hv, &hf, &sf);
        for (int i = 0; i < k; i++) {
            zp[2 * i] = z[(size_t) i](0);
            zp[2 * i + 1] = z[(size_t) i](1);
            for (int a = 0; a < 2; a++) {
                for (int b = 0; b < 3; b++) Hv[6 * i + 3 * a + b] = hv[(size_t) i](a, b);
                for (int b = 0; b < 2; b++) {
                    Hf[4 * i + 2 * a + b] = hf[(size_t) i](a, b);
                    Sf[4 * i + 2 * a + b] = sf[(size_t) i](a, b);
                }
            }
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// ---- the adapter, driven the way FastSLAM2Wrapper::run drives its algorithm object ----
void *cabi_algo_create(int method, int n_particles, int max_landmarks, int n_effective, int use_heading, int add_predict_noise,
                       float wheel_base, float sigma_phi, int tape, int math_mode, unsigned seed) {
    try {
        std::unique_ptr<Session> s(new Session());
        s->algo.addPredictNoise = add_predict_noise;   // fastslam2wrapper.cpp:18-23
        s->algo.useHeading = use_heading;
        s->algo.resample = true;
        s->algo.wheelBase = wheel_base;
        s->algo.sigmaPhi = sigma_phi;
        s->algo.nEffective = n_effective;
        if (tape) {  // libc rand() in the reference's order: libslamhost restates nRandMat::randn / stratifiedRandom
            const bool fs2 = method == SLAMGPU_FASTSLAM2;
            s->algo.drawTape = [fs2](int N, bool need_normals, float *normals, float *strata) {
                if (need_normals && fs2) slamhost_draw_normals(N, 3, normals);   // (FastSLAM1's update samples nothing)
                slamhost_draw_strata(N, strata);
            };
            s->algo.drawPredictNoise = [](int N, float *normals2) { slamhost_draw_normals(N, 2, normals2); };
        }
        s->algo.init(method, n_particles, max_landmarks, seed, math_mode);
        return s.release();
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}
void cabi_algo_destroy(void *h) { delete static_cast<Session *>(h); }

int cabi_algo_predict(void *h, const float xtrue[3], float V, float G, const float Q4[4], float dt) {
    try {
        Session *s = static_cast<Session *>(h);
        Vec x(3);
        for (int i = 0; i < 3; i++) x(i) = xtrue[i];
        Mat Q(2, 2);
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) Q(a, b) = Q4[2 * a + b];
        s->algo.predict(s->particles, x, V, G, Q, dt);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int cabi_algo_update(void *h, const float *zf, const int *idf, int m, const float *zn, int n, const float R4[4]) {
    try {
        Session *s = static_cast<Session *>(h);
        std::vector<Vec> f, nw, z;
        for (int k = 0; k < m; k++) {
            Vec v(2);
            v(0) = zf[2 * k];
            v(1) = zf[2 * k + 1];
            f.push_back(v);
        }
        for (int k = 0; k < n; k++) {
            Vec v(2);
            v(0) = zn[2 * k];
            v(1) = zn[2 * k + 1];
            nw.push_back(v);
        }
        std::vector<int> ids(idf, idf + m);
        Vec table(0);
        Mat R(2, 2);
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 2; b++) R(a, b) = R4[2 * a + b];
        s->algo.update(s->particles, f, nw, ids, z, table, R);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int cabi_algo_estimate(void *h, double xyt[3]) {
    try {
        static_cast<Session *>(h)->algo.estimate(xyt[0], xyt[1], xyt[2]);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

int cabi_algo_landmarks(void *h) {
    try {
        return static_cast<Session *>(h)->algo.landmarkCount();
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// every particle (stride 1): xv[3N], w[N], xf[2 nf N] (caller sizes xf for the capacity)
int cabi_algo_fetch(void *h, float *xv, float *w, float *xf) {
    try {
        Session *s = static_cast<Session *>(h);
        std::vector<float> a, b, c;
        s->algo.fetch(1, a, b, c);
        memcpy(xv, a.data(), 4 * a.size());
        memcpy(w, b.data(), 4 * b.size());
        if (!c.empty()) memcpy(xf, c.data(), 4 * c.size());
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}
}
