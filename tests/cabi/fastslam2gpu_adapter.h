// Seam 2, reference side: a device-backed algorithm object with the methods and public tunables of FastSLAM2 / FastSLAM1
// (src/backend/algorithms/fastslam2.h:20-31, fastslam1.h:29-39), as the wrappers call them (fastslam2wrapper.cpp:64,88;
// fastslam1wrapper.cpp:58,81).  The particle vector stays on the device; the wrapper's vector<Particle> argument is ignored.
// Written against the accessor syntax of the reference's types only (v(i), M(i, j)), so the reference tree instantiates it
// with its own Particle / Eigen::VectorXf / Eigen::MatrixXf:
//     typedef FastSLAMGpuT<Particle, VectorXf, MatrixXf> FastSLAM2Gpu;
// (tests/test_cabi_build.py compiles exactly that against the reference headers where they are present) and the GPU test
// (tests/cabi/cabi_driver.cpp) instantiates it with two 20-line stand-ins of the same syntax, because Eigen does not travel.
#ifndef SLAMGPU_FASTSLAM2GPU_ADAPTER_H
#define SLAMGPU_FASTSLAM2GPU_ADAPTER_H

#include <slamgpu.h>

#include <functional>
#include <stdexcept>
#include <vector>

template <class Particle, class VectorXf, class MatrixXf>
class FastSLAMGpuT {
public:
    // fastslam2.h:26-31 / fastslam1.h:34-39: the wrapper constructor copies the switches into these
    bool addPredictNoise = false, useHeading = false, resample = true;
    float wheelBase = 4.0f, sigmaPhi = 0.0f;
    int nEffective = 0;
    // Parity replay (SLAMGPU_RNG_TAPE): draws the tape in the reference's order -- per update N x nRandMat::randn(3,1) when zf
    // or zn is non-empty (FastSLAM2), then stratifiedRandom(N) (core.cpp:383-419, 751-769).  Unset: Philox on the device.
    std::function<void(int n_particles, bool need_normals, float *normals3N, float *strataN)> drawTape;
    // ... and, when addPredictNoise (FastSLAM1: always, fastslam1wrapper.cpp:20), per predict N x nRandMat::randn(2,1): the
    // normals multivariateGauss((V, G), Q) consumes per particle (fastslam1.cpp:44, fastslam2.cpp:91)
    std::function<void(int n_particles, float *normals2N)> drawPredictNoise;

    // called where the wrapper constructor has copied the switches; before the wrapper seeds rand()
    void init(int method, int nParticles, int maxLandmarks, unsigned seed, int mathMode = SLAMGPU_MATH_FAST) {
        slamgpu_config c = slamgpu_config();
        c.struct_size = sizeof c;
        c.method = method;
        c.n_particles = N = nParticles;
        c.max_landmarks = maxLandmarks;
        c.use_heading = useHeading;
        c.add_predict_noise = addPredictNoise;
        c.resample = resample;
        c.n_effective = nEffective;
        c.wheel_base = wheelBase;
        c.sigma_phi = sigmaPhi;
        c.rng_mode = drawTape ? SLAMGPU_RNG_TAPE : SLAMGPU_RNG_PHILOX;
        c.math_mode = mathMode;
        c.seed = seed;
        check(slamgpu_create(&c, &ctx));
        if (drawTape) {
            normals.resize(3 * (size_t) N);
            strata.resize((size_t) N);
        }
    }
    ~FastSLAMGpuT() { slamgpu_destroy(ctx); }

    // FastSLAM2::predict (fastslam2.cpp:51): xTrue(2) is only read when useHeading
    void predict(std::vector<Particle> &, VectorXf &xTrue, float V, float G, MatrixXf &Q, float dt) {
        float q[4] = {Q(0, 0), Q(0, 1), Q(1, 0), Q(1, 1)};
        const float *noise2 = nullptr;
        if (drawTape && addPredictNoise) {
            if (!drawPredictNoise) throw std::runtime_error("tape replay with addPredictNoise needs drawPredictNoise");
            pnoise.resize(2 * (size_t) N);
            drawPredictNoise(N, pnoise.data());
            noise2 = pnoise.data();
        }
        check(slamgpu_predict(ctx, V, G, q, dt, xTrue(2), noise2));
    }
    // FastSLAM2::update (fastslam2.cpp:21) / FastSLAM1::update (fastslam1.cpp:18)
    void update(std::vector<Particle> &, std::vector<VectorXf> &zf, std::vector<VectorXf> &zn, std::vector<int> &idf,
                std::vector<VectorXf> &, VectorXf &, MatrixXf &R) {
        std::vector<float> f, n;
        for (auto &z : zf) {
            f.push_back(z(0));
            f.push_back(z(1));
        }
        for (auto &z : zn) {
            n.push_back(z(0));
            n.push_back(z(1));
        }
        float r[4] = {R(0, 0), R(0, 1), R(1, 0), R(1, 1)};
        const float *nm = nullptr, *st = nullptr;
        if (drawTape) {
            drawTape(N, !zf.empty() || !zn.empty(), normals.data(), strata.data());
            nm = normals.data();
            st = strata.data();
        }
        check(slamgpu_update(ctx, f.data(), idf.data(), (int) idf.size(), n.data(), (int) zn.size(), r, nm, st));
    }
    // ParticleSLAMWrapper::computeEstimatedPosition (ParticleSLAMWrapper.cpp:56-77): replaces the loop over particles
    void estimate(double &x, double &y, double &t) {
        double e[3];
        check(slamgpu_estimate(ctx, e));
        x = e[0];
        y = e[1];
        t = e[2];
    }
    // particles[0].landmarkXs().size(), the Nf handed to dataAssociationKnown (fastslam2wrapper.cpp:84)
    int landmarkCount() {
        const int nf = slamgpu_num_landmarks(ctx);
        if (nf < 0) check(nf);
        return nf;
    }
    // drawParticles / drawFeatureParticles (ParticleSLAMWrapper.cpp:34-54): a decimated, READ-ONLY view (slamgpu_peek)
    void fetch(int stride, std::vector<float> &xv, std::vector<float> &w, std::vector<float> &xf) {
        const int nf = landmarkCount(), cnt = (N + stride - 1) / stride;
        xv.resize(3 * (size_t) cnt);
        w.resize((size_t) cnt);
        xf.resize(2 * (size_t) nf * cnt);
        check(slamgpu_peek(ctx, 0, stride, cnt, xv.data(), nullptr, w.data(), xf.data(), nullptr));
    }
    int particles() const { return N; }

private:
    void check(int rc) {
        if (rc) throw std::runtime_error(slamgpu_last_error());
    }
    slamgpu_ctx *ctx = nullptr;
    int N = 0;
    std::vector<float> normals, strata, pnoise;
};

#endif
