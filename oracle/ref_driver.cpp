// TEST INFRASTRUCTURE — not shipped, never on the product path.
//
// Headless C-ABI driver around the *reference's own objects* (matzipan/slam numeric core:
// src/backend/{core,Particle,utils}.cpp + src/backend/algorithms/*.cpp, compiled where they lie
// under /root/reference by oracle/Makefile into oracle/_ref/libslamref.so).  Nothing here is
// reference code: this file only #includes the reference headers and calls the reference
// functions, restating the wrapper loop (wrappers/fastslam2wrapper.cpp:31-122,
// wrappers/fastslam1wrapper.cpp:32-113, wrappers/slamwrapper.cpp:8-53,174-238,
// wrappers/ParticleSLAMWrapper.cpp:14-32,56-77) minus plotting, because the wrappers themselves
// do not compile here (QThread::wait leftover + ZeroMQ).  It exists to (1) pin the C restatement
// in oracle/slam_oracle.c and (2) generate the golden vectors under tests/golden/.
//
// Only built in the authoring container (needs /root/reference); the GPU box never sees it.

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <Eigen/Dense>

#include "src/backend/core.h"
#include "src/backend/Particle.h"
#include "src/backend/algorithms/fastslam1.h"
#include "src/backend/algorithms/fastslam2.h"
#include "src/backend/algorithms/ekfslam.h"

using namespace std;
using namespace Eigen;

#ifdef REF_DRIVER_ACCEL
// -DJACOBIAN_ACCELERATOR build (oracle/Makefile: refaccel): the global the reference defines in
// SLAMBackendApplication.cpp:11-14 and creates before its wrapper (:22-24); here on first use.
AcceleratorHandler *acceleratorHandler = nullptr;
extern "C" int ref_accel_init() {
    try {
        if (!acceleratorHandler) acceleratorHandler = new AcceleratorHandler();
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "ref_accel_init: %s\n", e.what());
        return -1;
    }
}
#endif

namespace {

struct OpenFS2 : FastSLAM2 {
    using FastSLAM2::deltaXv;
    using FastSLAM2::gaussEvaluate;
    using FastSLAM2::likelihoodGivenXv;
    using FastSLAM2::observeHeading;
    using FastSLAM2::predictState;
    using FastSLAM2::sampleProposal;
};

struct OpenFS1 : FastSLAM1 {
    using FastSLAM1::computeWeight;
    using FastSLAM1::observeHeading;
    using FastSLAM1::predictState;
};

struct OpenEKF : EKFSLAM {
    using EKFSLAM::augment;
    using EKFSLAM::batchUpdate;
    using EKFSLAM::dataAssociate;
    using EKFSLAM::observeHeading;
    using EKFSLAM::predict;
};

MatrixXf mat2(const float *rowmajor) {
    MatrixXf m(2, 2);
    m << rowmajor[0], rowmajor[1], rowmajor[2], rowmajor[3];
    return m;
}

MatrixXf mat3(const float *rowmajor) {
    MatrixXf m(3, 3);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) m(r, c) = rowmajor[3 * r + c];
    return m;
}

void put(const MatrixXf &m, float *rowmajor) {
    for (int r = 0; r < m.rows(); r++)
        for (int c = 0; c < m.cols(); c++) rowmajor[r * m.cols() + c] = m(r, c);
}

Particle makeParticle(const float *xv, const float *Pv9, float w, const float *xf, const float *Pf4, int nf) {
    Particle p;
    VectorXf x(3);
    x << xv[0], xv[1], xv[2];
    p.setXv(x);
    MatrixXf P = Pv9 ? mat3(Pv9) : MatrixXf::Zero(3, 3);
    p.setPv(P);
    p.setW(w);
    for (int i = 0; i < nf; i++) {
        VectorXf f(2);
        f << xf[2 * i], xf[2 * i + 1];
        MatrixXf Pf = mat2(Pf4 + 4 * i);
        p.setLandmarkX(i, f);
        p.setLandmarkP(i, Pf);
    }
    return p;
}

void readParticle(Particle &p, float *xv, float *Pv9, float *w, float *xf, float *Pf4) {
    if (xv)
        for (int i = 0; i < 3; i++) xv[i] = p.xv()(i);
    if (Pv9) put(p.Pv(), Pv9);
    if (w) *w = p.w();
    for (size_t i = 0; i < p.landmarkXs().size(); i++) {
        if (xf) {
            xf[2 * i] = p.landmarkXs()[i](0);
            xf[2 * i + 1] = p.landmarkXs()[i](1);
        }
        if (Pf4) put(p.landmarkPs()[i], Pf4 + 4 * i);
    }
}

vector<VectorXf> zlist(const float *z, int n) {
    vector<VectorXf> out;
    for (int i = 0; i < n; i++) {
        VectorXf v(2);
        v << z[2 * i], z[2 * i + 1];
        out.push_back(v);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------
// Whole-simulation handle
// ---------------------------------------------------------------------------------------------
struct RefSim {
    Conf conf;
    int method;  // 0 EKF, 1 FS1, 2 FS2
    MatrixXf landmarks, waypoints;
    MatrixXf Q, R, Qe, Re;
    float Vtrue, Gtrue, Vnoisy, Gnoisy, dt, dtSum;
    int nLoop, iwp;
    VectorXf xTrue;
    vector<int> landmarkIdentifiers;
    VectorXf table;
    vector<Particle> particles;
    OpenFS2 fs2;
    OpenFS1 fs1;
    // EKF
    OpenEKF ekf;
    VectorXf xEst;
    MatrixXf P;
    vector<int> ekfTable;
    // last observation
    vector<VectorXf> z, zf, zn;
    vector<int> idf, visible;
    long controlSteps, obsSteps;
};

}  // namespace

extern "C" {

// ---------------------------------------------------------------------------------------------
// Function-level entry points (known-answer tests)
// ---------------------------------------------------------------------------------------------

float ref_trig_offset(float a) { return trigonometricOffset(a); }

// core.cpp:579 (CPU branch :666-714).  R, Pf, Hf, Hv, Sf are row-major here.
void ref_compute_jacobians(const float *xv, const float *R4, const float *xf, const float *Pf4, int n, float *zp,
                           float *Hv6, float *Hf4, float *Sf4) {
    Particle p = makeParticle(xv, nullptr, 1.f, xf, Pf4, n);
    vector<int> idf;
    for (int i = 0; i < n; i++) idf.push_back(i);
    MatrixXf R = mat2(R4);
    vector<VectorXf> zpv;
    vector<MatrixXf> Hv, Hf, Sf;
    computeJacobians(p, idf, R, &zpv, &Hv, &Hf, &Sf);
    for (int i = 0; i < n; i++) {
        zp[2 * i] = zpv[i](0);
        zp[2 * i + 1] = zpv[i](1);
        put(Hv[i], Hv6 + 6 * i);
        put(Hf[i], Hf4 + 4 * i);
        put(Sf[i], Sf4 + 4 * i);
    }
}

// fastslam2.cpp:127 — D = 2 or 3, S row-major DxD.
// EKFSLAM::dataAssociate (algorithms/ekfslam.cpp:151-189) for ONE FastSLAM particle: pose known (zero pose covariance),
// landmarks with their own 2x2 covariances: x = [xv; xf_1; ..], P = blockdiag(0, Pf_1, ..).  One observation per call of
// the reference function (its decisions are independent per observation): labels[q] = landmark index, -1 new, -2 dropped.
void ref_data_associate(const float *xv, const float *xf, const float *Pf4, int nf, const float *z, int nz, const float *R4,
                        float gate1, float gate2, int *labels) {
    OpenEKF a;
    const int D = 3 + 2 * nf;
    VectorXf x(D);
    MatrixXf P = MatrixXf::Zero(D, D);
    for (int k = 0; k < 3; k++) x(k) = xv[k];
    for (int j = 0; j < nf; j++) {
        x(3 + 2 * j) = xf[2 * j];
        x(3 + 2 * j + 1) = xf[2 * j + 1];
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 2; c++) P(3 + 2 * j + r, 3 + 2 * j + c) = Pf4[4 * j + 2 * r + c];
    }
    MatrixXf R = mat2(R4);
    for (int q = 0; q < nz; q++) {
        vector<VectorXf> zz(1, VectorXf(2)), zf, zn;
        vector<int> idf;
        zz[0](0) = z[2 * q];
        zz[0](1) = z[2 * q + 1];
        a.dataAssociate(x, P, zz, R, gate1, gate2, zf, idf, zn);
        labels[q] = !idf.empty() ? idf[0] : (!zn.empty() ? -1 : -2);
    }
}

float ref_gauss_evaluate(const float *v, const float *S, int D, int logflag) {
    OpenFS2 a;
    VectorXf vv(D);
    MatrixXf SS(D, D);
    for (int i = 0; i < D; i++) {
        vv(i) = v[i];
        for (int j = 0; j < D; j++) SS(i, j) = S[i * D + j];
    }
    return a.gaussEvaluate(vv, SS, logflag);
}

// core.cpp:275 — 2x2 case.  in/out x[2], P[4]; in v[2], R[4], H[4] (row-major).
void ref_cholesky_update2(float *x, float *P4, const float *v, const float *R4, const float *H4) {
    VectorXf xx(2), vv(2);
    xx << x[0], x[1];
    vv << v[0], v[1];
    MatrixXf P = mat2(P4), R = mat2(R4), H = mat2(H4);
    choleskyUpdate(xx, P, vv, R, H);
    x[0] = xx(0);
    x[1] = xx(1);
    put(P, P4);
}

// core.cpp:294 with H = [0 0 1] (fastslam2.cpp:113-125).
void ref_observe_heading(float *xv, float *Pv9, float phi, float sigmaPhi) {
    OpenFS2 a;
    a.sigmaPhi = sigmaPhi;
    Particle p = makeParticle(xv, Pv9, 1.f, nullptr, nullptr, 0);
    a.observeHeading(p, phi);
    readParticle(p, xv, Pv9, nullptr, nullptr, nullptr);
}

// raw libc stream + core.cpp:383 randn
void ref_rand_stream(unsigned seed, int count, int *out) {
    srand(seed);
    for (int i = 0; i < count; i++) out[i] = rand();
}

void ref_randn(unsigned seed, int m, int n, float *out) {
    srand(seed);
    MatrixXf x = nRandMat::randn(m, n);
    put(x, out);
}

// core.cpp:452 — multivariateGauss for dimension D (2 or 3) after srand(seed).
void ref_multivariate_gauss(unsigned seed, const float *x, const float *P, int D, float *out) {
    srand(seed);
    VectorXf xx(D);
    MatrixXf PP(D, D);
    for (int i = 0; i < D; i++) {
        xx(i) = x[i];
        for (int j = 0; j < D; j++) PP(i, j) = P[i * D + j];
    }
    VectorXf r = multivariateGauss(xx, PP, 1);
    for (int i = 0; i < D; i++) out[i] = r(i);
}

// core.cpp:780 after srand(seed).  Returns number of strata produced (== N when the reference supports N).
int ref_stratified_count(int N) {
    float k = 1.0 / (float) N;
    float temp = k / 2;
    int c = 0;
    while (temp < (1 - k / 2)) {
        c++;
        temp = temp + k;
    }
    return c;
}

void ref_stratified_resample(unsigned seed, const float *w, int N, int *keep, float *neff) {
    srand(seed);
    VectorXf ww(N);
    for (int i = 0; i < N; i++) ww(i) = w[i];
    vector<int> k;
    float ne = 0;
    stratifiedResample(ww, k, ne);
    for (int i = 0; i < N; i++) keep[i] = k[i];
    *neff = ne;
}

// core.cpp:479
void ref_add_feature(const float *xv, const float *zn, int n, const float *R4, float *xf, float *Pf4) {
    Particle p = makeParticle(xv, nullptr, 1.f, nullptr, nullptr, 0);
    vector<VectorXf> z = zlist(zn, n);
    MatrixXf R = mat2(R4);
    addFeature(p, z, R);
    readParticle(p, nullptr, nullptr, nullptr, xf, Pf4);
}

// fastslam2.cpp:70 (no predict noise) — in/out xv[3], Pv[9].
void ref_fs2_predict_state(float *xv, float *Pv9, float V, float G, const float *Q4, float wheelBase, float dt) {
    OpenFS2 a;
    a.addPredictNoise = false;
    a.wheelBase = wheelBase;
    Particle p = makeParticle(xv, Pv9, 1.f, nullptr, nullptr, 0);
    MatrixXf Q = mat2(Q4);
    a.predictState(p, V, G, Q, dt);
    readParticle(p, xv, Pv9, nullptr, nullptr, nullptr);
}

// fastslam1.cpp:37 after srand(seed) (noise always on) — in/out xv[3].
void ref_fs1_predict_state(unsigned seed, float *xv, float V, float G, const float *Q4, float wheelBase, float dt) {
    OpenFS1 a;
    a.addPredictNoise = true;
    a.wheelBase = wheelBase;
    srand(seed);
    Particle p = makeParticle(xv, nullptr, 1.f, nullptr, nullptr, 0);
    MatrixXf Q = mat2(Q4);
    a.predictState(p, V, G, Q, dt);
    readParticle(p, xv, nullptr, nullptr, nullptr, nullptr);
}

// fastslam1.cpp:91
float ref_fs1_compute_weight(const float *xv, const float *xf, const float *Pf4, int nf, const float *zf,
                             const int *idf, int m, const float *R4) {
    OpenFS1 a;
    Particle p = makeParticle(xv, nullptr, 1.f, xf, Pf4, nf);
    vector<VectorXf> z = zlist(zf, m);
    vector<int> id(idf, idf + m);
    MatrixXf R = mat2(R4);
    return a.computeWeight(p, z, id, R);
}

// fastslam2.cpp:290 + core.cpp:132 on ONE particle after srand(seed): the per-particle body of
// FastSLAM2::update for zf != {} (fastslam2.cpp:28-32).  in/out xv, Pv, w, xf, Pf.
void ref_fs2_observe_particle(unsigned seed, float *xv, float *Pv9, float *w, float *xf, float *Pf4, int nf,
                              const float *zf, const int *idf, int m, const float *R4, int do_feature_update) {
    OpenFS2 a;
    srand(seed);
    Particle p = makeParticle(xv, Pv9, *w, xf, Pf4, nf);
    vector<VectorXf> z = zlist(zf, m);
    vector<int> id(idf, idf + m);
    MatrixXf R = mat2(R4);
    a.sampleProposal(p, z, id, R);
    if (do_feature_update) featureUpdate(p, z, id, R);
    readParticle(p, xv, Pv9, w, xf, Pf4);
}

// core.cpp:132 alone.
void ref_feature_update(const float *xv, float *xf, float *Pf4, int nf, const float *zf, const int *idf, int m,
                        const float *R4) {
    Particle p = makeParticle(xv, nullptr, 1.f, xf, Pf4, nf);
    vector<VectorXf> z = zlist(zf, m);
    vector<int> id(idf, idf + m);
    MatrixXf R = mat2(R4);
    featureUpdate(p, z, id, R);
    readParticle(p, nullptr, nullptr, nullptr, xf, Pf4);
}

// ---------------------------------------------------------------------------------------------
// Whole-simulation API
// ---------------------------------------------------------------------------------------------

void *ref_sim_create(int argc, char **argv) {
    RefSim *s = new RefSim();
    // SLAMBackendApplication.cpp:59-89
    string mapFilename = "example_webmap.mat";
    for (int i = 1; i < argc; i++)
        if (strcmp(argv[i], "-m") == 0 && i + 1 < argc) mapFilename = argv[i + 1];
    size_t dot = mapFilename.find_last_of(".");
    string ini = (dot == string::npos ? mapFilename : mapFilename.substr(0, dot)) + ".ini";
    s->conf.load(ini);
    s->conf.set_args(argc, argv);
    s->conf.parse();
    string method = s->conf.s("method");
    s->method = method == "FASTSLAM1" ? 1 : (method == "FASTSLAM2" ? 2 : 0);

    // slamwrapper.cpp:8-53
    readInputFile(s->conf.s("m"), &s->landmarks, &s->waypoints);
    Conf *c = &s->conf;
    s->Vtrue = c->V;
    s->Gtrue = 0;
    s->Vnoisy = 0;
    s->Gnoisy = 0;
    s->Q = MatrixXf(2, 2);
    s->Q << pow(c->sigmaV, 2), 0, 0, pow(c->sigmaG, 2);
    s->R = MatrixXf(2, 2);
    s->R << pow(c->sigmaR, 2), 0, 0, pow(c->sigmaB, 2);
    if (c->SWITCH_INFLATE_NOISE == 1) {
        s->Q = 2 * s->Q;
        s->R = 2 * s->R;
    } else {
        s->Qe = MatrixXf(s->Q);
        s->Re = MatrixXf(s->R);
    }
    s->nLoop = c->NUMBER_LOOPS;
    s->dt = c->DT_CONTROLS;
    s->dtSum = 0;
    s->iwp = 0;
    s->xTrue = VectorXf(3);
    s->xTrue.setZero(3);
    if (c->SWITCH_SEED_RANDOM != 0) srand(c->SWITCH_SEED_RANDOM);
    s->controlSteps = s->obsSteps = 0;

    for (int i = 0; i < s->landmarks.cols(); i++) s->landmarkIdentifiers.push_back(i);

    if (s->method != 0) {
        // ParticleSLAMWrapper.cpp:8-32
        s->particles = vector<Particle>(c->NPARTICLES);
        for (size_t i = 0; i < s->particles.size(); i++) s->particles[i] = Particle();
        float uniformw = 1.0 / (float) s->particles.size();
        for (size_t i = 0; i < s->particles.size(); i++) s->particles[i].setW(uniformw);
        s->table = VectorXf(s->landmarks.cols());
        for (int i = 0; i < s->table.size(); i++) s->table[i] = -1;
    }
    // fastslam2wrapper.cpp:18-23 / fastslam1wrapper.cpp:20-25
    s->fs2.addPredictNoise = c->SWITCH_PREDICT_NOISE == 1;
    s->fs2.useHeading = c->SWITCH_HEADING_KNOWN == 1;
    s->fs2.wheelBase = c->WHEELBASE;
    s->fs2.nEffective = c->NEFFECTIVE;
    s->fs2.resample = c->SWITCH_RESAMPLE == 1;
    s->fs2.sigmaPhi = c->sigmaT;
    s->fs1.addPredictNoise = 1;
    s->fs1.useHeading = c->SWITCH_HEADING_KNOWN == 1;
    s->fs1.wheelBase = c->WHEELBASE;
    s->fs1.nEffective = c->NEFFECTIVE;
    s->fs1.resample = c->SWITCH_RESAMPLE == 1;
    s->fs1.sigmaPhi = c->sigmaT;
    // ekfslamwrapper.cpp:17-27
    s->ekf.enableBatchUpdate = c->SWITCH_BATCH_UPDATE == 1;
    s->ekf.useHeading = c->SWITCH_HEADING_KNOWN == 1;
    s->ekf.wheelBase = c->WHEELBASE;
    s->ekf.gateReject = c->GATE_REJECT;
    s->ekf.gateAugment = c->GATE_AUGMENT;
    s->ekf.associationKnown = c->SWITCH_ASSOCIATION_KNOWN;
    s->ekf.sigmaPhi = c->sigmaT;
    if (s->method == 0) {
        s->xEst = VectorXf(3);
        s->xEst.setZero(3);
        s->P = MatrixXf(3, 3);
        s->P.setZero(3, 3);
        for (int i = 0; i < s->landmarks.cols(); i++) s->ekfTable.push_back(-1);
    }
    return s;
}

void ref_sim_destroy(void *h) { delete (RefSim *) h; }

// First half of one wrapper-loop iteration: control() + predict (fastslam2wrapper.cpp:54-66).
// Returns -1 finished, 0 no observation due, 1 observation due (call ref_sim_observe next).
int ref_sim_control(void *h) {
    RefSim *s = (RefSim *) h;
    Conf *c = &s->conf;
    // slamwrapper.cpp:174-238 control()
    if (s->iwp == -1) return -1;
    updateSteering(s->xTrue, s->waypoints, s->iwp, c->AT_WAYPOINT, s->Gtrue, c->RATEG, c->MAXG, s->dt);
    if (s->iwp == -1 && s->nLoop > 1) {
        s->iwp = 0;
        s->nLoop--;
    }
    if (s->iwp == -1 && s->nLoop == 1) return -1;
    predictTruePosition(s->xTrue, s->Vtrue, s->Gtrue, c->WHEELBASE, s->dt);
    if (c->SWITCH_CONTROL_NOISE) addControlNoise(s->Vtrue, s->Gtrue, s->Q, s->Vnoisy, s->Gnoisy);
    s->controlSteps++;

    if (s->method == 2) s->fs2.predict(s->particles, s->xTrue, s->Vnoisy, s->Gnoisy, s->Qe, s->dt);
    if (s->method == 1) s->fs1.predict(s->particles, s->xTrue, s->Vnoisy, s->Gnoisy, s->Qe, s->dt);

    s->dtSum += s->dt;
    if (s->dtSum >= c->DT_OBSERVE) {
        s->dtSum = 0;
        return 1;
    }
    return 0;
}

// Second half (fastslam2wrapper.cpp:72-90): observe, add noise, associate, update.
void ref_sim_observe(void *h) {
    RefSim *s = (RefSim *) h;
    Conf *c = &s->conf;
    s->visible = vector<int>(s->landmarkIdentifiers);
    s->z = getObservations(s->landmarks, s->xTrue, s->visible, c->MAX_RANGE);
    if (c->SWITCH_SENSOR_NOISE) addObservationNoise(s->z, s->R);
    if (s->method != 0) {
        unsigned long Nf = s->particles[0].landmarkXs().size();
        dataAssociationKnown(s->z, s->visible, s->table, Nf, s->zf, s->idf, s->zn);
        if (s->method == 2) s->fs2.update(s->particles, s->zf, s->zn, s->idf, s->z, s->table, s->Re);
        if (s->method == 1) s->fs1.update(s->particles, s->zf, s->zn, s->idf, s->visible, s->table, s->Re);
    }
    s->obsSteps++;
}

// One iteration of the wrapper main loop.  Returns -1 finished, 0 control step without observation,
// 1 control step with an observation update.
int ref_sim_step(void *h) {
    RefSim *s = (RefSim *) h;
    Conf *c = &s->conf;
    int r = ref_sim_control(h);
    if (r < 0) return r;
    bool observe = r == 1;
    if (observe) ref_sim_observe(h);
    if (s->method == 0) {
        // ekfslamwrapper.cpp:81-84 (zf/idf/zn/table are by-value scratch inside sim)
        vector<VectorXf> zf, zn;
        vector<int> idf;
        s->ekf.sim(s->landmarks, s->waypoints, s->xEst, s->P, s->Vnoisy, s->Gnoisy, s->Qe, s->dt,
                   s->xTrue(2) + c->sigmaT * unifRand(), s->landmarkIdentifiers, s->z, s->Re, observe, zf, idf, zn,
                   s->ekfTable, s->R);
    }
    return observe ? 1 : 0;
}

int ref_sim_nparticles(void *h) { return (int) ((RefSim *) h)->particles.size(); }
int ref_sim_nf(void *h) {
    RefSim *s = (RefSim *) h;
    if (s->method == 0) return (int) (s->xEst.size() - 3) / 2;
    return s->particles.empty() ? 0 : (int) s->particles[0].landmarkXs().size();
}
long ref_sim_control_steps(void *h) { return ((RefSim *) h)->controlSteps; }
long ref_sim_obs_steps(void *h) { return ((RefSim *) h)->obsSteps; }
int ref_sim_nlandmarks(void *h) { return (int) ((RefSim *) h)->landmarks.cols(); }

void ref_sim_true(void *h, float *x3, float *VnGn) {
    RefSim *s = (RefSim *) h;
    for (int i = 0; i < 3; i++) x3[i] = s->xTrue(i);
    if (VnGn) {
        VnGn[0] = s->Vnoisy;
        VnGn[1] = s->Gnoisy;
    }
}

// ParticleSLAMWrapper.cpp:56-77
void ref_sim_estimate(void *h, double *xyt) {
    RefSim *s = (RefSim *) h;
    if (s->method == 0) {
        for (int i = 0; i < 3; i++) xyt[i] = s->xEst(i);
        return;
    }
    double x = 0, y = 0, t = 0, wMax = -1e30;
    for (size_t i = 0; i < s->particles.size(); i++) {
        if (s->particles[i].w() > wMax) {
            wMax = s->particles[i].w();
            t = s->particles[i].xv()(2);
        }
        x += s->particles[i].xv()(0);
        y += s->particles[i].xv()(1);
    }
    xyt[0] = x / s->particles.size();
    xyt[1] = y / s->particles.size();
    xyt[2] = t;
}

// Copies of the particle set: xv[3N] (particle-major), Pv[9N] row-major, w[N], xf[2*Nf*N], Pf[4*Nf*N].
void ref_sim_get_particles(void *h, float *xv, float *Pv9, float *w, float *xf, float *Pf4) {
    RefSim *s = (RefSim *) h;
    int nf = ref_sim_nf(h);
    for (size_t i = 0; i < s->particles.size(); i++)
        readParticle(s->particles[i], xv ? xv + 3 * i : nullptr, Pv9 ? Pv9 + 9 * i : nullptr, w ? w + i : nullptr,
                     xf ? xf + 2 * nf * i : nullptr, Pf4 ? Pf4 + 4 * nf * i : nullptr);
}

// Last observation packet.  Returns m (=|zf|); *n_out = |zn|; *nz_out = |z|.
int ref_sim_last_obs(void *h, float *zf, int *idf, float *zn, int *n_out, float *z, int *vis, int *nz_out) {
    RefSim *s = (RefSim *) h;
    for (size_t i = 0; i < s->zf.size(); i++) {
        if (zf) {
            zf[2 * i] = s->zf[i](0);
            zf[2 * i + 1] = s->zf[i](1);
        }
        if (idf) idf[i] = s->idf[i];
    }
    for (size_t i = 0; i < s->zn.size(); i++)
        if (zn) {
            zn[2 * i] = s->zn[i](0);
            zn[2 * i + 1] = s->zn[i](1);
        }
    for (size_t i = 0; i < s->z.size(); i++) {
        if (z) {
            z[2 * i] = s->z[i](0);
            z[2 * i + 1] = s->z[i](1);
        }
        if (vis) vis[i] = s->visible[i];
    }
    if (n_out) *n_out = (int) s->zn.size();
    if (nz_out) *nz_out = (int) s->z.size();
    return (int) s->zf.size();
}

// EKF state: x[dim], diag(P)[dim], returns dim.
int ref_sim_ekf_state(void *h, float *x, float *P, int cap) {
    RefSim *s = (RefSim *) h;
    int d = (int) s->xEst.size();
    for (int i = 0; i < d && i < cap; i++) x[i] = s->xEst(i);
    if (P)
        for (int i = 0; i < d && i < cap; i++)
            for (int j = 0; j < d && j < cap; j++) P[i * cap + j] = s->P(i, j);
    return d;
}

}  // extern "C"
