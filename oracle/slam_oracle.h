/* TEST INFRASTRUCTURE — CPU restatement ("oracle") of the matzipan/slam FastSLAM/EKF inner loop.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker / reported baseline.  The product (slam_amd/, include/slamgpu.h) never links,
 * loads or calls it.
 *
 * Parity status: PINNED.  Every function below is checked against the reference's own objects
 * (oracle/_ref/libslamref.so, built from /root/reference by oracle/Makefile) and against the golden
 * vectors under tests/golden/ that were generated from those objects (tests/golden/make_golden.py).
 *
 * All matrices are row-major float32 unless said otherwise.  Citations are file:line under
 * /root/reference/.
 */
#ifndef SLAM_ORACLE_H
#define SLAM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- scalar / small-matrix pieces -------------------------------------------------------- */
float orc_trig_offset(float ang);                                   /* core.cpp:460-477 */
void orc_compute_jacobians(const float *xv, const float *R4, const float *xf, const float *Pf4, int n, float *zp,
                           float *Hv6, float *Hf4, float *Sf4);     /* core.cpp:666-714 */
float orc_gauss_evaluate(const float *v, const float *S, int D, int logflag); /* fastslam2.cpp:127-163 */
void orc_cholesky_update2(float *x, float *P4, const float *v, const float *R4, const float *H4); /* core.cpp:275 */
void orc_observe_heading(float *xv, float *Pv9, float phi, float sigmaPhi); /* fastslam2.cpp:113 + core.cpp:294 */
void orc_add_feature(const float *xv, const float *zn, int n, const float *R4, float *xf, float *Pf4); /* core.cpp:479 */
void orc_fs2_predict_state(float *xv, float *Pv9, float V, float G, const float *Q4, float wheelBase, float dt,
                           const float *noise2 /* NULL or 2 normals */);                /* fastslam2.cpp:70-105 */
void orc_fs1_predict_state(float *xv, float V, float G, const float *Q4, float wheelBase, float dt,
                           const float *noise2);                                        /* fastslam1.cpp:37-54 */
float orc_fs1_compute_weight(const float *xv, const float *xf, const float *Pf4, const float *zf, const int *idf,
                             int m, const float *R4);                                   /* fastslam1.cpp:91-118 */
void orc_feature_update(const float *xv, float *xf, float *Pf4, const float *zf, const int *idf, int m,
                        const float *R4);                                               /* core.cpp:132-175 */
/* fastslam2.cpp:290-368 on one particle; g3 = the three normals multivariateGauss would draw. */
void orc_fs2_sample_proposal(float *xv, float *Pv9, float *w, const float *xf, const float *Pf4, const float *zf,
                             const int *idf, int m, const float *R4, const float *g3);
/* 3x3 / 2x2 helpers exposed for tests */
int orc_llt_lower(int n, const float *A, float *L);                 /* Eigen LLT.h:260-287 order; returns -1 ok else k */
void orc_llt_solve_identity(int n, const float *A, float *X);       /* A.llt().solve(I) */
void orc_lu_inverse(int n, const float *A, float *X);               /* A.inverse() for dynamic sizes (PartialPivLU) */
float orc_lu_determinant(int n, const float *A);
void orc_multivariate_gauss(const float *x, const float *P, int D, const float *g, float *out); /* core.cpp:452 */

/* ---- libc-rand() tape in the reference's draw order --------------------------------------- */
void orc_srand(unsigned seed);
/* nRandMat::randn(m,n) (core.cpp:383-419): draws m*n+1 rand() values. */
void orc_randn(int m, int n, float *out);
/* stratifiedRandom (core.cpp:751-769): returns the number of strata the reference loop produces;
 * when it equals N (reference-supported N) sel[] holds the dithered strata, drawing N rand() values.
 * Otherwise falls back to the build's definition sel[i] = (i + u_i)/N (double arithmetic, N draws). */
int orc_stratified_random(int N, float *sel);
float orc_eigen_sum(const float *v, int n);                         /* Eigen Redux.h:200-240 + SSE2 predux order */
/* stratifiedResample (core.cpp:780-807) given precomputed strata sel[N]. */
void orc_stratified_resample(const float *w, int N, const float *sel, int *keep, float *neff);

/* ---- Philox4x32-10 (the build's throughput-mode RNG; identical on device) ----------------- */
void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]);
/* normals[3*N] particle-major, stream 0 of (seed, step); sel[N] = (i+u)/N, stream 1. */
void orc_philox_update_tape(uint64_t seed, uint32_t step, int first, int count, int Ntotal, float *normals,
                            float *sel);
void orc_philox_predict_tape(uint64_t seed, uint32_t step, int first, int count, float *normals2);

/* ---- particle set ------------------------------------------------------------------------- */
typedef struct orc_particles orc_particles;
orc_particles *orc_particles_create(int N, int cap_nf);
void orc_particles_destroy(orc_particles *p);
int orc_particles_n(const orc_particles *p);
int orc_particles_nf(const orc_particles *p);
/* particle-major copies: xv[3N], Pv[9N], w[N], xf[2*Nf*N], Pf[4*Nf*N]; NULLs skipped. */
void orc_particles_get(const orc_particles *p, float *xv, float *Pv9, float *w, float *xf, float *Pf4);
void orc_particles_set(orc_particles *p, int nf, const float *xv, const float *Pv9, const float *w, const float *xf,
                       const float *Pf4);
void orc_estimate(const orc_particles *p, double *xyt);             /* ParticleSLAMWrapper.cpp:56-77 */
/* log-weight extension (mirrors slamgpu_config.log_weights; see slam_oracle.c): w[] becomes log-weights */
void orc_particles_set_log_weights(orc_particles *p, int on);
int orc_particles_log_weights(const orc_particles *p);

typedef struct {
    int method;          /* 1 FASTSLAM1, 2 FASTSLAM2 */
    int use_heading;     /* SWITCH_HEADING_KNOWN */
    int add_predict_noise; /* FS2: SWITCH_PREDICT_NOISE; FS1: always 1 */
    int resample;        /* SWITCH_RESAMPLE */
    int n_effective;     /* NEFFECTIVE */
    float wheel_base;
    float sigma_phi;
} orc_algo;

/* FastSLAM{1,2}::predict over all particles (fastslam2.cpp:51-60, fastslam1.cpp:57-66).
 * noise2: NULL or 2N normals (particle-major) for the control-noise sample. */
void orc_predict(orc_particles *p, const orc_algo *a, float V, float G, const float *Q4, float dt, float phi_true,
                 const float *noise2);
/* FastSLAM{1,2}::update (fastslam2.cpp:21-48, fastslam1.cpp:18-35) incl. resampleParticles (core.cpp:718-749).
 * normals: 3N (used by FS2 when m>0, or n>0 && m==0), sel: N strata.  Outputs optional. */
void orc_update(orc_particles *p, const orc_algo *a, const float *zf, const int *idf, int m, const float *zn, int n,
                const float *R4, const float *normals, const float *sel, int *keep_out, float *neff_out,
                int *resampled_out);

/* resampleParticles alone, optionally with an externally supplied decision and ancestor list (see slam_oracle.c) */
void orc_resample_forced(orc_particles *p, const orc_algo *a, const float *sel, int forced_did, const int *forced_keep,
                         int *keep_out, float *neff_out, int *resampled_out);

/* The per-particle loop of the update only (no resampleParticles): used by the sharded-path tests. */
void orc_update_local(orc_particles *p, const orc_algo *a, const float *zf, const int *idf, int m, const float *zn, int n,
                      const float *R4, const float *normals);

/* ---- host front end (simulator) ----------------------------------------------------------- */
typedef struct orc_sim orc_sim;
/* Same CLI surface as slam-backend: -m map -method M -KEY value ... (SLAMBackendApplication.cpp:59-89). */
orc_sim *orc_sim_create(int argc, char **argv);
void orc_sim_destroy(orc_sim *s);
/* rng_mode 0: libc rand() in reference order (bit-parity with the reference);
 *          1: control/observation noise from libc rand(), particle noise from Philox(seed,step). */
void orc_sim_set_rng(orc_sim *s, int rng_mode, uint64_t seed);
int orc_sim_step(orc_sim *s);        /* -1 finished, 0 control step, 1 control step + observation update */
int orc_sim_control(orc_sim *s);     /* first half: control + predict; -1 finished, 1 = observation due */
void orc_sim_observe(orc_sim *s);    /* second half: observe + associate + update */
void orc_sim_observe_local(orc_sim *s);  /* ... split again: everything but resampleParticles */
void orc_sim_resample(orc_sim *s, int forced_did, const int *forced_keep, int *own_keep);
orc_particles *orc_sim_particles(orc_sim *s);
int orc_sim_nlandmarks(const orc_sim *s);
void orc_sim_true(const orc_sim *s, float *x3, float *VnGn);
int orc_sim_last_obs(const orc_sim *s, float *zf, int *idf, float *zn, int *n_out, float *z, int *vis, int *nz_out);
void orc_sim_last_resample(const orc_sim *s, float *neff, int *resampled);
/* Tape of the last update (valid after orc_sim_step returned 1): normals 3N, sel N. */
void orc_sim_last_tape(const orc_sim *s, float *normals, float *sel);
/* Control-noise normals of the last predict (2N, particle-major; valid when add_predict_noise). */
void orc_sim_last_noise2(const orc_sim *s, float *noise2);
const orc_algo *orc_sim_algo(const orc_sim *s);
void orc_sim_noise(const orc_sim *s, float *Q4, float *R4, float *dt);
/* EKF (config 1): state vector, covariance (row-major dim x dim into P with leading dimension cap). */
int orc_sim_ekf_state(const orc_sim *s, float *x, float *P, int cap);

/* map / ini helpers (core.cpp:855-962, utils.cpp:504-565) */
int orc_read_map(const char *path, float **lm, int *nlm, float **wp, int *nwp);
void orc_free(void *p);

#ifdef __cplusplus
}
#endif
/* bench.py cpu_baseline only: OpenMP threads for the per-particle loops (default 1; results are thread-count invariant) */
void orc_set_threads(int n);
int orc_get_threads(void);

#endif
