/* slamhost — C ABI of the host-side front end that surrounds the hot path in matzipan/slam's wrappers
 * (libslamhost.so, plain C++, no GPU code).  It is what slam-backend links and what bench.py / the
 * tests use to produce the control / observation tape.  Citations are file:line under the reference.
 *
 *  - Conf::parse keys + defaults (src/backend/core.cpp:971-1073), `<map stem>.ini` then `-KEY value`
 *    overrides (SLAMBackendApplication.cpp:59-89, utils.cpp:504-565,1032-1046)
 *  - map reader (core.cpp:855-962)
 *  - SLAMWrapper::control(): updateSteering, predictTruePosition, addControlNoise
 *    (wrappers/slamwrapper.cpp:174-238, core.cpp:24-78)
 *  - getObservations / addObservationNoise / dataAssociationKnown (core.cpp:91-120,185-273,438-449)
 *  - the libc rand() draws of nRandMat::randn and stratifiedRandom in the reference's order
 *    (core.cpp:383-419,751-769), used to feed slamgpu's TAPE mode.
 */
#ifndef SLAMHOST_H
#define SLAMHOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct slamhost_sim slamhost_sim;

typedef struct {
    float V, MAXG, RATEG, WHEELBASE, DT_CONTROLS, sigmaV, sigmaG, MAX_RANGE, DT_OBSERVE, sigmaR, sigmaB, sigmaT;
    float GATE_REJECT, GATE_AUGMENT, AT_WAYPOINT;
    int32_t NUMBER_LOOPS, NPARTICLES, NEFFECTIVE;
    int32_t SWITCH_CONTROL_NOISE, SWITCH_SENSOR_NOISE, SWITCH_INFLATE_NOISE, SWITCH_PREDICT_NOISE,
        SWITCH_SAMPLE_PROPOSAL, SWITCH_HEADING_KNOWN, SWITCH_RESAMPLE, SWITCH_PROFILE, SWITCH_SEED_RANDOM,
        SWITCH_ASSOCIATION_KNOWN, SWITCH_BATCH_UPDATE, SWITCH_USE_IEKF;
    int32_t method;      /* 0 EKF1, 1 FASTSLAM1, 2 FASTSLAM2 (anything else => EKF, SLAMBackendApplication.cpp:26-29) */
    int32_t n_landmarks; /* columns of lm */
    int32_t n_waypoints;
    float Q[4], R[4], Qe[4], Re[4]; /* slamwrapper.cpp:25-37 */
} slamhost_conf;

const char *slamhost_last_error(void);
/* argv as given to slam-backend: -m map.mat -method M [-KEY value]... ; seeds libc rand() (slamwrapper.cpp:48-52) */
slamhost_sim *slamhost_sim_create(int argc, char **argv);
void slamhost_sim_destroy(slamhost_sim *s);
int slamhost_sim_conf(const slamhost_sim *s, slamhost_conf *out);
/* landmarks / waypoints, 2 x n row-major */
int slamhost_sim_map(const slamhost_sim *s, float *lm, float *wp);
/* control(): -1 = run finished, 0 = step without observation, 1 = observation due.  Outputs the noisy
 * controls handed to predict and the true heading. */
int slamhost_sim_control(slamhost_sim *s, float *Vn, float *Gn, float *phi_true);
/* getObservations + noise + dataAssociationKnown against nf_known landmarks already in the map.
 * zf/zn sized 2*n_landmarks, idf n_landmarks.  Returns 0. */
int slamhost_sim_observe(slamhost_sim *s, int32_t nf_known, float *zf, int32_t *idf, int32_t *m, float *zn, int32_t *n);
/* raw observation of the last slamhost_sim_observe: z[2*nz], visible ids */
int slamhost_sim_last_z(const slamhost_sim *s, float *z, int32_t *vis, int32_t *nz);
void slamhost_sim_true(const slamhost_sim *s, float x[3]);
int64_t slamhost_sim_control_steps(const slamhost_sim *s);

/* EKF-SLAM on the host CPU (-method EKF1; BASELINE config 1): EKFSLAMWrapper::run's loop body
 * (wrappers/ekfslamwrapper.cpp:50-84) = control(), observation when due, EKFSLAM::sim (algorithms/ekfslam.cpp:17-43). */
typedef struct slamhost_ekf slamhost_ekf;
slamhost_ekf *slamhost_ekf_create(const slamhost_sim *s);
void slamhost_ekf_destroy(slamhost_ekf *e);
int slamhost_ekf_step(slamhost_ekf *e, slamhost_sim *s); /* -1 finished, 0 control step, 1 with observation */
int slamhost_ekf_state(const slamhost_ekf *e, float *x, float *P, int32_t cap); /* returns dim; P row-major, ld = cap */

/* libc rand() tape in the reference's draw order */
/* The policy that turns the vote of the gated per-particle association (slamgpu_associate: labels by EKFSLAM::dataAssociate,
 * ekfslam.cpp:151-189, per particle) into ONE association per step, as slam-backend -assoc gated applies it (slam_amd/csrc/host/
 * gated.h: supermajority to open, the second stage in world coordinates, landmark credits).  set: tunables by name -- "enabled",
 * "new_share", "match_share", "credit_start", "credit_max", "retire_below", "rescue", "rescue_base", "rescue_per_m",
 * "unique_ratio", "new_factor" (returns -1 for an unknown name).  step: z[2 nz], consensus / support of slamgpu_associate, the
 * pose xv[3] and landmark means xf[2 nf] of ONE particle (slamgpu_peek), MAX_RANGE, room = landmarks that may still be opened;
 * out (sized for nz / nz / nf): the packet of slamgpu_update and the landmarks to hand to slamgpu_retire_landmarks.
 * counts[6]: opened, retired, matched by the second stage, observations left unused, refused as new, landmarks in use. */
typedef struct slamhost_gated slamhost_gated;
slamhost_gated *slamhost_gated_create(void);
void slamhost_gated_destroy(slamhost_gated *g);
int slamhost_gated_set(slamhost_gated *g, const char *name, double value);
int slamhost_gated_step(slamhost_gated *g, const float *z, int32_t nz, const int32_t *consensus, const float *support, const float xv[3],
                        const float *xf, int32_t nf, float max_range, int32_t room, float *zf, int32_t *idf, int32_t *m, float *zn, int32_t *n,
                        int32_t *retire, int32_t *n_retire);
void slamhost_gated_counts(const slamhost_gated *g, int32_t counts[6]);

void slamhost_draw_normals(int32_t count, int32_t dim, float *out); /* count x randn(dim,1): dim+1 rand() each */
int32_t slamhost_draw_strata(int32_t N, float *out);                /* returns the reference's strata count (== N when supported) */
double slamhost_unif_rand(void);                                    /* unifRand (core.cpp:775) */

/* synthetic map generator for BASELINE config 5: n landmarks i.i.d. uniform on [x0,x1]x[y0,y1], SplitMix64(seed) */
void slamhost_synthetic_landmarks(uint64_t seed, int32_t n, float x0, float x1, float y0, float y1, float *lm /*2 x n*/);
int slamhost_write_map(const char *path, const float *lm, int32_t nlm, const float *wp, int32_t nwp);

/* ---- plot wire format + headless sink --------------------------------------------------
 * The backend -> GUI link of the reference: NetworkPlot (src/backend/plotting/NetworkPlot.cpp:22-218) sends one ZeroMQ
 * multipart message per command -- command name, then one frame per value in network byte order as the vendored zmqpp
 * serialises it (libs/zmqpp/message.cpp:225-328) -- over a PAIR socket to tcp://127.0.0.1:4242.  These entry points
 * produce exactly those frames and carry them to any of
 *     tcp://host:port   a ZMTP 3.0 PAIR client: the reference's slam-gui (libzmq) on the other side
 *     file:<path>       a frame file: u32 n_messages, per message u32 n_frames, per frame u32 length + bytes (LE)
 *     gather:<dir>      the GUI's DataGatherer (src/gui/plotting/DataGatherer.cpp:50-138), headless: results.txt,
 *                       errors.txt, times.txt, positions.txt, observedCounts.txt, averageLengthLandmark.txt under
 *                       <dir>/<simulation name>/
 * (several sinks separated by ','; "none" = no sink).  Every call returns 0 on success, -1 on failure
 * (slamhost_last_error). */
typedef struct slamhost_plot slamhost_plot;
slamhost_plot *slamhost_plot_open(const char *spec);
void slamhost_plot_close(slamhost_plot *p);
/* cmd = setLandmarks | setWaypoints | setParticles | setFeatureParticles (NetworkPlot.cpp:22-69) */
int slamhost_plot_xy(slamhost_plot *p, const char *cmd, const double *xs, int32_t nx, const double *ys, int32_t ny);
/* cmd = setLaserLines (idx ignored) | setCovEllipse; the matrix row-major (:71-101) */
int slamhost_plot_matrix(slamhost_plot *p, const char *cmd, uint32_t rows, uint32_t cols, const float *row_major, int32_t idx);
/* cmd = addTruePosition | addEstimatedPosition (2) | setCarTruePosition | setCarEstimatedPosition (3) | setPlotRange (4) */
int slamhost_plot_doubles(slamhost_plot *p, const char *cmd, const double *v, int32_t n);
int slamhost_plot_car_size(slamhost_plot *p, double s, uint32_t id);          /* setCarSize (:123-131) */
int slamhost_plot_u32(slamhost_plot *p, const char *cmd, uint32_t v);         /* loopTime | covEllipseAdd | setCurrentIteration (sends nothing, :176-186) */
int slamhost_plot_cmd(slamhost_plot *p, const char *cmd);                     /* clear | plot | endPlot */
int slamhost_plot_name(slamhost_plot *p, const char *name);                   /* setSimulationName (:168-174) */

#ifdef __cplusplus
}
#endif
#endif
