/* slamgpu — C ABI of the MI355X-native FastSLAM inner loop (libslamgpu.so).
 *
 * This is the drop-in boundary for matzipan/slam's hot path.  Citations are file:line under the
 * reference tree.  It replaces two seams of the reference:
 *
 *  Seam 1 — the accelerator object used by computeJacobians when built with
 *           -DJACOBIAN_ACCELERATOR=on: class AcceleratorHandler { getMemoryPointer(); setN(n);
 *           start(); isDone(); } (src/backend/AcceleratorHandler.h:10-23) with the packed float32
 *           window written/read at src/backend/core.cpp:586-664.  => slamgpu_jacobians().
 *           Built with -DMULTIPARTICLE_ACCELERATOR=on the object has setParticlesCount(count) in the place of setN
 *           (AcceleratorHandler.h:17-21) and the window holds `count` self-describing records
 *           (src/backend/algorithms/fastslam2.cpp:172-286).  => slamgpu_jacobians_multi().
 *
 *  Seam 2 — the algorithm objects the wrappers drive:
 *           FastSLAM2::predict / FastSLAM2::update (src/backend/algorithms/fastslam2.h:20-24, called at
 *           src/backend/wrappers/fastslam2wrapper.cpp:64,88) and FastSLAM1::predict / ::update
 *           (src/backend/algorithms/fastslam1.h:29-33, fastslam1wrapper.cpp:58,81), whose tunables are
 *           the public fields set in fastslam2wrapper.cpp:18-23, plus the per-step pose output
 *           ParticleSLAMWrapper::computeEstimatedPosition (ParticleSLAMWrapper.cpp:56-77).
 *           => slamgpu_create / _predict / _update / _estimate / _download / _destroy.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every entry point returns 0 on success
 * or a negative slamgpu_status, never throws or aborts; slamgpu_last_error() describes the last
 * failure on the calling thread.  Host pointers are never retained across calls.  One context is
 * driven by one host thread.  All 2x2 / 3x3 matrices are row-major float32 unless stated.
 * Work is enqueued on the context's HIP stream; calls that return data synchronise that stream.
 */
#ifndef SLAMGPU_H
#define SLAMGPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 4): SLAMGPU_STATUS_CAPACITY; slamgpu_run_observe; slamgpu_jacobians_multi; the round-1 exchange path, the push / fold collectives, the raw
 *    device-buffer helpers and slamgpu_debug_stamps moved behind SLAMGPU_EXPERIMENTAL (still exported, no longer stable).
 * Everything declared outside the SLAMGPU_EXPERIMENTAL block at the end of this file is STABLE: same name, same argument
 * meaning and same error behaviour for a given SLAMGPU_ABI_VERSION. */
#define SLAMGPU_ABI_VERSION 3

typedef enum {
    SLAMGPU_OK = 0,
    SLAMGPU_ERR_INVALID = -1,   /* bad argument / state */
    SLAMGPU_ERR_HIP = -2,       /* a HIP runtime call failed (message in slamgpu_last_error) */
    SLAMGPU_ERR_CAPACITY = -3,  /* landmark capacity exceeded */
    SLAMGPU_ERR_NO_DEVICE = -4, /* no usable GPU: the product path has no CPU fallback */
    SLAMGPU_ERR_ALLOC = -5,
    SLAMGPU_ERR_BARRIER = -6    /* distributed contexts, push / fold collective: a peer did not arrive at a flag barrier in
                                   time; every step since then ran unsynchronised and its results are void.  Sticky: destroy
                                   the contexts (or switch to SLAMGPU_DIST_GATHER and start the run again).  Also: the
                                   persistent step loop of slamgpu_run_observe abandoned a launch (a workgroup waited too long
                                   at the in-launch barrier); sticky, the steps of that launch are void */
} slamgpu_status;

enum { SLAMGPU_FASTSLAM1 = 1, SLAMGPU_FASTSLAM2 = 2 };
enum {
    SLAMGPU_RNG_TAPE = 0,  /* caller feeds the normals / strata it drew (parity with the reference's libc rand()) */
    SLAMGPU_RNG_PHILOX = 1 /* counter-based Philox4x32-10 keyed by (seed, step, global particle id) on device */
};
enum {
    SLAMGPU_MATH_STRICT = 0, /* no FMA contraction, IEEE divide/sqrt: closest to the reference's SSE2 float math */
    SLAMGPU_MATH_FAST = 1    /* FMA contraction allowed (results within the documented tolerance of STRICT) */
};

/* status bits of an update's resampling stage (slamgpu_step_status, slamgpu_history_fetch) */
enum {
    SLAMGPU_STATUS_DEGENERATE = 1, /* the sum of the weights was zero or not finite: the reference normalises to NaN here
                                      (core.cpp:726-729) and so does the device; the step is flagged instead of hidden */
    SLAMGPU_STATUS_CAPACITY = 4    /* slamgpu_step_observe: the device front end saw more new landmarks than the context has room
                                      for and dropped the surplus.  Sticky for the life of the context (slamgpu_step_status);
                                      the first call that brings the device's bookkeeping back to the host also returns
                                      SLAMGPU_ERR_CAPACITY, once */
};

typedef struct slamgpu_ctx slamgpu_ctx;

/* Mirrors the public fields of FastSLAM2 / FastSLAM1 (fastslam2.h:26-31) + sizes. Zero-initialise,
 * set struct_size = sizeof(slamgpu_config), then fill. */
typedef struct {
    uint32_t struct_size;
    int32_t device;            /* HIP device ordinal */
    int32_t method;            /* SLAMGPU_FASTSLAM1 / SLAMGPU_FASTSLAM2 */
    int32_t n_particles;       /* particles held by THIS context (its shard) */
    int32_t max_landmarks;     /* landmark capacity per particle */
    int32_t use_heading;       /* useHeading  (SWITCH_HEADING_KNOWN) */
    int32_t add_predict_noise; /* addPredictNoise (FS1: always 1, fastslam1wrapper.cpp:20) */
    int32_t resample;          /* resample (SWITCH_RESAMPLE) */
    int32_t n_effective;       /* nEffective (NEFFECTIVE), compared against the GLOBAL particle count's Neff */
    float wheel_base;          /* wheelBase */
    float sigma_phi;           /* sigmaPhi (sigmaT) */
    int32_t rng_mode;          /* SLAMGPU_RNG_* */
    int32_t math_mode;         /* SLAMGPU_MATH_* */
    uint64_t seed;             /* Philox key */
    /* sharding: this context holds global particles [first_particle, first_particle + n_particles) of
     * n_particles_global.  Single GPU: first_particle = 0, n_particles_global = n_particles (or 0). */
    int64_t first_particle;
    int64_t n_particles_global;
    /* 0: the context creates its own HIP stream.  Otherwise a hipStream_t owned by the caller (e.g. the stream
     * the caller's RCCL collectives are ordered on); the context launches on it and never destroys it. */
    uint64_t external_stream;
    /* 1: the context keeps LOG-weights: every weight factor of FastSLAM{1,2}::update enters as its logarithm -- the
     * reference's own gaussEvaluate(v, S, logflag = 1) branch (fastslam2.cpp:154-160), which upstream never calls --
     * and resampleParticles (core.cpp:718-749) works on exp(l - max l).  For maps where a step re-observes more than
     * ~20 landmarks (BASELINE config 5: ~1.3 k): the reference's float32 weight is a product of ~90 per landmark and
     * overflows to inf there.  w[] of slamgpu_download / _upload then holds log-weights (normalised so that
     * sum exp(l) = 1 after an update without resampling, log(1/N) after a resample); slamgpu_stats reports
     * log(sum of the raw weights) as weight_sum.  Single contexts only (not shards).  0 (default): the reference's
     * linear float32 weights, bit-compatible with rounds before this flag existed. */
    int32_t log_weights;
    /* SLAMGPU_FLAG_* bits (0: none) */
    int32_t flags;
} slamgpu_config;

/* slamgpu_config.flags */
enum {
    /* The context will be stepped with slamgpu_step_observe (the observation front end on the device).  Needed for landmark
     * capacities above 39: such a context keeps its observation packets in device memory (a ring written by the front-end
     * kernel).  Contexts of up to 39 landmarks make the observation inside the update launch and accept slamgpu_step_observe
     * with or without the flag. */
    SLAMGPU_FLAG_DEVICE_OBSERVE = 1,
    /* (round 5) A strict-build context of at most 5 000 particles (the largest set the reference itself can resample: its strata count is exact for N = 50, 100, 500, 1 000, 5 000 only, core.cpp:751-763) that takes the caller's draws (SLAMGPU_RNG_TAPE: the parity
     * configuration) runs its resampling stage in the REFERENCE'S OWN ORDER OF OPERATIONS -- float32 w / sum(w) with Eigen's
     * packet-order sum, Neff the same way, the serial float32 running prefix, `select < cum` (core.cpp:718-824) -- so that Neff,
     * the decision and every ancestor are the reference's bit for bit (two extra small launches per step).  This flag turns that
     * off: the context then scans in double like every other one (what a comparison with a sharded run of the same particles
     * needs: shards always do). */
    SLAMGPU_FLAG_NO_REFERENCE_RESAMPLE = 2,
    /* (round 6) The context is going to be stepped with a per-particle association (slamgpu_update_particle /
     * slamgpu_update_labels): plain genealogy rows and device-memory packets whatever the landmark capacity. */
    SLAMGPU_FLAG_PARTICLE_MAPS = 4
};

const char *slamgpu_last_error(void);
int slamgpu_abi_version(void);
int slamgpu_device_count(void);

/* ---- Seam 1: AcceleratorHandler-compatible batched computeJacobians -------------------------------
 * in : xv[3], R[4] (Eigen linear = column-major, core.cpp:600-602), then per feature xf[2], Pf[4]
 *      column-major (core.cpp:608-617)                                        => 7 + 6n floats
 * out: per feature zp[2], Hf[4] row-major, Hv[6] row-major, Sf[4] row-major (core.cpp:631-651)
 *                                                                              => 16n floats
 * Host pointers; synchronous (this is the start()/isDone() spin of core.cpp:619-622).  Both seam-1 calls take no context:
 * they run on the calling thread's CURRENT HIP device (hipSetDevice; device 0 by default), on its null stream, and keep a
 * grow-only device scratch per calling thread. */
int slamgpu_jacobians(const float *in, uint32_t n, float *out);
/* The same for the MULTIPARTICLE_ACCELERATOR form of the window (AcceleratorHandler.h:17-21: setParticlesCount + start;
 * written and read back at algorithms/fastslam2.cpp:172-286): `records` self-describing records back to back in ONE host
 * buffer, each  n, xv[3], R[4], n x (xf[2], Pf[4]), then n x 16 output floats (zp, Hf, Hv, Sf as above)  = 8 + 22 n floats
 * (the reference writes one record per particle and re-observed landmark, n = 1); the outputs are written IN PLACE, as the
 * accelerator wrote them into its memory block.  window_floats = size of the buffer (a record running beyond it is refused).
 * Synchronous.  (Upstream's caller of this form is unfinished -- it loops over copies of the particles and indexes Sf by the
 * observation -- so only the window itself is the contract here.) */
int slamgpu_jacobians_multi(float *window, uint32_t records, uint64_t window_floats);

/* Known-answer entry point for the scalar device functions the update kernels are built from, in the arithmetic of the
 * chosen build (SLAMGPU_MATH_*): op 0 = trigonometricOffset (core.cpp:460-477), in[n] -> out[n];
 * op 1 = gaussEvaluate D=2 (fastslam2.cpp:127-163), in = n x (v0 v1 S00 S10 S11); op 2 = gaussEvaluate D=3,
 * in = n x (v0 v1 v2 S00 S10 S11 S20 S21 S22); out[n].  Host pointers; synchronous.  Used by the parity tests to put
 * the reference's edge-case vectors (angles at +-pi, +-2pi, |a| > 2pi; near-singular S) through the device code. */
int slamgpu_kat(int32_t math_mode, int32_t op, const float *in, int32_t n, float *out);

/* ---- Seam 2: the algorithm object ------------------------------------------------------------------ */
int slamgpu_create(const slamgpu_config *cfg, slamgpu_ctx **out);
void slamgpu_destroy(slamgpu_ctx *ctx);

/* FastSLAM{1,2}::predict(particles, xTrue, V, G, Q, dt) (fastslam2.cpp:51-60 / fastslam1.cpp:57-66).
 * phi_true = xTrue(2), used only when use_heading.  noise2: TAPE mode + add_predict_noise: 2*N host
 * floats, the two normals multivariateGauss((V,G),Q) consumed per particle, particle-major; else NULL.
 * Consecutive predicts may be coalesced into one launch; any other call flushes them. */
int slamgpu_predict(slamgpu_ctx *ctx, float V, float G, const float Q[4], float dt, float phi_true,
                    const float *noise2);

/* FastSLAM{1,2}::update(particles, zf, zn, idf, z, table, R) (fastslam2.cpp:21-48 / fastslam1.cpp:18-35)
 * including resampleParticles (core.cpp:718-749).
 * zf[2m], idf[m]: re-observed landmarks; zn[2n]: new landmarks (appended at index Nf..Nf+n-1).
 * TAPE mode: normals = 3*N host floats (particle-major; the randn(3,1) of multivariateGauss, needed
 * when method==FASTSLAM2 and m+n>0), strata = N_global host floats (the dithered strata of
 * stratifiedRandom, core.cpp:751-769).  PHILOX mode: both NULL. */
int slamgpu_update(slamgpu_ctx *ctx, const float *zf, const int32_t *idf, int32_t m, const float *zn, int32_t n,
                   const float R[4], const float *normals, const float *strata);

/* One whole filter step in one call: the body of the reference's per-observation loop iteration
 * (FastSLAM2Wrapper.cpp / ParticleSLAMWrapper.cpp: k x predict at the control rate, then update, then
 * computeEstimatedPosition): n_controls x slamgpu_predict(V, G, Q, dt, phi_true) with controls = [n_controls][3]
 * host floats (V, G, phi_true), then slamgpu_update(zf, idf, m, zn, n, R, normals, strata), then, if
 * record_estimate != 0, slamgpu_estimate_async.  Exactly equivalent to making those calls one by one (same launches,
 * same results); it only saves the per-call overhead of a foreign-function boundary.  Not available with
 * TAPE-mode predict noise (per-particle noise2 buffers): use slamgpu_predict for that. */
int slamgpu_step(slamgpu_ctx *ctx, const float *controls, int32_t n_controls, const float Q[4], float dt,
                 const float *zf, const int32_t *idf, int32_t m, const float *zn, int32_t n, const float R[4],
                 const float *normals, const float *strata, int32_t record_estimate);

/* computeEstimatedPosition (ParticleSLAMWrapper.cpp:56-77): mean x, mean y, heading of the first
 * particle with the strictly greatest weight.  Synchronises. */
int slamgpu_estimate(slamgpu_ctx *ctx, double xyt[3]);

/* Same estimate, asynchronous: enqueue the reduction into a device-side history (capacity 4096 entries) and read
 * the whole history later (one synchronisation for many steps).  xyt holds 3 doubles per entry. */
int slamgpu_estimate_async(slamgpu_ctx *ctx);
int slamgpu_estimate_fetch(slamgpu_ctx *ctx, double *xyt, int32_t max_count, int32_t *count);
/* The same history with the per-step resampling record: neff[i] = Neff of the update that produced entry i and
 * resampled[i] = whether it resampled (core.cpp:781-788, :731).  Entries recorded after predicts only (no update
 * since the previous entry) repeat the last update's values.  Any output pointer may be NULL.  Reading the record
 * here, once per batch of steps, keeps the step loop free of host round trips (slamgpu_stats synchronises). */
/* status[i] = SLAMGPU_STATUS_* bits of that update (0 = healthy).  A fetch that takes fewer entries than were recorded
 * (max_count < recorded) keeps the rest for the next fetch. */
int slamgpu_history_fetch(slamgpu_ctx *ctx, double *xyt, float *neff, int32_t *resampled, int32_t *status, int32_t max_count,
                          int32_t *count);


/* Outcome of the last update: Neff, whether the resample fired, sum of the raw weights. Synchronises. */
int slamgpu_stats(slamgpu_ctx *ctx, float *neff, int32_t *resampled, double *weight_sum);
/* SLAMGPU_STATUS_* bits of the last update's resampling stage. Synchronises. */
int slamgpu_step_status(slamgpu_ctx *ctx, int32_t *status);
/* Ancestor indices of the last resample (keep[], core.cpp:800-806), N_local int32. Synchronises. */
int slamgpu_ancestors(slamgpu_ctx *ctx, int32_t *keep);

/* ---- observation front end on the device ------------------------------------------------
 * getObservations (core.cpp:185-273: visibility scan + range / bearing), addObservationNoise (:438-449) and
 * dataAssociationKnown (:91-120) as one kernel over a device-resident landmark map and association table.
 * slamgpu_set_map uploads the map (2 x nlm, row-major: xs then ys) and resets the table.  slamgpu_observe: xtrue = the
 * true vehicle pose; noise: 0 none, 1 tape (r1 / r2 = host arrays of at least nlm normals, consumed one per VISIBLE
 * landmark in visibility order, as the reference's two randn(1, len) calls are), 2 Philox(seed, observation step, landmark
 * id).  Outputs (host, any may be NULL): z[2 nz] + vis[nz] = the raw observation, and its split zf[2 m], idf[m], zn[2 n]
 * against the landmarks the context already holds (new ones are numbered Nf, Nf + 1, ... in visibility order).  The
 * caller hands zf / idf / zn to slamgpu_update (the genealogy bookkeeping of the update is host-side, so the ids have to
 * come back anyway); the kernel exists for maps where the visibility scan is the host's bottleneck (10^4 landmarks).
 * Synchronises. */
int slamgpu_set_map(slamgpu_ctx *ctx, const float *lm, int32_t nlm);
int slamgpu_observe(slamgpu_ctx *ctx, const float xtrue[3], float max_range, const float R[4], int32_t noise, const float *r1,
                    const float *r2, float *z, int32_t *vis, int32_t *nz, float *zf, int32_t *idf, int32_t *m, float *zn, int32_t *n);

/* One iteration of the wrapper's loop with the observation made ON THE DEVICE (fastslam2wrapper.cpp:64-88): the
 * n_controls control steps since the last observation (as slamgpu_step), then -- for the true vehicle pose xtrue --
 * getObservations + addObservationNoise + dataAssociationKnown (core.cpp:185-273, 438-449, 91-120) by one kernel that
 * leaves the observation packet (idf, zf, zn) AND the bookkeeping of the landmark genealogy in device memory, then the update
 * launch, which reads them there.  Per step the host sends the controls and the pose (<= 100 bytes) and learns nothing about
 * the observation: no visibility scan over the map on the host, no packet over PCIe (16 KB per step on the 10 000-landmark
 * map).  Needs slamgpu_set_map and known association; landmark capacities above 39 also SLAMGPU_FLAG_DEVICE_OBSERVE at
 * creation.  Maps of up to 39 landmarks: the observation is made inside the update launch itself (every block works the
 * packet out from the state the previous launch left: no front-end kernel, no second stream); bigger maps: a front-end
 * kernel on a stream of its own, a step ahead of the update launches.
 * noise: 0 none; 1 the caller's normals r1[k], r2[k] for the k-th visible landmark (parity with the reference's tape):
 * BOTH ARRAYS MUST HOLD AT LEAST nlm FLOATS (the map size given to slamgpu_set_map) -- the whole arrays are copied to the
 * device before the number of visible landmarks is known there; entries past the visible count are ignored;
 * 2 Philox(seed; landmark, step) on the device.  normals / strata: particle noise of TAPE-mode contexts, as slamgpu_update.
 * Landmarks beyond the context's capacity are dropped and reported by slamgpu_observe_fetch / slamgpu_num_landmarks
 * (SLAMGPU_ERR_CAPACITY).  Do not mix with slamgpu_observe / host-made observations of the same run: the association table
 * lives on the device (contexts of up to 39 landmarks refuse the mix: SLAMGPU_ERR_INVALID). */
int slamgpu_step_observe(slamgpu_ctx *ctx, const float *controls, int32_t n_controls, const float Q[4], float dt, const float xtrue[3],
                         float max_range, const float R[4], int32_t noise, const float *r1, const float *r2, const float *normals,
                         const float *strata, int32_t record_estimate);
/* K iterations of the wrapper's loop in ONE call (FastSLAM2Wrapper::run's loop body K times, fastslam2wrapper.cpp:51-117), the
 * observations made on the device: iteration k applies n_controls[k] control steps -- rows (V, G, phi_true) of `controls`, the
 * iterations' rows one after the other -- and observes from the true pose xtrue[3 k .. 3 k + 2]; the pose estimate of every
 * iteration is recorded (slamgpu_estimate_fetch / slamgpu_history_fetch: their capacity, 4 096 iterations, bounds K between two
 * fetches).  noise: 0 none or 2 Philox on the device (a caller's tape is per iteration: use slamgpu_step_observe).  Returns
 * without waiting for the GPU; the results are bit-identical to K calls of slamgpu_step_observe with record_estimate = 1.
 * Contexts of at most 39 landmarks and at most 2 048 particles (K >= 2, at most 16 controls per iteration) run the K
 * iterations as ONE launch: a persistent step loop whose workgroups sit on one XCD and meet at a counter in its L2 between two
 * iterations instead of at a kernel boundary (round 5; every wait is bounded, in time: a workgroup that waits 2 s -- another
 * process holding the CUs -- abandons the launch, and whatever synchronises next -- slamgpu_sync, _history_fetch, _estimate,
 * _stats, the next slamgpu_run_observe -- returns SLAMGPU_ERR_BARRIER, sticky; slamgpu_last_error and slamgpu_persist_status
 * say which launch it was and how many of its iterations had completed; the state is undefined from there: recreate the
 * context); everything else enqueues K update launches.  SLAMGPU_NO_PERSIST=1 in the environment selects the K launches
 * everywhere (diagnostic).
 * Arguments are validated before anything is enqueued: a call that is refused (bad pointers or counts, no map, a history that
 * K more estimates would overflow: SLAMGPU_ERR_CAPACITY) applies NO iteration.  Should an iteration fail later all the same
 * ON THE HOST SIDE (a HIP error while iteration k is being prepared), the iterations before it stay applied and
 * slamgpu_last_error names it; a launch abandoned on the DEVICE is the case above. */
int slamgpu_run_observe(slamgpu_ctx *ctx, int32_t K, const int32_t *n_controls, const float *controls, const float Q[4], float dt,
                        const float *xtrue, float max_range, const float R[4], int32_t noise);
/* The observation packet of the last slamgpu_step_observe, copied back for logging / tests (any pointer may be NULL;
 * arrays sized for the map): raw observations z[2 nz] and visible landmark ids vis[nz], re-observed zf[2 m] / idf[m],
 * new zn[2 n].  Synchronises. */
int slamgpu_observe_fetch(slamgpu_ctx *ctx, float *z, int32_t *vis, int32_t *nz, float *zf, int32_t *idf, int32_t *m, float *zn, int32_t *n);

/* Per-particle gated nearest-neighbour data association (the reference has it for EKF-SLAM only:
 * EKFSLAM::dataAssociate, algorithms/ekfslam.cpp:151-189; applied here to every particle with its own landmark
 * estimates: P = blockdiag(0, Pf_j), so S = Hf Pf Hf^T + R).  z[2*nz] = (range, bearing) of the nz observations.
 * labels[N*nz] (host, particle-major; may be NULL): landmark index >= 0, SLAMGPU_ASSOC_NEW or SLAMGPU_ASSOC_DISCARD.
 * consensus[nz] / support[nz] (may be NULL): the label carrying the largest total particle weight per observation and
 * that weight share -- what a caller feeds to slamgpu_update, whose association is per step, not per particle (two
 * observations claiming one landmark: the weaker one is discarded).  gate_reject / gate_augment = GATE_REJECT /
 * GATE_AUGMENT of the .ini.  Cost is O(N * nz * Nf): for maps with tens of landmarks.  Synchronises. */
enum { SLAMGPU_ASSOC_NEW = -1, SLAMGPU_ASSOC_DISCARD = -2 };
int slamgpu_associate(slamgpu_ctx *ctx, const float *z, int32_t nz, const float R[4], float gate_reject, float gate_augment,
                      int32_t *labels, int32_t *consensus, float *support);
/* The same with a choice of method and its cost.  mode SLAMGPU_ASSOC_AUTO: the spatial prefilter for maps of 64 landmarks or
 * more (single contexts), the exhaustive scan otherwise; _EXHAUSTIVE: every particle gates every observation against every
 * landmark, O(N nz Nf); _GRID: the landmarks are binned into a uniform grid by the bounding boxes of their estimates over all
 * particles, grown by a radius beyond which no particle's estimate can pass either gate; every (particle, observation) pair
 * then evaluates the landmarks of one cell only, O(N nz k) -- the labels are the exhaustive scan's, decision for decision.
 * (Round 6: a call with at most 4 096 observations builds one candidate LIST per observation instead of the grid -- the landmarks
 * whose box can pass a gate for some particle given THAT observation's range and the arc of the set's headings: ~3 entries where a
 * grid cell held ~33 on the 10 000-landmark map -- and walks it first with the bound of gate_reject alone; same labels.) 
 * stats (may be NULL): [0] (particle, observation, landmark) triples evaluated, [1] grid entries, [2] device milliseconds of
 * the association kernels, [3] 1 if the grid was used. */
enum { SLAMGPU_ASSOC_AUTO = 0, SLAMGPU_ASSOC_EXHAUSTIVE = 1, SLAMGPU_ASSOC_GRID = 2 };
int slamgpu_associate_ex(slamgpu_ctx *ctx, const float *z, int32_t nz, const float R[4], float gate_reject, float gate_augment, int32_t mode,
                         int32_t *labels, int32_t *consensus, float *support, double stats[4]);

/* (round 6) The per-particle association CARRIED INTO THE UPDATE: every particle acts on its own decisions, on a map of its own
 * (what the reference's data structure allows -- Particle.cpp:61-73 grows landmarkXs / landmarkPs per particle -- and what
 * SURVEY.md section 8(f4) asks for: EKFSLAM::dataAssociate, ekfslam.cpp:151-189, applied per particle; the reference has no
 * FastSLAM implementation of it, so parity is with the EKF's gating decisions and, for particles that agree, with slamgpu_update
 * bit for bit).  All particles share one index space of landmark SLOTS; a particle that has not opened the landmark of a slot holds
 * an ABSENT record there (slamgpu_download returns NaN for its xf).  One step:
 *   - every particle gates the nz observations against its own landmarks (as slamgpu_associate_ex does, with opt->mode);
 *   - a slot any particle matched is rewritten by every particle: updated with the observation THAT particle matched it with
 *     (one observation per landmark and particle: the first to claim it), or carried forward unchanged;
 *   - an observation that at least opt->new_share of the particles call new gets a slot (a dead slot first, otherwise the map
 *     grows; none left: the observation is dropped): those particles initialise it (core.cpp:479-509), the others hold it absent;
 *   - an observation a particle leaves unexplained (discarded between the gates, a second claim on one landmark, new) costs that
 *     particle the weight factor opt->p_new -- FastSLAM's constant likelihood of a new feature -- so that ignoring an
 *     observation never outweighs explaining it; a particle none of the observations concerns keeps its pose and Pv untouched;
 *   - every opt->census_every steps the particles holding each PARTIAL slot are counted (a slot every particle opened is held by
 *     every descendant for good and needs no counting): a slot nobody holds any more (its hypotheses died in a resample) is dead
 *     -- out of the association, reused by a later landmark.
 * At most 32 767 observations a step.
 * Resampling, estimates, history and downloads are the usual ones.  Single contexts on plain genealogy rows only: create the
 * context with SLAMGPU_FLAG_PARTICLE_MAPS (capacities of 40..256 landmarks are moved to plain rows at the first call).
 * normals / strata: as slamgpu_update.  report (may be NULL): [0] slots rewritten, [1] slots opened, [2] of them dead slots
 * reused, [3] observations dropped for want of a slot, [4] slots in use after the step (slamgpu_num_landmarks), [5] dead slots
 * waiting, [6] particles an observation needs to open a slot, [7] 1 if the holders were counted.  Synchronises. */
typedef struct slamgpu_particle_assoc {
    float gate_reject, gate_augment; /* GATE_REJECT / GATE_AUGMENT of the .ini */
    int32_t mode;                    /* SLAMGPU_ASSOC_AUTO / _EXHAUSTIVE / _GRID */
    float new_share;                 /* 0: one particle is enough to open a landmark */
    float p_new;                     /* > 0 */
    int32_t census_every;            /* 0: never */
    /* The EXCLUSION rule (excl_base + excl_per_m = 0: off, the gates alone decide).  The gates measure an observation against
     * S = Hf Pf Hf^T + R and know nothing of the particle's own pose error (the EKF's S carries it, ekfslam.cpp:160-176; a
     * particle's pose is a point), so a particle a metre off calls an observation of a mapped landmark new and opens a duplicate
     * beside it.  With the rule on, an observation no landmark gates is placed in the world from the particle's pose; if a
     * landmark the particle holds lies within excl_base + excl_per_m * range [m] of that point the observation cannot be new: it
     * is matched with that landmark when no other lies within unique_ratio times the distance, and discarded otherwise.
     * SLAMGPU_ASSOC_EXHAUSTIVE / _AUTO (which then scans exhaustively) only: O(N nz Nf), refused with SLAMGPU_ERR_CAPACITY beyond
     * 4e10 gate evaluations in a step (maps of a few hundred landmarks are its range). */
    float excl_base, excl_per_m, unique_ratio;
} slamgpu_particle_assoc;
int slamgpu_update_particle(slamgpu_ctx *ctx, const float *z, int32_t nz, const float R[4], const slamgpu_particle_assoc *opt,
                            const float *normals, const float *strata, int32_t report[8]);
/* The same step with the caller's labels[N*nz] (host, particle-major: slot >= 0, SLAMGPU_ASSOC_NEW or SLAMGPU_ASSOC_DISCARD)
 * instead of the gates' (opt->gate_* / mode are not read): what the tests drive, and the seam for an association made elsewhere. */
int slamgpu_update_labels(slamgpu_ctx *ctx, const float *z, int32_t nz, const float R[4], const int32_t *labels,
                          const slamgpu_particle_assoc *opt, const float *normals, const float *strata, int32_t report[8]);

/* Retire landmarks from the gated association (round 6): landmarks ids[0 .. count) take no part in slamgpu_associate /
 * _associate_ex from now on -- no particle gates an observation against them, nothing votes for them -- and, never being
 * re-observed, they are never written again.  They stay in the particles' maps (slamgpu_num_landmarks, slamgpu_download and the
 * landmark capacity count them): what the reference's Particle would allow -- dropping an entry of landmarkXs / landmarkPs,
 * Particle.cpp:61-73 -- is a renumbering of every later landmark, which a per-step association shared by all particles cannot do
 * in the middle of a run (slamgpu_upload, which replaces the whole particle set, clears the marks).  For a caller whose policy has given a landmark up (a duplicate opened by a wrong vote:
 * slam-backend -assoc gated, host/gated.h).  Single contexts.  Synchronises. */
int slamgpu_retire_landmarks(slamgpu_ctx *ctx, const int32_t *ids, int32_t count);
int slamgpu_num_landmarks(slamgpu_ctx *ctx);

/* Introspection: genealogy rows in use (what a resample composes per particle: 4 bytes each) and the context's row capacity.
 * Every update that writes landmarks opens a row; rows whose landmarks have all moved on are reused; updates consolidate the
 * landmarks of stale rows so that the rows in use stay bounded (a handful on maps of up to 39 landmarks, ~2 000 on big maps).
 * No counterpart in the reference (its resample copies whole particles, core.cpp:735-749).  Synchronises after
 * slamgpu_step_observe steps. */
int slamgpu_genealogy_rows(slamgpu_ctx *ctx, int32_t *in_use, int32_t *capacity);
/* How slamgpu_run_observe has run on this context so far: launches of the persistent step loop and the iterations they carried
 * (0 / 0: every iteration was a launch of its own); cross_xcd (may be NULL; synchronises): 1 if the last such launch found its
 * workgroups on more than one XCD and took the memory model's agent-scope release / acquire between iterations. */
int slamgpu_persist_info(slamgpu_ctx *ctx, int64_t *launches, int64_t *iterations, int32_t *cross_xcd);
/* How far an ABANDONED launch of the persistent step loop got (any pointer may be NULL; does not synchronise: call it after the
 * call that returned SLAMGPU_ERR_BARRIER).  abandoned: 1 once a launch has been abandoned (sticky), else 0 and the other outputs
 * are 0; launch: the number of that launch in this context (1 = the first slamgpu_run_observe call that took the loop; the
 * launches queued behind it did nothing); completed: iterations of THAT launch which every workgroup had completed (all its
 * barriers passed); handed: the iterations it had been handed.  The iteration
 * `completed` was in flight and is PARTIALLY applied (poses, records and weights of some tiles only), and the host's bookkeeping
 * (landmark book, history slots, RNG step) stands at the end of everything handed over: the context's state is undefined.
 * Recreate it and replay; the counts say from where.  The reference's own wait for its accelerator is an unbounded spin
 * (core.cpp:619-622): there the process hangs instead. */
int slamgpu_persist_status(slamgpu_ctx *ctx, int32_t *abandoned, int64_t *launch, int32_t *completed, int32_t *handed);
/* Particle-major host copies (any pointer may be NULL): xv[3N], Pv[9N] row-major, w[N],
 * xf[2*Nf*N], Pf[4*Nf*N] row-major — the layout of vector<Particle> flattened. Synchronises. */
int slamgpu_download(slamgpu_ctx *ctx, float *xv, float *Pv9, float *w, float *xf, float *Pf4);
/* The same for particles [first, first + count) only: what a plot sink with decimation, or a check of a context too
 * large to copy whole (config 5: 24 GB of landmark records), needs. */
int slamgpu_download_range(slamgpu_ctx *ctx, int32_t first, int32_t count, float *xv, float *Pv9, float *w, float *xf,
                           float *Pf4);
/* Read-only view of `count` particles first, first + stride, first + 2 stride, ... in the layout of slamgpu_download
 * (xv[3 count], Pv[9 count], w[count], xf[2 Nf count], Pf[4 Nf count]; any pointer may be NULL; landmarks = 0 skips the
 * records).  Unlike slamgpu_download it rewrites nothing: the pose is read through a pending resampling gather, the
 * landmark records through the genealogy, by one kernel, and the state the next step works on is bit for bit what it
 * would have been without the call.  What drawParticles / drawFeatureParticles need each iteration
 * (ParticleSLAMWrapper.cpp:34-54) with a decimation stride, in one launch and five copies.  Single contexts only.
 * Synchronises. */
int slamgpu_peek(slamgpu_ctx *ctx, int32_t first, int32_t stride, int32_t count, float *xv, float *Pv9, float *w, float *xf,
                 float *Pf4);
int slamgpu_upload(slamgpu_ctx *ctx, int32_t nf, const float *xv, const float *Pv9, const float *w, const float *xf,
                   const float *Pf4);
int slamgpu_sync(slamgpu_ctx *ctx);

/* ---- distributed operation (the multi-GPU path: nothing migrates) ---------------------------------------------------
 * One context per GPU; shard g holds the contiguous global particles [g*n, (g+1)*n), n a multiple of 256; contexts may
 * live in one process (peer access) or one process per GPU (hipIpc).  Every context maps every other shard's state arrays
 * (slamgpu_dist_export / slamgpu_dist_connect); from then on a filter step is
 *
 *     slamgpu_dist_step   ONE kernel launch per shard: the queued predicts, the resampling stage of the PREVIOUS step
 *                         (every shard scans the same all-gathered block totals: identical Neff, decision and ancestors
 *                         everywhere, independent of the number of shards) and the per-particle update.  A particle
 *                         whose ancestor lives on another GPU reads that ancestor's pose and genealogy in place over xGMI;
 *                         genealogy entries are global slot ids, so landmark records stay on the GPU that wrote them until
 *                         the landmark is observed again.  No pack / send / unpack, no host synchronisation.
 *     ALL-GATHER          of this step's block totals (slamgpu_dist_totals: 8 B per 256 particles per shard), stream-ordered
 *                         on the context's stream: by the library itself once slamgpu_dist_comm_init has run (RCCL), else
 *                         by the caller (torch.distributed, MPI, device copies for contexts of one process).  It is also
 *                         the only barrier the scheme needs: a shard's next launch starts after every shard's current
 *                         launch has finished.
 *
 * Together the two are resampleParticles (core.cpp:718-824) and the predict / update calls of the wrapper loop
 * (fastslam2wrapper.cpp:64,88) for a particle set that spans GPUs; results do not depend on the number of shards.
 * The pose estimate of a step (ParticleSLAMWrapper.cpp:56-77) is this shard's raw partial (slamgpu_dist_history_fetch: sum x,
 * sum y, heading and weight of the local maximum); combine across shards in shard order.  slamgpu_dist_settle (collective:
 * every shard, followed by the all-gather) applies the pending resampling stage so that slamgpu_download* can read the set.
 * Linear weights and SLAMGPU_RNG_PHILOX only. */
int slamgpu_dist_export_size(void);
int slamgpu_dist_export(slamgpu_ctx *ctx, void *blob);
int slamgpu_dist_connect(slamgpu_ctx *ctx, int32_t n_shards, int32_t shard, const void *blobs);
int slamgpu_dist_step(slamgpu_ctx *ctx, const float *controls, int32_t n_controls, const float Q[4], float dt, const float *zf,
                      const int32_t *idf, int32_t m, const float *zn, int32_t n, const float R[4], int32_t record_estimate);
/* buffers of the all-gather that follows the last slamgpu_dist_step / _settle: this shard's totals (floats_per_shard
 * floats) go to slot `shard` of every context's `gathered` (n_shards * floats_per_shard floats) */
int slamgpu_dist_totals(slamgpu_ctx *ctx, const float **local_dev, float **gathered_dev, int32_t *floats_per_shard);
int slamgpu_dist_settle(slamgpu_ctx *ctx);
/* the recorded steps of this shard (settled): raw partial of the estimate (sum x, sum y, heading and weight of the local
 * maximum-weight particle; combine in shard order, strict > on the weight) and the stage's Neff / resampled / status,
 * which are identical on every shard */
int slamgpu_dist_history_fetch(slamgpu_ctx *ctx, double *raw4, float *neff, int32_t *resampled, int32_t *status, int32_t max_count,
                               int32_t *count);

/* RCCL inside the library (bound at run time: dlopen librccl.so.1): one process per GPU.  Rank 0 makes an id
 * (SLAMGPU_DIST_COMM_ID_BYTES bytes) and hands it to every rank by whatever means the launcher has; after
 * slamgpu_dist_comm_init (collective), slamgpu_dist_step and slamgpu_dist_settle enqueue the all-gather themselves on
 * the context's stream: a filter step is one C call, one launch and one collective, and never waits on the host. */
#define SLAMGPU_DIST_COMM_ID_BYTES 128
int slamgpu_dist_comm_id(void *id, int32_t bytes);
int slamgpu_dist_comm_init(slamgpu_ctx *ctx, const void *id, int32_t n_ranks, int32_t rank);
/* the all-gather of the last step's totals once more (collective, idempotent): lets a harness time the collective alone */
int slamgpu_dist_gather(slamgpu_ctx *ctx);
/* what the communicator inside the library says about itself (ncclCommCount / ncclCommUserRank): a harness that claims an
 * N-GPU run checks n_ranks == N here instead of trusting its launcher's environment */
int slamgpu_dist_comm_info(slamgpu_ctx *ctx, int32_t *n_ranks, int32_t *rank);
/* particles of this shard whose ancestor at a resample lived on ANOTHER shard: their pose and genealogy were read in place out
 * of that GPU's memory (over xGMI between physical GPUs).  The counter costs the update launches an atomic per wave and resample,
 * so it is kept only FROM THE FIRST CALL of this function on (round 5): call it once before the steps of interest and take
 * differences.  Does not run outstanding stages; synchronises the stream. */
int slamgpu_dist_remote_reads(slamgpu_ctx *ctx, uint64_t *particles);


/* All shards in ONE process (the reference's single backend process driving k GPUs; or k logical shards on one GPU, which
 * must then share one stream): export + connect + the collective (shards on devices of their own: SLAMGPU_DIST_PUSH if its
 * barrier works, else ncclCommInitAll; a copy kernel on a shared device) in one call, then steps for every shard at once.  Contexts are created by the caller as for slamgpu_dist_connect and
 * destroyed by the caller after the group.  _history combines the shards' partial estimates (xyt[3] per recorded step);
 * _download concatenates the shards in shard order (buffers sized for all k * n particles). */
typedef struct slamgpu_dist_group slamgpu_dist_group;
int slamgpu_dist_group_create(slamgpu_ctx **ctxs, int32_t k, slamgpu_dist_group **out);
void slamgpu_dist_group_destroy(slamgpu_dist_group *g);
int slamgpu_dist_group_step(slamgpu_dist_group *g, const float *controls, int32_t n_controls, const float Q[4], float dt,
                            const float *zf, const int32_t *idf, int32_t m, const float *zn, int32_t n, const float R[4],
                            int32_t record_estimate);
int slamgpu_dist_group_settle(slamgpu_dist_group *g);
int slamgpu_dist_group_history(slamgpu_dist_group *g, double *xyt, float *neff, int32_t *resampled, int32_t *status,
                               int32_t max_count, int32_t *count);
int slamgpu_dist_group_download(slamgpu_dist_group *g, float *xv, float *Pv9, float *w, float *xf, float *Pf4);


/* ---- measurement / plumbing ------------------------------------------------------------------------ */
/* HIP stream the context launches on (hipStream_t as void*), so a harness can record events on it. */
void *slamgpu_stream(slamgpu_ctx *ctx);
/* Device time of a region of the context's stream: two HIP events recorded on that stream (start now / stop now);
 * stop synchronises on its event and returns the milliseconds between the two.  Unlike the per-launch event pairs of
 * slamgpu_profile this perturbs nothing inside the region. */
int slamgpu_timer_start(slamgpu_ctx *ctx);
int slamgpu_timer_stop(slamgpu_ctx *ctx, double *ms);

/* Device-time accounting: when enabled every kernel launch is bracketed by HIP events on the context
 * stream; slamgpu_kernel_time returns accumulated milliseconds and launch count for a kernel name
 * ("fs2_update", "weights_scan", "resample", "predict", "estimate", ...). */
int slamgpu_profile(slamgpu_ctx *ctx, int32_t enable);
int slamgpu_kernel_time(slamgpu_ctx *ctx, const char *kernel, double *ms, int64_t *launches);
/* Algorithmic bytes moved by the update path so far (SURVEY.md §8(d) formula, accumulated per step). */
int slamgpu_algorithmic_bytes(slamgpu_ctx *ctx, double *update_bytes, double *predict_bytes);

/* =================================================================================================================
 * EXPERIMENTAL / DIAGNOSTIC entry points.  Exported by the library, NOT part of the stable ABI: they may change or go
 * away without a bump of SLAMGPU_ABI_VERSION.  Declared only when SLAMGPU_EXPERIMENTAL is defined before this header is
 * included.  What is here and why:
 *   - slamgpu_shard_*           round 1's exchange path (pack / all-to-all / unpack of migrating offspring, the caller's
 *                               collectives).  Superseded by the distributed contexts above (nothing migrates); kept as the
 *                               fallback of bench.py --mgpu exchange when peer mappings cannot be set up.  EXPERIMENTAL: has
 *                               never run across physical GPUs (logical shards on one GPU and gloo ranks on the CPU only).
 *   - push / fold collectives   hand-made alternatives to the RCCL all-gather of slamgpu_dist_step.  EXPERIMENTAL: have never
 *                               run across physical GPUs (between contexts of ONE GPU only).  Default everywhere:
 *                               SLAMGPU_DIST_GATHER.  Which of the three generations stays is a question for the first
 *                               measured multi-GPU run (no 8-GPU node has been available to this build in six rounds).
 *   - slamgpu_dev_*             raw device buffers for the callers of the exchange path
 *   - slamgpu_debug_stamps      instrumented build only
 * ================================================================================================================= */
#ifdef SLAMGPU_EXPERIMENTAL

/* ---- sharded operation: particles partitioned over contexts (one per GPU) ---------------------------
 * The reference has no multi-device path; this is the build's data-parallel extension of
 * resampleParticles (core.cpp:718-824).  Shard g holds the contiguous global particles
 * [g*n, (g+1)*n), n a multiple of 256.  The collectives belong to the caller (RCCL through
 * torch.distributed in bench.py; any transport works, the buffers are plain device pointers):
 *
 *   1. slamgpu_shard_update        per-particle update only (the update half of slamgpu_update)
 *   2. slamgpu_shard_block_totals  -> device pointer of this shard's per-256-particle totals, one contiguous
 *                                  block [w(nb) | w^2(nb)]; ALL-GATHER it (8 B per 256 particles) so that every
 *                                  shard holds gtot = [shard 0: w|w2][shard 1: w|w2]...
 *   3. slamgpu_shard_plan          every shard runs the same scan of the gathered totals => identical
 *                                  sum w, Neff, decision and offspring boundaries K[0..G] on every shard
 *                                  (results are independent of the number of shards)
 *   4a. no resample: slamgpu_shard_finish normalises the weights
 *   4b. resample   : slamgpu_shard_pack gathers this shard's offspring K[g]..K[g+1] into per-destination
 *                    blocks; ALL-TO-ALL (send_counts/recv_counts in records of slamgpu_shard_record_floats
 *                    floats); slamgpu_shard_unpack scatters what arrived; slamgpu_shard_finish commits
 *   5. slamgpu_shard_estimate      local sum x, sum y, heading and weight of the local max-weight particle;
 *                                  combine across shards in shard order (sum; strict > for the maximum)
 */
typedef struct {
    double wsum, wsq;
    float neff;
    int32_t resampled;
    int64_t K[65]; /* K[r] = first global output particle whose ancestor lives on shard r; K[G] = N */
    int32_t status, pad; /* SLAMGPU_STATUS_* */
} slamgpu_shard_plan_t;

int slamgpu_shard_update(slamgpu_ctx *ctx, const float *zf, const int32_t *idf, int32_t m, const float *zn, int32_t n,
                         const float R[4], const float *normals, const float *strata);
/* slamgpu_step for a shard: n_controls x slamgpu_predict, then slamgpu_shard_update (the per-particle half of the
 * update); the caller continues with the slamgpu_shard_* resampling stage. */
int slamgpu_shard_step(slamgpu_ctx *ctx, const float *controls, int32_t n_controls, const float Q[4], float dt,
                       const float *zf, const int32_t *idf, int32_t m, const float *zn, int32_t n, const float R[4],
                       const float *normals, const float *strata);

/* Optional: make the update kernel write this shard's [w(nb) | w2(nb)] block totals straight into a caller-owned device
 * buffer (2*nb floats, e.g. the input tensor of the all-gather) instead of the context's own; NULL restores the default.
 * Saves the per-step device copy between slamgpu_shard_block_totals and the collective. */
int slamgpu_shard_set_totals_buffer(slamgpu_ctx *ctx, float *totals_dev);
int slamgpu_shard_block_totals(slamgpu_ctx *ctx, const float **totals_dev, int32_t *nblocks);
int slamgpu_shard_plan(slamgpu_ctx *ctx, const float *gtot_dev, int32_t nb_global, int32_t n_shards, slamgpu_shard_plan_t *out);
int slamgpu_shard_record_floats(slamgpu_ctx *ctx);
int slamgpu_shard_pack(slamgpu_ctx *ctx, const float *gtot_dev, int32_t nb_global, int32_t n_shards, int32_t shard,
                       const slamgpu_shard_plan_t *plan, float *send_dev, int64_t *send_counts, int64_t *recv_counts);
int slamgpu_shard_unpack(slamgpu_ctx *ctx, const float *recv_dev, int32_t n_shards, int32_t shard,
                         const slamgpu_shard_plan_t *plan);
int slamgpu_shard_finish(slamgpu_ctx *ctx, const slamgpu_shard_plan_t *plan);
int slamgpu_shard_estimate(slamgpu_ctx *ctx, double out[4]);
/* asynchronous form: the 4 raw doubles (sum x, sum y, heading, max w) of each call are kept in the device-side
 * history and fetched together (one synchronisation + one all-gather for many steps) */
int slamgpu_shard_estimate_async(slamgpu_ctx *ctx);
int slamgpu_shard_estimate_fetch(slamgpu_ctx *ctx, double *raw4, int32_t max_count, int32_t *count);

/* The collective between two launches, two ways:
 *   SLAMGPU_DIST_GATHER (default)  an all-gather of the block totals after every launch (RCCL inside the library after
 *                                  slamgpu_dist_comm_init, else the caller's);
 *   SLAMGPU_DIST_PUSH              the update launch stores its block totals straight into every shard's table (peer
 *                                  mappings), and a one-wave kernel is the barrier: it stores this step's sequence number
 *                                  into every peer's flag word (system-scope release) and polls its own flag words
 *                                  (fine-grained memory, bounded spin).  No collective library on the step's path.
 * Contexts driven from one thread need a stream of their own each for PUSH (a shared stream would queue a shard's flag
 * store behind the wave that waits for it).  slamgpu_dist_handshake_test (collective) runs `iters` barriers alone and
 * reports their device time and whether every peer arrived; slamgpu_dist_collective_status (synchronises) tells
 * afterwards whether any barrier of the run timed out.  Switch modes only between settled steps, on every shard alike. */
#define SLAMGPU_DIST_GATHER 0
#define SLAMGPU_DIST_PUSH 1
/* SLAMGPU_DIST_FOLD: as PUSH, but the barrier rides at the head of the NEXT update launch instead of in a kernel of its own:
 * that launch's helper block announces "my previous launch has completed" to every peer and waits for theirs, every other
 * block waits for the helper's go word before it requests anything.  One launch per step and nothing else; a flag kernel
 * only closes slamgpu_dist_settle.  Same requirements and the same bounded spins as PUSH. */
#define SLAMGPU_DIST_FOLD 2
int slamgpu_dist_set_collective(slamgpu_ctx *ctx, int32_t mode);
int slamgpu_dist_handshake_test(slamgpu_ctx *ctx, int32_t iters, double *usec, int32_t *ok);
int slamgpu_dist_collective_status(slamgpu_ctx *ctx, int32_t *ok);

/* Plain device-memory helpers for callers that have no allocator of their own (tests, the C++ host): buffers
 * for the gathered block totals and the send / receive records.  copy is device-to-device, ordered on the
 * context's stream and synchronised before returning. */
int slamgpu_dev_alloc(slamgpu_ctx *ctx, uint64_t bytes, void **ptr);
int slamgpu_dev_free(slamgpu_ctx *ctx, void *ptr);
int slamgpu_dev_copy(slamgpu_ctx *ctx, void *dst, const void *src, uint64_t bytes);
int slamgpu_dev_copy_async(slamgpu_ctx *ctx, void *dst, const void *src, uint64_t bytes); /* ordered on the stream, no wait */

/* Diagnostic: wall-clock stamps (100 MHz) of the LAST update launch at the levels of its dependent-load chain, 16 per
 * compute block (slot meaning: tools/stamps.py).  Only the instrumented build (slam_amd/libslamgpu_stamps.so, `make
 * stamps`) writes them, and only for contexts created with SLAMGPU_STAMPS=1 in the environment; otherwise an error
 * (or zeros).  Synchronises. */
int slamgpu_debug_stamps(slamgpu_ctx *ctx, uint64_t *out, int32_t max_blocks, int32_t *nblocks);

#endif /* SLAMGPU_EXPERIMENTAL */

#ifdef __cplusplus
}
#endif
#endif
