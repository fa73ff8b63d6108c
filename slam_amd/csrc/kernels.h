// Host-visible launch interface of the gfx950 kernels (kernels.hip is compiled twice: slam_strict / slam_fast).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slamgpu {

// ---- HBM layout ---------------------------------------------------------------------------------------
// Structure-of-arrays, particle index fastest, so that a wave's 64 particles read 256 contiguous bytes
// per component:
//   pose  : float[10][Ncap]   rows 0-2 xv (x, y, theta), rows 3-8 Pv lower triangle (p00,p10,p11,p20,p21,p22),
//                             row 9 weight w
//   lmk   : float[cap_nf][5][Ncap]   per landmark: xf.x, xf.y, Pf p00, p10, p11
// Two copies of each (ping-pong for the resampling gather); Ctrl.cur says which one is live and is only
// ever changed on the device, so no host round trip is needed to know whether a resample fired.
constexpr int kPoseRows = 10;
constexpr int kLmkRows = 5;
constexpr int kWave = 64;
constexpr int kMaxFusedPredict = 16;

struct Ctrl {
    int32_t cur;          // live buffer (0/1)
    int32_t resampled;    // 1 if the last update resampled
    uint32_t done;        // block-arrival counter for the last-block-flips-cur protocol
    int32_t pad;
    float neff;           // Neff of the last update
    float inv_n;          // 1/N_global
    double wsum;          // sum of raw weights (global)
    double wsq;           // sum of squared raw weights (global)
    double est[4];        // sum x, sum y, heading of max-w particle, max w
};

struct Buffers {
    float *pose[2];
    float *lmk[2];
    Ctrl *ctrl;
    int32_t n;        // local particles
    int32_t ncap;     // row stride (>= n, multiple of 64)
    int32_t cap_nf;
};

struct ObsPacket {          // lives in device memory, uploaded once per update
    int32_t m, n, nf, pad;  // re-observed, new, landmarks before this update
    float R[4];
    // followed by: int32 idf[m]; float zf[2m]; float zn[2n]   (offsets computed from m, n)
};

struct RngArgs {
    int32_t mode;            // 0 tape, 1 philox
    uint32_t step;           // observation-step (update) or control-step (predict) counter
    uint32_t k0, k1;         // philox key = seed
    int64_t first_particle;  // global id of local particle 0
    int64_t n_global;
    const float *normals;    // tape: [3][n] device (update) or [2][n] (predict), component-major
    const float *strata;     // tape: [n_global] device
};

struct PredictStep {
    float V, G, phi_true;
    uint32_t step;
};

struct PredictArgs {
    int32_t nsteps;
    int32_t method, use_heading, add_noise;
    float Q[4];
    float dt, wheel_base, sigma_phi;
    PredictStep steps[kMaxFusedPredict];
};

struct WeightScratch {
    float *lcum;      // [ncap]  inclusive in-wave prefix of the raw weights
    float *wave_w;    // [nwaves] wave totals of w
    float *wave_w2;   // [nwaves] wave totals of w^2
    double *wave_off; // [nwaves+1] exclusive prefix of wave totals
    int32_t *keep;    // [ncap] ancestors of the last resample (local index of global ancestor on this shard)
    int32_t nwaves;
};

struct KernelTable {
    void (*update)(hipStream_t, const Buffers &, int method, const ObsPacket *pkt_dev, int m, int n, int nf,
                   const RngArgs &, const WeightScratch &);
    void (*finalize)(hipStream_t, const Buffers &, const WeightScratch &, int do_resample, int n_effective);
    void (*resample)(hipStream_t, const Buffers &, const WeightScratch &, const RngArgs &, int nf);
    void (*predict)(hipStream_t, const Buffers &, const PredictArgs &, const RngArgs &);
    void (*estimate)(hipStream_t, const Buffers &, double *partials, int nblocks);
    void (*jacobians)(hipStream_t, const float *in_dev, uint32_t n, float *out_dev);
    void (*pack)(hipStream_t, const Buffers &, int nf, float *xv, float *Pv9, float *w, float *xf, float *Pf4);
    void (*unpack)(hipStream_t, const Buffers &, int nf, const float *xv, const float *Pv9, const float *w,
                   const float *xf, const float *Pf4);
};

const KernelTable *kernels_strict();
const KernelTable *kernels_fast();

}  // namespace slamgpu
