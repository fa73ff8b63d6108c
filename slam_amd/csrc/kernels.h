// Host-visible launch interface of the gfx950 kernels (kernels.hip is compiled twice: slam_strict / slam_fast).
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slamgpu {

// ---- HBM layout ---------------------------------------------------------------------------------------
// Array-of-16-byte-chunks, particle index fastest: every wave-instruction moves 64 x 16 B = 1 KiB
// contiguous (the widest coalesced access on gfx950), and the resampling gather fetches whole 16-B
// chunks per ancestor instead of 4-B words from many rows.
//   poseA : float4[Ncap]          x, y, theta, w
//   poseB : float4[Ncap]          Pv lower triangle p00, p10, p11, p20
//   poseC : float2[Ncap]          p21, p22                                      (40 B per particle)
//   lmkA  : float4[cap_nf][Ncap]  xf.x, xf.y, Pf p00, p10
//   lmkB  : float [cap_nf][Ncap]  Pf p11                                         (20 B per landmark)
//   gen   : int32 [cap_nf+1][Ncap] landmark GENEALOGY rows: which slot of its row holds particle k's record (4 B per row)
// Landmark records are not moved when the particle set is resampled: only slot indices are.  Every particle re-observes
// the same landmarks in a step (the association is per step, not per particle), so that step writes a fresh record for
// every particle into the landmark row's OTHER buffer at the particle's own slot and flips the row's live flag
// (lmk_live[j], uniform for all particles) -- nothing a sibling still reads is ever overwritten -- and offspring share
// their ancestor's records until the landmark is observed again.
// The slot of particle k's record of landmark j is the composition of the ancestor maps of every resample since j was
// last written -- the same for ALL landmarks last written in the same step.  So the genealogy is kept per EPOCH (= the
// step that last wrote a set of landmarks), not per landmark: row e of `gen` serves every landmark whose last write was
// epoch e; the host knows which row each landmark uses (the association is global) and hands the rows of the
// re-observed landmarks to the kernel with the observation packet.  A step that writes landmarks opens a new row
// (identity: own slot); a resample composes the live rows with the ancestor map (4 B per particle and LIVE ROW, not per
// landmark: ~25 rows on example_webmap, ~1 k rows after 1 000 steps of the 10 000-landmark map); a row is recycled when
// no landmark uses it any more.  Composition of non-decreasing ancestor maps is non-decreasing, so reads through a row
// coalesce like the plain gather did.
// Two copies of everything (ping-pong for the resampling gather).  Which one is live, and whether the live set
// still has to be read through the ancestor list of the last resample, is device-resident state (Ctrl.live /
// Ctrl.pend), so no host round trip is needed to know whether a resample fired.
constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kMaxFusedPredict = 16;
constexpr int kSmallRows = 40;      // contexts with at most this many genealogy rows (landmark capacity <= 39: the bundled maps)
                                    // are COMPACT: their rows are interleaved four to a 16-byte chunk per particle, so that a
                                    // resample composes ceil(rows / 4) chunks per particle with one wide load + store each,
                                    // and their observation packets always travel as kernel arguments
constexpr int kSmallObs = kSmallRows;  // ... so a packet holds up to this many zf / zn
constexpr int kMidLandmarks = 256;  // (round 5) single contexts of up to this many landmarks are compact too: with row consolidation a map
                                    // of a few hundred landmarks lives in a handful of genealogy rows (example_loop902: 117 landmarks,
                                    // <= 6 re-observed per step), and what sizes the compact layout is the ROWS and the packet, not the
                                    // landmarks.  A step that does not fit the packet (more than kSmallObs re-observed or new landmarks)
                                    // or the rows moves the context to plain rows for good (slamgpu.cpp: demote_to_plain)
constexpr int kRowsPerRole = 16;    // genealogy rows composed by one copy role (x 256 particles)
constexpr int kRowLiveBit = 1 << 30;  // packet row[k]: the landmark's live record buffer rides in bit 30 of its genealogy row
constexpr int kRowFreshBit = 1 << 29; // packet row[k]: the landmark was written by the PREVIOUS update and nothing has composed its
                                      // row since (the resample this launch applies excepted): its record sits in the source
                                      // slot itself, no genealogy lookup needed -- the usual case for a landmark in view
constexpr int kRowMask = kRowFreshBit - 1;
constexpr int kPoolBit = (int) 0x80000000;  // genealogy entry: the record lives in the arrival pool (Buffers::poolA/B)
constexpr int kHistStride = 6;      // doubles per pose-estimate history entry: sum x, sum y, heading, max w, Neff, resampled
constexpr int kMaxScanBlocks = 8192;
// single contexts on the compact layout with more tiles than this run update_kernel_wide (kernels.hip)
constexpr int kWideBlocks = 768;
inline bool update_wide_off() {  // (diagnostic: SLAMGPU_NO_WIDE=1 keeps update_kernel at every size)
    static const bool off = getenv("SLAMGPU_NO_WIDE") != nullptr;
    return off;
}
// (round 6: FastSLAM 1 only.  Without the SLP vectoriser update_kernel<2, 0, false> needs 154 registers and holds three waves per
// SIMD by itself; the wide kernel, squeezed under __launch_bounds__(256, 3), then LOSES to it: 10^6 particles 88.0 against 84.3 us per
// step, same box -- and 84.4 for round 5's build, where the SLP vectoriser had update_kernel at 171 registers and the wide kernel won
// 84.8 against 97.0.  tests/test_host_frontend.py holds update_kernel<2, 0, false> to 168 registers, the most that leaves three waves.)
inline bool update_is_wide(int method, int arrivals, bool big, int nblocks) {
    return method == 1 && arrivals == 0 && !big && nblocks > kWideBlocks && !update_wide_off();
}
// status bits of an update's resampling stage (slamgpu.h: SLAMGPU_STATUS_*)
constexpr int kStatusBadPacket = 2;   // the kernel did not find its packet where the kernel-argument layout says (never seen)
constexpr int kStatusCapacity = 4;    // device front end: more new landmarks than the context has room for (the surplus was dropped)
constexpr int kStatusDegenerate = 1;  // sum of the weights zero or not finite: the reference normalises to NaN (core.cpp:726-729)  // block totals scanned inside every resample block (LDS)

// Ctrl.live / Ctrl.pend are double-buffered by a host-side slot number (Buffers::slot): a kernel that changes the
// live buffer writes the NEW state into slot^1 while every block of that launch still reads slot, and the host
// flips its slot afterwards -- no block ever reads a word another block of the same launch writes.
//   pend[slot] = 1: the last update resampled but nothing has been moved yet (lazy gather): particle k of the
//   current set is particle keep[k] of buffer live[slot], with weight 1/N.  The next update kernel gathers while
//   it computes (or gather_kernel materialises the set when something else needs it).
struct Ctrl {
    int32_t live[2];      // live buffer (0/1), per slot
    int32_t pend[2];      // lazy gather pending, per slot
    int32_t resampled;    // 1 if the last update resampled
    float neff;           // Neff of the last update
    float inv_n;          // weight a resampled particle restarts with: 1/N_global (log-weight contexts: log(1/N_global))
    int32_t status;       // kStatus* bits of the last update's resampling stage
    double wsum;          // sum of raw weights (global)
    double wsq;           // sum of squared raw weights (global)
    double wmax;          // log-weight contexts: the largest log-weight M; wsum / wsq are sums of exp(l - M) and its square
    double est[4];        // sum x, sum y, heading of max-w particle, max w
    unsigned long long remote_reads;  // distributed contexts: particles, so far, whose ancestor lived on ANOTHER shard (its pose and
                                      // genealogy were read out of that GPU's memory over xGMI): one atomic per wave and resample
};

// Distributed operation (one context per GPU, particles sharded in contiguous blocks): the state arrays of every shard as
// mapped into THIS process (own allocations, peer-enabled pointers of contexts in the same process, or hipIpc mappings of
// other processes' contexts).  Nothing is ever migrated: a resampled particle reads its ancestor's pose and genealogy
// straight out of the owning GPU's memory over xGMI, genealogy entries are GLOBAL slot ids (shard * n + local slot), and
// landmark records stay where they were written until the landmark is observed again.
struct PeerPtrs {
    float4 *poseA[2];
    float4 *poseB[2];
    float2 *poseC[2];
    float4 *lmkA[2];
    float *lmkB[2];
    int32_t *gen[2];
    float *lcum[2];
    float *gtot[2];       // the shard's table of everybody's block totals: peers push their totals into it (push collective)
    uint32_t *flags;      // the shard's flag words (fine-grained): [h] = last sequence number shard h announced here,
                          // [kMaxShards] error word, [kGoBase + kGoStride * j] go word j of the folded barrier
};
constexpr int kGoWords = 16;    // copies of the go word (block b polls copy b % kGoWords: one uncached hot spot less)
constexpr int kGoStride = 64;   // ... 256 bytes apart
constexpr int kGoBase = 128;

struct Buffers {
    float4 *poseA[2];
    float4 *poseB[2];
    float2 *poseC[2];
    float4 *lmkA[2];
    float *lmkB[2];
    int32_t *gen[2];      // genealogy rows; the live one is the pose's (Ctrl.live): both are gathered together.
                          // plain  : [cap_rows][Ncap]              element (row, k) at row * Ncap + k
                          // compact: [ceil(cap_rows/4)][Ncap][4]   element (row, k) at ((row / 4) * Ncap + k) * 4 + row % 4
    int32_t compact;      // which (fixed at context creation: cap_rows <= kSmallRows)
    // device copies of the host's genealogy bookkeeping, refreshed before the kernels that need them (gather, flatten,
    // shard pack / unpack; the update kernel gets what it needs with the observation packet)
    const int32_t *erow;  // [cap_nf] row of every landmark
    const int32_t *rows;  // [n_rows] live rows
    int32_t n_rows;
    int32_t cap_rows;
    // Sharded runs: landmark records of particles that ARRIVED from another shard.  Such a record needs a place no
    // sibling shares; rather than settling the whole shard it goes into this side pool ([cap_nf][pool_cap]) and the
    // genealogy entry says so (kPoolBit | pool slot).  Descendants share pool records like any other; the entry is
    // replaced by "own slot" the next time the landmark is observed.  flatten / a settling unpack empty the pool.
    float4 *poolA;
    float *poolB;
    int32_t pool_cap;
    const int32_t *lmk_live;  // [cap_nf] which record buffer of every landmark row is live: device copy of the host's table
                              // (the association is global, so the host knows: a re-observed row flips, a flatten /
                              // settle flips all), refreshed for flatten / shard pack + unpack; the update kernel gets
                              // the flags of the landmarks it touches with the observation packet (bit 30 of row[k])
    // distributed contexts (slamgpu_dist_*): n_shards > 1 shards of n = ncap particles each; this one is `shard`, its slot 0
    // is global particle `first`.  Single contexts: n_shards = 1, first = 0 (global ids == local slots).
    const PeerPtrs *peers;    // [n_shards] device table (entry `shard` = this context's own arrays)
    const float *gtot[2];     // all-gathered block totals by step parity, shard-major [n_shards][rows][nblocks]
    int32_t n_shards, shard;
    int32_t first;
    uint32_t pad_dist;
    unsigned long long div_n; // floor(2^64 / ncap) + 1: global slot -> shard by one 64-bit multiply-high
    Ctrl *ctrl;
    int32_t n;        // local particles
    int32_t ncap;     // row stride (>= n, multiple of 256)
    int32_t cap_nf;
    int32_t slot;     // which Ctrl.live / Ctrl.pend entry this launch reads (host-tracked)
};

// index of genealogy element (row, particle k) in Buffers::gen[b]
__host__ __device__ inline size_t gen_index(int compact, size_t ncap, int row, size_t k) {
    return compact ? (((size_t) (row >> 2) * ncap + k) << 2) + (size_t) (row & 3) : (size_t) row * ncap + k;
}

struct ObsPacket {          // big packets live in device memory: uploaded once per update by the host, or written there by
                            // observe_book_kernel (the device observation front end: the host never sees them)
    int32_t m, n, nf, n_rows;   // re-observed, new, landmarks before this update, live genealogy rows a pending gather composes
    int32_t e_new;              // genealogy row this update opens (-1: it writes no landmark)
    int32_t status;             // device front end: kStatusCapacity if new landmarks had to be dropped
    int32_t cap;                // 0: dense layout (host packets): int32 idf[m]; float zf[2m]; float zn[2n]; int32 row[m]; int32 rows[n_rows]
                                // C > 0: fixed layout (device packets): idf[C] zf[2C] zn[2C] row[C] rows[..], C = landmarks of the map
    int32_t pad;                // device-made packets: landmarks the launch consolidates (idf / row entries behind the re-observed ones)
    // (row[k] = genealogy row of re-observed landmark k BEFORE this update | live buffer << 30 | fresh << 29; rows = the
    //  rows still in use after it, without the one this update opens: what the copy roles of a pending lazy gather compose)
};

struct SmallObs {           // compact contexts: the packet travels in the kernel argument segment
    int32_t idf[kSmallObs];
    int32_t row[kSmallObs];          // genealogy row (| live record buffer << 30) of each re-observed landmark before this update
    float zf[2 * kSmallObs];
    float zn[2 * kSmallObs];
    int32_t head[8];                 // front-end launches (FrontArgs): the header the launch worked out for itself (kFrontHead*)
    uint32_t magic, pad;             // kSmallMagic: the update kernel copies this struct out of its kernel-argument segment
                                     // with vector loads at a computed offset and refuses to run on anything else
};
enum { kFrontHeadM = 0, kFrontHeadN, kFrontHeadNf, kFrontHeadENew, kFrontHeadChunks, kFrontHeadFresh, kFrontHeadCons, kFrontHeadStatus };

// Compact contexts, slamgpu_step_observe: the observation front end runs INSIDE the update launch.  The maps of compact
// contexts have at most 39 landmarks: one wave holds one landmark of the map per lane, so get_observations (core.cpp:185-273,
// :438-449), dataAssociationKnown (core.cpp:91-120) and the genealogy bookkeeping of slamgpu.cpp: do_update are a few
// ballots.  EVERY block's first wave works the packet out for itself from the state the previous launch left (a pure
// function of that state and these arguments: no block waits for another, no second stream, no event) and parks it in the
// LDS words a host-made packet would occupy; block 0 also stores the successor state -- into the OTHER copy, which no block
// of this launch reads -- and the packet, for slamgpu_observe_fetch.
struct FrontLm {       // one landmark of the map (its coordinates ride in the kernel arguments: UpdateArgs::small.zn)
    int32_t idf;       // feature index (dataAssociationTable), -1: never seen
    int32_t row;       // genealogy row | live record buffer << 30 (kRowLiveBit)
};
struct FrontHdr {
    int32_t nf, fresh_row, status, pad;
};
constexpr int kFrontLanes = 64;
struct FrontState {    // what one launch leaves for the next: two copies, read / written alternately
    FrontHdr hdr;
    FrontLm lm[kFrontLanes];
};
struct ObserveOut;
struct FrontArgs {
    int32_t on;                 // 0: host-made packet (UpdateArgs::small)
    int32_t nlm, cap_nf, noise; // noise: 0 none, 1 tape (UpdateArgs::small.zf[c] / zf[kSmallObs + c]: one normal pair per visible landmark), 2 Philox
                                // (the map: UpdateArgs::small.zn[t] / zn[kSmallObs + t] = x / y of landmark t)
    float x, y, phi, max_range, sr, sb;
    uint32_t k0, k1, step;
    int32_t cons_above;         // consolidate stale rows when more than this many are alive (< 0: never)
    const FrontState *state_in; // (also the kernel's leading parameter h_front: requested before this struct has arrived)
    FrontState *state_out;
    ObserveOut *out;            // observe_kernel's output block (z, vis, ...), written by block 0
    ObsPacket *pkt;             // fixed-layout copy of the packet (cap = nlm), written by block 0
};
constexpr uint32_t kSmallMagic = 0x534c414du;
constexpr int kSmallWords = (int) (sizeof(SmallObs) / 4);

struct RngArgs {
    int32_t mode;            // 0 tape, 1 philox
    uint32_t step;           // observation-step counter
    uint32_t k0, k1;         // philox key = seed
    int64_t first_particle;  // global id of local particle 0
    int64_t n_global;
    const float *normals;    // tape: [3][ncap] device (update) or [2][ncap] (predict), component-major
    const float *strata;     // tape: [n_global] device
    uint32_t prev_step;      // observation-step counter of the PREVIOUS update (its resampling may be planned inline)
    const float *strata_prev;  // tape: the previous update's strata
};

struct PredictStep {
    float V, G, phi_true;
    uint32_t step;           // control-step counter (philox stream)
    // particle-independent trigonometry of the control, evaluated once on the host with the same float libm
    // calls the reference makes (sin(G), cos(G): fastslam2.cpp:79; sin(G/wheelBase): :103) — only valid when the
    // control is not perturbed per particle (add_noise == 0)
    float sinG, cosG, sinGw;
    uint32_t pad;
};

// The queued predicts folded into ONE step, for the fast build.  FastSLAM2::predictState without per-particle control
// noise and without the heading observation is the same map for every particle when written in the frame of the
// particle's own pose at the first queued step (the motion model is SE(2)-equivariant): with T = diag(R(theta0), 1),
//   xy += R(theta0) (ax, ay),  theta += dth,  Pv = F Pv F^T + T M T^T,  F = I + [J R(theta0)(ax, ay); 0] e3^T,
// where (ax, ay), dth and the accumulated process noise M (body frame, symmetric-packed) come from running
// fastslam2.cpp:70-105 once on the host, in double, for theta0 = 0.
struct PredictComposite {
    int32_t valid;
    float ax, ay, dth;
    float m00, m10, m11, m20, m21, m22;
};

struct PredictArgs {
    int32_t nsteps;
    int32_t method, use_heading, add_noise;
    float Q[4];
    float dt, wheel_base, sigma_phi;
    PredictStep steps[kMaxFusedPredict];
    PredictComposite comp;
};

// Persistent small-N step loop (slamgpu_run_observe on compact single contexts of at most kPersistMaxBlocks tiles; reference:
// the wrapper's loop, fastslam1wrapper.cpp:51-113 / fastslam2wrapper.cpp:51-117, one iteration = predicts + observe + update +
// estimate).  At 1 000 particles a step is one dependent chain on four workgroups, and a third of it is the kernel boundary
// (3.07 us per launch against 0.73-1.2 us for an exchange between four workgroups of one XCD: profiles/xcd_exchange_r04.txt).
// ONE launch runs K iterations: grid of 8 x (tiles + 1) workgroups of which those with blockIdx % 8 == 0 stay (the dispatcher
// deals workgroups round-robin to the 8 XCDs: the ones that stay share an L2; verified at run time through HW_REG_XCC_ID, a
// placement on several XCDs takes the agent-scope release / acquire of the memory model instead), the tiles' workgroups + one
// helper (Ctrl words, estimate reductions); between two iterations they meet at a counter in L2.  Every spin is bounded; a
// workgroup that waits too long sets the abort word, everybody leaves, and the call returns SLAMGPU_ERR_BARRIER.
// What an iteration needs from the host rides in a queue in device memory, written before the launch and never during it.
constexpr int kPersistMaxBlocks = 8;
constexpr int kPersistStride = 8;       // workgroups that stay: blockIdx % kPersistStride == 0
struct PersistStep {
    PredictArgs PA;                     // the iteration's queued predicts (composed, compose_predicts)
    float fx, fy, fphi;                 // true pose: FrontArgs::x / y / phi
    uint32_t fstep;                     // FrontArgs::step
    uint32_t rng_step, rng_prev_step;   // RngArgs::step / prev_step
    int32_t wpar;                       // WeightScratch::wpar
    int32_t plan_inline;                // UpdateArgs::plan_inline (0 only in a first iteration whose predecessor's plan already ran)
    int32_t finalize, finalize_par;     // UpdateArgs::finalize / finalize_par
    double *finalize_hist;              // UpdateArgs::finalize_hist
};
constexpr int kPersistStepWords = (int) (sizeof(PersistStep) / 4);
// words of PersistArgs::sync, on lines of their own: [0, kPersistSyncWords) zeroed before every launch; the abort word behind
// them is sticky (a launch that finds it set leaves at once)
enum { kPersistSyncCounter = 0, kPersistSyncXcc = 64, kPersistSyncDone = 96, kPersistSyncCross = 97, kPersistSyncWords = 128, kPersistSyncAbort = 128,
       kPersistSyncAlloc = 192, kPersistHostWords = 4 };
struct PersistArgs {
    const PersistStep *queue;           // [K], in PINNED HOST memory: every workgroup reads an entry an iteration before it needs it
    int32_t K;
    uint32_t max_spins;                 // bound of one wait at the counter, in polls (tests: SLAMGPU_PERSIST_MAX_SPINS) ...
    uint32_t *sync;                     // [kPersistSyncAlloc]
    uint32_t *host_status;              // [kPersistHostWords] pinned host words, written by the helper workgroup of the FIRST launch that
                                        // is abandoned (the host reads them behind any synchronisation, without a copy of its own):
                                        // [0] 1 = abandoned, [1] iterations of that launch every workgroup had completed (barriers passed),
                                        // [2] that launch's serial number, [3] the iterations it had been handed
    FrontState *state_final;            // where the front end's state is left for the next launch
    int32_t *packets;                   // [2][kSmallWords]: the observation packets the helper workgroup makes, an iteration ahead
    float4 *draws;                      // [2][6][ncap]: FastSLAM 1 (fast build): the pose-independent half of a particle's eight predicts,
                                        // made an iteration ahead by the drawer workgroups (component c of particle i at [c][i]: V dt of
                                        // steps 0-3, 4-7; the perturbed G of steps 0-3, 4-7; sin(G / wheelBase) of steps 0-3, 4-7)
    int32_t drawers, serial;            // drawer workgroups: one per tile, or none; the launch's number (from 1) in this context
    unsigned long long max_ticks;       // ... and in TIME: ticks of the 100 MHz constant clock (s_memrealtime), looked at every 256 polls
    int32_t abort_at, pad;              // tests (SLAMGPU_PERSIST_ABORT_AT): the helper workgroup abandons the launch in this iteration (-1: never)
    PersistStep *ring;                  // [4] in device memory: the helper workgroup copies entry it + 2 of the (host-resident) queue here
                                        // during iteration it; everybody else reads its entries from here (an L2 hit), so that no
                                        // PCIe read sits in front of a tile's loads on the in-order return path
};

// Per-particle data association (reference: EKFSLAM::dataAssociate, ekfslam.cpp:151-189, applied by every
// particle to its own map, which Particle.cpp:61-73 lets grow per particle).  The device layout keeps ONE index space of landmark
// slots for all particles; what differs per particle is WHICH observation (if any) it associates with a slot, and whether it holds
// the slot's landmark at all: a particle that has not opened a landmark carries an ABSENT record there (xf = NaN: every gate of the
// association compares false on it, so such a record is never matched).  One step: the packet lists the union of the slots any
// particle matched (idf[m], the "re-observed" landmarks: every particle writes a fresh record there, updated by ITS observation or
// copied unchanged) and the slots opened for observations some particles call new (idn[n]: initialised from the observation by
// those particles, absent for the others); obs[(k) * ncap + i] = the observation particle i associates with packet entry k
// (k < m: re-observed entry k; m <= k < m + n: new slot k - m), or -1.
struct PerParticle {
    const int16_t *obs;      // [m + n][ncap]
    const float *z;          // [2 nz] the step's observations (range, bearing)
    const int32_t *idn;      // [n] slots of the new landmarks (dead slots are reused before the map grows)
    const float *wf;         // [ncap] weight factor of the observations the particle leaves unexplained: p_new ^ count (log-weights: count * log p_new)
    const uint8_t *any;      // [ncap] bit 0: the particle matched a landmark, bit 1: it opens one (neither: the step leaves its pose alone)
    int32_t nz, z_lds;       // observations of the step; 1: the launch stages z in LDS (2 nz floats fit: launch_update_any)
};
constexpr int kPpLdsObs = 2048;  // most observations a launch stages in LDS (16 KB)
constexpr float kAbsent = __builtin_nanf("");  // xf.x / xf.y of an absent record

struct UpdateArgs {
    int32_t method, m, n, nf;
    float R[4];
    const ObsPacket *big;    // null => use `small`
    int32_t dev_packet;      // 1: `big` was written by the device front end (observe_book_kernel): m, n, nf, e_new, n_rows and the
                             // copy-role geometry are read from its header by the kernel; the fields of this struct hold upper bounds
    SmallObs small;
    FrontArgs front;
    int32_t lazy;            // 1: single-context pipeline (honour Ctrl.pend, launch the copy + finalise blocks)
    int32_t copy_lo, copy_hi;  // copy roles (particle tile x kRowsPerRole genealogy rows) of a pending lazy gather this launch carries
    int32_t e_new;           // genealogy row this update opens for the landmarks it writes (-1: it writes none)
    int32_t n_rows;          // device packets: live rows a pending gather composes (the packet's rows), by copy roles
    int32_t rows_per_role;   // ... of this many rows each.  (Compact contexts: the compute threads copy all their chunks.)
    int32_t live_chunks;     // compact contexts: chunks [0, live_chunks) hold every row in use (rows are opened lowest-first)
    int32_t n_cons;          // compact contexts: landmarks CONSOLIDATED by this launch: idf[m + c] / row[m + c], c < n_cons, name landmarks
                             // that are not observed but whose record is rewritten, unchanged, into the particle's own slot of the
                             // row's other buffer -- they join the row this update opens, and the stale rows they leave stop costing
                             // 4 bytes per particle and resample (slamgpu.cpp: do_update)
    int32_t all_fresh;       // compact contexts: every staged re-observed landmark carries kRowFreshBit: their records are
                             // requested together with the pose, one dependent round trip earlier
    // Inline planning: the resampling stage of the PREVIOUS update (Neff, decision, ancestors, pose-estimate partials)
    // has not run as a launch of its own; every block of this launch redoes its scan and every thread finds its own
    // ancestor, so a step is ONE launch.  0: that stage already ran (resample_kernel), honour Ctrl.pend / keep[].
    int32_t arrivals;        // 1: legacy shard context: particles may have arrived from other shards (keep[i] < 0, pool records)
                             // 2: distributed context (Buffers::peers): ancestors, genealogy and records may live on another GPU
    int32_t plan_inline;
    int32_t scan_global;     // the prefix of the previous step's block totals is in WeightScratch::scan (scan_kernel ran)
    int32_t do_resample, n_effective;  // SWITCH_RESAMPLE, NEFFECTIVE (for the inline plan)
    int32_t logw;            // the context keeps LOG-weights in poseA.w (slamgpu_config.log_weights)
    int32_t push_totals;     // distributed contexts, push collective: every block stores its totals into EVERY shard's table
                             // (Buffers::peers[h].gtot) instead of leaving them to an all-gather
    uint32_t fold_seq;       // != 0: the barrier rides at the head of THIS launch (folded collective): the helper block
                             // announces fold_seq to every peer ("my previous launch has completed") and waits for theirs,
                             // every other block waits for the helper's go word before it touches anything
    uint32_t fold_spins;     // bound of the helper's poll (the other blocks wait four times as long)
    int32_t finalize_par;    // parity of the estimate partials the helper block reduces
    int32_t finalize;        // 1: the extra block reduces the previous update's pose-estimate partials
    double *finalize_hist;   // history slot of that estimate (kHistStride doubles) or null
    unsigned long long *stamps;  // diagnostic build (-DSLAM_STAMPS, libslamgpu_stamps.so): [compute blocks][kStampSlots] wall-clock stamps
    PersistArgs persist;         // update_persist only
    int32_t count_remote;        // distributed contexts: count the particles whose ancestor lived on another shard (Ctrl::remote_reads): one
                                 // atomic per wave and resample on a single address, so only when somebody reads it (slamgpu_dist_remote_reads)
    int32_t pad_u;
};
constexpr int kStampSlots = 16;
constexpr int kAssocNew = -1, kAssocDiscard = -2;
constexpr int kAssocBatch = 8;  // observations associated per pass over a particle's landmarks (registers)

// The weight scratch is double-buffered by the parity of the observation step (wpar): the update launch of step t
// writes lcum / blk_w [wpar] while -- when it also plans the resampling of step t-1 inline -- its blocks are still
// searching lcum / blk_w [wpar ^ 1]; est_part[q] holds the pose-estimate partials of the last step of parity q.
struct WeightScratch {
    float *lcum[2];     // [ncap]    inclusive in-block (256 particles) prefix of the raw weights
    float *blk_w[2];    // [3*nblocks] block totals T of w, then q = sum (w/T)^2 (scale-free: w^2 may overflow float32): one
                        //           allocation, so a shard's totals travel as one contiguous message ([w(nb) | q(nb)]);
                        //           log-weight contexts: third row = the block's largest log-weight M_b, and w / q / lcum
                        //           are those of exp(l - M_b)
    double *est_part[2];  // [nblocks][4] pose-estimate partials (sum x, sum y, heading, max w)
    double *scan[2];      // [nblocks + 4] large contexts: exclusive prefix of the block totals, then sum w, sum w^2, max log-weight
                          // (scan_kernel), so that the update launch need not rescan the totals in every block
    int32_t wpar;       // parity of the step this launch belongs to
    int32_t *keep[2];   // [ncap] ancestors of the last resample, double-buffered by Buffers::slot: a launch reads the
                        // pending gather through keep[slot] and writes new ancestors into keep[slot ^ 1]
    int32_t nblocks;
};

struct ResampleArgs {
    int32_t nf;             // landmarks after this update
    int32_t do_resample;    // SWITCH_RESAMPLE
    int32_t n_effective;    // NEFFECTIVE
    int32_t logw;           // log-weight context
    int32_t planned;        // 1: resample_ref_kernel has planned this stage (decision in Ctrl.resampled, ancestors in keep[], the weights
                            // normalised): resample_kernel only publishes the state and reduces the estimate partials
};
// Reference-order resampling stage (strict build, TAPE draws, at most kRefResampleMax particles): one block replays
// resampleParticles / stratifiedResample (core.cpp:718-824) operation by operation -- float32 w / sum(w) with Eigen's packet-order
// sum, Neff = 1 / sum(w^2) the same way, the serial float32 running prefix, `select[ctr] < cum[i]` -- so that Neff, the decision
// and every ancestor are the reference's bit for bit (a double-precision block scan puts a stratum within a few float32 ulps of a
// boundary on the other side: 8 of 5 000 ancestors at N = 5 000, rounds 3-4).
constexpr int kRefResampleMax = 5000;  // (the largest N the reference can resample -- stratifiedRandom yields N strata for 50, 100, 500, 1 000, 5 000 only,
                                       // core.cpp:751-763 -- and the largest the KAT vectors hold: no untested range; 40 KB of dynamic LDS)

// ---- sharded resampling (see kernels.hip) ---------------------------------------------------------------
constexpr int kMaxShards = 64;

struct ShardPlan {           // written by shard_plan_kernel, read back by the host
    double wsum, wsq;
    float neff;
    int32_t resampled;
    int64_t K[kMaxShards + 1];  // K[r] = first global output particle whose ancestor lives on shard r
    int32_t status, pad;        // kStatus* bits
};

struct ShardPlanArgs {
    const float *gblk;  // all-gathered block totals, shard-major [G][w(nb_per_shard) | w2(nb_per_shard)]
    int32_t nb_global, nb_per_shard, n_shards;
    int32_t do_resample, n_effective;
};

struct ShardPackArgs {
    const float *gblk;
    int32_t nb_global, first_block;  // this shard's first block in the global numbering
    int32_t nb_per_shard, pad0;
    int64_t k_lo, k_hi;              // offspring [K[g], K[g+1]) of this shard
    int64_t n_per_shard;
    int32_t nf, fields;              // fields = 10 + 5*nf floats per record
    int32_t shard, pad;              // offspring whose output slot is on this shard are gathered in place, not sent
    float *send;                     // device buffer: the offspring bound for OTHER shards, per-destination blocks
};

struct ShardUnpackArgs {
    int32_t pool_base;               // lazy arrivals: first free slot of the arrival pool; < 0: settle the whole shard
    int32_t own_lo, own_hi, pad1;    // local outputs [own_lo, own_hi) have local ancestors; the rest are arrivals
    const float *recv;               // device buffer, n_local*fields floats, blocks in source-shard order
    int32_t n_shards, nf, fields, shard;
    int64_t src_lo[kMaxShards + 1];  // local output index boundaries per source shard (own block: written by pack)
};

// ---- observation front end -------------------------------------------------------------
struct DevBook {           // device-resident genealogy bookkeeping header (slamgpu_step_observe)
    int32_t nf;            // landmarks known
    int32_t fresh_row;     // row the last update opened (records of its landmarks sit in the source slot itself), -1: none
    int32_t status;        // sticky kStatus* bits of the front end
    int32_t pad;
};

struct ObserveOut {
    int32_t nz, m, n, nf_after;
};

struct ObserveArgs {
    const float *lm;       // [2][nlm] landmark map, device
    int32_t *table;        // [nlm] dataAssociationTable: landmark -> feature index, -1 = never seen (device-resident)
    int32_t nlm, nf;       // map size; features known before this observation (observe_book_kernel: read from the book instead)
    float x, y, phi;       // true vehicle pose
    float max_range;
    float sr, sb;          // sqrt(R(0,0)), sqrt(R(1,1))
    int32_t noise;         // 0 none; 1 tape (r1 / r2: one normal per visible landmark, in visibility order); 2 Philox
    const float *r1, *r2;  // device, tape mode
    uint32_t k0, k1, step; // Philox key / observation step
    ObserveOut *out;       // device: header, then z[2 nlm], vis[nlm], zf[2 nlm], idf[nlm], zn[2 nlm] (4-byte units, in this order)
    // observe_book_kernel only: the observation packet of the update launch that follows, and the genealogy bookkeeping the
    // host otherwise does (slamgpu.cpp: do_update), device-resident
    ObsPacket *pkt;        // fixed-layout packet (cap = nlm)
    DevBook *book;
    int32_t *erow;         // [cap_nf] genealogy row of every landmark
    int32_t *live;         // [cap_nf] live record buffer of every landmark row
    int32_t *refcnt;       // [cap_rows] landmarks using each row
    int32_t cap_nf, cap_rows;
    int32_t cons_target;   // consolidate stale rows when more than this many are in use (< 0: never), at most
    int32_t cons_budget;   // max(cons_budget, visible / 16) landmarks per update
    int32_t *take;         // [cap_rows] scratch
};

// all-gather of the block totals between distributed contexts that share one device and one stream (rehearsal of the
// multi-GPU path on a single GPU): block (h, g) copies shard g's totals into slot g of shard h's table
struct DistGatherArgs {
    const float *local[kMaxShards];
    float *gathered[kMaxShards];
    int32_t n_shards, floats_per_shard;
};

// Barrier of the push collective: one wave per shard.  Lane t stores `seq` into shard t's flag word for this shard (a
// release at system scope: this shard's update launch has completed, its totals are in every table), then polls this
// shard's own flag word for shard t until it says `seq` too.  Flags are fine-grained device memory; the spin is bounded.
struct DistFlagArgs {
    uint32_t *peer_flags[kMaxShards];  // [h] = shard h's flag array (n_shards words), as mapped here
    uint32_t *my_flags;
    uint32_t *err;                     // set to seq if a peer did not arrive within max_spins polls
    int32_t n_shards, shard;
    uint32_t seq, max_spins;
};

// Read-only view of particles first, first + stride, ... (count of them) as the reference's vector<Particle> would hold
// them: pose through a pending lazy gather (keep[]), landmark records through the genealogy.  Nothing is written to the
// particle state (slamgpu_peek): a plot sink's decimated view, and the parity tests' window onto a genealogy that has been
// accumulating for hundreds of resamples.
struct PeekArgs {
    int32_t first, stride, count, nf;
    float4 *oa, *ob;     // [count] poseA / poseB as stored
    float2 *oc;          // [count]
    float4 *la;          // [nf][count] landmark records
    float *lb;
};

// ---- gated association with a spatial prefilter (slamgpu_associate_ex) ------------------------------------------------------
// Per landmark j, over ALL particles: the bounding box of its position estimates and the largest trace of its covariance
// (lmk_box_kernel, recomputed when the landmark is written), and from them a radius rho_j such that a particle's estimate of j
// farther than rho_j from the world point an observation implies cannot pass either gate for that particle
// (nis >= v_i^2 / S_ii, S_00 <= tr Pf + R00, S_11 <= tr Pf / d^2 + R11).  The landmarks are binned into a uniform grid over
// the region the observations can point into; every (particle, observation) pair then evaluates the landmarks of ONE cell.
struct LmkBox {
    float xmin, xmax, ymin, ymax, tmax, pad[3];
};
struct AssocGeom {
    float x0, y0, inv_cs, cs;     // grid origin, 1 / cell size, cell size
    int32_t nx, ny;
    float px0, px1, py0, py1;     // bounding box of the particle poses
    float zmax;                   // largest observed range
    float th_ref, dth0, dth1;     // headings of the particle poses: th_ref + [dth0, dth1] (dth1 - dth0 >= 2 pi: no bound)
    int32_t total, overflow;      // grid entries written / the entry buffer was too small (the caller falls back to brute force)
    unsigned long long pairs;     // (particle, observation, landmark) triples evaluated by associate_grid_kernel
};
constexpr int kAssocMaxCells = 64;  // per axis
constexpr int kAssocObsPerBlock = 4;  // AssocGridArgs::obs_per_block
// weighted vote per observation, on the device: a small open-addressing table per observation (the labels one observation
// draws from 10^5 particles are the handful of landmarks of one grid cell, NEW and DISCARD)
constexpr int kVoteSlots = 32;
constexpr int kVoteEmpty = (int) 0x80000000;
struct VoteSlot {
    int32_t key;   // label, kVoteEmpty: unused
    float w;       // sum of the weights of the particles that chose it
};
struct AssocGridArgs {
    LmkBox *box;                  // [nf]; pad[0]: the radial bound of this call's gates (assoc_count_kernel writes it, associate_grid_kernel reads it)
    AssocGeom *geom;
    float *geom_part;             // [64][8] partial pose boxes / heading ranges (assoc_geom_partial_kernel)
    int32_t *cell_start;          // [nx * ny + 1] exclusive prefix of the cell populations
    int32_t *cell_fill;           // [nx * ny] cursors of the fill pass
    float4 *items;                // [2 cap_items] cell after cell, an entry = two float4: (xmin, xmax, ymin, ymax) of the landmark's box and
                                  // (radial bound of this call's gates, landmark id as bits, tmax, the same bound for G1): what the walk needs of a landmark in ONE
                                  // contiguous read (round 6: an id here and the box behind it were two dependent trips per entry)
    int32_t cap_items, nf, nz;
    const float *z;               // [2 nz] observations (range, bearing), device
    float r00, r11, G;            // R diagonal; G = max(gate_reject, gate_augment) with the safety margin
    float G1;                     // gate_reject with the safety margin: the bound of the walk's FIRST pass (associate_grid_kernel)
    int32_t obs_per_block;        // observations one thread of associate_grid_kernel works through (blockIdx.y = a group of that many)
    float g1_ratio, pad_r;        // sqrt(G1 / G): a LIST entry's radial bound for the first pass is derived from its bound for G (its fourth word
                                  // carries the landmark's genealogy row | live buffer << 30 instead: one dependent trip less per evaluated triple)
    int32_t lab_by_obs;           // labels laid out [nz][ncap] (the per-particle update's) instead of [n][nz]
    int32_t lcap;                 // candidate LISTS per observation instead of grid cells (assoc_lists_kernel): entries of observation q at
                                  // items[2 (q lcap + k)], k < cell_start[q]; 0: the grid
    VoteSlot *votes;              // [nz][kVoteSlots] or null: the weighted vote per observation (AssocGeom::overflow bit 1: a table filled up)
    int32_t *census_first, *census_news;  // per-particle update (or null): the census of the labels taken right here, where they are made --
                                  // first[l] = the lowest observation naming landmark slot l (preset to INT_MAX), news[q] = particles calling
                                  // observation q new (preset to 0): what pp_census_kernel works out from the label array otherwise
    float *vote_w;                // candidate lists only (lcap > 0), or null: [nz][lcap + 2] weights addressed DIRECTLY -- [0] new, [1] discard,
                                  // [2 + k] entry k of the observation's list -- so that a vote is one atomic add nobody waits for (the hash
                                  // table's look-up was a trip to L2 in every wave's dependent chain, once per observation: 1.2 ms of a
                                  // 3.4 ms association at config 5); vote_compact_kernel turns them into `votes` afterwards
    int32_t logw;                 // the context keeps log-weights
};

struct KernelTable {
    // the step: [resampling stage of the previous update, inline] + [gather] + [fused predicts] + per-particle observation
    // update + in-block weight prefix / totals  (+ helper blocks: genealogy copy, Ctrl words, estimate reduction)
    void (*update)(hipStream_t, const Buffers &, const PredictArgs &, const UpdateArgs &, const RngArgs &,
                   const WeightScratch &);
    // the same step with a per-particle association (PerParticle: update_kernel<.., PP = true>; single contexts on plain rows)
    void (*update_particle)(hipStream_t, const Buffers &, const PredictArgs &, const UpdateArgs &, const RngArgs &, const WeightScratch &,
                            const PerParticle &);
    // K steps of a compact single context in ONE launch (PersistArgs): U = what every iteration shares (front end on: the map, R)
    void (*update_persist)(hipStream_t, const Buffers &, const PredictArgs &, const UpdateArgs &, const RngArgs &, const WeightScratch &);
    // the resampling stage as a launch of its own (on demand): Neff + decision; normalise, or the ancestors of a
    // stratified resample into keep[] (nothing is moved); estimate partials
    void (*resample)(hipStream_t, const Buffers &, const WeightScratch &, const RngArgs &, const ResampleArgs &,
                     const UpdateArgs &);
    // the plan of that stage in the reference's own order of operations (kernels.h: kRefResampleMax); followed by `resample` with
    // ResampleArgs::planned = 1
    void (*resample_ref)(hipStream_t, const Buffers &, const WeightScratch &, const RngArgs &, const ResampleArgs &);
    // large contexts: prefix of this step's block totals into WeightScratch::scan[wpar] (one block)
    void (*scan)(hipStream_t, const WeightScratch &, int logw);
    // materialise a pending lazy gather (needed before anything but the next update touches the particle set)
    void (*gather)(hipStream_t, const Buffers &, const WeightScratch &);
    // rewrite every landmark record into its particle's own slot, genealogy row 0 = identity (download, sharded arrivals)
    void (*flatten)(hipStream_t, const Buffers &, int nf);
    // identity ("own slot") in genealogy row `row` of gen[which]
    void (*identity)(hipStream_t, const Buffers &, int which, int row);
    // genealogy rows from the compact layout (src: [ceil(rows / 4)][ncap][4]) into plain rows (dst: [rows][ncap]); src != dst
    void (*decompact)(hipStream_t, const int32_t *src, int32_t *dst, int ncap, int rows);
    // reduce the estimate partials est_part[par] now (-> Ctrl.est, history slot)
    void (*finish)(hipStream_t, const Buffers &, const WeightScratch &, double *hist, int par);
    void (*predict)(hipStream_t, const Buffers &, const PredictArgs &, const RngArgs &);
    void (*estimate)(hipStream_t, const Buffers &, const WeightScratch &, double *hist);
    void (*jacobians)(hipStream_t, const float *in_dev, uint32_t n, float *out_dev);
    // known-answer entry point for the scalar device functions (slamgpu_kat): op 0 trigonometricOffset, 1 gaussEvaluate D=2,
    // 2 gaussEvaluate D=3, in the arithmetic this build's update kernel uses
    void (*kat)(hipStream_t, int op, const float *in_dev, int n, float *out_dev);
    // observation front end on the device (slamgpu_observe): visibility scan + range / bearing + sensor noise + known data
    // association, one block; results into `out` (ObserveOut header, then z[2 cap], vis[cap], zf[2 cap], idf[cap], zn[2 cap])
    void (*observe)(hipStream_t, const ObserveArgs &);
    // the same + the genealogy bookkeeping of the update that consumes the observation: everything into device memory
    // (ObserveArgs::pkt / book / erow / live / refcnt); nothing comes back to the host
    void (*observe_book)(hipStream_t, const ObserveArgs &);
    // per-particle gated nearest-neighbour association of nz observations against every landmark of every particle
    // (slamgpu_associate): labels [n][nz] = landmark index, kAssocNew or kAssocDiscard.  Plain set required (no pending gather).
    // retired (may be null): bit j set = landmark j takes no part (slamgpu_retire_landmarks)
    // excl3 (may be null): excl_base, excl_per_m, unique_ratio of the exclusion rule (slamgpu_particle_assoc; base + per_m = 0: off)
    void (*associate)(hipStream_t, const Buffers &, int nf, const float *z_dev, int nz, const float *R4, float gate_reject,
                      float gate_augment, const float *excl3, const uint32_t *retired_dev, int32_t *labels_dev, int labels_by_obs);
    // seq_out != null: `out` and `seq_out` are pinned host memory; the kernel stores `seq` there last (system-scope fence)
    void (*shard_plan)(hipStream_t, const ShardPlanArgs &, const RngArgs &, ShardPlan *out, uint32_t *seq_out, uint32_t seq);
    void (*shard_pack)(hipStream_t, const Buffers &, const WeightScratch &, const ShardPackArgs &, const RngArgs &);
    void (*shard_unpack)(hipStream_t, const Buffers &, const WeightScratch &, const ShardUnpackArgs &);
    // normalise or leave the lazy gather pending; this shard's pose-estimate partials; outcome into Ctrl
    void (*shard_finish)(hipStream_t, const Buffers &, const WeightScratch &, double W, double Q, float neff, int resampled);
    void (*dist_gather)(hipStream_t, const DistGatherArgs &);
    void (*dist_flags)(hipStream_t, const DistFlagArgs &);
    void (*peek)(hipStream_t, const Buffers &, const WeightScratch &, const PeekArgs &);
    // bounding boxes of the listed landmarks (ids on the device) over all particle slots of their live record buffer
    // (a retired landmark gets the EMPTY box: it enters no grid cell and is never evaluated)
    void (*lmk_box)(hipStream_t, const Buffers &, const int32_t *ids_dev, int count, const uint32_t *retired_dev, LmkBox *box_dev);
    // geometry + grid of the landmark boxes for one association call (four small launches)
    void (*assoc_grid)(hipStream_t, const Buffers &, const AssocGridArgs &);
    // ... or geometry + one candidate list per observation (AssocGridArgs::lcap > 0; nz <= kAssocMaxCells^2)
    void (*assoc_lists)(hipStream_t, const Buffers &, const AssocGridArgs &);
    // AssocGridArgs::vote_w -> votes (after associate_grid; one block per observation)
    void (*vote_compact)(hipStream_t, const AssocGridArgs &);
    // slamgpu_associate through the grid: the same labels as `associate`, evaluating only the landmarks of one cell per
    // (particle, observation)
    void (*associate_grid)(hipStream_t, const Buffers &, const AssocGridArgs &, const float *R4, float gate_reject, float gate_augment,
                           int32_t *labels_dev);
    // seam 1, MULTIPARTICLE_ACCELERATOR form: self-describing records back to back, outputs in place; tab: 3 words per feature
    void (*jacobians_multi)(hipStream_t, float *win_dev, const uint32_t *tab_dev, uint32_t nfeat);
    // per-particle association (PerParticle; labels BY OBSERVATION here, [nz][ncap]): census of the labels (first[l]: lowest observation naming landmark slot l, preset
    // to INT_MAX; news[j]: particles calling observation j new, preset to 0), the labels resolved into PerParticle::obs / wf / any, and
    // the number of particles that hold each landmark slot (holders[l], preset to 0; plain set, tables in sync)
    void (*pp_census)(hipStream_t, const int32_t *labels_dev, int n, int nz, int ncap, int32_t *first_dev, int32_t *news_dev);
    void (*pp_resolve)(hipStream_t, const int32_t *labels_dev, int n, int nz, int ncap, const int32_t *uidx_dev, const int32_t *newk_dev, int m, int nn,
                       float p_new, int logw, int16_t *obs_dev, float *wf_dev, uint8_t *any_dev);
    void (*pp_holders)(hipStream_t, const Buffers &, int count, const int32_t *ids_dev, int32_t *holders_dev);  // ids (may be null: 0 .. count - 1)
};

const KernelTable *kernels_strict();
const KernelTable *kernels_fast();

}  // namespace slamgpu
