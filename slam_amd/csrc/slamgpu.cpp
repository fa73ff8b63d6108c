// libslamgpu.so — C ABI over the gfx950 kernels (include/slamgpu.h).  Host side only: owns the device
// buffers, the HIP stream, the pinned staging rings and the lazy predict queue.  No CPU compute path:
// if there is no usable GPU every entry point fails with SLAMGPU_ERR_NO_DEVICE.
#define SLAMGPU_EXPERIMENTAL 1  // (the library defines every entry point, the experimental ones included)
#include "../../include/slamgpu.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

using namespace slamgpu;

namespace {

thread_local char g_err[1024] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// RCCL is bound at run time (dlopen): the library of the process (torch's when torch is loaded, /opt/rocm's otherwise), and no
// link-time dependency for single-GPU users.
struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    bool ok = false;
};
const Rccl *rccl() {
    static Rccl R;
    static bool tried = false;
    if (tried) return R.ok ? &R : nullptr;
    tried = true;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return nullptr;
#define SLAM_SYM(f) R.f = reinterpret_cast<decltype(R.f)>(dlsym(h, "nccl" #f))
    SLAM_SYM(GetUniqueId);
    SLAM_SYM(CommInitRank);
    SLAM_SYM(CommInitAll);
    SLAM_SYM(CommDestroy);
    SLAM_SYM(AllGather);
    SLAM_SYM(GroupStart);
    SLAM_SYM(GroupEnd);
    SLAM_SYM(GetErrorString);
    SLAM_SYM(CommCount);
    SLAM_SYM(CommUserRank);
#undef SLAM_SYM
    R.ok = R.GetUniqueId && R.CommInitRank && R.CommInitAll && R.CommDestroy && R.AllGather && R.GroupStart && R.GroupEnd && R.GetErrorString;
    return R.ok ? &R : nullptr;
}
#define RCCL_TRY(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) return fail(SLAMGPU_ERR_HIP, "%s: %s", #expr, rccl()->GetErrorString(r_)); \
    } while (0)


#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(SLAMGPU_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

constexpr int kRing = 64;       // big-packet staging slots
constexpr int kPlainRowsTarget = 2048;  // plain-row contexts: genealogy rows in use before updates start consolidating the emptiest ones
constexpr int kPlainConsBudget = 32;  // ... landmarks moved per update, at least
constexpr int kMidRowsHigh = 24, kMidRowsLow = 12;  // compact contexts of mid-size maps: consolidate the emptiest rows from ... down to ... rows in use
constexpr int kStageBound = 8;  // = kStage of kernels.hip: re-observed landmarks whose records an update launch stages in LDS
constexpr int kConsolidateAbove = 6;  // compact contexts: genealogy rows alive before stale rows are consolidated (3..8 measure alike; profiles/consolidate_sweep_r03.txt)
constexpr int kHistCap = 4096;  // asynchronous pose-estimate history entries

struct EventPair {
    hipEvent_t a, b;
};

struct KernelStat {
    std::vector<EventPair> pending;
    double ms = 0;
    int64_t launches = 0;
};

}  // namespace

struct slamgpu_ctx {
    slamgpu_config cfg{};
    const KernelTable *k = nullptr;
    hipStream_t stream = nullptr;
    Buffers B{};
    WeightScratch ws{};
    int nf = 0;
    uint32_t obs_step = 0, ctl_step = 0;
    uint32_t rng_skew = 0;  // update launches that were not filter steps (slamgpu_dist_settle): they draw nothing
    // big-packet ring (observation packets that do not fit the kernel-argument form)
    size_t pkt_bytes = 0;
    char *pkt_host = nullptr;  // pinned [kRing][pkt_bytes]
    char *pkt_dev = nullptr;
    hipEvent_t pkt_ev[kRing]{};
    bool pkt_ev_used[kRing]{};
    uint64_t pkt_seq = 0;
    // tape staging (TAPE mode)
    float *tape_host = nullptr;  // pinned: normals [3][ncap] (or predict [2][ncap]) + strata [n_global]
    float *normals_dev = nullptr;
    float *strata_dev[2] = {nullptr, nullptr};  // by step parity: an update launch may still need the previous step's
    // lazy predict queue
    PredictArgs pending{};
    // host mirror of ctrl for readback
    Ctrl *ctrl_host = nullptr;  // pinned
    // pose-estimate history
    double *hist_dev = nullptr;  // [kHistCap][kHistStride]
    double *hist_host = nullptr; // pinned mirror of it (history_to_host)
    int hist_n = 0;
    bool est_fresh = false;  // Ctrl.est / hist slot hist_n were written by the last update and nothing changed since
    // profiling
    bool profile = false;
    std::map<std::string, KernelStat> stats;
    std::vector<hipEvent_t> ev_pool;
    hipEvent_t timer_a = nullptr, timer_b = nullptr;  // slamgpu_timer_start / _stop
    double predict_bytes = 0;
    bool own_stream = true;
    ShardPlan *plan_dev = nullptr, *plan_host = nullptr;  // sharded resampling plan (device + pinned mirror)
    uint32_t *plan_seq_host = nullptr;  // pinned: sequence number the plan kernel stores after the plan
    uint32_t plan_seq = 0;
    // Ctrl.live / Ctrl.pend slot the next launch reads (kernels.h: Ctrl); flipped after every launch that may
    // change the live buffer (resample_kernel, gather_kernel, shard_commit_kernel)
    int64_t pool_used = 0;        // arrival-pool slots handed out since the pool was last emptied (flatten / settle / upload)
    bool shard_settled = false;   // the sharded resampling stage of this step moved everything physically (records arrived)
    int slot = 0;
    int keep_slot = 0;            // which WeightScratch::keep buffer holds the ancestors of the last update
    bool maybe_pending = false;   // the last update may have left a lazy gather (only the device knows)
    bool shard_est_fresh = false; // sharded: est_part holds this shard's partials of the last update (shard_finalize_kernel)
    bool own_totals = true;       // ws.blk_w is this context's allocation (not a caller-provided collective buffer)
    float *own_blk_w = nullptr;
    // Pose-estimate pipeline of the single-context path.  The resampling stage of update t (Neff, decision, ancestors,
    // estimate partials) normally runs INSIDE the launch of update t+1 (UpdateArgs::plan_inline) and its partials are
    // reduced by the helper block of launch t+2; anything that needs results earlier runs them as launches of their own.
    struct EstStage {
        bool has = false;
        int par = 0;              // step parity: which est_part / lcum / blk_w buffers
        uint32_t step = 0;        // observation-step counter of that update (Philox stream of its strata)
        int nf = 0;               // landmarks after that update
        double *hist = nullptr;   // history slot its estimate belongs to (or null)
    };
    bool scan_ready = false;      // scan_kernel ran on the last update's block totals (large contexts)
    bool mid_compact = false;     // compact layout on a map of more than 39 landmarks (kernels.h: kMidLandmarks): host-made packets only
    bool ref_resample = false;    // the resampling stage replays the reference's order of operations (kernels.h: kRefResampleMax):
                                  // strict build, the caller's draws (TAPE), a single context of at most 5 000 particles, linear weights
    bool consolidate = true;      // row consolidation of compact contexts (do_update); SLAMGPU_NO_CONSOLIDATE=1 turns it off
    int consolidate_above = kConsolidateAbove;  // (SLAMGPU_CONSOLIDATE_ABOVE: diagnostic)
    int plain_rows_target = kPlainRowsTarget;   // (SLAMGPU_PLAIN_ROWS_TARGET: diagnostic / tests)
    int scan_min_blocks = 1024;   // contexts with more blocks of 256 particles than this use scan_kernel (262 144 particles)
    // Genealogy bookkeeping (kernels.h: gen).  The association is global, so the host knows which genealogy row every
    // landmark uses: a step that writes landmarks opens a new row for them; a row whose last landmark moved on is recycled.
    std::vector<uint32_t> seen_step; // [cap_nf] observation step that last re-observed each landmark (duplicate check)
    std::vector<int32_t> live_flag;  // [cap_nf] which record buffer of every landmark row is live (flips when re-observed)
    int32_t *live_dev = nullptr;     // device copy for flatten / shard pack + unpack
    std::vector<int32_t> erow;       // [cap_nf] row of every landmark
    std::vector<int32_t> refcnt;     // [cap_rows] landmarks using each row
    std::vector<int32_t> free_rows;  // stack of unused rows
    std::vector<int32_t> live_rows;  // rows with refcnt > 0
    std::vector<int32_t> live_pos;   // [cap_rows] position in live_rows, -1 if not live
    int32_t *erow_dev = nullptr, *rows_dev = nullptr;  // device copies for gather / flatten / shard pack + unpack
    // distributed operation (slamgpu_dist_*)
    bool dist = false, dist_clean = false;
    PeerPtrs *peers_dev = nullptr;
    float *gtot_dev[2] = {nullptr, nullptr};
    std::vector<void *> ipc_opened;
    void *comm = nullptr;  // ncclComm_t: when set, slamgpu_dist_step / _settle run the all-gather themselves
    // push collective (slamgpu_dist_set_collective): the update launch stores its totals into every shard's table and a
    // one-wave flag kernel is the barrier; flags_dev = [kMaxShards] flag words + the error word, fine-grained memory
    bool count_remote = false;   // slamgpu_dist_remote_reads has been asked for: the update launches keep the counter from then on
    bool dist_push = false;
    bool dist_fold = false;  // push + the barrier folded into the head of the next update launch (SLAMGPU_DIST_FOLD)
    uint32_t *flags_dev = nullptr;
    uint32_t *peer_flags[kMaxShards] = {};
    uint32_t flag_seq = 0;
    // observation front end (slamgpu_set_map / slamgpu_observe)
    float *map_dev = nullptr, *obs_r_dev = nullptr;
    int32_t *table_dev = nullptr;
    ObserveOut *obs_out_dev = nullptr;
    int32_t map_n = 0, obs_nf = 0;
    uint32_t observe_step = 0;
    int fresh_row = -1;              // row the last update opened, while nothing but the resample the next update launch
                                     // applies has touched it: records of its landmarks sit in the source slot itself
    bool tables_dirty = true;
    // device-resident genealogy bookkeeping (slamgpu_step_observe): while book_on_device the tables erow_dev / live_dev /
    // refcnt_dev / book_dev are the truth and the host's vectors are stale; book_pull / book_push hand the ownership over
    bool book_on_device = false;
    int32_t front_status = 0;        // sticky kStatus* bits of the device front end seen by book_pull (carried back by book_push)
    DevBook *book_dev = nullptr;
    int32_t *refcnt_dev = nullptr, *take_dev = nullptr;
    int32_t *book_host = nullptr;    // pinned staging of book_pull / book_push
    hipStream_t obs_stream = nullptr;  // the front-end kernels run here, a step ahead of the update launches (events order them)
    hipEvent_t obs_ev[kRing]{};        // observe_book of the packet in ring slot k has finished
    char *last_pkt_dev = nullptr;    // packet of the last slamgpu_step_observe (slamgpu_observe_fetch)
    // compact contexts: the front end runs inside the update launch (kernels.h: FrontArgs); its state lives in two device
    // copies, read / written alternately (front_par: the one the next launch reads)
    std::vector<float> map_host;     // [2][map_n], as slamgpu_set_map received it
    FrontState *front_dev = nullptr, *front_host = nullptr;
    ObsPacket *front_pkt_dev = nullptr;
    int front_par = 0;
    bool front_ready = false;
    // gated association with the spatial prefilter (slamgpu_associate_ex): per-landmark boxes over all particles, refreshed
    // for the landmarks written since (box_dirty), and the grid buffers
    LmkBox *box_dev = nullptr;
    std::vector<char> box_dirty;
    // landmarks the caller has retired from the gated association (slamgpu_retire_landmarks): host flags + the device's bit mask
    std::vector<char> retired;
    uint32_t *retired_dev = nullptr;
    int n_retired = 0;
    int32_t *assoc_ids_dev = nullptr, *cell_start_dev = nullptr, *cell_fill_dev = nullptr;
    // per-particle association (slamgpu_update_particle / _labels; kernels.h: PerParticle): device scratch, grown on demand
    int32_t *pp_lab_dev = nullptr;   // labels BY OBSERVATION, [nz][ncap]
    size_t pp_lab_cap = 0;
    int16_t *pp_obs_dev = nullptr;   // PerParticle::obs [rows][ncap]
    size_t pp_obs_rows = 0;
    float *pp_z_dev = nullptr;       // [2 pp_nz_cap]
    int32_t *pp_tab_dev = nullptr;   // [cap_nf] first / uidx | [cap_nf] holders | [pp_nz_cap] news / newk | [pp_nz_cap] idn
    int pp_nz_cap = 0;
    float *pp_wf_dev = nullptr;      // [ncap]
    uint8_t *pp_any_dev = nullptr;   // [ncap]
    std::vector<char> pp_partial;    // slots that NOT every particle opened: the only ones that can lose their last holder (a slot every particle
                                     // opened is held by every descendant for good): what the holders census counts
    std::vector<char> pp_dead;       // landmark slots no particle holds any more (their hypotheses died in a resample): out of the
    std::vector<int32_t> pp_dead_list;  // association (retired) until a later landmark opens them again
    float *vote_w_dev = nullptr;     // AssocGridArgs::vote_w, grown on demand
    size_t vote_w_cap = 0;
    float *assoc_z_dev = nullptr;    // the observations of an association call / its vote tables: kept between calls (an allocation and a release
    VoteSlot *assoc_votes_dev = nullptr;  // per call each), grown on demand
    int assoc_nz_cap = 0;
    bool retired_stale = false;      // the host's retired flags have changed since the device's mask was written (retired_upload clears it)
    uint64_t pp_steps = 0;
    bool pp_census_done = false;     // the association kernel took the census of the labels itself (AssocGridArgs::census_first): pp_census_kernel is skipped
    const PerParticle *pp_launch = nullptr;  // set around issue_update by do_update_particle: the launch takes update_kernel<.., PP = true>
    float4 *items_dev = nullptr;  // [2 cap_items]: kernels.h: AssocGridArgs::items
    AssocGeom *geom_dev = nullptr;
    int32_t cap_items = 0;
    char *peek_dev = nullptr;        // staging of slamgpu_peek, grown on demand
    size_t peek_bytes = 0;
    unsigned long long *stamps_dev = nullptr;  // diagnostic (SLAMGPU_STAMPS=1 + libslamgpu_stamps.so): UpdateArgs::stamps
    // persistent small-N step loop (slamgpu_run_observe, kernels.h: PersistArgs)
    bool persist_ok = true;              // SLAMGPU_NO_PERSIST=1 turns it off (diagnostic / tests: the per-step loop)
    struct PersistCollect {              // while set, issue_update queues its launch instead of making it
        std::vector<PersistStep> steps;
        bool have_first = false;
        Buffers B{};
        UpdateArgs U{};
        RngArgs rng{};
        WeightScratch ws{};
    } *collect = nullptr;
    // the queue of a launch lives in PINNED HOST memory and the kernel reads it there (an entry an iteration ahead: the PCIe trip is
    // hidden): kPqBufs buffers of pq_cap entries used in turn; a buffer is rewritten once the launch that read it has finished
    static constexpr int kPqBufs = 4;
    PersistStep *pq_host = nullptr;
    size_t pq_cap = 0;
    hipEvent_t pq_kev[kPqBufs] = {};
    bool pq_kev_used[kPqBufs] = {};
    int pq_next = 0;
    uint32_t *psync_dev = nullptr, *pstatus_host = nullptr;
    int32_t *ppk_dev = nullptr;          // [2][kSmallWords] observation packets of the loop's helper workgroup
    PersistStep *pring_dev = nullptr;    // [4] the loop's ring of queue entries in device memory (kernels.h: PersistArgs::ring)
    float4 *pdraw_dev = nullptr;         // [2][6][ncap] draws of the loop's drawer workgroups (FastSLAM 1, fast build)
    int64_t persist_launches = 0, persist_steps = 0;
    EstStage unplanned;           // the last update: resampling stage not run yet
    EstStage unreduced;           // an update whose partials exist (est_part[par]) but are not reduced yet
};

namespace {

int64_t n_global(const slamgpu_ctx *c) { return c->cfg.n_particles_global > 0 ? c->cfg.n_particles_global : c->cfg.n_particles; }

hipEvent_t get_event(slamgpu_ctx *c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

struct Timed {
    slamgpu_ctx *c;
    KernelStat *st = nullptr;
    EventPair ep{};
    Timed(slamgpu_ctx *ctx, const char *name) : c(ctx) {
        if (!c->profile) return;
        st = &c->stats[name];
        ep.a = get_event(c);
        ep.b = get_event(c);
        if (ep.a) (void) hipEventRecord(ep.a, c->stream);
    }
    ~Timed() {
        if (!st) return;
        if (ep.b) (void) hipEventRecord(ep.b, c->stream);
        st->pending.push_back(ep);
        st->launches++;
    }
};

void drain_stats(slamgpu_ctx *c) {
    for (auto &kv : c->stats) {
        for (auto &ep : kv.second.pending) {
            float ms = 0;
            if (ep.a && ep.b && hipEventSynchronize(ep.b) == hipSuccess && hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess)
                kv.second.ms += ms;
            if (ep.a) c->ev_pool.push_back(ep.a);
            if (ep.b) c->ev_pool.push_back(ep.b);
        }
        kv.second.pending.clear();
    }
}

RngArgs rng_args(const slamgpu_ctx *c, uint32_t step) {
    RngArgs r{};
    r.mode = c->cfg.rng_mode;
    r.step = step;
    r.k0 = (uint32_t) c->cfg.seed;
    r.k1 = (uint32_t) (c->cfg.seed >> 32);
    r.first_particle = c->cfg.first_particle;
    r.n_global = n_global(c);
    r.normals = c->normals_dev;
    r.strata = c->strata_dev[step & 1];
    r.prev_step = step - 1;
    r.strata_prev = c->strata_dev[(step & 1) ^ 1];
    return r;
}

// Fold the queued predicts into PredictArgs::comp (see kernels.h): FastSLAM2::predictState (fastslam2.cpp:70-105)
// run once in double for a particle at the origin with heading 0.
void compose_predicts(PredictArgs &P) {
    PredictComposite &C = P.comp;
    C = PredictComposite{};
    if (P.method != SLAMGPU_FASTSLAM2 || P.add_noise || P.use_heading || P.nsteps == 0) return;
    static const bool off = getenv("SLAMGPU_NO_COMPOSITE") != nullptr;  // diagnostic: sequential predicts in the fast build
    if (off) return;
    const double dt = P.dt, wb = P.wheel_base;
    const double Q[2][2] = {{P.Q[0], P.Q[1]}, {P.Q[2], P.Q[3]}};
    double ax = 0, ay = 0, dth = 0;
    double M[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int k = 0; k < P.nsteps; k++) {
        const double V = P.steps[k].V, G = P.steps[k].G;
        const double sn = sin(G + dth), cs = cos(G + dth);
        const double Gv[3][3] = {{1, 0, -V * dt * sn}, {0, 1, V * dt * cs}, {0, 0, 1}};
        const double Gu[3][2] = {{dt * cs, -V * dt * sn}, {dt * sn, V * dt * cs}, {dt * sin(G) / wb, V * dt * cos(G) / wb}};
        double T[3][3], N[3][3], U[3][2];
        for (int r = 0; r < 3; r++)
            for (int c2 = 0; c2 < 3; c2++) T[r][c2] = Gv[r][0] * M[0][c2] + Gv[r][1] * M[1][c2] + Gv[r][2] * M[2][c2];
        for (int r = 0; r < 3; r++)
            for (int c2 = 0; c2 < 3; c2++) N[r][c2] = T[r][0] * Gv[c2][0] + T[r][1] * Gv[c2][1] + T[r][2] * Gv[c2][2];
        for (int r = 0; r < 3; r++)
            for (int c2 = 0; c2 < 2; c2++) U[r][c2] = Gu[r][0] * Q[0][c2] + Gu[r][1] * Q[1][c2];
        for (int r = 0; r < 3; r++)
            for (int c2 = 0; c2 < 3; c2++) M[r][c2] = N[r][c2] + U[r][0] * Gu[c2][0] + U[r][1] * Gu[c2][1];
        ax += V * dt * cs;
        ay += V * dt * sn;
        dth += V * dt * sin(G / wb);  // sin(G/wheelBase): upstream quirk (fastslam2.cpp:103)
    }
    C.valid = 1;
    C.ax = (float) ax;
    C.ay = (float) ay;
    C.dth = (float) dth;
    C.m00 = (float) M[0][0];
    C.m10 = (float) (0.5 * (M[1][0] + M[0][1]));
    C.m11 = (float) M[1][1];
    C.m20 = (float) (0.5 * (M[2][0] + M[0][2]));
    C.m21 = (float) (0.5 * (M[2][1] + M[1][2]));
    C.m22 = (float) M[2][2];
}

// ---- genealogy rows (host bookkeeping) ----
void rows_add_live(slamgpu_ctx *c, int r) {
    c->live_pos[r] = (int32_t) c->live_rows.size();
    c->live_rows.push_back(r);
}

void rows_remove_live(slamgpu_ctx *c, int r) {
    const int p = c->live_pos[r], last = c->live_rows.back();
    c->live_rows[p] = last;
    c->live_pos[last] = p;
    c->live_rows.pop_back();
    c->live_pos[r] = -1;
}

// every landmark [0, nf) in row 0 (own slot): after upload, flatten, a settling unpack
void rows_reset(slamgpu_ctx *c, int nf) {
    const int cap_rows = c->B.cap_rows;
    c->fresh_row = -1;
    std::fill(c->refcnt.begin(), c->refcnt.end(), 0);
    std::fill(c->live_pos.begin(), c->live_pos.end(), -1);
    std::fill(c->erow.begin(), c->erow.end(), 0);
    c->live_rows.clear();
    c->free_rows.clear();
    for (int r = cap_rows - 1; r >= (nf > 0 ? 1 : 0); r--) c->free_rows.push_back(r);  // back() = lowest free row
    if (nf > 0) {
        c->refcnt[0] = nf;
        rows_add_live(c, 0);
    }
    c->tables_dirty = true;
}

// ---- genealogy bookkeeping on the device (slamgpu_step_observe) <-> on the host ----
// The device-driven steps keep erow / live flags / reference counts / nf / fresh row in device memory (observe_book_kernel).
// Anything host-side that needs them (a host-made packet, download, flatten, peek, the landmark count) first pulls them back
// -- one synchronisation -- and the host is authoritative again; the next device-driven step pushes them.
// (All copies go through one pinned staging buffer.)
int book_staging(slamgpu_ctx *c) {
    if (c->book_host) return 0;
    const size_t words = sizeof(DevBook) / 4 + 2 * (size_t) c->B.cap_nf + (size_t) c->B.cap_rows;
    HIP_TRY(hipHostMalloc((void **) &c->book_host, 4 * words, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **) &c->book_dev, sizeof(DevBook)));
    HIP_TRY(hipMalloc((void **) &c->refcnt_dev, sizeof(int32_t) * (size_t) c->B.cap_rows));
    HIP_TRY(hipMalloc((void **) &c->take_dev, sizeof(int32_t) * (size_t) c->B.cap_rows));
    HIP_TRY(hipStreamCreateWithFlags(&c->obs_stream, hipStreamNonBlocking));
    for (int i = 0; i < kRing; i++) HIP_TRY(hipEventCreateWithFlags(&c->obs_ev[i], hipEventDisableTiming));
    return 0;
}

// the row lists follow from the reference counts (book_pull)
void book_rebuild(slamgpu_ctx *c) {
    std::fill(c->live_pos.begin(), c->live_pos.end(), -1);
    c->live_rows.clear();
    c->free_rows.clear();
    for (int r = c->B.cap_rows - 1; r >= 0; r--)
        if (c->refcnt[r] == 0) c->free_rows.push_back(r);  // back() = lowest free row
    for (int r = 0; r < c->B.cap_rows; r++)
        if (c->refcnt[r] > 0) rows_add_live(c, r);
    c->tables_dirty = true;
    c->book_on_device = false;
    std::fill(c->box_dirty.begin(), c->box_dirty.end(), 1);  // (which landmarks the device-driven steps wrote is not known here)
}

// compact contexts: the front end's state (kernels.h: FrontState), both copies holding the map, no landmark seen yet
int front_setup(slamgpu_ctx *c) {
    if (c->front_ready) return 0;
    if (c->map_n > kFrontLanes || c->map_n > kSmallObs - 1) return fail(SLAMGPU_ERR_CAPACITY, "map of %d landmarks in a compact context", c->map_n);
    if (!c->front_dev) {
        HIP_TRY(hipMalloc((void **) &c->front_dev, 2 * sizeof(FrontState)));
        HIP_TRY(hipHostMalloc((void **) &c->front_host, sizeof(FrontState), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **) &c->front_pkt_dev, sizeof(ObsPacket) + 4 * (6 * (size_t) kFrontLanes + kSmallRows)));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    FrontState *h = c->front_host;
    memset(h, 0, sizeof *h);
    h->hdr = FrontHdr{0, -1, 0, 0};
    for (int t = 0; t < kFrontLanes; t++) h->lm[t] = FrontLm{-1, 0};
    for (int q = 0; q < 2; q++) HIP_TRY(hipMemcpy(c->front_dev + q, h, sizeof *h, hipMemcpyHostToDevice));
    c->front_par = 0;
    c->front_ready = true;
    return 0;
}

int book_pull(slamgpu_ctx *c) {
    if (!c->book_on_device) return 0;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (c->B.compact) {
        FrontState *h = c->front_host;
        HIP_TRY(hipMemcpyAsync(h, c->front_dev + c->front_par, sizeof *h, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const int nf = h->hdr.nf;
        if (nf < 0 || nf > c->B.cap_nf) return fail(SLAMGPU_ERR_INVALID, "device bookkeeping corrupt: nf = %d", nf);
        std::fill(c->refcnt.begin(), c->refcnt.end(), 0);
        int seen = 0;
        for (int t = 0; t < c->map_n; t++) {
            const FrontLm &l = h->lm[t];
            if (l.idf < 0) continue;
            const int r = l.row & kRowMask;
            if (l.idf >= nf || r >= c->B.cap_rows) return fail(SLAMGPU_ERR_INVALID, "device bookkeeping corrupt: landmark %d -> feature %d, row %d", t, l.idf, r);
            c->erow[l.idf] = r;
            c->live_flag[l.idf] = (l.row & kRowLiveBit) ? 1 : 0;
            c->refcnt[r]++;
            seen++;
        }
        if (seen != nf) return fail(SLAMGPU_ERR_INVALID, "device bookkeeping corrupt: %d landmarks of the map seen, nf = %d", seen, nf);
        c->nf = nf;
        c->fresh_row = h->hdr.fresh_row;
        book_rebuild(c);
        const bool first_seen = (h->hdr.status & kStatusCapacity) && !(c->front_status & kStatusCapacity);
        c->front_status |= h->hdr.status;
        if (first_seen)  // (reported as an error once; the bit stays: slamgpu_step_status)
            return fail(SLAMGPU_ERR_CAPACITY, "the device front end dropped new landmarks: landmark capacity %d exceeded", c->B.cap_nf);
        return 0;
    }
    const size_t cn = (size_t) c->B.cap_nf, cr = (size_t) c->B.cap_rows;
    DevBook *hb = reinterpret_cast<DevBook *>(c->book_host);
    int32_t *h_erow = c->book_host + sizeof(DevBook) / 4, *h_live = h_erow + cn, *h_ref = h_live + cn;
    HIP_TRY(hipStreamSynchronize(c->obs_stream));  // (the last front-end kernel writes these tables)
    HIP_TRY(hipMemcpyAsync(hb, c->book_dev, sizeof(DevBook), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_erow, c->erow_dev, 4 * cn, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_live, c->live_dev, 4 * cn, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(h_ref, c->refcnt_dev, 4 * cr, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int nf = hb->nf;
    if (nf < 0 || nf > c->B.cap_nf) return fail(SLAMGPU_ERR_INVALID, "device bookkeeping corrupt: nf = %d", nf);
    std::copy(h_erow, h_erow + nf, c->erow.begin());
    std::copy(h_live, h_live + nf, c->live_flag.begin());
    std::copy(h_ref, h_ref + cr, c->refcnt.begin());
    c->nf = nf;
    c->fresh_row = hb->fresh_row;
    book_rebuild(c);
    const bool first_seen = (hb->status & kStatusCapacity) && !(c->front_status & kStatusCapacity);
    c->front_status |= hb->status;
    if (first_seen)
        return fail(SLAMGPU_ERR_CAPACITY, "the device front end dropped new landmarks: landmark capacity %d exceeded", c->B.cap_nf);
    return 0;
}

int book_push(slamgpu_ctx *c) {
    if (c->book_on_device) return 0;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (c->B.compact) {
        // the landmark -> feature table lives in the front end's state and nowhere else: every feature the context knows must
        // have come from it (a context is driven by one front end, the host's or the device's, from its first landmark on)
        if (int rc = front_setup(c)) return rc;
        FrontState *h = c->front_host;
        HIP_TRY(hipMemcpyAsync(h, c->front_dev + c->front_par, sizeof *h, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // (and nothing in flight still reads this copy)
        int seen = 0;
        for (int t = 0; t < c->map_n; t++) {
            FrontLm &l = h->lm[t];
            if (l.idf < 0) continue;
            if (l.idf >= c->nf) return fail(SLAMGPU_ERR_INVALID, "the device's landmark table knows feature %d, the context only %d", l.idf, c->nf);
            l.row = c->erow[l.idf] | (c->live_flag[l.idf] ? kRowLiveBit : 0);
            seen++;
        }
        if (seen != c->nf)
            return fail(SLAMGPU_ERR_INVALID, "slamgpu_step_observe on a compact context: %d of its %d landmarks came from host-made packets, "
                        "which the device's landmark table does not know", c->nf - seen, c->nf);
        h->hdr = FrontHdr{c->nf, c->fresh_row, c->front_status, 0};
        HIP_TRY(hipMemcpyAsync(c->front_dev + c->front_par, h, sizeof *h, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->book_on_device = true;
        return 0;
    }
    if (int rc = book_staging(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));  // (nothing in flight may still read the old tables or the staging buffer)
    const size_t cn = (size_t) c->B.cap_nf, cr = (size_t) c->B.cap_rows;
    DevBook *hb = reinterpret_cast<DevBook *>(c->book_host);
    int32_t *h_erow = c->book_host + sizeof(DevBook) / 4, *h_live = h_erow + cn, *h_ref = h_live + cn;
    *hb = DevBook{};
    hb->nf = c->nf;
    hb->fresh_row = c->fresh_row;
    hb->status = c->front_status;
    std::copy(c->erow.begin(), c->erow.begin() + cn, h_erow);
    std::copy(c->live_flag.begin(), c->live_flag.begin() + cn, h_live);
    std::copy(c->refcnt.begin(), c->refcnt.begin() + cr, h_ref);
    HIP_TRY(hipMemcpyAsync(c->book_dev, hb, sizeof(DevBook), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->erow_dev, h_erow, 4 * cn, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->live_dev, h_live, 4 * cn, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->refcnt_dev, h_ref, 4 * cr, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->book_on_device = true;
    return 0;
}

// device copies of the row tables for the kernels that run outside the update launch
int sync_tables(slamgpu_ctx *c) {
    if (int rc = book_pull(c)) return rc;
    c->B.erow = c->erow_dev;
    c->B.rows = c->rows_dev;
    c->B.lmk_live = c->live_dev;
    c->B.n_rows = (int32_t) c->live_rows.size();
    if (!c->tables_dirty) return 0;
    if (c->nf > 0) {
        HIP_TRY(hipMemcpyAsync(c->erow_dev, c->erow.data(), sizeof(int32_t) * (size_t) c->nf, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->live_dev, c->live_flag.data(), sizeof(int32_t) * (size_t) c->nf, hipMemcpyHostToDevice, c->stream));
    }
    if (!c->live_rows.empty())
        HIP_TRY(hipMemcpyAsync(c->rows_dev, c->live_rows.data(), sizeof(int32_t) * c->live_rows.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // pageable sources: the vectors may change right after
    c->tables_dirty = false;
    return 0;
}

// Make the particle set plain again (particle k in slot k of the live buffers) if the last update may have left a
// lazy gather: everything except the next update launch needs that.
int flush_stages(slamgpu_ctx *c);

int materialize(slamgpu_ctx *c) {
    if (int rc = flush_stages(c)) return rc;  // the plan of the last update decides whether anything is pending
    if (!c->maybe_pending) return 0;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_tables(c)) return rc;
    c->B.slot = c->slot;
    {
        Timed t(c, "gather");
        c->k->gather(c->stream, c->B, c->ws);
    }
    c->fresh_row = -1;  // rows may have been composed: no shortcut past the genealogy for the next update
    HIP_TRY(hipGetLastError());
    c->slot ^= 1;
    c->B.slot = c->slot;
    c->maybe_pending = false;
    return 0;
}

// Every landmark record into its particle's own slot (identity genealogy): what download hands out.
int flatten(slamgpu_ctx *c) {
    if (int rc = materialize(c)) return rc;
    if (c->nf == 0) return 0;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = sync_tables(c)) return rc;
    c->B.slot = c->slot;
    {
        Timed t(c, "flatten");
        c->k->flatten(c->stream, c->B, c->nf);
    }
    HIP_TRY(hipGetLastError());
    for (int j = 0; j < c->nf; j++) c->live_flag[j] ^= 1;  // every row's records now live in its other buffer
    c->pool_used = 0;  // every record is in its particle's own slot again
    rows_reset(c, c->nf);  // ... which is what genealogy row 0 says now, for every landmark
    return 0;
}

// Run, as launches of their own, whatever part of the last updates' resampling / estimate stages is still outstanding
// (normally the next update launches do it on the side): first the reduction of complete partials, then the plan of
// the last update (resample_kernel: Neff, decision, ancestors into keep[], partials) and its reduction.
int flush_stages(slamgpu_ctx *c) {
    if (!c->unreduced.has && !c->unplanned.has) return 0;
    HIP_TRY(hipSetDevice(c->cfg.device));
    c->B.slot = c->slot;
    if (c->unreduced.has) {
        Timed t(c, "finish");
        c->k->finish(c->stream, c->B, c->ws, c->unreduced.hist, c->unreduced.par);
        c->unreduced.has = false;
    }
    if (c->unplanned.has && c->dist)
        // the resampling stage of a distributed context needs every shard's totals: it only ever runs inside the next
        // update launch (slamgpu_dist_step / slamgpu_dist_settle)
        return fail(SLAMGPU_ERR_INVALID, "distributed context: call slamgpu_dist_settle on every shard (and all-gather) before reading results");
    if (c->unplanned.has) {
        ResampleArgs ra{};
        ra.nf = c->unplanned.nf;
        ra.do_resample = c->cfg.resample;
        ra.n_effective = c->cfg.n_effective;
        ra.logw = c->cfg.log_weights;
        c->ws.wpar = c->unplanned.par;
        if (c->ref_resample) {
            // the plan in the reference's own order of operations (one block), then the usual stage for what is left of it
            {
                Timed t(c, "resample_ref");
                c->k->resample_ref(c->stream, c->B, c->ws, rng_args(c, c->unplanned.step), ra);
            }
            HIP_TRY(hipGetLastError());  // (a refused launch must not leave `resample` a stale plan: ADVICE r5)
            ra.planned = 1;
        }
        {
            Timed t(c, "resample");
            c->k->resample(c->stream, c->B, c->ws, rng_args(c, c->unplanned.step), ra, UpdateArgs{});
        }
        // resample_kernel published the new live / pending state (and the ancestors) in the other slot
        c->keep_slot = c->slot ^ 1;
        c->slot ^= 1;
        c->B.slot = c->slot;
        c->maybe_pending = true;
        {
            Timed t(c, "finish");
            c->k->finish(c->stream, c->B, c->ws, c->unplanned.hist, c->unplanned.par);
        }
        c->unplanned.has = false;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int flush_predict(slamgpu_ctx *c) {
    if (c->pending.nsteps == 0) return 0;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = materialize(c)) return rc;
    compose_predicts(c->pending);
    {
        Timed t(c, "predict");
        c->k->predict(c->stream, c->B, c->pending, rng_args(c, 0));
    }
    HIP_TRY(hipGetLastError());
    // algorithmic bytes: xv + Pv read and written once per particle-predict (72 B)
    c->predict_bytes += 72.0 * c->cfg.n_particles * c->pending.nsteps;
    c->pending.nsteps = 0;
    c->est_fresh = false;
    return 0;
}

// the recorded history entries, behind everything enqueued, through the pinned mirror (a pageable destination cost the FIRST fetch of
// a process 8.4 ms for 104 KB: 3.9 us per observation step of a whole example_webmap run of the drop-in binary, round 5)
int history_to_host(slamgpu_ctx *c, std::vector<double> &h) {
    h.assign((size_t) kHistStride * (c->hist_n > 0 ? c->hist_n : 1), 0.0);
    if (c->hist_n > 0)
        HIP_TRY(hipMemcpyAsync(c->hist_host, c->hist_dev, sizeof(double) * kHistStride * (size_t) c->hist_n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->hist_n > 0) memcpy(h.data(), c->hist_host, sizeof(double) * kHistStride * (size_t) c->hist_n);
    return 0;
}

// A fetch that took fewer entries than were recorded (max_count < hist_n) keeps the rest: the unfetched tail moves to
// the front of the device-side history (all stages have been flushed: nothing points into it).
int keep_history_tail(slamgpu_ctx *c, const std::vector<double> &h, int taken) {
    const int left = c->hist_n - taken;
    if (left > 0 && taken > 0)
        HIP_TRY(hipMemcpy(c->hist_dev, h.data() + (size_t) kHistStride * taken, sizeof(double) * kHistStride * (size_t) left,
                          hipMemcpyHostToDevice));
    c->hist_n = left > 0 ? left : 0;
    return 0;
}

// The abort word of the persistent step loop is sticky, like the barrier error of the push collective: whoever synchronises with
// the device reports it (the kernel stores it into pinned host memory: no copy needed here).
int persist_check(slamgpu_ctx *c) {
    if (c->pstatus_host && __atomic_load_n(c->pstatus_host, __ATOMIC_ACQUIRE) != 0) {
        const volatile uint32_t *h = c->pstatus_host;
        return fail(SLAMGPU_ERR_BARRIER, "persistent step loop (slamgpu_run_observe): a workgroup waited too long at the in-launch barrier and launch %u of "
                                         "this context was abandoned after %u of its %u iterations (every workgroup had completed those; the iteration in "
                                         "flight is partially applied, so the state is undefined and everything handed over since is void: recreate the "
                                         "context and replay from there; slamgpu_persist_status returns these counts; SLAMGPU_NO_PERSIST=1 selects the "
                                         "per-step loop)", (unsigned) h[2], (unsigned) h[1], (unsigned) h[3]);
    }
    return 0;
}

int check_ctx(slamgpu_ctx *c) {
    if (!c) return fail(SLAMGPU_ERR_INVALID, "null context");
    return 0;
}

// need_set: the caller is going to touch the particle buffers (not only the Ctrl words)
int read_ctrl(slamgpu_ctx *c, bool need_set = false) {
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = flush_predict(c)) return rc;
    if (need_set)
        if (int rc = materialize(c)) return rc;
    if (int rc = flush_stages(c)) return rc;
    HIP_TRY(hipMemcpyAsync(c->ctrl_host, c->B.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return persist_check(c);
}

}  // namespace

extern "C" {

const char *slamgpu_last_error(void) { return g_err; }
int slamgpu_abi_version(void) { return SLAMGPU_ABI_VERSION; }

int slamgpu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int slamgpu_create(const slamgpu_config *cfg, slamgpu_ctx **out) {
    if (!cfg || !out) return fail(SLAMGPU_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(slamgpu_config))
        return fail(SLAMGPU_ERR_INVALID, "slamgpu_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size, sizeof(slamgpu_config));
    if (cfg->n_particles <= 0 || cfg->max_landmarks < 0) return fail(SLAMGPU_ERR_INVALID, "bad sizes");
    if (cfg->method != SLAMGPU_FASTSLAM1 && cfg->method != SLAMGPU_FASTSLAM2)
        return fail(SLAMGPU_ERR_INVALID, "method must be SLAMGPU_FASTSLAM1 or SLAMGPU_FASTSLAM2 (EKF1 runs on the host)");
    if (cfg->log_weights && cfg->n_particles_global > 0 && cfg->n_particles_global != cfg->n_particles)
        return fail(SLAMGPU_ERR_INVALID, "log_weights is not available for shards (n_particles_global != n_particles)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SLAMGPU_ERR_NO_DEVICE, "no HIP device: libslamgpu has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(SLAMGPU_ERR_INVALID, "device %d out of range (%d devices)", cfg->device, ndev);
    const int n = cfg->n_particles;
    const int ncap = ((n + kBlock - 1) / kBlock) * kBlock;
    if (ncap / kBlock > kMaxScanBlocks)
        return fail(SLAMGPU_ERR_INVALID, "n_particles %d exceeds %d per context (shard across contexts)", n, kMaxScanBlocks * kBlock);
    HIP_TRY(hipSetDevice(cfg->device));

    slamgpu_ctx *c = new slamgpu_ctx();
    c->cfg = *cfg;
    if (c->cfg.n_particles_global <= 0) c->cfg.n_particles_global = c->cfg.n_particles;
    c->k = cfg->math_mode == SLAMGPU_MATH_FAST ? kernels_fast() : kernels_strict();
    if (const char *e = getenv("SLAMGPU_SCAN_MIN_BLOCKS")) c->scan_min_blocks = atoi(e);  // diagnostic
    c->consolidate = getenv("SLAMGPU_NO_CONSOLIDATE") == nullptr;                          // diagnostic / tests
    c->ref_resample = cfg->math_mode != SLAMGPU_MATH_FAST && cfg->rng_mode == SLAMGPU_RNG_TAPE && n <= kRefResampleMax && !cfg->log_weights &&
                      c->cfg.n_particles_global == c->cfg.n_particles && !(cfg->flags & SLAMGPU_FLAG_NO_REFERENCE_RESAMPLE) &&
                      getenv("SLAMGPU_NO_REF_RESAMPLE") == nullptr;
    if (c->ref_resample) {  // (its one block keeps 2 n floats in dynamic LDS: ask the device rather than assume gfx950's 160 KB)
        int lds_max = 0;
        if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device) != hipSuccess || (size_t) lds_max < sizeof(float) * 2 * (size_t) n + 4096)
            c->ref_resample = false;
    }
    c->persist_ok = getenv("SLAMGPU_NO_PERSIST") == nullptr;                               // diagnostic / tests: slamgpu_run_observe as a loop of launches
    if (const char *e = getenv("SLAMGPU_CONSOLIDATE_ABOVE")) c->consolidate_above = atoi(e);
    if (const char *e = getenv("SLAMGPU_PLAIN_ROWS_TARGET")) c->plain_rows_target = atoi(e);
    const bool want_stamps = getenv("SLAMGPU_STAMPS") != nullptr;                         // diagnostic
    const int cap_nf = cfg->max_landmarks > 0 ? cfg->max_landmarks : 1;
    c->B.n = n;
    c->B.ncap = ncap;
    c->B.cap_nf = cap_nf;
#define CTX_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            int rc_ = fail(e_ == hipErrorOutOfMemory ? SLAMGPU_ERR_ALLOC : SLAMGPU_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
            slamgpu_destroy(c);                                                                        \
            return rc_;                                                                                \
        }                                                                                              \
    } while (0)
    if (cfg->external_stream) {
        c->stream = reinterpret_cast<hipStream_t>(cfg->external_stream);
        c->own_stream = false;
    } else {
        CTX_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    }
    CTX_TRY(hipMalloc((void **) &c->plan_dev, sizeof(ShardPlan)));
    CTX_TRY(hipHostMalloc((void **) &c->plan_host, sizeof(ShardPlan), hipHostMallocDefault));
    CTX_TRY(hipHostMalloc((void **) &c->plan_seq_host, sizeof(uint32_t), hipHostMallocDefault));
    *c->plan_seq_host = 0;
    const size_t S = (size_t) ncap;
    for (int b = 0; b < 2; b++) {
        CTX_TRY(hipMalloc((void **) &c->B.poseA[b], sizeof(float4) * S));
        CTX_TRY(hipMalloc((void **) &c->B.poseB[b], sizeof(float4) * S));
        CTX_TRY(hipMalloc((void **) &c->B.poseC[b], sizeof(float2) * S));
        CTX_TRY(hipMalloc((void **) &c->B.lmkA[b], sizeof(float4) * S * cap_nf));
        CTX_TRY(hipMalloc((void **) &c->B.lmkB[b], sizeof(float) * S * cap_nf));
        CTX_TRY(hipMemsetAsync(c->B.poseA[b], 0, sizeof(float4) * S, c->stream));
        CTX_TRY(hipMemsetAsync(c->B.poseB[b], 0, sizeof(float4) * S, c->stream));
        CTX_TRY(hipMemsetAsync(c->B.poseC[b], 0, sizeof(float2) * S, c->stream));
        CTX_TRY(hipMemsetAsync(c->B.lmkA[b], 0, sizeof(float4) * S * cap_nf, c->stream));
        CTX_TRY(hipMemsetAsync(c->B.lmkB[b], 0, sizeof(float) * S * cap_nf, c->stream));
        // genealogy rows: plain [cap_rows][S], or compact [ceil(cap_rows / 4)][S][4] (kernels.h: Buffers::gen)
        CTX_TRY(hipMalloc((void **) &c->B.gen[b], sizeof(int32_t) * S * (size_t) ((cap_nf + 1 + 3) / 4 * 4)));
    }
    c->B.slot = 0;
    c->B.cap_rows = cap_nf + 1;  // at most one row per landmark, plus the one a step opens while the old ones are still read
    c->B.compact = c->B.cap_rows <= kSmallRows ? 1 : 0;
    // maps of 40 .. kMidLandmarks landmarks (round 5): compact as well -- kSmallRows genealogy rows, packets in the kernel
    // arguments -- unless the context is going to make its observations on the device (that front end writes plain-row packets)
    if (!c->B.compact && cap_nf <= kMidLandmarks && !(cfg->flags & SLAMGPU_FLAG_DEVICE_OBSERVE) && c->cfg.n_particles_global == c->cfg.n_particles &&
        getenv("SLAMGPU_NO_MID_COMPACT") == nullptr) {
        c->B.compact = 1;
        c->B.cap_rows = kSmallRows;
        c->mid_compact = true;
    }
    if (getenv("SLAMGPU_NO_COMPACT") || (cfg->flags & SLAMGPU_FLAG_PARTICLE_MAPS)) {  // plain rows for a small map (diagnostic; per-particle association)
        c->B.compact = 0;
        c->B.cap_rows = cap_nf + 1;
        c->mid_compact = false;
    }
    c->erow.assign((size_t) cap_nf, 0);
    c->live_flag.assign((size_t) cap_nf, 0);
    c->seen_step.assign((size_t) cap_nf, 0);
    c->box_dirty.assign((size_t) cap_nf, 1);
    CTX_TRY(hipMalloc((void **) &c->live_dev, sizeof(int32_t) * (size_t) cap_nf));
    c->refcnt.assign((size_t) c->B.cap_rows, 0);
    c->live_pos.assign((size_t) c->B.cap_rows, -1);
    rows_reset(c, 0);
    CTX_TRY(hipMalloc((void **) &c->erow_dev, sizeof(int32_t) * (size_t) cap_nf));
    CTX_TRY(hipMalloc((void **) &c->rows_dev, sizeof(int32_t) * (size_t) c->B.cap_rows));
    CTX_TRY(hipMalloc((void **) &c->B.ctrl, sizeof(Ctrl)));
    CTX_TRY(hipHostMalloc((void **) &c->ctrl_host, sizeof(Ctrl), hipHostMallocDefault));
    memset(c->ctrl_host, 0, sizeof(Ctrl));
    c->ctrl_host->inv_n = 1.0f / (float) n_global(c);  // core.cpp:745
    if (cfg->log_weights) c->ctrl_host->inv_n = logf(c->ctrl_host->inv_n);
    CTX_TRY(hipMemcpyAsync(c->B.ctrl, c->ctrl_host, sizeof(Ctrl), hipMemcpyHostToDevice, c->stream));
    // weight scratch
    c->ws.nblocks = ncap / kBlock;
    for (int b = 0; b < 2; b++) {
        CTX_TRY(hipMalloc((void **) &c->ws.lcum[b], sizeof(float) * S));
        CTX_TRY(hipMalloc((void **) &c->ws.blk_w[b], sizeof(float) * 3 * (size_t) c->ws.nblocks));  // [w | w2 (| max log-weight)] contiguous
        CTX_TRY(hipMalloc((void **) &c->ws.est_part[b], sizeof(double) * (4 * (size_t) c->ws.nblocks + 2)));  // + Neff, resampled
        CTX_TRY(hipMalloc((void **) &c->ws.scan[b], sizeof(double) * ((size_t) c->ws.nblocks + 4)));
    }
    for (int b = 0; b < 2; b++) {
        CTX_TRY(hipMalloc((void **) &c->ws.keep[b], sizeof(int32_t) * S));
        CTX_TRY(hipMemsetAsync(c->ws.keep[b], 0, sizeof(int32_t) * S, c->stream));
    }
    CTX_TRY(hipMalloc((void **) &c->hist_dev, sizeof(double) * kHistStride * (size_t) kHistCap));
    CTX_TRY(hipHostMalloc((void **) &c->hist_host, sizeof(double) * kHistStride * (size_t) kHistCap, hipHostMallocDefault));
    if (want_stamps) {
        CTX_TRY(hipMalloc((void **) &c->stamps_dev, sizeof(unsigned long long) * kStampSlots * (size_t) c->ws.nblocks));
        CTX_TRY(hipMemsetAsync(c->stamps_dev, 0, sizeof(unsigned long long) * kStampSlots * (size_t) c->ws.nblocks, c->stream));
    }
    // big-packet ring: header + idf[cap] + zf[2cap] + zn[2cap]
    c->pkt_bytes = ((sizeof(ObsPacket) + sizeof(int32_t) * cap_nf + sizeof(float) * 4 * cap_nf + sizeof(int32_t) * (2 * (size_t) cap_nf + 1)) + 255) / 256 * 256;
    CTX_TRY(hipHostMalloc((void **) &c->pkt_host, c->pkt_bytes * kRing, hipHostMallocDefault));
    CTX_TRY(hipMalloc((void **) &c->pkt_dev, c->pkt_bytes * kRing));
    for (int i = 0; i < kRing; i++) CTX_TRY(hipEventCreateWithFlags(&c->pkt_ev[i], hipEventDisableTiming));
    if (cfg->rng_mode == SLAMGPU_RNG_TAPE) {
        const size_t tape_floats = 3 * S + (size_t) n_global(c);
        CTX_TRY(hipHostMalloc((void **) &c->tape_host, sizeof(float) * tape_floats, hipHostMallocDefault));
        CTX_TRY(hipMalloc((void **) &c->normals_dev, sizeof(float) * 3 * S));
        for (int b = 0; b < 2; b++) CTX_TRY(hipMalloc((void **) &c->strata_dev[b], sizeof(float) * (size_t) n_global(c)));
    }
    // initial particle set: Particle() then w = 1/N (ParticleSLAMWrapper.cpp:14-25)
    {
        std::vector<float4> a(S, make_float4(0.f, 0.f, 0.f, 0.f));
        const float uw = cfg->log_weights ? c->ctrl_host->inv_n : (float) (1.0 / (float) n_global(c));
        for (int i = 0; i < n; i++) a[i].w = uw;
        CTX_TRY(hipMemcpyAsync(c->B.poseA[0], a.data(), sizeof(float4) * S, hipMemcpyHostToDevice, c->stream));
        // one launch out of the library's code object, so that the runtime loads it HERE and not inside the caller's first step
        // (the first launch of a process pays for the module: ~10 ms, seven microseconds per observation step of a whole
        // example_webmap run of the drop-in binary, round 5): genealogy row 0 = "own slot", which is what it means before any landmark
        c->k->identity(c->stream, c->B, 0, 0);
        c->k->identity(c->stream, c->B, 1, 0);
        CTX_TRY(hipGetLastError());
        CTX_TRY(hipStreamSynchronize(c->stream));
    }
#undef CTX_TRY
    *out = c;
    return 0;
}

void slamgpu_destroy(slamgpu_ctx *c) {
    if (!c) return;
    (void) hipSetDevice(c->cfg.device);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    drain_stats(c);
    for (auto e : c->ev_pool) (void) hipEventDestroy(e);
    for (int b = 0; b < 2; b++) {
        if (c->B.poseA[b]) (void) hipFree(c->B.poseA[b]);
        if (c->B.poseB[b]) (void) hipFree(c->B.poseB[b]);
        if (c->B.poseC[b]) (void) hipFree(c->B.poseC[b]);
        if (c->B.lmkA[b]) (void) hipFree(c->B.lmkA[b]);
        if (c->B.lmkB[b]) (void) hipFree(c->B.lmkB[b]);
        if (c->B.gen[b]) (void) hipFree(c->B.gen[b]);
        if (b == 0 && c->B.poolA) (void) hipFree(c->B.poolA);
        if (b == 0 && c->B.poolB) (void) hipFree(c->B.poolB);
    }
    if (c->B.ctrl) (void) hipFree(c->B.ctrl);
    if (c->ctrl_host) (void) hipHostFree(c->ctrl_host);
    if (!c->own_totals) c->ws.blk_w[0] = c->own_blk_w;
    for (int b = 0; b < 2; b++) {
        if (c->ws.lcum[b]) (void) hipFree(c->ws.lcum[b]);
        if (c->ws.blk_w[b]) (void) hipFree(c->ws.blk_w[b]);
        if (c->ws.est_part[b]) (void) hipFree(c->ws.est_part[b]);
        if (c->ws.scan[b]) (void) hipFree(c->ws.scan[b]);
    }
    for (int b = 0; b < 2; b++)
        if (c->ws.keep[b]) (void) hipFree(c->ws.keep[b]);
    if (c->hist_dev) (void) hipFree(c->hist_dev);
    if (c->hist_host) (void) hipHostFree(c->hist_host);
    if (c->stamps_dev) (void) hipFree(c->stamps_dev);
    if (c->peek_dev) (void) hipFree(c->peek_dev);
    if (c->retired_dev) (void) hipFree(c->retired_dev);
    if (c->vote_w_dev) (void) hipFree(c->vote_w_dev);
    if (c->assoc_z_dev) (void) hipFree(c->assoc_z_dev);
    if (c->assoc_votes_dev) (void) hipFree(c->assoc_votes_dev);
    if (c->pp_lab_dev) (void) hipFree(c->pp_lab_dev);
    if (c->pp_obs_dev) (void) hipFree(c->pp_obs_dev);
    if (c->pp_z_dev) (void) hipFree(c->pp_z_dev);
    if (c->pp_tab_dev) (void) hipFree(c->pp_tab_dev);
    if (c->pp_wf_dev) (void) hipFree(c->pp_wf_dev);
    if (c->pp_any_dev) (void) hipFree(c->pp_any_dev);
    for (void *p_ : {(void *) c->box_dev, (void *) c->assoc_ids_dev, (void *) c->cell_start_dev, (void *) c->cell_fill_dev, (void *) c->items_dev,
                     (void *) c->geom_dev})
        if (p_) (void) hipFree(p_);
    if (c->pq_host) (void) hipHostFree(c->pq_host);
    for (int b = 0; b < slamgpu_ctx::kPqBufs; b++)
        if (c->pq_kev[b]) (void) hipEventDestroy(c->pq_kev[b]);
    if (c->psync_dev) (void) hipFree(c->psync_dev);
    if (c->ppk_dev) (void) hipFree(c->ppk_dev);
    if (c->pring_dev) (void) hipFree(c->pring_dev);
    if (c->pdraw_dev) (void) hipFree(c->pdraw_dev);
    if (c->pstatus_host) (void) hipHostFree(c->pstatus_host);
    if (c->book_dev) (void) hipFree(c->book_dev);
    if (c->refcnt_dev) (void) hipFree(c->refcnt_dev);
    if (c->take_dev) (void) hipFree(c->take_dev);
    if (c->book_host) (void) hipHostFree(c->book_host);
    for (int i = 0; i < kRing; i++)
        if (c->obs_ev[i]) (void) hipEventDestroy(c->obs_ev[i]);
    if (c->obs_stream) {
        (void) hipStreamSynchronize(c->obs_stream);
        (void) hipStreamDestroy(c->obs_stream);
    }
    if (c->comm && rccl()) (void) rccl()->CommDestroy((ncclComm_t) c->comm);
    for (void *p : c->ipc_opened) (void) hipIpcCloseMemHandle(p);
    if (c->peers_dev) (void) hipFree(c->peers_dev);
    for (int b = 0; b < 2; b++)
        if (c->gtot_dev[b]) (void) hipFree(c->gtot_dev[b]);
    if (c->flags_dev) (void) hipFree(c->flags_dev);
    if (c->front_dev) (void) hipFree(c->front_dev);
    if (c->front_host) (void) hipHostFree(c->front_host);
    if (c->front_pkt_dev) (void) hipFree(c->front_pkt_dev);
    if (c->map_dev) (void) hipFree(c->map_dev);
    if (c->obs_r_dev) (void) hipFree(c->obs_r_dev);
    if (c->table_dev) (void) hipFree(c->table_dev);
    if (c->obs_out_dev) (void) hipFree(c->obs_out_dev);
    if (c->erow_dev) (void) hipFree(c->erow_dev);
    if (c->live_dev) (void) hipFree(c->live_dev);
    if (c->rows_dev) (void) hipFree(c->rows_dev);
    if (c->pkt_host) (void) hipHostFree(c->pkt_host);
    if (c->pkt_dev) (void) hipFree(c->pkt_dev);
    for (int i = 0; i < kRing; i++)
        if (c->pkt_ev[i]) (void) hipEventDestroy(c->pkt_ev[i]);
    if (c->timer_a) (void) hipEventDestroy(c->timer_a);
    if (c->timer_b) (void) hipEventDestroy(c->timer_b);
    if (c->tape_host) (void) hipHostFree(c->tape_host);
    if (c->normals_dev) (void) hipFree(c->normals_dev);
    for (int b = 0; b < 2; b++)
        if (c->strata_dev[b]) (void) hipFree(c->strata_dev[b]);
    if (c->plan_dev) (void) hipFree(c->plan_dev);
    if (c->plan_host) (void) hipHostFree(c->plan_host);
    if (c->plan_seq_host) (void) hipHostFree(c->plan_seq_host);
    if (c->stream && c->own_stream) (void) hipStreamDestroy(c->stream);
    delete c;
}

int slamgpu_predict(slamgpu_ctx *c, float V, float G, const float Q[4], float dt, float phi_true, const float *noise2) {
    if (int rc = check_ctx(c)) return rc;
    if (!Q) return fail(SLAMGPU_ERR_INVALID, "Q is null");
    const bool noise = c->cfg.add_predict_noise != 0;
    const bool tape_noise = noise && c->cfg.rng_mode == SLAMGPU_RNG_TAPE;
    if (tape_noise && !noise2) return fail(SLAMGPU_ERR_INVALID, "TAPE mode with add_predict_noise needs noise2[2N]");
    PredictArgs &P = c->pending;
    // parameters must be uniform across a fused launch; flush when they change or the queue is full
    if (P.nsteps > 0 && (P.dt != dt || memcmp(P.Q, Q, sizeof P.Q) != 0 || P.nsteps == kMaxFusedPredict || tape_noise))
        if (int rc = flush_predict(c)) return rc;
    if (P.nsteps == 0) {
        P.method = c->cfg.method;
        P.use_heading = c->cfg.use_heading;
        P.add_noise = noise ? 1 : 0;
        memcpy(P.Q, Q, sizeof P.Q);
        P.dt = dt;
        P.wheel_base = c->cfg.wheel_base;
        P.sigma_phi = c->cfg.sigma_phi;
    }
    c->ctl_step++;
    PredictStep &s = P.steps[P.nsteps++];
    s.V = V;
    s.G = G;
    s.phi_true = phi_true;
    s.step = c->ctl_step;
    s.sinG = sinf(G);
    s.cosG = cosf(G);
    s.sinGw = sinf(G / c->cfg.wheel_base);
    s.pad = 0;
    c->est_fresh = false;
    c->shard_est_fresh = false;
    if (tape_noise) {
        HIP_TRY(hipSetDevice(c->cfg.device));
        HIP_TRY(hipStreamSynchronize(c->stream));  // tape_host is single-buffered (parity mode only)
        const int n = c->B.n, S = c->B.ncap;
        for (int i = 0; i < n; i++) {
            c->tape_host[i] = noise2[2 * i];
            c->tape_host[S + i] = noise2[2 * i + 1];
        }
        HIP_TRY(hipMemcpyAsync(c->normals_dev, c->tape_host, sizeof(float) * 2 * (size_t) S, hipMemcpyHostToDevice, c->stream));
        return flush_predict(c);
    }
    return 0;
}

}  // extern "C"

namespace {

// FastSLAM{1,2}::update: the per-particle stage, then (single-context case) the resampling stage
// The launch of an update whose packet is in place (host-made or device-made): queued predicts, the particle-noise tape,
// the stage bookkeeping.  n_new: new landmarks if the host knows (device packets: -1); n_rows: live genealogy rows a pending
// gather composes (device packets: -1: the launch carries the copy roles of the largest geometry).
int issue_update(slamgpu_ctx *c, UpdateArgs &U, int n_new, int n_rows, bool need_normals, const float *normals, const float *strata, bool sharded) {
    const bool tape = c->cfg.rng_mode == SLAMGPU_RNG_TAPE;
    const int n = n_new;
    // reference-order resampling (strict build, TAPE draws, small contexts): the stage of the previous update never rides in this
    // launch: it runs now, as launches of its own, through resample_ref_kernel
    if (c->ref_resample && !sharded && !c->dist && c->unplanned.has)
        if (int rc = flush_stages(c)) return rc;
    // pending predicts ride inside the update launch (state stays in registers) unless their noise is a host tape
    PredictArgs PA{};
    if (c->pending.nsteps > 0) {
        compose_predicts(c->pending);
        PA = c->pending;
        c->predict_bytes += 72.0 * c->cfg.n_particles * c->pending.nsteps;
        c->pending.nsteps = 0;
    }

    if (tape) {
        HIP_TRY(hipStreamSynchronize(c->stream));  // single-buffered tape staging (parity mode)
        const int N = c->B.n, S = c->B.ncap;
        if (need_normals) {
            for (int i = 0; i < N; i++) {
                c->tape_host[i] = normals[3 * i];
                c->tape_host[S + i] = normals[3 * i + 1];
                c->tape_host[2 * S + i] = normals[3 * i + 2];
            }
            HIP_TRY(hipMemcpyAsync(c->normals_dev, c->tape_host, sizeof(float) * 3 * (size_t) S, hipMemcpyHostToDevice, c->stream));
        }
        float *sh = c->tape_host + 3 * (size_t) S;
        memcpy(sh, strata, sizeof(float) * (size_t) n_global(c));
        HIP_TRY(hipMemcpyAsync(c->strata_dev[c->obs_step & 1], sh, sizeof(float) * (size_t) n_global(c), hipMemcpyHostToDevice, c->stream));
    }

    RngArgs rng = rng_args(c, c->obs_step);
    if (c->dist) {
        rng.step = c->obs_step - c->rng_skew;  // (device noise only: the tape buffers above do not apply)
        rng.prev_step = c->unplanned.step;
    }
    if (sharded && c->unplanned.has)
        if (int rc = flush_stages(c)) return rc;  // (a context is driven either way, not both; be safe)
    c->B.slot = c->slot;
    c->ws.wpar = sharded ? 0 : (int) (c->obs_step & 1);
    U.lazy = 1;
    U.arrivals = c->dist ? 2 : (sharded ? 1 : 0);
    if (c->dist) {
        c->B.gtot[0] = c->gtot_dev[0];
        c->B.gtot[1] = c->gtot_dev[1];
        c->dist_clean = false;
        U.push_totals = (c->dist_push || c->dist_fold) ? 1 : 0;
        U.count_remote = c->count_remote ? 1 : 0;
        U.fold_seq = c->dist_fold ? ++c->flag_seq : 0;
        U.fold_spins = 1u << 20;
    }
    // a pending gather's genealogy composition: small packets: by the compute threads themselves; device packets: by copy
    // roles (one role = 256 particles x rows_per_role live rows; at most ~4 roles per particle tile: every role block
    // redoes the plan's scan and search)
    U.rows_per_role = std::max(16, ((std::max(n_rows, 0) + 3) / 4 + 3) / 4 * 4);
    const int roles = !U.big ? 0 : (n_rows < 0 ? c->ws.nblocks * 4 : c->ws.nblocks * ((n_rows + U.rows_per_role - 1) / U.rows_per_role));
    U.copy_lo = 0;
    U.copy_hi = roles;
    // the resampling stage of the previous update rides in this launch unless something already ran it
    U.plan_inline = (!sharded && c->unplanned.has) ? 1 : 0;
    U.scan_global = (U.plan_inline && c->scan_ready) ? 1 : 0;
    U.do_resample = c->cfg.resample;
    U.n_effective = c->cfg.n_effective;
    U.logw = c->cfg.log_weights;
    U.stamps = c->stamps_dev;
    U.finalize = c->unreduced.has ? 1 : 0;  // (sharded: this shard's partials of the previous step, shard_finalize_kernel)
    U.finalize_hist = c->unreduced.hist;
    U.finalize_par = c->unreduced.par;
    if (c->collect) {
        // persistent loop: this iteration rides in the queue of ONE launch (slamgpu_run_observe); everything the host keeps
        // track of moves on exactly as if the launch had been made
        PersistStep q{};
        q.PA = PA;
        q.fx = U.front.x;
        q.fy = U.front.y;
        q.fphi = U.front.phi;
        q.fstep = U.front.step;
        q.rng_step = rng.step;
        q.rng_prev_step = rng.prev_step;
        q.wpar = c->ws.wpar;
        q.plan_inline = U.plan_inline;
        q.finalize = U.finalize;
        q.finalize_par = U.finalize_par;
        q.finalize_hist = U.finalize_hist;
        if (!c->collect->have_first) {
            c->collect->have_first = true;
            c->collect->B = c->B;
            c->collect->U = U;
            c->collect->rng = rng;
            c->collect->ws = c->ws;
        }
        c->collect->steps.push_back(q);
    } else {
        Timed t(c, c->cfg.method == SLAMGPU_FASTSLAM2 ? "fs2_update" : "fs1_update");
        if (c->pp_launch) c->k->update_particle(c->stream, c->B, PA, U, rng, c->ws, *c->pp_launch);
        else c->k->update(c->stream, c->B, PA, U, rng, c->ws);
    }
    HIP_TRY(hipGetLastError());
    c->slot ^= 1;   // ... and where it left the set (Ctrl.live / pend of the other slot)
    c->B.slot = c->slot;
    c->maybe_pending = false;  // whatever gather was pending, this launch performed it
    if (n >= 0) c->nf += n;
    if (sharded) {
        // the resampling stage is driven by the caller through slamgpu_shard_* (needs collectives)
        c->unreduced.has = false;  // reduced by the helper block of this launch
        c->est_fresh = false;
        c->shard_est_fresh = false;
        return 0;
    }
    // stage bookkeeping: the helper block reduced `unreduced`; the inline plan left the partials of `unplanned`;
    // this update's own resampling stage is now the outstanding one
    c->unreduced.has = false;
    if (U.plan_inline) c->unreduced = c->unplanned;
    c->unplanned.has = true;
    c->unplanned.par = c->ws.wpar;
    c->unplanned.step = c->obs_step - c->rng_skew;
    c->unplanned.nf = c->nf;
    c->unplanned.hist = c->hist_n < kHistCap ? c->hist_dev + kHistStride * (size_t) c->hist_n : nullptr;
    c->est_fresh = c->unplanned.hist != nullptr;
    // large contexts: one block prepares the prefix of this step's block totals for the next launch, instead of every
    // block of that launch redoing it (a second, tiny launch; negligible at these sizes)
    c->scan_ready = false;
    if (c->ws.nblocks > c->scan_min_blocks && !c->dist) {
        Timed t(c, "scan");
        c->k->scan(c->stream, c->ws, c->cfg.log_weights);
        c->scan_ready = true;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// A compact context on a map of more than 39 landmarks (mid_compact) meets a step its layout cannot take -- more re-observed or
// new landmarks than a kernel-argument packet holds, or no genealogy row left: it becomes a plain-row context for good.  The
// records stay where they are (both layouts index the same slots); the genealogy rows are rewritten from the interleaved form
// (four rows to a 16-byte chunk) into whole rows, and the host's row tables grow to one row per landmark.
int demote_to_plain(slamgpu_ctx *c) {
    if (!c->mid_compact) return fail(SLAMGPU_ERR_INVALID, "not a compact context of a mid-size map");
    if (int rc = materialize(c)) return rc;  // (no pending gather: the set is plain in the live buffers)
    HIP_TRY(hipSetDevice(c->cfg.device));
    const size_t S = (size_t) c->B.ncap, words = S * (size_t) kSmallRows;
    int32_t *tmp = nullptr;
    HIP_TRY(hipMalloc((void **) &tmp, sizeof(int32_t) * words));
    hipError_t er = hipSuccess;  // (no early return between the allocation and its release: ADVICE r5)
    for (int b = 0; b < 2 && er == hipSuccess; b++) {
        er = hipMemcpyAsync(tmp, c->B.gen[b], sizeof(int32_t) * words, hipMemcpyDeviceToDevice, c->stream);
        if (er != hipSuccess) break;
        c->k->decompact(c->stream, tmp, c->B.gen[b], c->B.ncap, kSmallRows);
        er = hipGetLastError();
    }
    const hipError_t er2 = hipStreamSynchronize(c->stream);
    (void) hipFree(tmp);
    HIP_TRY(er);
    HIP_TRY(er2);
    const int old_rows = c->B.cap_rows, new_rows = c->B.cap_nf + 1;
    c->B.compact = 0;
    c->B.cap_rows = new_rows;
    c->mid_compact = false;
    c->refcnt.resize((size_t) new_rows, 0);
    c->live_pos.resize((size_t) new_rows, -1);
    for (int r = old_rows; r < new_rows; r++) c->free_rows.push_back(r);
    std::sort(c->free_rows.begin(), c->free_rows.end(), std::greater<int32_t>());  // back() = lowest free row
    (void) hipFree(c->rows_dev);
    c->rows_dev = nullptr;
    HIP_TRY(hipMalloc((void **) &c->rows_dev, sizeof(int32_t) * (size_t) new_rows));
    c->fresh_row = -1;
    c->tables_dirty = true;
    return 0;
}

int do_update(slamgpu_ctx *c, const float *zf, const int32_t *idf, int32_t m, const float *zn, int32_t n,
              const float R[4], const float *normals, const float *strata, bool sharded) {
    if (m < 0 || n < 0 || !R || (m > 0 && (!zf || !idf)) || (n > 0 && !zn)) return fail(SLAMGPU_ERR_INVALID, "bad observation packet");
    if (int rc = book_pull(c)) return rc;  // (after device-driven steps the bookkeeping comes back first)
    if (m > c->nf) return fail(SLAMGPU_ERR_INVALID, "m=%d re-observed landmarks but only %d known", m, c->nf);
    if (c->nf + n > c->B.cap_nf) return fail(SLAMGPU_ERR_CAPACITY, "landmark capacity exceeded: %d + %d > %d", c->nf, n, c->B.cap_nf);
    for (int k = 0; k < m; k++)
        if (idf[k] < 0 || idf[k] >= c->nf) return fail(SLAMGPU_ERR_INVALID, "idf[%d]=%d out of range [0,%d)", k, idf[k], c->nf);
    const bool tape = c->cfg.rng_mode == SLAMGPU_RNG_TAPE;
    const bool need_normals = c->cfg.method == SLAMGPU_FASTSLAM2 && (m > 0 || n > 0);
    if (tape && ((need_normals && !normals) || !strata)) return fail(SLAMGPU_ERR_INVALID, "TAPE mode needs normals[3N] and strata[N]");
    HIP_TRY(hipSetDevice(c->cfg.device));
    // a compact context's packet rides in the kernel arguments: at most kSmallObs re-observed and kSmallObs new landmarks, and its
    // genealogy has kSmallRows rows.  Maps of up to 39 landmarks cannot exceed either; a mid-size map that does goes to plain rows
    if (c->mid_compact && (m > kSmallObs || n > kSmallObs || c->free_rows.size() < 2))
        if (int rc = demote_to_plain(c)) return rc;
    c->obs_step++;

    // genealogy rows: the landmarks this update writes move to a row it opens; the rows they leave may become unused
    for (int k = 0; k < m; k++) {
        if (c->seen_step[idf[k]] == c->obs_step) return fail(SLAMGPU_ERR_INVALID, "landmark %d re-observed twice in one update", idf[k]);
        c->seen_step[idf[k]] = c->obs_step;
    }
    // Row consolidation (compact contexts): when more than kConsolidateAbove genealogy rows are alive, the landmarks this
    // update does NOT observe are rewritten by the launch, unchanged, into the particles' own slots and join the row it opens
    // (the kernel treats them like re-observed landmarks with a no-op update: UpdateArgs::n_cons).  A landmark goes out of
    // view every ~60 steps on example_webmap, so after one consolidation the map lives in one or two rows for hundreds of
    // steps, and a resample composes one 16-byte chunk per particle instead of eight.  Values never change: results are bit
    // for bit those of a run without it (SLAMGPU_NO_CONSOLIDATE=1, tests/test_gpu_parity.py).
    std::vector<int32_t> cons;
    if (c->B.compact && !c->mid_compact && !sharded && c->consolidate && (int) c->live_rows.size() > c->consolidate_above) {
        for (int j = 0; j < c->nf && m + (int) cons.size() < kSmallObs; j++)
            if (c->seen_step[j] != c->obs_step) cons.push_back(j);
        // (only worth a launch's while if it empties rows: every row but the one opened now, or as many landmarks as fit)
    }
    // Plain rows (big maps): every update opens a row, and the landmarks that went out of view stay behind in the rows of the
    // steps that saw them last (~40 per row on BASELINE config 5's map), one more stale row per step, 8 bytes per particle and
    // row for every resample from then on: a copy that grows with the length of the run.  Up to ~1 000 rows the copy roles
    // hide it behind the compute blocks (measured at config 5, steps 1008..1028: 994 rows in use 1.239 ms per step; held at
    // 117 rows by consolidating 81 landmarks per step 1.314 ms: the moves ride in the compute blocks, the copies do not), so
    // the bound is set where the copy would start to show: past kPlainRowsTarget rows in use each update also moves the
    // landmarks of the emptiest rows (at most max(kPlainConsBudget, m / 16), 40 bytes per particle each) into the row it
    // opens, and the rows in use stop growing.
    // Compact contexts of mid-size maps (kernels.h: kMidLandmarks) keep far more landmarks than a packet can move at once, most of
    // them out of view for good (example_loop902: a loop of 117): moving "everything stale" every time seven rows are alive moved
    // ~34 landmarks on EVERY step (53 us per step at 10^5 particles against 30 with plain rows, round 5).  They take the plain
    // rows' policy instead, with bounds that fit the layout: past kMidRowsHigh rows in use the emptiest rows are emptied -- one or two
    // landmarks each on that map -- down to kMidRowsLow, as many as the packet has room for; 40 rows are never exceeded (past
    // kSmallRows - 2 the context would go to plain rows: do_update's guard).
    const bool mid_cons = c->mid_compact && !sharded && c->consolidate && (int) c->live_rows.size() > kMidRowsHigh;
    if (mid_cons || (!c->B.compact && !sharded && !c->dist && c->consolidate && (int) c->live_rows.size() > c->plain_rows_target)) {
        const int budget = mid_cons ? std::max(0, kSmallObs - m) : std::max(kPlainConsBudget, m / 16);
        const int target = mid_cons ? kMidRowsLow : c->plain_rows_target;
        std::vector<int32_t> order(c->live_rows);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return c->refcnt[a] != c->refcnt[b] ? c->refcnt[a] < c->refcnt[b] : a < b; });
        std::vector<char> take((size_t) c->B.cap_rows, 0);
        int planned = 0, rows_taken = 0;
        for (int r : order) {
            if (planned >= budget || (int) c->live_rows.size() - rows_taken <= target) break;
            if (mid_cons && planned + c->refcnt[r] > budget) break;  // (whole rows only: a row half emptied stays in use)
            take[(size_t) r] = 1;
            planned += c->refcnt[r];
            rows_taken++;
        }
        if (planned > 0)
            for (int j = 0; j < c->nf && (int) cons.size() < budget; j++)
                if (take[(size_t) c->erow[j]] && c->seen_step[j] != c->obs_step) cons.push_back(j);
    }
    const int nc = (int) cons.size();
    int e_new = -1;
    std::vector<int32_t> rows_of((size_t) m + nc), dropped;
    if (m + n + nc > 0) {
        e_new = c->free_rows.back();
        c->free_rows.pop_back();
    }
    for (int q = 0; q < nc; q++) {
        const int j = cons[(size_t) q], r = c->erow[j];
        rows_of[(size_t) m + q] = r | (c->live_flag[j] ? kRowLiveBit : 0) | (r == c->fresh_row ? kRowFreshBit : 0);
        c->live_flag[j] ^= 1;
        if (--c->refcnt[r] == 0) {
            rows_remove_live(c, r);
            dropped.push_back(r);
        }
        c->erow[j] = e_new;
        c->refcnt[e_new]++;
    }
    for (int k = 0; k < m; k++) {
        const int r = c->erow[idf[k]];
        rows_of[k] = r | (c->live_flag[idf[k]] ? kRowLiveBit : 0);  // the landmark's live record buffer rides in bit 30
        if (r == c->fresh_row && !sharded) rows_of[k] |= kRowFreshBit;
        c->live_flag[idf[k]] ^= 1;                                   // this update writes its records into the other one
        if (--c->refcnt[r] == 0) {
            rows_remove_live(c, r);
            dropped.push_back(r);
        }
        c->erow[idf[k]] = e_new;
        c->refcnt[e_new]++;
        c->box_dirty[idf[k]] = 1;
    }
    for (int k = 0; k < n; k++) {
        c->box_dirty[c->nf + k] = 1;
        c->erow[c->nf + k] = e_new;
        c->refcnt[e_new]++;
        c->live_flag[c->nf + k] = 0;  // a new row's first records go to buffer 0
    }
    const int n_rows = (int) c->live_rows.size();  // still in use, without e_new: what a pending gather has to compose
    c->tables_dirty = true;

    UpdateArgs U{};
    U.method = c->cfg.method;
    U.m = m;
    U.n = n;
    U.nf = c->nf;
    U.e_new = e_new;
    U.n_rows = n_rows;
    {
        int top = e_new;
        for (int r : c->live_rows) top = std::max(top, r);
        U.live_chunks = (top >> 2) + 1;  // (top = -1: 0 chunks)
    }
    U.all_fresh = 1;
    for (int k = 0; k < std::min(m, 8); k++) U.all_fresh &= (rows_of[k] & kRowFreshBit) ? 1 : 0;
    c->fresh_row = sharded ? -1 : e_new;
    memcpy(U.R, R, sizeof U.R);
    if (c->B.compact) {
        // compact context (at most kSmallRows - 1 landmarks): the packet rides in the kernel-argument segment, no staging
        // copy on the stream
        for (int k = 0; k < m; k++) {
            U.small.idf[k] = idf[k];
            U.small.row[k] = rows_of[k];
            U.small.zf[2 * k] = zf[2 * k];
            U.small.zf[2 * k + 1] = zf[2 * k + 1];
        }
        for (int q = 0; q < nc; q++) {  // the consolidated landmarks ride behind the re-observed ones
            U.small.idf[m + q] = cons[(size_t) q];
            U.small.row[m + q] = rows_of[(size_t) m + q];
        }
        U.n_cons = nc;
        for (int k = 0; k < n; k++) {
            U.small.zn[2 * k] = zn[2 * k];
            U.small.zn[2 * k + 1] = zn[2 * k + 1];
        }
        U.small.magic = kSmallMagic;
        U.big = nullptr;
    } else {
        const int slot = (int) (c->pkt_seq++ % kRing);
        if (c->pkt_ev_used[slot]) HIP_TRY(hipEventSynchronize(c->pkt_ev[slot]));
        char *ph = c->pkt_host + (size_t) slot * c->pkt_bytes;
        ObsPacket *hp = reinterpret_cast<ObsPacket *>(ph);
        hp->m = m;
        hp->n = n;
        hp->nf = c->nf;
        hp->n_rows = n_rows;
        hp->e_new = e_new;
        hp->status = 0;
        hp->cap = 0;  // dense layout
        hp->pad = 0;
        int32_t *hidf = reinterpret_cast<int32_t *>(hp + 1);
        float *hzf = reinterpret_cast<float *>(hidf + m + nc);
        float *hzn = hzf + 2 * m;
        if (m) {
            memcpy(hidf, idf, sizeof(int32_t) * m);
            memcpy(hzf, zf, sizeof(float) * 2 * m);
        }
        if (nc) memcpy(hidf + m, cons.data(), sizeof(int32_t) * nc);  // the consolidated landmarks ride behind the re-observed ones
        if (n) memcpy(hzn, zn, sizeof(float) * 2 * n);
        int32_t *hrow = reinterpret_cast<int32_t *>(hzn + 2 * n);
        if (m + nc) memcpy(hrow, rows_of.data(), sizeof(int32_t) * ((size_t) m + nc));
        if (n_rows) memcpy(hrow + m + nc, c->live_rows.data(), sizeof(int32_t) * n_rows);
        const size_t used = sizeof(ObsPacket) + sizeof(int32_t) * ((size_t) m + nc) + sizeof(float) * 2 * (m + n) + sizeof(int32_t) * ((size_t) m + nc + n_rows);
        U.n_cons = nc;
        char *pd = c->pkt_dev + (size_t) slot * c->pkt_bytes;
        HIP_TRY(hipMemcpyAsync(pd, ph, used, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->pkt_ev[slot], c->stream));
        c->pkt_ev_used[slot] = true;
        U.big = reinterpret_cast<const ObsPacket *>(pd);
    }
    // the row this update opens is in use from now on; the rows it emptied can be opened again by a later update
    if (e_new >= 0) {
        if (c->refcnt[e_new] > 0) rows_add_live(c, e_new);
        else c->free_rows.push_back(e_new);
    }
    for (int r : dropped) c->free_rows.push_back(r);
    // rows are opened lowest-first, so that the rows in use stay dense at the bottom (compact contexts copy chunks
    // [0, live_chunks) on a resample)
    if (c->B.compact && (!dropped.empty() || (e_new >= 0 && c->refcnt[e_new] == 0)))
        std::sort(c->free_rows.begin(), c->free_rows.end(), std::greater<int32_t>());
    return issue_update(c, U, n, n_rows, need_normals, normals, strata, sharded);
}

// FastSLAM{1,2}::update with the observation made on the device (slamgpu_step_observe): observe_book_kernel writes the packet
// and the genealogy bookkeeping into device memory, the update launch reads them there.
int do_update_dev(slamgpu_ctx *c, const float xtrue[3], float max_range, const float R[4], int32_t noise, const float *r1,
                  const float *r2, const float *normals, const float *strata) {
    if (!c->map_dev) return fail(SLAMGPU_ERR_INVALID, "no map: call slamgpu_set_map first");
    if ((!c->B.compact || c->mid_compact) && !(c->cfg.flags & SLAMGPU_FLAG_DEVICE_OBSERVE))
        return fail(SLAMGPU_ERR_INVALID, "create the context with SLAMGPU_FLAG_DEVICE_OBSERVE (its observation packets live in device memory)");
    if (c->dist || c->cfg.n_particles_global != c->cfg.n_particles) return fail(SLAMGPU_ERR_INVALID, "slamgpu_step_observe: single contexts only");
    if (c->map_n > c->B.cap_nf) return fail(SLAMGPU_ERR_CAPACITY, "map of %d landmarks, capacity %d", c->map_n, c->B.cap_nf);
    if (!xtrue || !R || noise < 0 || noise > 2 || (noise == 1 && (!r1 || !r2))) return fail(SLAMGPU_ERR_INVALID, "bad arguments");
    const bool tape = c->cfg.rng_mode == SLAMGPU_RNG_TAPE;
    if (tape && !strata) return fail(SLAMGPU_ERR_INVALID, "TAPE mode needs strata[N] (and normals[3N] for FastSLAM2)");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = book_push(c)) return rc;
    c->obs_step++;
    if (c->B.compact) {
        // compact context: no front-end launch at all: every block of the update launch works the packet out for itself from
        // the state the previous launch left (kernels.h: FrontArgs; kernels.hip: front_make)
        UpdateArgs U{};
        U.method = c->cfg.method;
        U.m = kStageBound;  // (sizes the launch's record staging: the kernel learns m from the packet it makes)
        U.e_new = -1;
        U.small.magic = kSmallMagic;
        for (int q = 0; q < c->map_n; q++) {  // the map rides in the words of the new landmarks' observations
            U.small.zn[q] = c->map_host[(size_t) q];
            U.small.zn[kSmallObs + q] = c->map_host[(size_t) c->map_n + q];
        }
        if (noise == 1)
            for (int q = 0; q < c->map_n; q++) {
                U.small.zf[q] = r1[q];
                U.small.zf[kSmallObs + q] = r2[q];
            }
        FrontArgs &F = U.front;
        F.on = 1;
        F.nlm = c->map_n;
        F.cap_nf = c->B.cap_nf;
        F.noise = noise;
        F.x = xtrue[0];
        F.y = xtrue[1];
        F.phi = xtrue[2];
        F.max_range = max_range;
        F.sr = sqrtf(R[0]);
        F.sb = sqrtf(R[3]);
        F.k0 = (uint32_t) c->cfg.seed;
        F.k1 = (uint32_t) (c->cfg.seed >> 32);
        F.step = ++c->observe_step;
        F.cons_above = c->consolidate ? c->consolidate_above : -1;
        F.state_in = c->front_dev + c->front_par;
        F.state_out = c->front_dev + (c->front_par ^ 1);
        F.out = c->obs_out_dev;
        F.pkt = c->front_pkt_dev;
        memcpy(U.R, R, sizeof U.R);
        const bool need_normals = c->cfg.method == SLAMGPU_FASTSLAM2 && normals != nullptr;
        if (int rc = issue_update(c, U, -1, -1, need_normals, normals, strata, false)) return rc;
        c->front_par ^= 1;
        c->last_pkt_dev = reinterpret_cast<char *>(c->front_pkt_dev);
        c->fresh_row = -1;  // (the device's book knows)
        return 0;
    }
    const size_t nl = (size_t) c->map_n;
    // The front-end kernel runs on a stream of its own, ahead of the update launches: packet t is made while update t - 1
    // still computes (it depends on the true pose and on the previous front-end kernel only); events order the two streams:
    // update t waits for packet t, and a ring slot is rewritten only after the update that read it has finished.
    if (noise == 1) {
        HIP_TRY(hipStreamSynchronize(c->obs_stream));  // (parity mode only: single-buffered normals, pageable sources)
        HIP_TRY(hipMemcpyAsync(c->obs_r_dev, r1, sizeof(float) * nl, hipMemcpyHostToDevice, c->obs_stream));
        HIP_TRY(hipMemcpyAsync(c->obs_r_dev + nl, r2, sizeof(float) * nl, hipMemcpyHostToDevice, c->obs_stream));
        HIP_TRY(hipStreamSynchronize(c->obs_stream));
    }
    const int slot = (int) (c->pkt_seq++ % kRing);
    if (c->pkt_ev_used[slot]) HIP_TRY(hipStreamWaitEvent(c->obs_stream, c->pkt_ev[slot], 0));
    char *pd = c->pkt_dev + (size_t) slot * c->pkt_bytes;
    ObserveArgs A{};
    A.lm = c->map_dev;
    A.table = c->table_dev;
    A.nlm = c->map_n;
    A.x = xtrue[0];
    A.y = xtrue[1];
    A.phi = xtrue[2];
    A.max_range = max_range;
    A.sr = sqrtf(R[0]);
    A.sb = sqrtf(R[3]);
    A.noise = noise;
    A.r1 = c->obs_r_dev;
    A.r2 = c->obs_r_dev + nl;
    A.k0 = (uint32_t) c->cfg.seed;
    A.k1 = (uint32_t) (c->cfg.seed >> 32);
    A.step = ++c->observe_step;
    A.out = c->obs_out_dev;
    A.pkt = reinterpret_cast<ObsPacket *>(pd);
    A.book = c->book_dev;
    A.erow = c->erow_dev;
    A.live = c->live_dev;
    A.refcnt = c->refcnt_dev;
    A.cap_nf = c->B.cap_nf;
    A.cap_rows = c->B.cap_rows;
    A.cons_target = c->consolidate ? c->plain_rows_target : -1;
    A.cons_budget = kPlainConsBudget;
    A.take = c->take_dev;
    c->k->observe_book(c->obs_stream, A);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->obs_ev[slot], c->obs_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->obs_ev[slot], 0));
    c->last_pkt_dev = pd;
    c->fresh_row = -1;  // (the device's book knows)

    UpdateArgs U{};
    U.method = c->cfg.method;
    U.m = c->map_n;  // upper bounds: the kernel reads the packet's header
    U.n = c->map_n;
    U.nf = 0;
    U.e_new = -1;
    U.n_rows = c->B.cap_rows;
    U.dev_packet = 1;
    U.big = reinterpret_cast<const ObsPacket *>(pd);
    memcpy(U.R, R, sizeof U.R);
    const bool need_normals = c->cfg.method == SLAMGPU_FASTSLAM2 && normals != nullptr;
    if (int rc = issue_update(c, U, -1, -1, need_normals, normals, strata, false)) return rc;
    HIP_TRY(hipEventRecord(c->pkt_ev[slot], c->stream));  // the launch that reads this ring slot: the slot may be rewritten after it
    c->pkt_ev_used[slot] = true;
    return 0;
}

}  // namespace

extern "C" {

int slamgpu_update(slamgpu_ctx *c, const float *zf, const int32_t *idf, int32_t m, const float *zn, int32_t n,
                   const float R[4], const float *normals, const float *strata) {
    if (int rc = check_ctx(c)) return rc;
    if (c->dist) return fail(SLAMGPU_ERR_INVALID, "distributed context: use slamgpu_dist_step (an update here would skip the all-gather)");
    if (c->cfg.n_particles_global != c->cfg.n_particles)
        return fail(SLAMGPU_ERR_INVALID, "this context is a shard (%d of %lld particles): use slamgpu_shard_update + slamgpu_shard_*",
                    c->cfg.n_particles, (long long) c->cfg.n_particles_global);
    return do_update(c, zf, idf, m, zn, n, R, normals, strata, false);
}

int slamgpu_step(slamgpu_ctx *c, const float *controls, int32_t n_controls, const float Q[4], float dt, const float *zf,
                 const int32_t *idf, int32_t m, const float *zn, int32_t n, const float R[4], const float *normals,
                 const float *strata, int32_t record_estimate) {
    if (int rc = check_ctx(c)) return rc;
    if (n_controls < 0 || (n_controls > 0 && !controls)) return fail(SLAMGPU_ERR_INVALID, "bad control list");
    if (n_controls > 0 && c->cfg.add_predict_noise && c->cfg.rng_mode == SLAMGPU_RNG_TAPE)
        return fail(SLAMGPU_ERR_INVALID, "slamgpu_step cannot carry TAPE-mode predict noise: call slamgpu_predict per control");
    for (int k = 0; k < n_controls; k++)
        if (int rc = slamgpu_predict(c, controls[3 * k], controls[3 * k + 1], Q, dt, controls[3 * k + 2], nullptr)) return rc;
    if (int rc = slamgpu_update(c, zf, idf, m, zn, n, R, normals, strata)) return rc;
    if (record_estimate) return slamgpu_estimate_async(c);
    return 0;
}

int slamgpu_step_observe(slamgpu_ctx *c, const float *controls, int32_t n_controls, const float Q[4], float dt, const float xtrue[3],
                         float max_range, const float R[4], int32_t noise, const float *r1, const float *r2, const float *normals,
                         const float *strata, int32_t record_estimate) {
    if (int rc = check_ctx(c)) return rc;
    if (n_controls < 0 || (n_controls > 0 && (!controls || !Q))) return fail(SLAMGPU_ERR_INVALID, "bad control list");
    if (n_controls > 0 && c->cfg.add_predict_noise && c->cfg.rng_mode == SLAMGPU_RNG_TAPE)
        return fail(SLAMGPU_ERR_INVALID, "slamgpu_step_observe cannot carry TAPE-mode predict noise: call slamgpu_predict per control");
    for (int k = 0; k < n_controls; k++)
        if (int rc = slamgpu_predict(c, controls[3 * k], controls[3 * k + 1], Q, dt, controls[3 * k + 2], nullptr)) return rc;
    if (int rc = do_update_dev(c, xtrue, max_range, R, noise, r1, r2, normals, strata)) return rc;
    if (record_estimate) return slamgpu_estimate_async(c);
    return 0;
}

// K iterations of the wrapper's loop as ONE launch (kernels.h: PersistArgs): the K calls of slamgpu_step_observe are made with
// the context in collect mode -- every piece of host bookkeeping moves on as usual, the update launches are queued instead of
// made -- then the queue is uploaded and update_persist runs it.
// what the persistent loop needs beside the context's own state (allocated once: at slamgpu_set_map for contexts that qualify, so
// that a caller's first slamgpu_run_observe does not pay for it; K: entries per queue buffer)
static int persist_setup(slamgpu_ctx *c, int32_t K) {
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (!c->psync_dev) {
        HIP_TRY(hipMalloc((void **) &c->psync_dev, sizeof(uint32_t) * kPersistSyncAlloc));
        HIP_TRY(hipMemsetAsync(c->psync_dev, 0, sizeof(uint32_t) * kPersistSyncAlloc, c->stream));
        HIP_TRY(hipMalloc((void **) &c->ppk_dev, sizeof(int32_t) * 2 * kSmallWords));
        HIP_TRY(hipMalloc((void **) &c->pring_dev, sizeof(PersistStep) * 4));
        HIP_TRY(hipMalloc((void **) &c->pdraw_dev, sizeof(float4) * 2 * 6 * (size_t) c->B.ncap));
        HIP_TRY(hipHostMalloc((void **) &c->pstatus_host, kPersistHostWords * sizeof(uint32_t), hipHostMallocDefault));
        for (int w = 0; w < kPersistHostWords; w++) c->pstatus_host[w] = 0;
        for (int b = 0; b < slamgpu_ctx::kPqBufs; b++) HIP_TRY(hipEventCreateWithFlags(&c->pq_kev[b], hipEventDisableTiming));
    }
    if ((size_t) K > c->pq_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));  // (a launch in flight may still read the old queue)
        if (c->pq_host) (void) hipHostFree(c->pq_host);
        c->pq_host = nullptr;
        c->pq_cap = 0;
        const size_t cap = std::max<size_t>((size_t) K, 256);
        HIP_TRY(hipHostMalloc((void **) &c->pq_host, sizeof(PersistStep) * slamgpu_ctx::kPqBufs * cap, hipHostMallocDefault));
        c->pq_cap = cap;
        for (int b = 0; b < slamgpu_ctx::kPqBufs; b++) c->pq_kev_used[b] = false;
    }
    return 0;
}

static bool persist_eligible(const slamgpu_ctx *c) {
    return c->persist_ok && c->B.compact && !c->mid_compact && !c->dist && c->cfg.n_particles_global == c->cfg.n_particles &&
           c->cfg.rng_mode == SLAMGPU_RNG_PHILOX && c->ws.nblocks <= kPersistMaxBlocks && c->ws.nblocks <= c->scan_min_blocks;
}

static int run_observe_persist(slamgpu_ctx *c, int32_t K, const int32_t *n_controls, const float *controls, const float Q[4], float dt,
                               const float *xtrue, float max_range, const float R[4], int32_t noise) {
    if (int rc = persist_setup(c, K)) return rc;
    // everything that can fail without a launch having been made comes BEFORE the host's bookkeeping moves on by K iterations
    // (a failing hipEventSynchronize / hipMemsetAsync behind the collect loop left the host K iterations ahead of the
    // device).  The queue buffer of kPqBufs calls ago must be free (bounds the host's run-ahead), and the loop's meeting words
    // are zeroed in stream order behind the previous launch.
    const int b = c->pq_next;
    if (c->pq_kev_used[b]) HIP_TRY(hipEventSynchronize(c->pq_kev[b]));
    HIP_TRY(hipMemsetAsync(c->psync_dev, 0, sizeof(uint32_t) * kPersistSyncWords, c->stream));
    slamgpu_ctx::PersistCollect col;
    col.steps.reserve((size_t) K);
    c->collect = &col;
    int rc = 0;
    std::string why;
    size_t row = 0;
    int32_t k = 0;
    for (; k < K; k++) {
        const int32_t nc = n_controls[k];
        rc = slamgpu_step_observe(c, nc ? controls + 3 * row : nullptr, nc, Q, dt, xtrue + 3 * (size_t) k, max_range, R, noise, nullptr, nullptr, nullptr,
                                  nullptr, 1);
        if (rc) {
            why = slamgpu_last_error();
            break;
        }
        row += (size_t) nc;
    }
    c->collect = nullptr;
    if (!col.steps.empty()) {  // (after a failure: the iterations before the failing one are applied, as the header promises)
        const size_t n = col.steps.size();
        // The queue is NOT uploaded (round 5, measured with the drop-in binary at 1 000 particles): a hipMemcpyAsync enqueued behind a
        // running launch made the CALL wait for that launch -- 270-520 us per call of 32 iterations, on the context's stream or on one
        // of its own -- so the host simulated its next iterations only after the GPU had finished the last ones.  The kernel reads
        // the entries out of pinned host memory instead, each an iteration before it needs it.
        c->pq_next = (c->pq_next + 1) % slamgpu_ctx::kPqBufs;
        PersistStep *qh = c->pq_host + (size_t) b * c->pq_cap;
        memcpy(qh, col.steps.data(), sizeof(PersistStep) * n);
        UpdateArgs U = col.U;
        U.persist.queue = qh;
        U.persist.K = (int32_t) n;
        // one wait is bounded in time (2 s of the 100 MHz constant clock: a member workgroup that is dispatched that late -- another
        // process holding the CUs -- abandons the launch) and, for the tests of the abandon path, in polls
        U.persist.max_spins = 0xffffffffu;
        U.persist.max_ticks = 200000000ull;
        if (const char *e = getenv("SLAMGPU_PERSIST_MAX_SPINS")) U.persist.max_spins = (uint32_t) std::max(0, atoi(e));
        if (const char *e = getenv("SLAMGPU_PERSIST_MAX_MS")) U.persist.max_ticks = 100000ull * (unsigned long long) std::max(0, atoi(e));
        U.persist.serial = (int32_t) (c->persist_launches + 1);
        U.persist.abort_at = -1;
        if (const char *e = getenv("SLAMGPU_PERSIST_ABORT_AT")) U.persist.abort_at = atoi(e);
        U.persist.sync = c->psync_dev;
        U.persist.host_status = c->pstatus_host;
        U.persist.state_final = c->front_dev + c->front_par;  // (the copy the next launch reads)
        U.persist.packets = c->ppk_dev;
        U.persist.ring = c->pring_dev;
        U.persist.draws = c->pdraw_dev;
        // drawer workgroups (one per tile): FastSLAM 1 in the fast build, whose predicts draw eight Philox blocks per particle and step
        U.persist.drawers = (c->cfg.method == SLAMGPU_FASTSLAM1 && c->cfg.math_mode == SLAMGPU_MATH_FAST && c->cfg.add_predict_noise &&
                             !c->cfg.use_heading && getenv("SLAMGPU_NO_DRAWERS") == nullptr) ? c->ws.nblocks : 0;
        {
            Timed t(c, "persist_loop");
            c->k->update_persist(c->stream, col.B, PredictArgs{}, U, col.rng, col.ws);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->pq_kev[b], c->stream));
        c->pq_kev_used[b] = true;
        c->persist_launches++;
        c->persist_steps += (int64_t) n;
    }
    if (rc) return fail(rc, "slamgpu_run_observe: iteration %d of %d: %s", (int) k, (int) K, why.c_str());
    return 0;
}

int slamgpu_run_observe(slamgpu_ctx *c, int32_t K, const int32_t *n_controls, const float *controls, const float Q[4], float dt,
                        const float *xtrue, float max_range, const float R[4], int32_t noise) {
    if (int rc = check_ctx(c)) return rc;
    // everything that can be refused is refused BEFORE the first device call: a failing call applies no iteration at all
    if (K < 0 || (K > 0 && (!n_controls || !xtrue || !R))) return fail(SLAMGPU_ERR_INVALID, "slamgpu_run_observe: bad arguments");
    if (noise != 0 && noise != 2) return fail(SLAMGPU_ERR_INVALID, "slamgpu_run_observe: noise must be 0 or 2 (a tape is per iteration: slamgpu_step_observe)");
    if (c->cfg.rng_mode == SLAMGPU_RNG_TAPE) return fail(SLAMGPU_ERR_INVALID, "slamgpu_run_observe: TAPE-mode contexts take their draws per iteration (slamgpu_step_observe)");
    if (K == 0) return 0;
    if (!c->map_dev) return fail(SLAMGPU_ERR_INVALID, "slamgpu_run_observe: no map: call slamgpu_set_map first");
    if ((!c->B.compact || c->mid_compact) && !(c->cfg.flags & SLAMGPU_FLAG_DEVICE_OBSERVE))
        return fail(SLAMGPU_ERR_INVALID, "slamgpu_run_observe: create the context with SLAMGPU_FLAG_DEVICE_OBSERVE (its observation packets live in device memory)");
    if (c->dist || c->cfg.n_particles_global != c->cfg.n_particles) return fail(SLAMGPU_ERR_INVALID, "slamgpu_run_observe: single contexts only");
    if (c->map_n > c->B.cap_nf) return fail(SLAMGPU_ERR_CAPACITY, "slamgpu_run_observe: map of %d landmarks, capacity %d", c->map_n, c->B.cap_nf);
    size_t total = 0;
    int32_t max_nc = 0;
    for (int32_t k = 0; k < K; k++) {
        if (n_controls[k] < 0) return fail(SLAMGPU_ERR_INVALID, "slamgpu_run_observe: iteration %d has a negative control count", (int) k);
        total += (size_t) n_controls[k];
        max_nc = std::max(max_nc, n_controls[k]);
    }
    if (total > 0 && (!controls || !Q)) return fail(SLAMGPU_ERR_INVALID, "slamgpu_run_observe: %zu controls but no control list / Q", total);
    if (total > 0 && c->cfg.add_predict_noise && c->cfg.rng_mode == SLAMGPU_RNG_TAPE)
        return fail(SLAMGPU_ERR_INVALID, "slamgpu_run_observe cannot carry TAPE-mode predict noise");
    if ((int64_t) c->hist_n + K > kHistCap)
        return fail(SLAMGPU_ERR_CAPACITY, "slamgpu_run_observe: %d iterations would overflow the estimate history (%d of %d entries in use): call "
                                          "slamgpu_history_fetch first, or hand over fewer iterations", (int) K, c->hist_n, kHistCap);
    if (int rc = persist_check(c)) return rc;
    // small compact contexts: ONE launch for all K iterations (kernels.h: PersistArgs)
    if (K >= 2 && persist_eligible(c) && max_nc <= kMaxFusedPredict)
        return run_observe_persist(c, K, n_controls, controls, Q, dt, xtrue, max_range, R, noise);
    size_t row = 0;
    for (int32_t k = 0; k < K; k++) {
        const int32_t nc = n_controls[k];
        if (int rc = slamgpu_step_observe(c, nc ? controls + 3 * row : nullptr, nc, Q, dt, xtrue + 3 * (size_t) k, max_range, R, noise, nullptr, nullptr,
                                          nullptr, nullptr, 1)) {
            std::string why = slamgpu_last_error();
            return fail(rc, "slamgpu_run_observe: iteration %d of %d: %s", (int) k, (int) K, why.c_str());
        }
        row += (size_t) nc;
    }
    return 0;
}

int slamgpu_observe_fetch(slamgpu_ctx *c, float *z, int32_t *vis, int32_t *nz, float *zf, int32_t *idf, int32_t *m, float *zn, int32_t *n) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->last_pkt_dev) return fail(SLAMGPU_ERR_INVALID, "no device-made observation yet (slamgpu_step_observe)");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (c->obs_stream) HIP_TRY(hipStreamSynchronize(c->obs_stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const size_t C = (size_t) c->map_n;
    std::vector<char> pk(sizeof(ObsPacket) + 4 * 6 * C), ob(sizeof(ObserveOut) + 4 * 3 * C);
    HIP_TRY(hipMemcpy(pk.data(), c->last_pkt_dev, pk.size(), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ob.data(), c->obs_out_dev, ob.size(), hipMemcpyDeviceToHost));
    const ObsPacket *h = reinterpret_cast<const ObsPacket *>(pk.data());
    const int32_t *base = reinterpret_cast<const int32_t *>(h + 1);
    const ObserveOut *o = reinterpret_cast<const ObserveOut *>(ob.data());
    const float *hz = reinterpret_cast<const float *>(o + 1);
    const int32_t *hvis = reinterpret_cast<const int32_t *>(hz + 2 * C);
    if (z) memcpy(z, hz, sizeof(float) * 2 * (size_t) o->nz);
    if (vis) memcpy(vis, hvis, sizeof(int32_t) * (size_t) o->nz);
    if (nz) *nz = o->nz;
    if (idf) memcpy(idf, base, sizeof(int32_t) * (size_t) h->m);
    if (zf) memcpy(zf, base + C, sizeof(float) * 2 * (size_t) h->m);
    if (zn) memcpy(zn, base + 3 * C, sizeof(float) * 2 * (size_t) h->n);
    if (m) *m = h->m;
    if (n) *n = h->n;
    if (h->status & kStatusCapacity) return fail(SLAMGPU_ERR_CAPACITY, "the device front end dropped new landmarks: landmark capacity %d exceeded", c->B.cap_nf);
    return 0;
}

int slamgpu_shard_step(slamgpu_ctx *c, const float *controls, int32_t n_controls, const float Q[4], float dt, const float *zf,
                       const int32_t *idf, int32_t m, const float *zn, int32_t n, const float R[4], const float *normals,
                       const float *strata) {
    if (int rc = check_ctx(c)) return rc;
    if (n_controls < 0 || (n_controls > 0 && !controls)) return fail(SLAMGPU_ERR_INVALID, "bad control list");
    if (n_controls > 0 && c->cfg.add_predict_noise && c->cfg.rng_mode == SLAMGPU_RNG_TAPE)
        return fail(SLAMGPU_ERR_INVALID, "slamgpu_shard_step cannot carry TAPE-mode predict noise: call slamgpu_predict per control");
    for (int k = 0; k < n_controls; k++)
        if (int rc = slamgpu_predict(c, controls[3 * k], controls[3 * k + 1], Q, dt, controls[3 * k + 2], nullptr)) return rc;
    return slamgpu_shard_update(c, zf, idf, m, zn, n, R, normals, strata);
}

// ---- sharded operation -------------------------------------------------------------------------------
int slamgpu_shard_update(slamgpu_ctx *c, const float *zf, const int32_t *idf, int32_t m, const float *zn, int32_t n,
                         const float R[4], const float *normals, const float *strata) {
    if (int rc = check_ctx(c)) return rc;
    if (c->cfg.n_particles % kBlock != 0 || c->cfg.first_particle % kBlock != 0)
        return fail(SLAMGPU_ERR_INVALID, "shards must hold a multiple of %d particles", kBlock);
    if (c->cfg.log_weights) return fail(SLAMGPU_ERR_INVALID, "the sharded resampling stage works on linear weights (log_weights contexts: slamgpu_update)");
    if (c->dist) return fail(SLAMGPU_ERR_INVALID, "distributed context: use slamgpu_dist_step");
    return do_update(c, zf, idf, m, zn, n, R, normals, strata, true);
}

int slamgpu_shard_set_totals_buffer(slamgpu_ctx *c, float *totals_dev) {
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->own_totals) c->own_blk_w = c->ws.blk_w[0];
    float *p = totals_dev ? totals_dev : c->own_blk_w;
    c->own_totals = totals_dev == nullptr;
    c->ws.blk_w[0] = p;  // shard contexts only use parity 0 of the weight scratch
    return 0;
}

int slamgpu_shard_block_totals(slamgpu_ctx *c, const float **totals_dev, int32_t *nblocks) {
    if (int rc = check_ctx(c)) return rc;
    if (totals_dev) *totals_dev = c->ws.blk_w[0];  // [w(nblocks) | w2(nblocks)]
    if (nblocks) *nblocks = c->ws.nblocks;
    return 0;
}

int slamgpu_shard_record_floats(slamgpu_ctx *c) { return c ? 10 + 5 * c->nf : SLAMGPU_ERR_INVALID; }

int slamgpu_shard_plan(slamgpu_ctx *c, const float *gtot, int32_t nb_global, int32_t n_shards, slamgpu_shard_plan_t *out) {
    if (int rc = check_ctx(c)) return rc;
    if (!gtot || !out) return fail(SLAMGPU_ERR_INVALID, "null argument");
    if (n_shards < 1 || n_shards > kMaxShards) return fail(SLAMGPU_ERR_INVALID, "n_shards %d out of range [1,%d]", n_shards, kMaxShards);
    if (nb_global > kMaxScanBlocks || nb_global % n_shards != 0 || (int64_t) nb_global * kBlock != n_global(c))
        return fail(SLAMGPU_ERR_INVALID, "nb_global %d inconsistent with %lld particles over %d shards", nb_global, (long long) n_global(c), n_shards);
    static_assert(sizeof(slamgpu_shard_plan_t) == sizeof(ShardPlan), "public / device plan layout");
    HIP_TRY(hipSetDevice(c->cfg.device));
    ShardPlanArgs A{};
    A.gblk = gtot;
    A.nb_global = nb_global;
    A.nb_per_shard = nb_global / n_shards;
    A.n_shards = n_shards;
    A.do_resample = c->cfg.resample;
    A.n_effective = c->cfg.n_effective;
    // the plan goes straight into pinned host memory and the host polls a sequence word the kernel stores last: a
    // stream synchronisation costs several microseconds more than the store takes to arrive.  Bounded: after 2 ms
    // without the word (e.g. a profiler serialising the queue) fall back to synchronising the stream.
    const uint32_t seq = ++c->plan_seq;
    {
        Timed t(c, "shard_plan");
        c->k->shard_plan(c->stream, A, rng_args(c, c->obs_step), c->plan_host, c->plan_seq_host, seq);
    }
    HIP_TRY(hipGetLastError());
    const auto t0 = std::chrono::steady_clock::now();
    volatile uint32_t *flag = c->plan_seq_host;
    bool arrived = false;
    for (uint64_t spin = 0;; spin++) {
        if (*flag == seq) {
            arrived = true;
            break;
        }
        if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    if (!arrived) HIP_TRY(hipStreamSynchronize(c->stream));
    std::atomic_thread_fence(std::memory_order_acquire);
    memcpy(out, c->plan_host, sizeof(ShardPlan));
    return 0;
}

int slamgpu_shard_pack(slamgpu_ctx *c, const float *gtot, int32_t nb_global, int32_t n_shards, int32_t shard,
                       const slamgpu_shard_plan_t *plan, float *send_dev, int64_t *send_counts, int64_t *recv_counts) {
    if (int rc = check_ctx(c)) return rc;
    if (!gtot || !plan || !send_counts || !recv_counts) return fail(SLAMGPU_ERR_INVALID, "null argument");
    if (n_shards < 1 || nb_global % n_shards != 0) return fail(SLAMGPU_ERR_INVALID, "bad shard geometry");
    if (shard < 0 || shard >= n_shards || n_shards > kMaxShards) return fail(SLAMGPU_ERR_INVALID, "bad shard index");
    const int64_t n = c->cfg.n_particles;
    const int64_t k_lo = plan->K[shard], k_hi = plan->K[shard + 1];
    // records this shard sends to d / receives from s: overlaps of offspring ranges with output ranges
    for (int d = 0; d < n_shards; d++) {
        const int64_t a = std::max(k_lo, (int64_t) d * n), b = std::min(k_hi, (int64_t) (d + 1) * n);
        send_counts[d] = (b > a && d != shard) ? b - a : 0;  // own offspring are gathered in place by the pack kernel
        const int64_t ra = std::max(plan->K[d], (int64_t) shard * n), rb = std::min(plan->K[d + 1], (int64_t) (shard + 1) * n);
        recv_counts[d] = (rb > ra && d != shard) ? rb - ra : 0;
    }
    int64_t to_send = 0;
    for (int d = 0; d < n_shards; d++) to_send += send_counts[d];
    if (to_send > 0 && !send_dev) return fail(SLAMGPU_ERR_INVALID, "null send buffer");
    HIP_TRY(hipSetDevice(c->cfg.device));
    ShardPackArgs A{};
    A.gblk = gtot;
    A.nb_global = nb_global;
    A.nb_per_shard = nb_global / n_shards;
    A.first_block = (int32_t) (c->cfg.first_particle / kBlock);
    A.k_lo = k_lo;
    A.k_hi = k_hi;
    A.n_per_shard = n;
    A.nf = c->nf;
    A.fields = 10 + 5 * c->nf;
    A.shard = shard;
    A.send = send_dev;
    if (int rc = sync_tables(c)) return rc;  // the pack kernel reads records through the genealogy (B.erow)
    c->B.slot = c->slot;
    c->shard_settled = false;
    {
        Timed t(c, "shard_pack");
        c->k->shard_pack(c->stream, c->B, c->ws, A, rng_args(c, c->obs_step));
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int slamgpu_shard_unpack(slamgpu_ctx *c, const float *recv_dev, int32_t n_shards, int32_t shard,
                         const slamgpu_shard_plan_t *plan) {
    if (int rc = check_ctx(c)) return rc;
    if (!plan) return fail(SLAMGPU_ERR_INVALID, "null argument");
    if (shard < 0 || shard >= n_shards || n_shards > kMaxShards) return fail(SLAMGPU_ERR_INVALID, "bad shard index");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const int64_t n = c->cfg.n_particles, first = (int64_t) shard * n;
    {
        // nothing to scatter when every output slot of this shard was produced locally
        const int64_t own_lo = std::min(std::max(plan->K[shard] - first, (int64_t) 0), n);
        const int64_t own_hi = std::min(std::max(plan->K[shard + 1] - first, (int64_t) 0), n);
        if (own_lo == 0 && own_hi == n) return 0;
        if (!recv_dev) return fail(SLAMGPU_ERR_INVALID, "null receive buffer");
    }
    ShardUnpackArgs A{};
    A.recv = recv_dev;
    A.n_shards = n_shards;
    A.nf = c->nf;
    A.fields = 10 + 5 * c->nf;
    A.shard = shard;
    for (int s = 0; s <= n_shards; s++) A.src_lo[s] = std::min(std::max(plan->K[s] - first, (int64_t) 0), n);
    // the arrivals' landmark records go into the arrival pool (allocated on first use); when it cannot take this step's,
    // the kernel settles the whole shard instead, which also empties the pool
    const int64_t arrivals = n - (A.src_lo[shard + 1] - A.src_lo[shard]);
    if (!c->B.poolA) {
        const int64_t budget = (int64_t) 256 << 20;  // bytes
        int64_t cap = budget / (20 * (int64_t) c->B.cap_nf);
        cap = std::max<int64_t>(256, std::min<int64_t>(cap, 32768));
        if (const char *e = getenv("SLAMGPU_POOL_CAP")) cap = std::max(1, atoi(e));  // diagnostic / tests
        HIP_TRY(hipMalloc((void **) &c->B.poolA, sizeof(float4) * (size_t) cap * c->B.cap_nf));
        HIP_TRY(hipMalloc((void **) &c->B.poolB, sizeof(float) * (size_t) cap * c->B.cap_nf));
        c->B.pool_cap = (int32_t) cap;
        c->pool_used = 0;
    }
    const bool settle = c->pool_used + arrivals > c->B.pool_cap;
    A.pool_base = settle ? -1 : (int32_t) c->pool_used;
    A.own_lo = (int32_t) A.src_lo[shard];
    A.own_hi = (int32_t) A.src_lo[shard + 1];
    if (int rc = sync_tables(c)) return rc;  // arrivals: every live row points at the pool; settling reads through B.erow
    c->B.slot = c->slot;
    {
        Timed t(c, "shard_unpack");
        c->k->shard_unpack(c->stream, c->B, c->ws, A);
    }
    HIP_TRY(hipGetLastError());
    if (settle) {
        // the kernel rewrote the whole shard physically and flipped every landmark row; nothing references the pool now
        c->shard_settled = true;
        for (int j = 0; j < c->nf; j++) c->live_flag[j] ^= 1;
        c->pool_used = 0;
        rows_reset(c, c->nf);  // every landmark of every output particle in its own slot: genealogy row 0
    } else {
        c->pool_used += arrivals;
    }
    return 0;
}

int slamgpu_shard_finish(slamgpu_ctx *c, const slamgpu_shard_plan_t *plan) {
    if (int rc = check_ctx(c)) return rc;
    if (!plan) return fail(SLAMGPU_ERR_INVALID, "null plan");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const int mode = !plan->resampled ? 0 : (c->shard_settled ? 2 : 1);
    c->B.slot = c->slot;
    {
        Timed t(c, "shard_finish");
        c->k->shard_finish(c->stream, c->B, c->ws, plan->wsum, plan->wsq, plan->neff, mode);
    }
    c->keep_slot = c->slot ^ 1;  // pack / unpack wrote the ancestors there
    c->slot ^= 1;  // shard_finalize_kernel published the live / pending state in the other slot
    c->B.slot = c->slot;
    c->maybe_pending = mode == 1;
    c->shard_settled = false;
    c->est_fresh = false;
    c->shard_est_fresh = true;  // est_part holds this shard's partials of this update
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- distributed operation ---------------------------------------------------------------------------------------------
namespace {
// the all-gather that follows an update launch of a distributed context, on the context's stream
int launch_flags(slamgpu_ctx *c);
int gather_totals(slamgpu_ctx *c) {
    if (c->dist_fold) return 0;                // ... and the barrier rides at the head of the next launch
    if (c->dist_push) return launch_flags(c);  // the totals are in every table already: only the barrier is left
    if (!c->comm) return 0;
    const int par = (int) (c->obs_step & 1);
    Timed t(c, "allgather");  // (slamgpu_profile: an event pair around the collective, like around every launch)
    RCCL_TRY(rccl()->AllGather(c->ws.blk_w[par], c->gtot_dev[par], (size_t) 2 * c->ws.nblocks, ncclFloat, (ncclComm_t) c->comm, c->stream));
    return 0;
}

// The barrier error word of the push / fold collective is sticky (a kernel that finds it set does not wait again): whoever
// synchronises with the device anyway -- history fetch, settle + read, the status query -- reports it, so that a C caller
// that never polls slamgpu_dist_collective_status still cannot read results of unsynchronised steps as valid.
int barrier_check(slamgpu_ctx *c) {
    if (!c->dist || !c->flags_dev || !(c->dist_push || c->dist_fold)) return 0;
    uint32_t err = 0;
    HIP_TRY(hipMemcpy(&err, c->flags_dev + kMaxShards, sizeof err, hipMemcpyDeviceToHost));
    if (err != 0)
        return fail(SLAMGPU_ERR_BARRIER, "flag barrier %u of the push collective timed out: a peer did not arrive; the steps since then are void "
                                         "(recreate the contexts, or use SLAMGPU_DIST_GATHER)", err);
    return 0;
}

constexpr int kDistArrays = 17;
constexpr int kFlagWords = kGoBase + kGoStride * kGoWords;  // flags [0, kMaxShards), error word at kMaxShards, then the go words
struct DistBlob {
    int64_t pid;
    int32_t device, ncap, cap_nf, compact;
    void *ptr[kDistArrays];
    hipIpcMemHandle_t handle[kDistArrays];
};
void dist_arrays(slamgpu_ctx *c, void **a) {
    int k = 0;
    for (int b = 0; b < 2; b++) a[k++] = c->B.poseA[b];
    for (int b = 0; b < 2; b++) a[k++] = c->B.poseB[b];
    for (int b = 0; b < 2; b++) a[k++] = c->B.poseC[b];
    for (int b = 0; b < 2; b++) a[k++] = c->B.lmkA[b];
    for (int b = 0; b < 2; b++) a[k++] = c->B.lmkB[b];
    for (int b = 0; b < 2; b++) a[k++] = c->B.gen[b];
    for (int b = 0; b < 2; b++) a[k++] = c->ws.lcum[b];
    for (int b = 0; b < 2; b++) a[k++] = c->gtot_dev[b];
    a[k++] = c->flags_dev;
}

// the table of everybody's block totals (by step parity) and the flag words: allocated before the export, because the
// peers map them too (push collective)
int dist_alloc_shared(slamgpu_ctx *c) {
    if (c->gtot_dev[0]) return 0;
    const int64_t n_shards = std::max<int64_t>(1, n_global(c) / std::max(1, c->cfg.n_particles));
    for (int b = 0; b < 2; b++) {
        HIP_TRY(hipMalloc((void **) &c->gtot_dev[b], sizeof(float) * 3 * (size_t) c->ws.nblocks * (size_t) n_shards));
        HIP_TRY(hipMemset(c->gtot_dev[b], 0, sizeof(float) * 3 * (size_t) c->ws.nblocks * (size_t) n_shards));
    }
    // (fine-grained device memory: polled by a running kernel while another GPU stores into it.  A stack that cannot
    // provide it loses the push collective, nothing else.)
    if (hipExtMallocWithFlags((void **) &c->flags_dev, sizeof(uint32_t) * kFlagWords, hipDeviceMallocFinegrained) != hipSuccess) {
        (void) hipGetLastError();
        c->flags_dev = nullptr;
    } else {
        HIP_TRY(hipMemset(c->flags_dev, 0, sizeof(uint32_t) * kFlagWords));
    }
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

int launch_flags(slamgpu_ctx *c) {
    DistFlagArgs A{};
    for (int h = 0; h < c->B.n_shards; h++) A.peer_flags[h] = c->peer_flags[h];
    A.my_flags = c->flags_dev;
    A.err = c->flags_dev + kMaxShards;
    A.n_shards = c->B.n_shards;
    A.shard = c->B.shard;
    A.seq = ++c->flag_seq;
    A.max_spins = 1u << 20;  // (~a second: a peer that never arrives is reported, not waited for)
    c->k->dist_flags(c->stream, A);
    HIP_TRY(hipGetLastError());
    return 0;
}
}  // namespace

int slamgpu_dist_export_size(void) { return (int) sizeof(DistBlob); }

int slamgpu_dist_export(slamgpu_ctx *c, void *blob) {
    if (int rc = check_ctx(c)) return rc;
    if (!blob) return fail(SLAMGPU_ERR_INVALID, "null blob");
    HIP_TRY(hipSetDevice(c->cfg.device));
    DistBlob b{};
    b.pid = (int64_t) getpid();
    b.device = c->cfg.device;
    b.ncap = c->B.ncap;
    b.cap_nf = c->B.cap_nf;
    b.compact = c->B.compact;
    if (int rc = dist_alloc_shared(c)) return rc;
    dist_arrays(c, b.ptr);
    for (int k = 0; k < kDistArrays; k++)
        if (b.ptr[k]) HIP_TRY(hipIpcGetMemHandle(&b.handle[k], b.ptr[k]));  // (null: no flag words on that shard)
    memcpy(blob, &b, sizeof b);
    return 0;
}

int slamgpu_dist_connect(slamgpu_ctx *c, int32_t n_shards, int32_t shard, const void *blobs) {
    if (int rc = check_ctx(c)) return rc;
    if (!blobs || n_shards < 1 || n_shards > kMaxShards || shard < 0 || shard >= n_shards) return fail(SLAMGPU_ERR_INVALID, "bad shard geometry");
    if (c->cfg.log_weights && n_shards > 1) return fail(SLAMGPU_ERR_INVALID, "log_weights is not available for distributed contexts");
    const int64_t n = c->cfg.n_particles;
    if (n_shards > 1 && (n % kBlock != 0 || c->cfg.first_particle != (int64_t) shard * n || n_global(c) != n * n_shards))
        return fail(SLAMGPU_ERR_INVALID, "shard %d of %d must hold global particles [%lld, %lld) of %lld, a multiple of %d each", shard, n_shards,
                    (long long) (shard * n), (long long) ((shard + 1) * n), (long long) (n * n_shards), kBlock);
    if ((int64_t) c->ws.nblocks * n_shards > kMaxScanBlocks)
        return fail(SLAMGPU_ERR_INVALID, "%lld particles exceed %d blocks of 256", (long long) (n * n_shards), kMaxScanBlocks);
    if (c->obs_step != 0 || c->nf != 0) return fail(SLAMGPU_ERR_INVALID, "connect a distributed context before its first update");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const DistBlob *all = static_cast<const DistBlob *>(blobs);
    std::vector<PeerPtrs> table((size_t) n_shards);
    for (int h = 0; h < n_shards; h++) {
        const DistBlob &b = all[h];
        if (b.ncap != c->B.ncap || b.cap_nf != c->B.cap_nf || b.compact != c->B.compact)
            return fail(SLAMGPU_ERR_INVALID, "shard %d was created with different sizes", h);
        void *p[kDistArrays];
        if (h == shard) {
            if (int rc = dist_alloc_shared(c)) return rc;
            dist_arrays(c, p);
        } else if (b.pid == (int64_t) getpid()) {
            for (int k = 0; k < kDistArrays; k++) p[k] = b.ptr[k];
            if (b.device != c->cfg.device) {
                hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    return fail(SLAMGPU_ERR_HIP, "no peer access from device %d to device %d: %s", c->cfg.device, b.device, hipGetErrorString(e));
                (void) hipGetLastError();
            }
        } else {
            for (int k = 0; k < kDistArrays; k++) {
                if (!b.ptr[k]) {
                    p[k] = nullptr;
                    continue;
                }
                hipError_t e = hipIpcOpenMemHandle(&p[k], b.handle[k], hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) return fail(SLAMGPU_ERR_HIP, "hipIpcOpenMemHandle (shard %d, array %d): %s", h, k, hipGetErrorString(e));
                c->ipc_opened.push_back(p[k]);
            }
        }
        PeerPtrs &t = table[(size_t) h];
        int k = 0;
        for (int q = 0; q < 2; q++) t.poseA[q] = (float4 *) p[k++];
        for (int q = 0; q < 2; q++) t.poseB[q] = (float4 *) p[k++];
        for (int q = 0; q < 2; q++) t.poseC[q] = (float2 *) p[k++];
        for (int q = 0; q < 2; q++) t.lmkA[q] = (float4 *) p[k++];
        for (int q = 0; q < 2; q++) t.lmkB[q] = (float *) p[k++];
        for (int q = 0; q < 2; q++) t.gen[q] = (int32_t *) p[k++];
        for (int q = 0; q < 2; q++) t.lcum[q] = (float *) p[k++];
        for (int q = 0; q < 2; q++) t.gtot[q] = (float *) p[k++];
        c->peer_flags[h] = (uint32_t *) p[k++];
        t.flags = c->peer_flags[h];
    }
    HIP_TRY(hipMalloc((void **) &c->peers_dev, sizeof(PeerPtrs) * (size_t) n_shards));
    HIP_TRY(hipMemcpy(c->peers_dev, table.data(), sizeof(PeerPtrs) * (size_t) n_shards, hipMemcpyHostToDevice));
    HIP_TRY(hipDeviceSynchronize());
    c->B.peers = c->peers_dev;
    c->B.n_shards = n_shards;
    c->B.shard = shard;
    c->B.first = (int32_t) c->cfg.first_particle;
    c->B.div_n = (unsigned long long) (~0ull / (unsigned long long) c->B.ncap) + 1ull;
    c->dist = true;
    return 0;
}

int slamgpu_dist_step(slamgpu_ctx *c, const float *controls, int32_t n_controls, const float Q[4], float dt, const float *zf,
                      const int32_t *idf, int32_t m, const float *zn, int32_t n, const float R[4], int32_t record_estimate) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->dist) return fail(SLAMGPU_ERR_INVALID, "not a distributed context: call slamgpu_dist_connect first");
    if (c->cfg.rng_mode != SLAMGPU_RNG_PHILOX) return fail(SLAMGPU_ERR_INVALID, "distributed contexts draw their noise on the device (SLAMGPU_RNG_PHILOX)");
    if (n_controls < 0 || (n_controls > 0 && !controls)) return fail(SLAMGPU_ERR_INVALID, "bad control list");
    for (int k = 0; k < n_controls; k++)
        if (int rc = slamgpu_predict(c, controls[3 * k], controls[3 * k + 1], Q, dt, controls[3 * k + 2], nullptr)) return rc;
    if (record_estimate && c->hist_n >= kHistCap) return fail(SLAMGPU_ERR_CAPACITY, "estimate history full (%d): fetch it", kHistCap);
    if (int rc = do_update(c, zf, idf, m, zn, n, R, nullptr, nullptr, false)) return rc;
    if (record_estimate) {
        c->hist_n++;  // slot filled when the partials of this update are reduced (next launch / fetch)
        c->est_fresh = false;
    }
    return gather_totals(c);
}

int slamgpu_dist_comm_id(void *id, int32_t bytes) {
    if (!id || bytes < (int32_t) sizeof(ncclUniqueId)) return fail(SLAMGPU_ERR_INVALID, "id buffer must hold %d bytes", (int) sizeof(ncclUniqueId));
    if (!rccl()) return fail(SLAMGPU_ERR_HIP, "librccl.so.1 not found: %s", dlerror());
    ncclUniqueId u;
    RCCL_TRY(rccl()->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return (int) sizeof u > 0 ? 0 : 0;
}

int slamgpu_dist_comm_init(slamgpu_ctx *c, const void *id, int32_t n_ranks, int32_t rank) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->dist) return fail(SLAMGPU_ERR_INVALID, "not a distributed context: call slamgpu_dist_connect first");
    if (!id || n_ranks != c->B.n_shards || rank != c->B.shard) return fail(SLAMGPU_ERR_INVALID, "communicator geometry differs from the shard geometry");
    if (c->comm) return fail(SLAMGPU_ERR_INVALID, "communicator already initialised");
    if (!rccl()) return fail(SLAMGPU_ERR_HIP, "librccl.so.1 not found: %s", dlerror());
    HIP_TRY(hipSetDevice(c->cfg.device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    RCCL_TRY(rccl()->CommInitRank(&comm, n_ranks, u, rank));
    c->comm = comm;
    return 0;
}

int slamgpu_dist_comm_info(slamgpu_ctx *c, int32_t *n_ranks, int32_t *rank) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->comm) return fail(SLAMGPU_ERR_INVALID, "no communicator inside the library (slamgpu_dist_comm_init)");
    if (!rccl() || !rccl()->CommCount || !rccl()->CommUserRank) return fail(SLAMGPU_ERR_HIP, "this librccl has no ncclCommCount / ncclCommUserRank");
    int n = 0, r = 0;
    RCCL_TRY(rccl()->CommCount(static_cast<ncclComm_t>(c->comm), &n));
    RCCL_TRY(rccl()->CommUserRank(static_cast<ncclComm_t>(c->comm), &r));
    if (n_ranks) *n_ranks = n;
    if (rank) *rank = r;
    return 0;
}

int slamgpu_dist_remote_reads(slamgpu_ctx *c, uint64_t *particles) {
    if (int rc = check_ctx(c)) return rc;
    if (!particles) return fail(SLAMGPU_ERR_INVALID, "null output");
    if (!c->dist) return fail(SLAMGPU_ERR_INVALID, "not a distributed context");
    HIP_TRY(hipSetDevice(c->cfg.device));
    // (only this word, and without running any outstanding stage: a distributed context's stages need the other shards)
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, &c->B.ctrl->remote_reads, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *particles = v;
    c->count_remote = true;  // (the counter is kept from the first time it is asked for: ADVICE r4)
    return 0;
}

int slamgpu_dist_totals(slamgpu_ctx *c, const float **local_dev, float **gathered_dev, int32_t *floats_per_shard) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->dist) return fail(SLAMGPU_ERR_INVALID, "not a distributed context");
    const int par = (int) (c->obs_step & 1);
    if (local_dev) *local_dev = c->ws.blk_w[par];
    if (gathered_dev) *gathered_dev = c->gtot_dev[par];
    if (floats_per_shard) *floats_per_shard = 2 * c->ws.nblocks;  // [w | q] (linear weights)
    return 0;
}

int slamgpu_dist_set_collective(slamgpu_ctx *c, int32_t mode) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->dist) return fail(SLAMGPU_ERR_INVALID, "not a distributed context");
    if (mode != SLAMGPU_DIST_GATHER && mode != SLAMGPU_DIST_PUSH && mode != SLAMGPU_DIST_FOLD)
        return fail(SLAMGPU_ERR_INVALID, "unknown collective %d", mode);
    if (mode != SLAMGPU_DIST_GATHER)
        for (int h = 0; h < c->B.n_shards; h++)
            if (!c->peer_flags[h]) return fail(SLAMGPU_ERR_INVALID, "push collective unavailable: shard %d has no flag words (fine-grained memory)", h);
    if (c->unplanned.has && !c->dist_clean) return fail(SLAMGPU_ERR_INVALID, "switch the collective between settled steps (slamgpu_dist_settle)");
    c->dist_push = mode == SLAMGPU_DIST_PUSH;
    c->dist_fold = mode == SLAMGPU_DIST_FOLD;
    return 0;
}

int slamgpu_dist_handshake_test(slamgpu_ctx *c, int32_t iters, double *usec, int32_t *ok) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->dist) return fail(SLAMGPU_ERR_INVALID, "not a distributed context");
    for (int h = 0; h < c->B.n_shards; h++)
        if (!c->peer_flags[h]) return fail(SLAMGPU_ERR_INVALID, "push collective unavailable: shard %d has no flag words (fine-grained memory)", h);
    if (iters < 1 || iters > 100000) return fail(SLAMGPU_ERR_INVALID, "iters out of range");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (!usec && !ok) {  // enqueue only: several contexts driven by one thread must all have theirs queued before anybody waits
        for (int i = 0; i < iters; i++)
            if (int rc = launch_flags(c)) return rc;
        return 0;
    }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, c->stream));
    for (int i = 0; i < iters; i++)
        if (int rc = launch_flags(c)) return rc;
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    uint32_t err = 0;
    HIP_TRY(hipMemcpy(&err, c->flags_dev + kMaxShards, sizeof err, hipMemcpyDeviceToHost));
    if (usec) *usec = 1e3 * (double) ms / iters;
    if (ok) *ok = err == 0 ? 1 : 0;
    return 0;
}

int slamgpu_dist_collective_status(slamgpu_ctx *c, int32_t *ok) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->dist || !ok) return fail(SLAMGPU_ERR_INVALID, "not a distributed context / null output");
    if (!c->flags_dev) {
        *ok = 1;  // (no flag words: the push collective was never in use)
        return 0;
    }
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t err = 0;
    HIP_TRY(hipMemcpy(&err, c->flags_dev + kMaxShards, sizeof err, hipMemcpyDeviceToHost));
    *ok = err == 0 ? 1 : 0;
    return 0;
}

int slamgpu_dist_gather(slamgpu_ctx *c) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->dist || !c->comm) return fail(SLAMGPU_ERR_INVALID, "no communicator: slamgpu_dist_comm_init first");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const int par = (int) (c->obs_step & 1);
    Timed t(c, "allgather");  // (slamgpu_profile: an event pair around the collective, like around every launch)
    RCCL_TRY(rccl()->AllGather(c->ws.blk_w[par], c->gtot_dev[par], (size_t) 2 * c->ws.nblocks, ncclFloat, (ncclComm_t) c->comm, c->stream));
    return 0;
}

int slamgpu_dist_settle(slamgpu_ctx *c) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->dist) return fail(SLAMGPU_ERR_INVALID, "not a distributed context");
    if (c->dist_clean || !c->unplanned.has) return 0;
    // an update without observations and without predicts: applies the pending resampling stage (inline plan) and leaves
    // a stage that is a no-op.  It must not consume an observation-step number: the Philox streams of later steps would shift.
    const float R[4] = {1.0f, 0.0f, 0.0f, 1.0f};
    if (int rc = flush_predict(c)) return rc;
    if (int rc = do_update(c, nullptr, nullptr, 0, nullptr, 0, R, nullptr, nullptr, false)) return rc;
    c->rng_skew++;
    c->unplanned.has = false;  // the stage this launch leaves is a no-op: weights normalised, nothing to resample
    c->dist_clean = true;
    if (c->dist_fold) return launch_flags(c);  // no next launch to carry the barrier: a flag kernel closes the step
    return gather_totals(c);
}

// ---- all shards of a distributed run in ONE process (slam-backend -gpus k, rehearsals on one GPU) ------------------------
struct slamgpu_dist_group {
    std::vector<slamgpu_ctx *> ctx;
    std::vector<ncclComm_t> comms;  // one per context when every context has a device of its own
    bool shared = false;            // all contexts on one device and one stream: dist_gather_kernel instead of RCCL
    bool push = false;              // push collective in use (slamgpu_dist_step runs the flag barrier itself)
};

namespace {
int group_gather(slamgpu_dist_group *g) {
    const int k = (int) g->ctx.size();
    if (g->push) return 0;
    if (g->shared) {
        DistGatherArgs A{};
        A.n_shards = k;
        A.floats_per_shard = 2 * g->ctx[0]->ws.nblocks;
        for (int i = 0; i < k; i++) {
            const int par = (int) (g->ctx[i]->obs_step & 1);
            A.local[i] = g->ctx[i]->ws.blk_w[par];
            A.gathered[i] = g->ctx[i]->gtot_dev[par];
        }
        HIP_TRY(hipSetDevice(g->ctx[0]->cfg.device));
        g->ctx[0]->k->dist_gather(g->ctx[0]->stream, A);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    RCCL_TRY(rccl()->GroupStart());
    for (int i = 0; i < k; i++) {
        slamgpu_ctx *c = g->ctx[i];
        const int par = (int) (c->obs_step & 1);
        RCCL_TRY(rccl()->AllGather(c->ws.blk_w[par], c->gtot_dev[par], (size_t) 2 * c->ws.nblocks, ncclFloat, g->comms[i], c->stream));
    }
    RCCL_TRY(rccl()->GroupEnd());
    return 0;
}
}  // namespace

int slamgpu_dist_group_create(slamgpu_ctx **ctxs, int32_t k, slamgpu_dist_group **out) {
    if (!ctxs || !out || k < 1 || k > kMaxShards) return fail(SLAMGPU_ERR_INVALID, "bad context list");
    for (int i = 0; i < k; i++)
        if (!ctxs[i]) return fail(SLAMGPU_ERR_INVALID, "null context %d", i);
    bool same = true, distinct = true;
    for (int i = 0; i < k; i++)
        for (int j = 0; j < i; j++) {
            if (ctxs[i]->cfg.device == ctxs[j]->cfg.device) distinct = false;
            else same = false;
        }
    if (k > 1 && !same && !distinct) return fail(SLAMGPU_ERR_INVALID, "contexts must all share one device or each have a device of its own");
    const bool shared = k > 1 ? same : true;
    if (shared)
        for (int i = 1; i < k; i++)
            if (ctxs[i]->stream != ctxs[0]->stream)
                return fail(SLAMGPU_ERR_INVALID, "contexts sharing a device must share a stream (slamgpu_config.external_stream)");
    std::vector<DistBlob> blobs((size_t) k);
    for (int i = 0; i < k; i++)
        if (int rc = slamgpu_dist_export(ctxs[i], &blobs[(size_t) i])) return rc;
    for (int i = 0; i < k; i++)
        if (int rc = slamgpu_dist_connect(ctxs[i], k, i, blobs.data())) return rc;
    auto *g = new slamgpu_dist_group();
    g->ctx.assign(ctxs, ctxs + k);
    g->shared = shared;
    if (!shared) {
        // every shard has a device (hence hardware queues) of its own: try the push collective first -- a few barriers with
        // everybody's queued before anybody is waited for -- and keep the RCCL all-gather for when a peer does not arrive
        bool push = getenv("SLAMGPU_GROUP_NO_PUSH") == nullptr;
        for (int i = 0; i < k && push; i++) push = slamgpu_dist_handshake_test(ctxs[i], 5, nullptr, nullptr) == 0;
        for (int i = 0; i < k; i++) {
            int32_t ok = 0;
            if (slamgpu_dist_collective_status(ctxs[i], &ok) != 0 || !ok) push = false;
        }
        if (push) {
            for (int i = 0; i < k; i++)
                if (int rc = slamgpu_dist_set_collective(ctxs[i], SLAMGPU_DIST_PUSH)) {
                    delete g;
                    return rc;
                }
            g->push = true;
            *out = g;
            return 0;
        }
        if (!rccl()) {
            delete g;
            return fail(SLAMGPU_ERR_HIP, "librccl.so.1 not found: %s", dlerror());
        }
        std::vector<int> devs((size_t) k);
        for (int i = 0; i < k; i++) devs[(size_t) i] = ctxs[i]->cfg.device;
        g->comms.resize((size_t) k);
        ncclResult_t r = rccl()->CommInitAll(g->comms.data(), k, devs.data());
        if (r != ncclSuccess) {
            delete g;
            return fail(SLAMGPU_ERR_HIP, "ncclCommInitAll: %s", rccl()->GetErrorString(r));
        }
    }
    *out = g;
    return 0;
}

void slamgpu_dist_group_destroy(slamgpu_dist_group *g) {
    if (!g) return;
    for (slamgpu_ctx *c : g->ctx) {
        (void) hipSetDevice(c->cfg.device);
        (void) hipStreamSynchronize(c->stream);
    }
    for (ncclComm_t c : g->comms) (void) rccl()->CommDestroy(c);
    delete g;
}

int slamgpu_dist_group_step(slamgpu_dist_group *g, const float *controls, int32_t n_controls, const float Q[4], float dt, const float *zf,
                            const int32_t *idf, int32_t m, const float *zn, int32_t n, const float R[4], int32_t record_estimate) {
    if (!g) return fail(SLAMGPU_ERR_INVALID, "null group");
    for (slamgpu_ctx *c : g->ctx)
        if (int rc = slamgpu_dist_step(c, controls, n_controls, Q, dt, zf, idf, m, zn, n, R, record_estimate)) return rc;
    return group_gather(g);
}

int slamgpu_dist_group_settle(slamgpu_dist_group *g) {
    if (!g) return fail(SLAMGPU_ERR_INVALID, "null group");
    for (slamgpu_ctx *c : g->ctx)
        if (int rc = slamgpu_dist_settle(c)) return rc;
    if (int rc = group_gather(g)) return rc;
    for (slamgpu_ctx *c : g->ctx) {  // reads of one shard (flatten) follow: every shard's launch must have finished
        HIP_TRY(hipSetDevice(c->cfg.device));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (slamgpu_ctx *c : g->ctx)
        if (int rc = barrier_check(c)) return rc;
    return 0;
}

int slamgpu_dist_group_history(slamgpu_dist_group *g, double *xyt, float *neff, int32_t *resampled, int32_t *status, int32_t max_count,
                               int32_t *count) {
    if (!g || !count) return fail(SLAMGPU_ERR_INVALID, "null group / count");
    if (int rc = slamgpu_dist_group_settle(g)) return rc;
    const int k = (int) g->ctx.size();
    int nmin = std::max(max_count, 0);
    for (slamgpu_ctx *c : g->ctx) nmin = std::min(nmin, c->hist_n);
    std::vector<double> raw((size_t) 4 * std::max(nmin, 1)), best((size_t) std::max(nmin, 1), -1.0e300);
    if (xyt) std::fill(xyt, xyt + 3 * (size_t) nmin, 0.0);
    const double n_all = (double) g->ctx[0]->B.n * k;
    for (int i = 0; i < k; i++) {
        int32_t got = 0;
        // (Neff / decision / status of a stage are the same on every shard: the last shard's copy stays)
        if (int rc = slamgpu_dist_history_fetch(g->ctx[i], raw.data(), neff, resampled, status, nmin, &got)) return rc;
        if (got != nmin) return fail(SLAMGPU_ERR_INVALID, "shard %d recorded %d steps, expected %d", i, got, nmin);
        if (xyt)
            for (int t = 0; t < nmin; t++) {
                xyt[3 * t] += raw[4 * (size_t) t];
                xyt[3 * t + 1] += raw[4 * (size_t) t + 1];
                if (raw[4 * (size_t) t + 3] > best[(size_t) t]) {  // strict: ties keep the lowest global index
                    best[(size_t) t] = raw[4 * (size_t) t + 3];
                    xyt[3 * t + 2] = raw[4 * (size_t) t + 2];
                }
            }
    }
    if (xyt)
        for (int t = 0; t < nmin; t++) {
            xyt[3 * t] /= n_all;
            xyt[3 * t + 1] /= n_all;
        }
    *count = nmin;
    return 0;
}

int slamgpu_dist_group_download(slamgpu_dist_group *g, float *xv, float *Pv9, float *w, float *xf, float *Pf4) {
    if (!g) return fail(SLAMGPU_ERR_INVALID, "null group");
    if (int rc = slamgpu_dist_group_settle(g)) return rc;
    const size_t n = (size_t) g->ctx[0]->B.n, nf = (size_t) g->ctx[0]->nf;
    for (size_t i = 0; i < g->ctx.size(); i++)
        if (int rc = slamgpu_download(g->ctx[i], xv ? xv + 3 * n * i : nullptr, Pv9 ? Pv9 + 9 * n * i : nullptr, w ? w + n * i : nullptr,
                                      xf ? xf + 2 * nf * n * i : nullptr, Pf4 ? Pf4 + 4 * nf * n * i : nullptr))
            return rc;
    for (slamgpu_ctx *c : g->ctx) {  // a shard's flatten reads its peers: nobody may step before everybody has read
        HIP_TRY(hipSetDevice(c->cfg.device));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return 0;
}

int slamgpu_dev_alloc(slamgpu_ctx *c, uint64_t bytes, void **ptr) {
    if (int rc = check_ctx(c)) return rc;
    if (!ptr) return fail(SLAMGPU_ERR_INVALID, "null pointer");
    HIP_TRY(hipSetDevice(c->cfg.device));
    hipError_t e = hipMalloc(ptr, bytes ? bytes : 4);
    if (e != hipSuccess) return fail(SLAMGPU_ERR_ALLOC, "hipMalloc(%llu): %s", (unsigned long long) bytes, hipGetErrorString(e));
    return 0;
}

int slamgpu_dev_free(slamgpu_ctx *c, void *ptr) {
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (ptr) HIP_TRY(hipFree(ptr));
    return 0;
}

int slamgpu_dev_copy(slamgpu_ctx *c, void *dst, const void *src, uint64_t bytes) {
    if (int rc = check_ctx(c)) return rc;
    if (bytes == 0) return 0;
    if (!dst || !src) return fail(SLAMGPU_ERR_INVALID, "null pointer");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int slamgpu_dev_copy_async(slamgpu_ctx *c, void *dst, const void *src, uint64_t bytes) {
    if (int rc = check_ctx(c)) return rc;
    if (bytes == 0) return 0;
    if (!dst || !src) return fail(SLAMGPU_ERR_INVALID, "null pointer");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int slamgpu_shard_estimate_async(slamgpu_ctx *c) {
    if (int rc = check_ctx(c)) return rc;
    if (c->hist_n >= kHistCap) return fail(SLAMGPU_ERR_CAPACITY, "estimate history full (%d)", kHistCap);
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = flush_predict(c)) return rc;
    if (c->shard_est_fresh) {
        // nothing moved since shard_finalize_kernel left this update's partials: their one-block reduction rides in the
        // next update launch (or runs when the history is fetched)
        if (int rc = flush_stages(c)) return rc;  // (an older reduction still outstanding)
        c->unreduced.has = true;
        c->unreduced.par = 0;
        c->unreduced.hist = c->hist_dev + kHistStride * (size_t) c->hist_n;
        c->shard_est_fresh = false;
    } else {
        if (int rc = materialize(c)) return rc;
        if (int rc = flush_stages(c)) return rc;
        Timed t(c, "estimate");
        c->k->estimate(c->stream, c->B, c->ws, c->hist_dev + kHistStride * (size_t) c->hist_n);
    }
    c->hist_n++;
    HIP_TRY(hipGetLastError());
    return 0;
}

int slamgpu_shard_estimate_fetch(slamgpu_ctx *c, double *raw4, int32_t max_count, int32_t *count) {
    if (int rc = check_ctx(c)) return rc;
    if (!count) return fail(SLAMGPU_ERR_INVALID, "null count");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = flush_stages(c)) return rc;
    std::vector<double> h;
    if (int rc = history_to_host(c, h)) return rc;
    const int n = c->hist_n < std::max(max_count, 0) ? c->hist_n : std::max(max_count, 0);
    if (raw4)
        for (int i = 0; i < n; i++)
            for (int k = 0; k < 4; k++) raw4[4 * (size_t) i + k] = h[(size_t) kHistStride * i + k];
    *count = n;
    if (int rc = keep_history_tail(c, h, n)) return rc;
    c->est_fresh = false;
    return 0;
}

int slamgpu_shard_estimate(slamgpu_ctx *c, double out[4]) {
    if (int rc = check_ctx(c)) return rc;
    if (!out) return fail(SLAMGPU_ERR_INVALID, "null output");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = flush_predict(c)) return rc;
    if (int rc = materialize(c)) return rc;
    if (int rc = flush_stages(c)) return rc;
    {
        Timed t(c, "estimate");
        c->k->estimate(c->stream, c->B, c->ws, nullptr);
    }
    HIP_TRY(hipGetLastError());
    if (int rc = read_ctrl(c)) return rc;
    for (int i = 0; i < 4; i++) out[i] = c->ctrl_host->est[i];
    return 0;
}

int slamgpu_estimate(slamgpu_ctx *c, double xyt[3]) {
    if (int rc = check_ctx(c)) return rc;
    if (!xyt) return fail(SLAMGPU_ERR_INVALID, "null output");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = flush_predict(c)) return rc;
    if (!c->est_fresh) {
        if (int rc = materialize(c)) return rc;
        if (int rc = flush_stages(c)) return rc;
        Timed t(c, "estimate");
        c->k->estimate(c->stream, c->B, c->ws, nullptr);
    }
    HIP_TRY(hipGetLastError());
    if (int rc = read_ctrl(c)) return rc;
    xyt[0] = c->ctrl_host->est[0] / (double) c->B.n;
    xyt[1] = c->ctrl_host->est[1] / (double) c->B.n;
    xyt[2] = c->ctrl_host->est[2];
    return 0;
}

int slamgpu_estimate_async(slamgpu_ctx *c) {
    if (int rc = check_ctx(c)) return rc;
    if (c->hist_n >= kHistCap) return fail(SLAMGPU_ERR_CAPACITY, "estimate history full (%d): call slamgpu_estimate_fetch", kHistCap);
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = flush_predict(c)) return rc;
    if (!c->est_fresh) {
        // the particle set changed since the last update (predicts / upload): reduce it now
        if (int rc = materialize(c)) return rc;
        if (int rc = flush_stages(c)) return rc;
        Timed t(c, "estimate");
        c->k->estimate(c->stream, c->B, c->ws, c->hist_dev + kHistStride * (size_t) c->hist_n);
        HIP_TRY(hipGetLastError());
    }
    // else: slot hist_n is filled when the partials of the last update are reduced (next update launch / finish)
    c->hist_n++;
    c->est_fresh = false;
    return 0;
}

int slamgpu_history_fetch(slamgpu_ctx *c, double *xyt, float *neff, int32_t *resampled, int32_t *status, int32_t max_count,
                          int32_t *count) {
    if (int rc = check_ctx(c)) return rc;
    if (!count) return fail(SLAMGPU_ERR_INVALID, "null count");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = flush_stages(c)) return rc;
    std::vector<double> h;
    if (int rc = history_to_host(c, h)) return rc;
    if (int rc = persist_check(c)) return rc;
    const int n = c->hist_n < std::max(max_count, 0) ? c->hist_n : std::max(max_count, 0);
    for (int i = 0; i < n; i++) {
        const double *e = h.data() + (size_t) kHistStride * i;
        if (xyt) {
            xyt[3 * i] = e[0] / (double) c->B.n;
            xyt[3 * i + 1] = e[1] / (double) c->B.n;
            xyt[3 * i + 2] = e[2];
        }
        if (neff) neff[i] = (float) e[4];
        if (resampled) resampled[i] = ((int32_t) e[5]) & 1;
        if (status) status[i] = ((int32_t) e[5]) >> 1;
    }
    *count = n;
    if (int rc = keep_history_tail(c, h, n)) return rc;
    c->est_fresh = false;
    return 0;
}

int slamgpu_dist_history_fetch(slamgpu_ctx *c, double *raw4, float *neff, int32_t *resampled, int32_t *status, int32_t max_count,
                               int32_t *count) {
    if (int rc = check_ctx(c)) return rc;
    if (!count) return fail(SLAMGPU_ERR_INVALID, "null count");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = flush_stages(c)) return rc;
    std::vector<double> h;
    if (int rc = history_to_host(c, h)) return rc;
    const int n = c->hist_n < std::max(max_count, 0) ? c->hist_n : std::max(max_count, 0);
    for (int i = 0; i < n; i++) {
        const double *e = h.data() + (size_t) kHistStride * i;
        if (raw4)
            for (int k = 0; k < 4; k++) raw4[4 * (size_t) i + k] = e[k];
        if (neff) neff[i] = (float) e[4];
        if (resampled) resampled[i] = ((int32_t) e[5]) & 1;
        if (status) status[i] = ((int32_t) e[5]) >> 1;
    }
    *count = n;
    if (int rc = keep_history_tail(c, h, n)) return rc;
    c->est_fresh = false;
    return barrier_check(c);
}

int slamgpu_estimate_fetch(slamgpu_ctx *c, double *xyt, int32_t max_count, int32_t *count) {
    return slamgpu_history_fetch(c, xyt, nullptr, nullptr, nullptr, max_count, count);
}

int slamgpu_stats(slamgpu_ctx *c, float *neff, int32_t *resampled, double *weight_sum) {
    if (int rc = check_ctx(c)) return rc;
    if (int rc = read_ctrl(c)) return rc;
    if (neff) *neff = c->ctrl_host->neff;
    if (resampled) *resampled = c->ctrl_host->resampled;
    if (weight_sum) *weight_sum = c->cfg.log_weights ? c->ctrl_host->wmax + log(c->ctrl_host->wsum) : c->ctrl_host->wsum;
    return 0;
}

int slamgpu_step_status(slamgpu_ctx *c, int32_t *status) {
    if (int rc = check_ctx(c)) return rc;
    if (!status) return fail(SLAMGPU_ERR_INVALID, "null output");
    if (int rc = read_ctrl(c)) return rc;
    *status = c->ctrl_host->status | (c->front_status & kStatusCapacity);
    return 0;
}

int slamgpu_ancestors(slamgpu_ctx *c, int32_t *keep) {
    if (int rc = check_ctx(c)) return rc;
    if (!keep) return fail(SLAMGPU_ERR_INVALID, "null output");
    if (int rc = read_ctrl(c)) return rc;
    if (!c->ctrl_host->resampled) {
        for (int i = 0; i < c->B.n; i++) keep[i] = i;
        return 0;
    }
    HIP_TRY(hipMemcpy(keep, c->ws.keep[c->keep_slot], sizeof(int32_t) * (size_t) c->B.n, hipMemcpyDeviceToHost));
    if (c->cfg.n_particles_global != c->cfg.n_particles) {
        // shard: device entries are local indices (>= 0) or -(global id + 1) for records that came from another shard
        for (int i = 0; i < c->B.n; i++) keep[i] = keep[i] >= 0 ? (int32_t) (c->cfg.first_particle + keep[i]) : -(keep[i] + 1);
    }
    return 0;
}

int slamgpu_set_map(slamgpu_ctx *c, const float *lm, int32_t nlm) {
    if (int rc = check_ctx(c)) return rc;
    if (!lm || nlm <= 0) return fail(SLAMGPU_ERR_INVALID, "empty map");
    if (int rc = book_pull(c)) return rc;  // (a new map starts a new landmark table: the book goes back to the host first)
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (void *p : {(void *) c->map_dev, (void *) c->obs_r_dev, (void *) c->table_dev, (void *) c->obs_out_dev})
        if (p) (void) hipFree(p);
    c->map_dev = c->obs_r_dev = nullptr;
    c->table_dev = nullptr;
    c->obs_out_dev = nullptr;
    const size_t n = (size_t) nlm;
    HIP_TRY(hipMalloc((void **) &c->map_dev, sizeof(float) * 2 * n));
    HIP_TRY(hipMalloc((void **) &c->obs_r_dev, sizeof(float) * 2 * n));
    HIP_TRY(hipMalloc((void **) &c->table_dev, sizeof(int32_t) * n));
    HIP_TRY(hipMalloc((void **) &c->obs_out_dev, sizeof(ObserveOut) + 4 * 8 * n));  // z 2n, vis n, zf 2n, idf n, zn 2n
    HIP_TRY(hipMemcpy(c->map_dev, lm, sizeof(float) * 2 * n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(c->table_dev, 0xff, sizeof(int32_t) * n, c->stream));  // -1: never seen
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->map_n = nlm;
    c->obs_nf = 0;
    c->map_host.assign(lm, lm + 2 * n);
    c->front_ready = false;  // (a new map starts a new landmark table)
    // a context that may run slamgpu_run_observe as the persistent loop gets that loop's buffers now, not inside its first call
    if (persist_eligible(c) && nlm <= kSmallObs - 1)
        if (int rc = persist_setup(c, 256)) return rc;
    return 0;
}

int slamgpu_observe(slamgpu_ctx *c, const float xtrue[3], float max_range, const float R[4], int32_t noise, const float *r1,
                    const float *r2, float *z, int32_t *vis, int32_t *nz, float *zf, int32_t *idf, int32_t *m, float *zn, int32_t *n) {
    if (int rc = check_ctx(c)) return rc;
    if (!c->map_dev) return fail(SLAMGPU_ERR_INVALID, "no map: call slamgpu_set_map first");
    if (!xtrue || !R || noise < 0 || noise > 2 || (noise == 1 && (!r1 || !r2))) return fail(SLAMGPU_ERR_INVALID, "bad arguments");
    HIP_TRY(hipSetDevice(c->cfg.device));
    const size_t nl = (size_t) c->map_n;
    if (noise == 1) {
        HIP_TRY(hipMemcpyAsync(c->obs_r_dev, r1, sizeof(float) * nl, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(c->obs_r_dev + nl, r2, sizeof(float) * nl, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // pageable sources
    }
    ObserveArgs A{};
    A.lm = c->map_dev;
    A.table = c->table_dev;
    A.nlm = c->map_n;
    A.nf = c->obs_nf;
    A.x = xtrue[0];
    A.y = xtrue[1];
    A.phi = xtrue[2];
    A.max_range = max_range;
    A.sr = sqrtf(R[0]);
    A.sb = sqrtf(R[3]);
    A.noise = noise;
    A.r1 = c->obs_r_dev;
    A.r2 = c->obs_r_dev + nl;
    A.k0 = (uint32_t) c->cfg.seed;
    A.k1 = (uint32_t) (c->cfg.seed >> 32);
    A.step = ++c->observe_step;
    A.out = c->obs_out_dev;
    {
        Timed t(c, "observe");
        c->k->observe(c->stream, A);
    }
    HIP_TRY(hipGetLastError());
    std::vector<char> host(sizeof(ObserveOut) + 4 * 8 * nl);
    HIP_TRY(hipMemcpyAsync(host.data(), c->obs_out_dev, host.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const ObserveOut *o = reinterpret_cast<const ObserveOut *>(host.data());
    const float *hz = reinterpret_cast<const float *>(o + 1);
    const int32_t *hvis = reinterpret_cast<const int32_t *>(hz + 2 * nl);
    const float *hzf = reinterpret_cast<const float *>(hvis + nl);
    const int32_t *hidf = reinterpret_cast<const int32_t *>(hzf + 2 * nl);
    const float *hzn = reinterpret_cast<const float *>(hidf + nl);
    if (z) memcpy(z, hz, sizeof(float) * 2 * (size_t) o->nz);
    if (vis) memcpy(vis, hvis, sizeof(int32_t) * (size_t) o->nz);
    if (zf) memcpy(zf, hzf, sizeof(float) * 2 * (size_t) o->m);
    if (idf) memcpy(idf, hidf, sizeof(int32_t) * (size_t) o->m);
    if (zn) memcpy(zn, hzn, sizeof(float) * 2 * (size_t) o->n);
    if (nz) *nz = o->nz;
    if (m) *m = o->m;
    if (n) *n = o->n;
    c->obs_nf = o->nf_after;
    return 0;
}

int slamgpu_associate(slamgpu_ctx *c, const float *z, int32_t nz, const float R[4], float gate_reject, float gate_augment,
                      int32_t *labels, int32_t *consensus, float *support) {
    return slamgpu_associate_ex(c, z, nz, R, gate_reject, gate_augment, SLAMGPU_ASSOC_AUTO, labels, consensus, support, nullptr);
}

namespace {
// the update's association is per step: one observation per landmark (the better-supported one wins)
void assoc_resolve(int nz, std::vector<int32_t> &best, const std::vector<double> &share, int32_t *consensus, float *support) {
    // (by landmark, not pairwise: 1.3 k observations per step on the 10 000-landmark map)
    std::map<int32_t, int> owner;
    for (int q = 0; q < nz; q++) {
        if (best[q] < 0) continue;
        auto it = owner.find(best[q]);
        if (it == owner.end()) {
            owner[best[q]] = q;
        } else if (share[q] > share[(size_t) it->second]) {
            best[(size_t) it->second] = SLAMGPU_ASSOC_DISCARD;
            it->second = q;
        } else {
            best[q] = SLAMGPU_ASSOC_DISCARD;
        }
    }
    for (int q = 0; q < nz; q++) {
        if (consensus) consensus[q] = best[q];
        if (support) support[q] = (float) share[q];
    }
}
}  // namespace

namespace {
// lab_ext: a device array the labels are left in BY OBSERVATION, [nz][ncap] (slamgpu_update_particle: they never visit the host), or null
int associate_impl(slamgpu_ctx *c, const float *z, int32_t nz, const float R[4], float gate_reject, float gate_augment, int32_t mode,
                   int32_t *labels, int32_t *consensus, float *support, double stats[4], int32_t *lab_ext, const float *excl3 = nullptr) {
    if (int rc = check_ctx(c)) return rc;
    if (mode < SLAMGPU_ASSOC_AUTO || mode > SLAMGPU_ASSOC_GRID) return fail(SLAMGPU_ERR_INVALID, "unknown association mode %d", mode);
    if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0.0;
    c->pp_census_done = false;
    if (nz < 0 || (nz > 0 && !z) || !R) return fail(SLAMGPU_ERR_INVALID, "bad observation list");
    if (nz == 0) return 0;
    static_assert(SLAMGPU_ASSOC_NEW == kAssocNew && SLAMGPU_ASSOC_DISCARD == kAssocDiscard, "public / device labels");
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = flush_predict(c)) return rc;
    if (int rc = materialize(c)) return rc;  // plain set: particle k in slot k
    if (int rc = sync_tables(c)) return rc;
    const int N = c->B.n;
    const bool single = !c->dist && c->cfg.n_particles_global == c->cfg.n_particles && c->pool_used == 0;
    if (mode == SLAMGPU_ASSOC_GRID && !single)
        return fail(SLAMGPU_ERR_INVALID, "the association grid needs a single context (shards: SLAMGPU_ASSOC_EXHAUSTIVE)");
    const bool excl = excl3 && excl3[0] + excl3[1] > 0.0f;  // (the exclusion rule: the exhaustive scan only)
    if (excl && mode == SLAMGPU_ASSOC_GRID) return fail(SLAMGPU_ERR_INVALID, "the exclusion rule needs SLAMGPU_ASSOC_EXHAUSTIVE (or _AUTO)");
    // (... and an exhaustive scan is O(N nz Nf): refused where it would run for seconds -- 10^5 particles x 865 observations x 10 000
    // landmarks would be 10^12 gate evaluations in one launch -- instead of looking like a hang)
    if (excl && (double) c->B.n * (double) nz * (double) c->nf > 4e10)
        return fail(SLAMGPU_ERR_CAPACITY, "the exclusion rule scans exhaustively: %d particles x %d observations x %d landmarks is too much for one launch; "
                    "turn it off (excl_base = excl_per_m = 0) on maps of this size", c->B.n, nz, c->nf);
    bool grid = single && !excl && (mode == SLAMGPU_ASSOC_GRID || (mode == SLAMGPU_ASSOC_AUTO && c->nf >= 64));
    const bool want_vote = consensus || support;
    float *z_dev = nullptr;
    int32_t *lab_dev = nullptr;
    VoteSlot *votes_dev = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int rc = 0;
    auto step = [&](hipError_t er, const char *what) {
        if (!rc && er != hipSuccess) rc = fail(er == hipErrorOutOfMemory ? SLAMGPU_ERR_ALLOC : SLAMGPU_ERR_HIP, "%s: %s", what, hipGetErrorString(er));
    };
    auto need_labels = [&]() {  // [N][nz] labels on the device: only when the caller wants them, or the vote is taken on the host
        if (!lab_dev && lab_ext) lab_dev = lab_ext;
        if (!lab_dev) step(hipMalloc((void **) &lab_dev, sizeof(int32_t) * (size_t) N * nz), "hipMalloc(labels)");
    };
    // (the per-particle update keeps the observations in a buffer of the context: no allocation per step)
    if (nz > c->assoc_nz_cap) {  // (the call's observations and vote tables live in buffers of the context)
        if (c->assoc_z_dev) (void) hipFree(c->assoc_z_dev);
        if (c->assoc_votes_dev) (void) hipFree(c->assoc_votes_dev);
        c->assoc_z_dev = nullptr;
        c->assoc_votes_dev = nullptr;
        c->assoc_nz_cap = 0;
        const int cap = std::max(64, 2 * nz);
        step(hipMalloc((void **) &c->assoc_z_dev, sizeof(float) * 2 * (size_t) cap), "hipMalloc");
        step(hipMalloc((void **) &c->assoc_votes_dev, sizeof(VoteSlot) * kVoteSlots * (size_t) cap), "hipMalloc");
        if (!rc) c->assoc_nz_cap = cap;
    }
    const bool z_own = false;
    z_dev = (lab_ext && c->pp_z_dev && nz <= c->pp_nz_cap) ? c->pp_z_dev : c->assoc_z_dev;
    step(hipMemcpyAsync(z_dev, z, sizeof(float) * 2 * (size_t) nz, hipMemcpyHostToDevice, c->stream), "H2D");
    step(hipStreamSynchronize(c->stream), "sync");
    if (stats) {
        step(hipEventCreate(&ev0), "event");
        step(hipEventCreate(&ev1), "event");
    }
    std::vector<int32_t> best((size_t) nz, SLAMGPU_ASSOC_DISCARD);
    std::vector<double> share((size_t) nz, 0.0);
    bool voted = false;
    c->B.slot = c->slot;
    if (!rc && grid) {
        // boxes of the landmarks written since the last call, then geometry + grid, then the pruned scan with the vote
        const int cap_nf = c->B.cap_nf;
        if (!c->box_dev) {
            // (a landmark whose estimates are scattered over the whole region lands in every cell: room for that on small
            // maps, 64 cells per landmark + 4 Mi entries on large ones; beyond it the exhaustive scan takes over)
            c->cap_items = (int32_t) std::min<int64_t>((int64_t) cap_nf * kAssocMaxCells * kAssocMaxCells, 64 * (int64_t) cap_nf + (4 << 20));
            step(hipMalloc((void **) &c->box_dev, sizeof(LmkBox) * (size_t) cap_nf), "hipMalloc");
            step(hipMalloc((void **) &c->assoc_ids_dev, sizeof(int32_t) * (size_t) cap_nf), "hipMalloc");
            step(hipMalloc((void **) &c->cell_start_dev, sizeof(int32_t) * (kAssocMaxCells * kAssocMaxCells + 1)), "hipMalloc");
            step(hipMalloc((void **) &c->cell_fill_dev, sizeof(int32_t) * (kAssocMaxCells * kAssocMaxCells)), "hipMalloc");
            step(hipMalloc((void **) &c->items_dev, sizeof(float4) * 2 * (size_t) c->cap_items), "hipMalloc");
            step(hipMalloc((void **) &c->geom_dev, sizeof(AssocGeom) + sizeof(float) * 8 * 64), "hipMalloc");  // (+ the partial boxes behind it)
        }
        std::vector<int32_t> ids;
        for (int j = 0; j < c->nf; j++)
            if (c->box_dirty[j]) ids.push_back(j);
        if (!rc && !ids.empty()) {
            step(hipMemcpyAsync(c->assoc_ids_dev, ids.data(), sizeof(int32_t) * ids.size(), hipMemcpyHostToDevice, c->stream), "H2D");
            step(hipStreamSynchronize(c->stream), "sync");  // (pageable source)
        }
        if (labels || lab_ext) need_labels();
        if (want_vote) {
            votes_dev = c->assoc_votes_dev;
            // key = kVoteEmpty (0x80000000), weight = -0.0f (the same bits): -0.0 + w = w
            if (!rc) step(hipMemsetD32Async((hipDeviceptr_t) votes_dev, (int) 0x80000000, 2 * kVoteSlots * (size_t) nz, c->stream), "memset");
        }
        AssocGridArgs G{};
        G.box = c->box_dev;
        G.geom = c->geom_dev;
        G.geom_part = reinterpret_cast<float *>(c->geom_dev + 1);
        G.cell_start = c->cell_start_dev;
        G.cell_fill = c->cell_fill_dev;
        G.items = c->items_dev;
        G.cap_items = c->cap_items;
        G.nf = c->nf;
        G.nz = nz;
        G.z = z_dev;
        G.r00 = R[0];
        G.r11 = R[3];
        G.G = std::max(gate_reject, gate_augment) * 1.001f;
        G.G1 = std::min(gate_reject * 1.001f, G.G);
        // observations per thread: as many as still leave ~2 000 workgroups (a workgroup's head -- Ctrl word, pose, geometry: dependent
        // trips -- is paid once per group: at config 5, 10^5 particles x 865 observations, groups of 1 / 4 / 32 / 128 take 16.9 / 5.1 / 3.6 /
        // 3.5 ms; round 5's fixed 4 dated from the grid's long walks)
        G.obs_per_block = (int) std::min<int64_t>(64, std::max<int64_t>(kAssocObsPerBlock, (int64_t) (c->B.ncap / kBlock) * nz / 2048));
        if (const char *e = getenv("SLAMGPU_ASSOC_OBS_PER_BLOCK")) G.obs_per_block = std::max(1, atoi(e));  // (diagnostic)
        G.votes = votes_dev;
        G.logw = c->cfg.log_weights;
        G.lab_by_obs = lab_ext ? 1 : 0;
        if (lab_ext && c->pp_tab_dev && nz <= c->pp_nz_cap) {
            // the per-particle update's census rides in this launch: first | holders | news of the context's table (do_update_particle's layout)
            G.census_first = c->pp_tab_dev;
            G.census_news = c->pp_tab_dev + 2 * (size_t) c->B.cap_nf;
            step(hipMemsetD32Async((hipDeviceptr_t) G.census_first, 0x7fffffff, (size_t) c->B.cap_nf, c->stream), "memset");
            step(hipMemsetAsync(G.census_news, 0, sizeof(int32_t) * (size_t) nz, c->stream), "memset");
        }
        AssocGeom hg{};
        // few enough observations: a candidate list per observation instead of the grid (kernels.hip: assoc_lists_kernel); a list that
        // does not fit its share of the entry buffer sends the call to the grid (SLAMGPU_NO_ASSOC_LISTS=1: always the grid -- A/B, tests)
        int lcap = (nz > 0 && nz <= kAssocMaxCells * kAssocMaxCells && (double) nz * (double) c->nf <= 4e8 && getenv("SLAMGPU_NO_ASSOC_LISTS") == nullptr)
                       ? (int) std::min<int64_t>(2048, (int64_t) c->cap_items / nz) : 0;
        bool lcap_forced = false;
        if (const char *e = getenv("SLAMGPU_ASSOC_LCAP")) {  // (tests: lists too short -> the call must take the grid by itself)
            lcap = std::min(lcap, std::max(1, atoi(e)));
            lcap_forced = lcap >= 1;
        }
        if (!rc) {
            if (ev0) step(hipEventRecord(ev0, c->stream), "event");
            Timed t(c, "associate");
            if (!ids.empty()) c->k->lmk_box(c->stream, c->B, c->assoc_ids_dev, (int) ids.size(), c->retired_dev, c->box_dev);
            for (int attempt = (lcap >= 16 || lcap_forced) ? 0 : 1; attempt < 2 && !rc; attempt++) {
                G.lcap = attempt == 0 ? lcap : 0;
                G.vote_w = nullptr;
                if (G.lcap && want_vote) {  // the lists' votes are addressed directly: [nz][lcap + 2] weights, compacted into `votes` afterwards
                    const size_t words = (size_t) nz * ((size_t) G.lcap + 2);
                    if (words > c->vote_w_cap) {
                        if (c->vote_w_dev) (void) hipFree(c->vote_w_dev);
                        c->vote_w_dev = nullptr;
                        c->vote_w_cap = 0;
                        step(hipMalloc((void **) &c->vote_w_dev, sizeof(float) * words), "hipMalloc(votes)");
                        if (!rc) c->vote_w_cap = words;
                    }
                    if (!rc) step(hipMemsetAsync(c->vote_w_dev, 0, sizeof(float) * words, c->stream), "memset");
                    if (rc) break;
                    G.vote_w = c->vote_w_dev;
                }
                // (the first pass bounded by the match gate pays on the lists' short walks; on the grid's long ones it cost more than
                // it saved -- 8.52 against 8.32 ms at config 5 --: there the one wide pass)
                G.G1 = G.lcap ? std::min(gate_reject * 1.001f, G.G) : G.G;
                G.g1_ratio = sqrtf(G.G1 / G.G);
                if (G.lcap) c->k->assoc_lists(c->stream, c->B, G);
                else c->k->assoc_grid(c->stream, c->B, G);
                c->k->associate_grid(c->stream, c->B, G, R, gate_reject, gate_augment, lab_dev);
                if (G.vote_w) c->k->vote_compact(c->stream, G);
                step(hipGetLastError(), "launch");
                if (attempt == 0) {  // (did every list fit?  the association kernel has left at once if not)
                    step(hipMemcpyAsync(&hg, c->geom_dev, sizeof hg, hipMemcpyDeviceToHost, c->stream), "D2H");
                    step(hipStreamSynchronize(c->stream), "sync");
                    if (!(hg.overflow & 1)) break;
                }
            }
        }
        step(hipGetLastError(), "launch");
        if (ev1) step(hipEventRecord(ev1, c->stream), "event");
        step(hipMemcpyAsync(&hg, c->geom_dev, sizeof hg, hipMemcpyDeviceToHost, c->stream), "D2H");
        step(hipStreamSynchronize(c->stream), "sync");
        if (!rc) {
            for (int j : ids) c->box_dirty[j] = 0;
            if (hg.overflow & 1) {
                if (mode == SLAMGPU_ASSOC_GRID) rc = fail(SLAMGPU_ERR_CAPACITY, "association grid: %d entries exceed the buffer of %d", hg.total, c->cap_items);
                grid = false;  // (auto: the exhaustive scan instead)
            } else {
                c->pp_census_done = G.census_first != nullptr;
                if (stats) {
                    stats[0] = (double) hg.pairs;
                    stats[1] = (double) hg.total;
                    stats[3] = 1.0;
                }
                if (want_vote && !(hg.overflow & 2)) {
                    // the vote was taken on the device: one small table per observation comes back
                    std::vector<VoteSlot> tab((size_t) kVoteSlots * nz);
                    step(hipMemcpy(tab.data(), votes_dev, sizeof(VoteSlot) * tab.size(), hipMemcpyDeviceToHost), "D2H");
                    for (int q = 0; q < nz && !rc; q++) {
                        double wsum = 0, bw = -1;
                        int32_t bk = SLAMGPU_ASSOC_DISCARD;
                        for (int p = 0; p < kVoteSlots; p++) {
                            const VoteSlot &v = tab[(size_t) q * kVoteSlots + p];
                            if (v.key == kVoteEmpty) continue;
                            wsum += (double) v.w;
                            // (the largest share; ties go to the smaller label, as the ordered map of the host vote does)
                            if ((double) v.w > bw || ((double) v.w == bw && v.key < bk)) {
                                bw = (double) v.w;
                                bk = v.key;
                            }
                        }
                        best[(size_t) q] = bk;
                        share[(size_t) q] = wsum > 0 ? bw / wsum : 0.0;
                    }
                    voted = true;
                } else if (want_vote) {
                    // (more than 32 distinct labels for one observation: take the vote on the host from the full label array)
                    need_labels();
                    if (!rc) {
                        G.votes = nullptr;  // (G.lcap: whichever of lists / grid the call ended with)
                        c->k->associate_grid(c->stream, c->B, G, R, gate_reject, gate_augment, lab_dev);
                        step(hipGetLastError(), "launch");
                    }
                }
            }
        }
    }
    if (!rc && !grid) {
        need_labels();
        if (!rc) {
            if (ev0) step(hipEventRecord(ev0, c->stream), "event");
            {
                Timed t(c, "associate");
                c->k->associate(c->stream, c->B, c->nf, z_dev, nz, R, gate_reject, gate_augment, excl3, c->retired_dev, lab_dev, lab_ext ? 1 : 0);
            }
            if (ev1) step(hipEventRecord(ev1, c->stream), "event");
            if (stats) stats[0] = (double) N * (double) nz * (double) c->nf;
            step(hipGetLastError(), "launch");
        }
    }
    std::vector<int32_t> lab;
    if (!rc && lab_dev && (labels || (want_vote && !voted))) {
        lab.resize((size_t) N * nz);
        step(hipMemcpyAsync(lab.data(), lab_dev, sizeof(int32_t) * lab.size(), hipMemcpyDeviceToHost, c->stream), "D2H");
    }
    step(hipStreamSynchronize(c->stream), "sync");
    if (!rc && stats && ev0 && ev1) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) stats[2] = ms;
    }
    if (ev0) (void) hipEventDestroy(ev0);
    if (ev1) (void) hipEventDestroy(ev1);
    if (z_dev && z_own) (void) hipFree(z_dev);
    if (lab_dev && lab_dev != lab_ext) (void) hipFree(lab_dev);
    if (rc) return rc;
    if (labels) memcpy(labels, lab.data(), sizeof(int32_t) * lab.size());
    if (want_vote && !voted) {
        if (int rc2 = read_ctrl(c, true)) return rc2;
        const int cur = c->ctrl_host->live[c->slot];
        std::vector<float4> pa((size_t) N);
        HIP_TRY(hipMemcpy(pa.data(), c->B.poseA[cur], sizeof(float4) * (size_t) N, hipMemcpyDeviceToHost));
        // weights (log-weight contexts: exp(l - max l)), normalised
        std::vector<double> w((size_t) N);
        double wmax = -1e300, wsum = 0;
        for (int i = 0; i < N; i++) wmax = std::max(wmax, (double) pa[i].w);
        for (int i = 0; i < N; i++) {
            w[i] = c->cfg.log_weights ? exp((double) pa[i].w - wmax) : (double) pa[i].w;
            wsum += w[i];
        }
        for (int q = 0; q < nz; q++) {
            std::map<int32_t, double> votes;
            for (int i = 0; i < N; i++) votes[lab[(size_t) i * nz + q]] += w[i];
            int32_t b = SLAMGPU_ASSOC_DISCARD;
            double bw = -1;
            for (auto &kv : votes)
                if (kv.second > bw) {
                    bw = kv.second;
                    b = kv.first;
                }
            best[(size_t) q] = b;
            share[(size_t) q] = wsum > 0 ? bw / wsum : 0.0;
        }
    }
    if (want_vote) assoc_resolve(nz, best, share, consensus, support);
    return 0;
}
}  // namespace

int slamgpu_associate_ex(slamgpu_ctx *c, const float *z, int32_t nz, const float R[4], float gate_reject, float gate_augment, int32_t mode,
                         int32_t *labels, int32_t *consensus, float *support, double stats[4]) {
    return associate_impl(c, z, nz, R, gate_reject, gate_augment, mode, labels, consensus, support, stats, nullptr);
}

namespace {
int retired_upload(slamgpu_ctx *c) {
    const size_t words = ((size_t) c->B.cap_nf + 31) / 32;
    if (c->retired.empty()) c->retired.assign((size_t) c->B.cap_nf, 0);
    if (!c->retired_dev) HIP_TRY(hipMalloc((void **) &c->retired_dev, sizeof(uint32_t) * words));
    std::vector<uint32_t> mask(words, 0u);
    for (int j = 0; j < c->B.cap_nf; j++)
        if (c->retired[(size_t) j]) mask[(size_t) j >> 5] |= 1u << (j & 31);
    HIP_TRY(hipStreamSynchronize(c->stream));  // (an association in flight may still read the old mask)
    HIP_TRY(hipMemcpy(c->retired_dev, mask.data(), sizeof(uint32_t) * words, hipMemcpyHostToDevice));
    c->retired_stale = false;
    return 0;
}
}  // namespace

int slamgpu_retire_landmarks(slamgpu_ctx *c, const int32_t *ids, int32_t count) {
    if (int rc = check_ctx(c)) return rc;
    if (count < 0 || (count > 0 && !ids)) return fail(SLAMGPU_ERR_INVALID, "slamgpu_retire_landmarks: bad list");
    if (c->dist || c->cfg.n_particles_global != c->cfg.n_particles) return fail(SLAMGPU_ERR_INVALID, "slamgpu_retire_landmarks: single contexts only");
    if (int rc = book_pull(c)) return rc;
    for (int32_t k = 0; k < count; k++)
        if (ids[k] < 0 || ids[k] >= c->nf) return fail(SLAMGPU_ERR_INVALID, "slamgpu_retire_landmarks: landmark %d of %d", (int) ids[k], c->nf);
    if (count == 0) return 0;
    HIP_TRY(hipSetDevice(c->cfg.device));
    const size_t words = ((size_t) c->B.cap_nf + 31) / 32;
    if (c->retired.empty()) c->retired.assign((size_t) c->B.cap_nf, 0);
    if (!c->retired_dev) HIP_TRY(hipMalloc((void **) &c->retired_dev, sizeof(uint32_t) * words));
    for (int32_t k = 0; k < count; k++) {
        if (!c->retired[(size_t) ids[k]]) c->n_retired++;
        c->retired[(size_t) ids[k]] = 1;
        c->box_dirty[(size_t) ids[k]] = 1;  // (its box becomes the empty one at the next grid call)
    }
    return retired_upload(c);
}

namespace {
int pp_reserve(slamgpu_ctx *c, int nz, size_t rows) {
    const size_t S = (size_t) c->B.ncap, cap_nf = (size_t) c->B.cap_nf;
    if (S * (size_t) nz > c->pp_lab_cap) {  // (by observation: [nz][ncap])
        if (c->pp_lab_dev) (void) hipFree(c->pp_lab_dev);
        c->pp_lab_dev = nullptr;
        c->pp_lab_cap = 0;
        HIP_TRY(hipMalloc((void **) &c->pp_lab_dev, sizeof(int32_t) * S * (size_t) nz));
        c->pp_lab_cap = S * (size_t) nz;
    }
    if (nz > c->pp_nz_cap || !c->pp_tab_dev) {
        const int cap = std::max(64, 2 * nz);
        if (c->pp_z_dev) (void) hipFree(c->pp_z_dev);
        if (c->pp_tab_dev) (void) hipFree(c->pp_tab_dev);
        c->pp_z_dev = nullptr;
        c->pp_tab_dev = nullptr;
        c->pp_nz_cap = 0;
        HIP_TRY(hipMalloc((void **) &c->pp_z_dev, sizeof(float) * 2 * (size_t) cap));
        HIP_TRY(hipMalloc((void **) &c->pp_tab_dev, sizeof(int32_t) * (2 * cap_nf + 2 * (size_t) cap)));
        c->pp_nz_cap = cap;
    }
    if (rows > c->pp_obs_rows) {
        const size_t want = std::max<size_t>(rows, 2 * c->pp_obs_rows);
        if (c->pp_obs_dev) (void) hipFree(c->pp_obs_dev);
        c->pp_obs_dev = nullptr;
        c->pp_obs_rows = 0;
        HIP_TRY(hipMalloc((void **) &c->pp_obs_dev, sizeof(int16_t) * want * S));
        c->pp_obs_rows = want;
    }
    if (!c->pp_wf_dev) HIP_TRY(hipMalloc((void **) &c->pp_wf_dev, sizeof(float) * S));
    if (!c->pp_any_dev) HIP_TRY(hipMalloc((void **) &c->pp_any_dev, S));
    return 0;
}

int pp_check(slamgpu_ctx *c, const float *z, int32_t nz, const float R[4], const slamgpu_particle_assoc *opt) {
    if (int rc = check_ctx(c)) return rc;
    if (!opt || !R || nz < 0 || (nz > 0 && !z)) return fail(SLAMGPU_ERR_INVALID, "per-particle association: bad arguments");
    if (nz > 32767) return fail(SLAMGPU_ERR_CAPACITY, "per-particle association: %d observations in one step (at most 32767)", nz);
    if (!(opt->p_new > 0.0f) || !(opt->new_share >= 0.0f && opt->new_share <= 1.0f) || opt->census_every < 0)
        return fail(SLAMGPU_ERR_INVALID, "per-particle association: p_new > 0, 0 <= new_share <= 1, census_every >= 0");
    if (!(opt->excl_base >= 0.0f) || !(opt->excl_per_m >= 0.0f) || !(opt->unique_ratio >= 0.0f))
        return fail(SLAMGPU_ERR_INVALID, "per-particle association: excl_base, excl_per_m, unique_ratio >= 0");
    if (c->dist || c->cfg.n_particles_global != c->cfg.n_particles || c->pool_used != 0)
        return fail(SLAMGPU_ERR_INVALID, "per-particle association: single contexts only");
    if (c->mid_compact)
        if (int rc = demote_to_plain(c)) return rc;
    if (c->B.compact)
        return fail(SLAMGPU_ERR_INVALID, "per-particle association needs plain genealogy rows: create the context with SLAMGPU_FLAG_PARTICLE_MAPS");
    return 0;
}

// One observation step in which every particle acts on ITS OWN association (kernels.h: PerParticle): labels [N][nz] are in
// c->pp_lab_dev (landmark slot, SLAMGPU_ASSOC_NEW or _DISCARD per particle and observation).
int do_update_particle(slamgpu_ctx *c, const float *z, int32_t nz, const float R[4], const slamgpu_particle_assoc *opt, const float *normals,
                       const float *strata, int32_t report[8]) {
    if (report) memset(report, 0, sizeof(int32_t) * 8);
    if (int rc = book_pull(c)) return rc;
    if (int rc = flush_predict(c)) return rc;
    if (int rc = materialize(c)) return rc;  // plain set: particle k in slot k (the holders census reads it through the genealogy)
    if (int rc = sync_tables(c)) return rc;
    const bool tape = c->cfg.rng_mode == SLAMGPU_RNG_TAPE;
    if (tape && !strata) return fail(SLAMGPU_ERR_INVALID, "TAPE mode needs strata[N] (and normals[3N] for FastSLAM2)");
    const int N = c->B.n, cap_nf = c->B.cap_nf, nf0 = c->nf;
    c->B.slot = c->slot;
    int32_t *first_dev = c->pp_tab_dev, *hold_dev = first_dev + cap_nf, *news_dev = hold_dev + cap_nf, *idn_dev = news_dev + c->pp_nz_cap;
    if (c->pp_partial.empty()) c->pp_partial.assign((size_t) cap_nf, 0);
    if (c->pp_dead.empty()) c->pp_dead.assign((size_t) cap_nf, 0);
    std::vector<int32_t> partial;  // the slots whose holders are counted: partial ones that are not dead already
    const bool due = opt->census_every > 0 && nf0 > 0 && (c->pp_steps % (uint64_t) opt->census_every) == 0;
    if (due)
        for (int l = 0; l < nf0; l++)
            if (c->pp_partial[(size_t) l] && !c->pp_dead[(size_t) l]) partial.push_back(l);
    const bool census = due && !partial.empty();
    c->pp_steps++;
    // (first | holders | news are one stretch of the table: one copy down)
    std::vector<int32_t> down(2 * (size_t) cap_nf + (size_t) nz);
    int32_t *const first = down.data(), *const hold = first + cap_nf, *const news = hold + cap_nf;
    HIP_TRY(hipMemcpyAsync(c->pp_z_dev, z, sizeof(float) * 2 * (size_t) nz, hipMemcpyHostToDevice, c->stream));
    const bool census_taken = c->pp_census_done;  // (by the association kernel itself: slamgpu_update_particle through the grid / lists)
    c->pp_census_done = false;
    if (!census_taken) {
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t) first_dev, 0x7fffffff, (size_t) cap_nf, c->stream));
        HIP_TRY(hipMemsetAsync(news_dev, 0, sizeof(int32_t) * (size_t) nz, c->stream));
    }
    HIP_TRY(hipMemsetAsync(hold_dev, 0, sizeof(int32_t) * (size_t) cap_nf, c->stream));
    if (census) {  // (the list rides in the words the new slots' ids will take later in this call)
        if ((int) partial.size() > c->pp_nz_cap) partial.resize((size_t) c->pp_nz_cap);  // (more partial slots than the table has words: the rest next time)
        HIP_TRY(hipMemcpyAsync(idn_dev, partial.data(), sizeof(int32_t) * partial.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // (pageable source)
    }
    {
        Timed t(c, "particle_census");
        if (!census_taken) c->k->pp_census(c->stream, c->pp_lab_dev, N, nz, c->B.ncap, first_dev, news_dev);
        if (census) c->k->pp_holders(c->stream, c->B, (int) partial.size(), idn_dev, hold_dev);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(down.data(), first_dev, sizeof(int32_t) * down.size(), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));

    // the packet's re-observed entries: every slot some particle matched, in the order of the first observation that names it
    std::vector<std::pair<int32_t, int32_t>> touched;
    for (int l = 0; l < nf0; l++)
        if (first[l] != 0x7fffffff) touched.push_back({first[l], l});
    std::sort(touched.begin(), touched.end());
    const int m = (int) touched.size();
    if (c->retired.empty()) c->retired.assign((size_t) cap_nf, 0);
    bool &mask_dirty = c->retired_stale;  // (sticky: a call that fails between a change of the flags and the upload leaves it set for the next one)
    if (census) {
        // a slot nobody holds any more is dead: out of the association, free for a later landmark
        for (int l : partial)
            if (hold[l] == N) c->pp_partial[(size_t) l] = 0;  // (every particle holds it: so will every descendant)
        for (int l : partial)
            if (hold[l] == 0 && !c->pp_dead[(size_t) l] && first[l] == 0x7fffffff) {
                c->pp_dead[(size_t) l] = 1;
                c->pp_dead_list.push_back(l);
                if (!c->retired[(size_t) l]) {
                    c->retired[(size_t) l] = 1;
                    c->n_retired++;
                    mask_dirty = true;
                }
                c->box_dirty[(size_t) l] = 1;
            }
        std::sort(c->pp_dead_list.begin(), c->pp_dead_list.end(), std::greater<int32_t>());  // back() = lowest dead slot
    }
    // new landmarks: an observation enough particles call new gets a slot (a dead one first, then the map grows)
    const int need = std::max(1, (int) ceil((double) opt->new_share * (double) N));
    std::vector<int32_t> newk((size_t) nz, -1), idn;
    int reused = 0, fresh = 0, dropped = 0;
    for (int j = 0; j < nz; j++) {
        if (news[j] < need) continue;
        int slot = -1;
        if (!c->pp_dead_list.empty()) {
            slot = c->pp_dead_list.back();
            c->pp_dead_list.pop_back();
            reused++;
        } else if (nf0 + fresh < cap_nf) {
            slot = nf0 + fresh++;
        } else {
            dropped++;
            continue;
        }
        newk[(size_t) j] = (int32_t) idn.size();
        idn.push_back(slot);
        c->pp_partial[(size_t) slot] = news[j] < N ? 1 : 0;
    }
    const int n = (int) idn.size();
    if (int rc = pp_reserve(c, nz, (size_t) m + n + 1)) return rc;
    first_dev = c->pp_tab_dev, hold_dev = first_dev + cap_nf, news_dev = hold_dev + cap_nf, idn_dev = news_dev + c->pp_nz_cap;
    // (uidx | holders (not read again) | newk | idn: the table's layout, one copy up)
    std::vector<int32_t> up(2 * (size_t) cap_nf + 2 * (size_t) c->pp_nz_cap, -1);
    int32_t *const uidx = up.data(), *const newk_up = uidx + 2 * (size_t) cap_nf, *const idn_up = newk_up + c->pp_nz_cap;
    for (int k = 0; k < m; k++) uidx[touched[(size_t) k].second] = k;
    for (int j = 0; j < nz; j++) newk_up[j] = newk[(size_t) j];
    for (int q = 0; q < n; q++) idn_up[q] = idn[(size_t) q];
    for (int q = 0; q < n; q++)
        if (idn[(size_t) q] < nf0) {  // a dead slot comes back into the association
            const int l = idn[(size_t) q];
            c->pp_dead[(size_t) l] = 0;
            if (c->retired[(size_t) l]) {
                c->retired[(size_t) l] = 0;
                c->n_retired--;
                mask_dirty = true;
            }
        }
    if (mask_dirty)
        if (int rc = retired_upload(c)) return rc;

    const bool need_normals = c->cfg.method == SLAMGPU_FASTSLAM2 && (m > 0 || n > 0);
    if (tape && need_normals && !normals) return fail(SLAMGPU_ERR_INVALID, "TAPE mode needs normals[3N] and strata[N]");
    // the labels resolved into what the launch reads (everything that can fail on the way there comes BEFORE the host's bookkeeping moves)
    HIP_TRY(hipMemcpyAsync(first_dev, up.data(), sizeof(int32_t) * up.size(), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // (pageable source)
    {
        Timed t(c, "particle_resolve");
        c->k->pp_resolve(c->stream, c->pp_lab_dev, N, nz, c->B.ncap, first_dev, news_dev, m, n, opt->p_new, c->cfg.log_weights, c->pp_obs_dev,
                         c->pp_wf_dev, c->pp_any_dev);
    }
    HIP_TRY(hipGetLastError());
    c->obs_step++;
    // genealogy bookkeeping, as do_update's: every packet entry is written by every particle (updated or copied forward) and moves
    // to the row this update opens
    int e_new = -1;
    std::vector<int32_t> rows_of((size_t) m), left;
    if (m + n > 0) {
        e_new = c->free_rows.back();
        c->free_rows.pop_back();
    }
    auto leave_row = [&](int l) {
        const int r = c->erow[(size_t) l];
        if (--c->refcnt[(size_t) r] == 0) {
            rows_remove_live(c, r);
            left.push_back(r);
        }
        c->erow[(size_t) l] = e_new;
        c->refcnt[(size_t) e_new]++;
        c->box_dirty[(size_t) l] = 1;
    };
    for (int k = 0; k < m; k++) {
        const int l = touched[(size_t) k].second, r = c->erow[(size_t) l];
        rows_of[(size_t) k] = r | (c->live_flag[(size_t) l] ? kRowLiveBit : 0) | (r == c->fresh_row ? kRowFreshBit : 0);
        c->live_flag[(size_t) l] ^= 1;
        c->seen_step[(size_t) l] = c->obs_step;
        leave_row(l);
    }
    for (int q = 0; q < n; q++) {
        const int l = idn[(size_t) q];
        if (l < nf0) {
            leave_row(l);
        } else {
            c->erow[(size_t) l] = e_new;
            c->refcnt[(size_t) e_new]++;
            c->box_dirty[(size_t) l] = 1;
        }
        c->live_flag[(size_t) l] = 0;  // a new landmark's first records go to buffer 0
    }
    const int n_rows = (int) c->live_rows.size();
    c->tables_dirty = true;

    UpdateArgs U{};
    U.method = c->cfg.method;
    U.m = m;
    U.n = n;
    U.nf = nf0;
    U.e_new = e_new;
    U.n_rows = n_rows;
    {
        int top = e_new;
        for (int r : c->live_rows) top = std::max(top, r);
        U.live_chunks = (top >> 2) + 1;
    }
    U.all_fresh = 0;
    c->fresh_row = e_new;
    memcpy(U.R, R, sizeof U.R);
    {
        const int slot = (int) (c->pkt_seq++ % kRing);
        if (c->pkt_ev_used[slot]) HIP_TRY(hipEventSynchronize(c->pkt_ev[slot]));
        char *ph = c->pkt_host + (size_t) slot * c->pkt_bytes;
        ObsPacket *hp = reinterpret_cast<ObsPacket *>(ph);
        hp->m = m;
        hp->n = n;
        hp->nf = nf0;
        hp->n_rows = n_rows;
        hp->e_new = e_new;
        hp->status = 0;
        hp->cap = 0;  // dense layout
        hp->pad = 0;
        int32_t *hidf = reinterpret_cast<int32_t *>(hp + 1);
        float *hzf = reinterpret_cast<float *>(hidf + m);
        float *hzn = hzf + 2 * m;
        for (int k = 0; k < m; k++) hidf[k] = touched[(size_t) k].second;
        memset(hzf, 0, sizeof(float) * 2 * ((size_t) m + n));  // (the observations are per particle: PerParticle::z)
        int32_t *hrow = reinterpret_cast<int32_t *>(hzn + 2 * n);
        if (m) memcpy(hrow, rows_of.data(), sizeof(int32_t) * (size_t) m);
        if (n_rows) memcpy(hrow + m, c->live_rows.data(), sizeof(int32_t) * (size_t) n_rows);
        const size_t used = sizeof(ObsPacket) + sizeof(int32_t) * (size_t) m + sizeof(float) * 2 * ((size_t) m + n) + sizeof(int32_t) * ((size_t) m + n_rows);
        char *pd = c->pkt_dev + (size_t) slot * c->pkt_bytes;
        HIP_TRY(hipMemcpyAsync(pd, ph, used, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->pkt_ev[slot], c->stream));
        c->pkt_ev_used[slot] = true;
        U.big = reinterpret_cast<const ObsPacket *>(pd);
    }
    if (e_new >= 0) {
        if (c->refcnt[(size_t) e_new] > 0) rows_add_live(c, e_new);
        else c->free_rows.push_back(e_new);
    }
    for (int r : left) c->free_rows.push_back(r);

    PerParticle ppa{c->pp_obs_dev, c->pp_z_dev, idn_dev, c->pp_wf_dev, c->pp_any_dev, nz, 0};
    c->pp_launch = &ppa;
    const int rc = issue_update(c, U, fresh, n_rows, need_normals, normals, strata, false);
    c->pp_launch = nullptr;
    if (report) {
        report[0] = m;
        report[1] = n;
        report[2] = reused;
        report[3] = dropped;
        report[4] = c->nf;
        report[5] = (int32_t) c->pp_dead_list.size();
        report[6] = need;
        report[7] = census ? 1 : 0;
    }
    return rc;
}
}  // namespace

int slamgpu_update_particle(slamgpu_ctx *c, const float *z, int32_t nz, const float R[4], const slamgpu_particle_assoc *opt, const float *normals,
                            const float *strata, int32_t report[8]) {
    if (report) memset(report, 0, sizeof(int32_t) * 8);
    if (int rc = pp_check(c, z, nz, R, opt)) return rc;
    if (opt->mode < SLAMGPU_ASSOC_AUTO || opt->mode > SLAMGPU_ASSOC_GRID) return fail(SLAMGPU_ERR_INVALID, "unknown association mode %d", opt->mode);
    if (nz == 0) return 0;  // (no observation, no update: fastslam2wrapper.cpp:84-95)
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = pp_reserve(c, nz, 1)) return rc;
    const float excl3[3] = {opt->excl_base, opt->excl_per_m, opt->unique_ratio};
    if (int rc = associate_impl(c, z, nz, R, opt->gate_reject, opt->gate_augment, opt->mode, nullptr, nullptr, nullptr, nullptr, c->pp_lab_dev, excl3)) return rc;
    return do_update_particle(c, z, nz, R, opt, normals, strata, report);
}

int slamgpu_update_labels(slamgpu_ctx *c, const float *z, int32_t nz, const float R[4], const int32_t *labels, const slamgpu_particle_assoc *opt,
                          const float *normals, const float *strata, int32_t report[8]) {
    if (report) memset(report, 0, sizeof(int32_t) * 8);
    if (int rc = pp_check(c, z, nz, R, opt)) return rc;
    if (nz > 0 && !labels) return fail(SLAMGPU_ERR_INVALID, "slamgpu_update_labels: labels[N * nz]");
    if (nz == 0) return 0;
    if (int rc = book_pull(c)) return rc;
    const size_t count = (size_t) c->B.n * (size_t) nz;
    for (size_t q = 0; q < count; q++)
        if (labels[q] >= c->nf || labels[q] < SLAMGPU_ASSOC_DISCARD)
            return fail(SLAMGPU_ERR_INVALID, "slamgpu_update_labels: label %d of particle %d, observation %d (%d landmarks)", (int) labels[q], (int) (q / nz), (int) (q % nz), c->nf);
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = pp_reserve(c, nz, 1)) return rc;
    c->pp_census_done = false;  // (the caller's labels: nobody has taken their census)
    {
        // (the device reads the labels by observation: [nz][ncap])
        const size_t S = (size_t) c->B.ncap;
        std::vector<int32_t> t(S * (size_t) nz, SLAMGPU_ASSOC_DISCARD);
        for (int i = 0; i < c->B.n; i++)
            for (int q = 0; q < nz; q++) t[(size_t) q * S + i] = labels[(size_t) i * nz + q];
        HIP_TRY(hipMemcpyAsync(c->pp_lab_dev, t.data(), sizeof(int32_t) * t.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // (pageable source)
    }
    return do_update_particle(c, z, nz, R, opt, normals, strata, report);
}

int slamgpu_num_landmarks(slamgpu_ctx *c) {
    if (!c) return SLAMGPU_ERR_INVALID;
    if (int rc = book_pull(c)) return rc;  // (device-driven steps: the count lives on the device; synchronises)
    return c->nf;
}

int slamgpu_genealogy_rows(slamgpu_ctx *c, int32_t *in_use, int32_t *capacity) {
    if (int rc = check_ctx(c)) return rc;
    if (int rc = book_pull(c)) return rc;
    if (in_use) *in_use = (int32_t) c->live_rows.size();
    if (capacity) *capacity = c->B.cap_rows;
    return 0;
}

int slamgpu_persist_info(slamgpu_ctx *c, int64_t *launches, int64_t *iterations, int32_t *cross_xcd) {
    if (int rc = check_ctx(c)) return rc;
    if (launches) *launches = c->persist_launches;
    if (iterations) *iterations = c->persist_steps;
    if (cross_xcd) {
        *cross_xcd = 0;
        if (c->psync_dev) {
            HIP_TRY(hipSetDevice(c->cfg.device));
            HIP_TRY(hipStreamSynchronize(c->stream));
            uint32_t v = 0;
            HIP_TRY(hipMemcpy(&v, c->psync_dev + kPersistSyncCross, sizeof v, hipMemcpyDeviceToHost));
            *cross_xcd = (int32_t) v;
            return persist_check(c);
        }
    }
    return 0;
}

int slamgpu_persist_status(slamgpu_ctx *c, int32_t *abandoned, int64_t *launch, int32_t *completed, int32_t *handed) {
    if (int rc = check_ctx(c)) return rc;
    const bool ab = c->pstatus_host && __atomic_load_n(c->pstatus_host, __ATOMIC_ACQUIRE) != 0;
    if (abandoned) *abandoned = ab ? 1 : 0;
    if (launch) *launch = ab ? (int64_t) c->pstatus_host[2] : 0;
    if (completed) *completed = ab ? (int32_t) c->pstatus_host[1] : 0;
    if (handed) *handed = ab ? (int32_t) c->pstatus_host[3] : 0;
    return 0;
}

int slamgpu_sync(slamgpu_ctx *c) {
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (!c->dist) {  // (a distributed context's outstanding stage needs the other shards: it stays queued)
        if (int rc = flush_predict(c)) return rc;
        if (int rc = flush_stages(c)) return rc;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return persist_check(c);
}

int slamgpu_download_range(slamgpu_ctx *c, int32_t first, int32_t count, float *xv, float *Pv9, float *w, float *xf, float *Pf4) {
    if (int rc = check_ctx(c)) return rc;
    if (first < 0 || count < 0 || (int64_t) first + count > c->B.n)
        return fail(SLAMGPU_ERR_INVALID, "particle range [%d, %d) outside [0, %d)", first, first + count, c->B.n);
    // after device-driven steps the landmark count and the genealogy book live on the device: bring them back BEFORE deciding
    // from c->nf whether the records need flattening (a C caller need not have asked for the count first)
    if (int rc = book_pull(c)) return rc;
    if ((xf || Pf4) && c->nf > 0) {
        if (int rc = flush_predict(c)) return rc;
        if (int rc = flatten(c)) return rc;  // records into their particles' own slots
    }
    if (int rc = read_ctrl(c, true)) return rc;
    const int cur = c->ctrl_host->live[c->slot], nf = c->nf;
    const size_t S = (size_t) c->B.ncap, M = (size_t) count;
    if (count == 0) return 0;
    std::vector<float4> pa(M), pb(M);
    std::vector<float2> pc(M);
    HIP_TRY(hipMemcpy(pa.data(), c->B.poseA[cur] + first, sizeof(float4) * M, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pb.data(), c->B.poseB[cur] + first, sizeof(float4) * M, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pc.data(), c->B.poseC[cur] + first, sizeof(float2) * M, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < M; i++) {
        if (xv) {
            xv[3 * i] = pa[i].x;
            xv[3 * i + 1] = pa[i].y;
            xv[3 * i + 2] = pa[i].z;
        }
        if (Pv9) {
            const float p00 = pb[i].x, p10 = pb[i].y, p11 = pb[i].z, p20 = pb[i].w, p21 = pc[i].x, p22 = pc[i].y;
            float *P = Pv9 + 9 * i;
            P[0] = p00; P[1] = p10; P[2] = p20;
            P[3] = p10; P[4] = p11; P[5] = p21;
            P[6] = p20; P[7] = p21; P[8] = p22;
        }
        if (w) w[i] = pa[i].w;
    }
    if ((xf || Pf4) && nf > 0) {
        // every landmark row has its own live buffer (kernels.h: lmk_live); rows are fetched a bounded batch at a time so
        // that a 10 000-landmark context does not need the whole slice on the host twice
        const std::vector<int32_t> &live = c->live_flag;
        const int batch = (int) std::max<size_t>(1, std::min<size_t>((size_t) nf, ((size_t) 64 << 20) / (20 * M)));
        std::vector<float4> la(M * batch);
        std::vector<float> lb(M * batch);
        for (int j0 = 0; j0 < nf; j0 += batch) {
            const int jn = std::min(batch, nf - j0);
            for (int j = 0; j < jn; j++) {
                const size_t row = (size_t) (j0 + j) * S + first;
                HIP_TRY(hipMemcpyAsync(la.data() + (size_t) j * M, c->B.lmkA[live[j0 + j]] + row, sizeof(float4) * M, hipMemcpyDeviceToHost, c->stream));
                HIP_TRY(hipMemcpyAsync(lb.data() + (size_t) j * M, c->B.lmkB[live[j0 + j]] + row, sizeof(float) * M, hipMemcpyDeviceToHost, c->stream));
            }
            HIP_TRY(hipStreamSynchronize(c->stream));
            for (size_t i = 0; i < M; i++)
                for (int j = 0; j < jn; j++) {
                    const float4 a = la[(size_t) j * M + i];
                    const float b = lb[(size_t) j * M + i];
                    const size_t at = i * nf + (size_t) (j0 + j);
                    if (xf) {
                        xf[at * 2] = a.x;
                        xf[at * 2 + 1] = a.y;
                    }
                    if (Pf4) {
                        float *P = Pf4 + at * 4;
                        P[0] = a.z;
                        P[1] = a.w;
                        P[2] = a.w;
                        P[3] = b;
                    }
                }
        }
    }
    return 0;
}

int slamgpu_peek(slamgpu_ctx *c, int32_t first, int32_t stride, int32_t count, float *xv, float *Pv9, float *w, float *xf,
                 float *Pf4) {
    if (int rc = check_ctx(c)) return rc;
    if (c->dist || c->cfg.n_particles_global != c->cfg.n_particles)
        return fail(SLAMGPU_ERR_INVALID, "slamgpu_peek: single contexts only (shards: slamgpu_download / slamgpu_dist_group_download)");
    if (first < 0 || count < 0 || stride < 1 || (count > 0 && (int64_t) first + (int64_t) (count - 1) * stride >= c->B.n))
        return fail(SLAMGPU_ERR_INVALID, "particles %d + k * %d, k < %d, outside [0, %d)", first, stride, count, c->B.n);
    if (count == 0) return 0;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (int rc = flush_predict(c)) return rc;   // (queued predicts belong to the state the caller asks about)
    if (int rc = flush_stages(c)) return rc;    // the plan of the last update: weights normalised, or ancestors in keep[]
    if (int rc = sync_tables(c)) return rc;
    const int nf = (xf || Pf4) ? c->nf : 0;
    const size_t M = (size_t) count;
    // device staging: [oa M][ob M][oc M][la nf M][lb nf M], every part 16-byte aligned
    const size_t o_a = 0, o_b = o_a + 16 * M, o_c = o_b + 16 * M, o_la = (o_c + 8 * M + 15) & ~(size_t) 15,
                 o_lb = o_la + 16 * M * nf, total = o_lb + 4 * M * nf + 16;
    if (total > c->peek_bytes) {
        if (c->peek_dev) (void) hipFree(c->peek_dev);
        c->peek_dev = nullptr;
        c->peek_bytes = 0;
        hipError_t e = hipMalloc((void **) &c->peek_dev, total);
        if (e != hipSuccess) return fail(SLAMGPU_ERR_ALLOC, "hipMalloc(%zu): %s", total, hipGetErrorString(e));
        c->peek_bytes = total;
    }
    PeekArgs A{};
    A.first = first;
    A.stride = stride;
    A.count = count;
    A.nf = nf;
    A.oa = reinterpret_cast<float4 *>(c->peek_dev + o_a);
    A.ob = reinterpret_cast<float4 *>(c->peek_dev + o_b);
    A.oc = reinterpret_cast<float2 *>(c->peek_dev + o_c);
    A.la = reinterpret_cast<float4 *>(c->peek_dev + o_la);
    A.lb = reinterpret_cast<float *>(c->peek_dev + o_lb);
    c->B.slot = c->slot;
    {
        Timed t(c, "peek");
        c->k->peek(c->stream, c->B, c->ws, A);
    }
    HIP_TRY(hipGetLastError());
    std::vector<float4> pa(M), pb(M), la(M * (size_t) nf);
    std::vector<float2> pc(M);
    std::vector<float> lb(M * (size_t) nf);
    HIP_TRY(hipMemcpyAsync(pa.data(), A.oa, 16 * M, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(pb.data(), A.ob, 16 * M, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(pc.data(), A.oc, 8 * M, hipMemcpyDeviceToHost, c->stream));
    if (nf > 0) {
        HIP_TRY(hipMemcpyAsync(la.data(), A.la, 16 * M * nf, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(lb.data(), A.lb, 4 * M * nf, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < M; i++) {
        if (xv) {
            xv[3 * i] = pa[i].x;
            xv[3 * i + 1] = pa[i].y;
            xv[3 * i + 2] = pa[i].z;
        }
        if (Pv9) {
            float *P = Pv9 + 9 * i;
            P[0] = pb[i].x; P[1] = pb[i].y; P[2] = pb[i].w;
            P[3] = pb[i].y; P[4] = pb[i].z; P[5] = pc[i].x;
            P[6] = pb[i].w; P[7] = pc[i].x; P[8] = pc[i].y;
        }
        if (w) w[i] = pa[i].w;
        for (int j = 0; j < nf; j++) {
            const float4 a = la[(size_t) j * M + i];
            const size_t at = i * (size_t) nf + (size_t) j;
            if (xf) {
                xf[at * 2] = a.x;
                xf[at * 2 + 1] = a.y;
            }
            if (Pf4) {
                float *P = Pf4 + at * 4;
                P[0] = a.z;
                P[1] = a.w;
                P[2] = a.w;
                P[3] = lb[(size_t) j * M + i];
            }
        }
    }
    return 0;
}

int slamgpu_download(slamgpu_ctx *c, float *xv, float *Pv9, float *w, float *xf, float *Pf4) {
    if (int rc = check_ctx(c)) return rc;
    return slamgpu_download_range(c, 0, c->B.n, xv, Pv9, w, xf, Pf4);
}

int slamgpu_upload(slamgpu_ctx *c, int32_t nf, const float *xv, const float *Pv9, const float *w, const float *xf,
                   const float *Pf4) {
    if (int rc = check_ctx(c)) return rc;
    if (nf < 0 || nf > c->B.cap_nf) return fail(SLAMGPU_ERR_CAPACITY, "nf=%d exceeds capacity %d", nf, c->B.cap_nf);
    if (nf > 0 && (!xf || !Pf4)) return fail(SLAMGPU_ERR_INVALID, "nf>0 needs xf and Pf");
    if (int rc = book_pull(c)) return rc;
    if (int rc = read_ctrl(c, true)) return rc;
    const int cur = c->ctrl_host->live[c->slot], N = c->B.n;
    const size_t S = (size_t) c->B.ncap;
    std::vector<float4> pa(S), pb(S);
    std::vector<float2> pc(S);
    HIP_TRY(hipMemcpy(pa.data(), c->B.poseA[cur], sizeof(float4) * S, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pb.data(), c->B.poseB[cur], sizeof(float4) * S, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pc.data(), c->B.poseC[cur], sizeof(float2) * S, hipMemcpyDeviceToHost));
    for (int i = 0; i < N; i++) {
        if (xv) {
            pa[i].x = xv[3 * i];
            pa[i].y = xv[3 * i + 1];
            pa[i].z = xv[3 * i + 2];
        }
        if (Pv9) {  // lower triangle of the reference's full matrix
            const float *P = Pv9 + 9 * (size_t) i;
            pb[i] = make_float4(P[0], P[3], P[4], P[6]);
            pc[i] = make_float2(P[7], P[8]);
        }
        if (w) pa[i].w = w[i];
    }
    HIP_TRY(hipMemcpy(c->B.poseA[cur], pa.data(), sizeof(float4) * S, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->B.poseB[cur], pb.data(), sizeof(float4) * S, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->B.poseC[cur], pc.data(), sizeof(float2) * S, hipMemcpyHostToDevice));
    if (nf > 0) {
        std::vector<float4> la(S * nf, make_float4(0.f, 0.f, 0.f, 0.f));
        std::vector<float> lb(S * nf, 0.0f);
        for (int i = 0; i < N; i++)
            for (int j = 0; j < nf; j++) {
                const float *P = Pf4 + ((size_t) i * nf + j) * 4;
                la[(size_t) j * S + i] = make_float4(xf[((size_t) i * nf + j) * 2], xf[((size_t) i * nf + j) * 2 + 1], P[0], P[2]);
                lb[(size_t) j * S + i] = P[3];
            }
        HIP_TRY(hipMemcpy(c->B.lmkA[0], la.data(), sizeof(float4) * la.size(), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c->B.lmkB[0], lb.data(), sizeof(float) * lb.size(), hipMemcpyHostToDevice));
    }
    // every landmark row live in buffer 0, every record in its particle's own slot
    c->pool_used = 0;
    std::fill(c->live_flag.begin(), c->live_flag.end(), 0);
    c->B.slot = c->slot;
    c->k->identity(c->stream, c->B, cur, 0);  // every landmark in genealogy row 0: own slot
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->nf = nf;
    rows_reset(c, nf);
    std::fill(c->box_dirty.begin(), c->box_dirty.end(), 1);
    // (a new particle set is a new map: nothing of it has been retired from the association)
    c->pp_dead.clear();
    c->pp_dead_list.clear();
    // (an uploaded set may hold absent records anywhere: every slot counts as partial until a census has seen it)
    c->pp_partial.assign((size_t) c->B.cap_nf, 1);
    if (c->n_retired > 0) {
        std::fill(c->retired.begin(), c->retired.end(), 0);
        c->n_retired = 0;
        if (c->retired_dev) HIP_TRY(hipMemset(c->retired_dev, 0, sizeof(uint32_t) * (((size_t) c->B.cap_nf + 31) / 32)));
    }
    c->est_fresh = false;
    c->shard_est_fresh = false;
    return 0;
}

void *slamgpu_stream(slamgpu_ctx *c) { return c ? (void *) c->stream : nullptr; }

int slamgpu_profile(slamgpu_ctx *c, int32_t enable) {
    if (int rc = check_ctx(c)) return rc;
    if (int rc = slamgpu_sync(c)) return rc;
    drain_stats(c);
    c->profile = enable != 0;
    return 0;
}

int slamgpu_timer_start(slamgpu_ctx *c) {
    if (int rc = check_ctx(c)) return rc;
    HIP_TRY(hipSetDevice(c->cfg.device));
    if (!c->timer_a) HIP_TRY(hipEventCreate(&c->timer_a));
    if (!c->timer_b) HIP_TRY(hipEventCreate(&c->timer_b));
    HIP_TRY(hipEventRecord(c->timer_a, c->stream));
    return 0;
}

int slamgpu_timer_stop(slamgpu_ctx *c, double *ms) {
    if (int rc = check_ctx(c)) return rc;
    if (!ms || !c->timer_a || !c->timer_b) return fail(SLAMGPU_ERR_INVALID, "timer not started");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipEventRecord(c->timer_b, c->stream));
    HIP_TRY(hipEventSynchronize(c->timer_b));
    float t = 0;
    HIP_TRY(hipEventElapsedTime(&t, c->timer_a, c->timer_b));
    *ms = t;
    return 0;
}

int slamgpu_kernel_time(slamgpu_ctx *c, const char *kernel, double *ms, int64_t *launches) {
    if (int rc = check_ctx(c)) return rc;
    if (!kernel) return fail(SLAMGPU_ERR_INVALID, "null kernel name");
    if (int rc = slamgpu_sync(c)) return rc;
    drain_stats(c);
    auto it = c->stats.find(kernel);
    if (ms) *ms = it == c->stats.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == c->stats.end() ? 0 : it->second.launches;
    return 0;
}

int slamgpu_algorithmic_bytes(slamgpu_ctx *c, double *update_bytes, double *predict_bytes) {
    if (int rc = check_ctx(c)) return rc;
    if (update_bytes) *update_bytes = 0;  // accounted by the harness from (m, n, nf, resampled) per step
    if (predict_bytes) *predict_bytes = c->predict_bytes;
    return 0;
}

// ---- Seam 1 -----------------------------------------------------------------------------------------
// Both entry points work on the CALLER's current HIP device and its null stream (they take no context).  Their device
// buffers are a grow-only scratch per calling thread (the reference's accelerator loop calls once per particle window: a
// hipMalloc / hipFree pair per call cost more than the 256 KB window's transfer).  The scratch is re-made when the thread's
// current device changes and is left to the runtime at process exit.
namespace {
struct SeamScratch {
    int device = -1;
    void *buf[2] = {nullptr, nullptr};
    size_t bytes[2] = {0, 0};
};
thread_local SeamScratch g_seam;

int seam_reserve(int which, size_t bytes, void **out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    SeamScratch &s = g_seam;
    if (s.device != dev) {  // buffers of another device: give them back there
        for (int k = 0; k < 2; k++)
            if (s.buf[k]) {
                if (hipSetDevice(s.device) == hipSuccess) (void) hipFree(s.buf[k]);
                s.buf[k] = nullptr;
                s.bytes[k] = 0;
            }
        HIP_TRY(hipSetDevice(dev));
        s.device = dev;
    }
    if (s.bytes[which] < bytes) {
        if (s.buf[which]) (void) hipFree(s.buf[which]);
        s.buf[which] = nullptr;
        s.bytes[which] = 0;
        const size_t want = std::max(bytes + bytes / 2, (size_t) 1 << 16);
        hipError_t e = hipMalloc(&s.buf[which], want);
        if (e != hipSuccess) return fail(SLAMGPU_ERR_ALLOC, "hipMalloc(%zu): %s", want, hipGetErrorString(e));
        s.bytes[which] = want;
    }
    *out = s.buf[which];
    return 0;
}
}  // namespace

int slamgpu_jacobians(const float *in, uint32_t n, float *out) {
    if (!in || (n > 0 && !out)) return fail(SLAMGPU_ERR_INVALID, "null buffer");
    if (n == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SLAMGPU_ERR_NO_DEVICE, "no HIP device: libslamgpu has no CPU fallback");
    float *din = nullptr, *dout = nullptr;
    const size_t nin = 7 + 6 * (size_t) n, nout = 16 * (size_t) n;
    if (int rc = seam_reserve(0, sizeof(float) * nin, (void **) &din)) return rc;
    if (int rc = seam_reserve(1, sizeof(float) * nout, (void **) &dout)) return rc;
    hipError_t e;
    if ((e = hipMemcpy(din, in, sizeof(float) * nin, hipMemcpyHostToDevice)) != hipSuccess) return fail(SLAMGPU_ERR_HIP, "H2D: %s", hipGetErrorString(e));
    kernels_strict()->jacobians(nullptr, din, n, dout);
    if ((e = hipGetLastError()) != hipSuccess) return fail(SLAMGPU_ERR_HIP, "launch: %s", hipGetErrorString(e));
    if ((e = hipMemcpy(out, dout, sizeof(float) * nout, hipMemcpyDeviceToHost)) != hipSuccess) return fail(SLAMGPU_ERR_HIP, "D2H: %s", hipGetErrorString(e));
    return 0;
}

int slamgpu_jacobians_multi(float *window, uint32_t records, uint64_t window_floats) {
    if (!window && records > 0) return fail(SLAMGPU_ERR_INVALID, "null window");
    if (records == 0) return 0;
    // walk the self-describing records: [n][xv 3][R 4][n x 6 in][n x 16 out]
    thread_local std::vector<uint32_t> tab;
    tab.clear();
    uint64_t pos = 0;
    for (uint32_t r = 0; r < records; r++) {
        if (pos + 8 > window_floats) return fail(SLAMGPU_ERR_INVALID, "record %u starts beyond the window (%llu floats)", r, (unsigned long long) window_floats);
        const float nf = window[pos];
        if (!(nf >= 0.0f) || nf > 65536.0f || nf != (float) (uint32_t) nf) return fail(SLAMGPU_ERR_INVALID, "record %u: feature count %g", r, (double) nf);
        const uint32_t n = (uint32_t) nf;
        const uint64_t len = 8 + 22 * (uint64_t) n;
        if (pos + len > window_floats || pos + len > 0xffffffffull) return fail(SLAMGPU_ERR_INVALID, "record %u (%u features) runs beyond the window", r, n);
        for (uint32_t k = 0; k < n; k++) {
            tab.push_back((uint32_t) pos);
            tab.push_back(k);
            tab.push_back(n);
        }
        pos += len;
    }
    const uint32_t nfeat = (uint32_t) (tab.size() / 3);
    if (nfeat == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SLAMGPU_ERR_NO_DEVICE, "no HIP device: libslamgpu has no CPU fallback");
    float *dwin = nullptr;
    uint32_t *dtab = nullptr;
    if (int rc = seam_reserve(0, sizeof(float) * pos, (void **) &dwin)) return rc;
    if (int rc = seam_reserve(1, sizeof(uint32_t) * tab.size(), (void **) &dtab)) return rc;
    hipError_t e;
    if ((e = hipMemcpy(dwin, window, sizeof(float) * pos, hipMemcpyHostToDevice)) != hipSuccess) return fail(SLAMGPU_ERR_HIP, "H2D: %s", hipGetErrorString(e));
    if ((e = hipMemcpy(dtab, tab.data(), sizeof(uint32_t) * tab.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail(SLAMGPU_ERR_HIP, "H2D: %s", hipGetErrorString(e));
    kernels_strict()->jacobians_multi(nullptr, dwin, dtab, nfeat);
    if ((e = hipGetLastError()) != hipSuccess) return fail(SLAMGPU_ERR_HIP, "launch: %s", hipGetErrorString(e));
    // the whole window comes back (the outputs are interleaved with the inputs record by record; one copy of 256 KB costs
    // less than a copy per record)
    if ((e = hipMemcpy(window, dwin, sizeof(float) * pos, hipMemcpyDeviceToHost)) != hipSuccess) return fail(SLAMGPU_ERR_HIP, "D2H: %s", hipGetErrorString(e));
    return 0;
}

int slamgpu_debug_stamps(slamgpu_ctx *c, uint64_t *out, int32_t max_blocks, int32_t *nblocks) {
    if (int rc = check_ctx(c)) return rc;
    if (!nblocks) return fail(SLAMGPU_ERR_INVALID, "null output");
    *nblocks = 0;
    if (!c->stamps_dev) return fail(SLAMGPU_ERR_INVALID, "no stamps: create the context with SLAMGPU_STAMPS=1 in the environment");
    HIP_TRY(hipSetDevice(c->cfg.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const int n = std::min(max_blocks, c->ws.nblocks);
    if (n > 0 && out) HIP_TRY(hipMemcpy(out, c->stamps_dev, sizeof(uint64_t) * kStampSlots * (size_t) n, hipMemcpyDeviceToHost));
    *nblocks = n;
    return 0;
}

int slamgpu_kat(int32_t math_mode, int32_t op, const float *in, int32_t n, float *out) {
    if (op < 0 || op > 2 || n < 0 || (n > 0 && (!in || !out))) return fail(SLAMGPU_ERR_INVALID, "bad arguments");
    if (n == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SLAMGPU_ERR_NO_DEVICE, "no HIP device: libslamgpu has no CPU fallback");
    const size_t per = op == 0 ? 1 : (op == 1 ? 5 : 9);
    float *din = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc((void **) &din, sizeof(float) * per * n));
    hipError_t e = hipMalloc((void **) &dout, sizeof(float) * n);
    if (e != hipSuccess) {
        (void) hipFree(din);
        return fail(SLAMGPU_ERR_ALLOC, "hipMalloc: %s", hipGetErrorString(e));
    }
    int rc = 0;
    if ((e = hipMemcpy(din, in, sizeof(float) * per * n, hipMemcpyHostToDevice)) != hipSuccess) rc = fail(SLAMGPU_ERR_HIP, "H2D: %s", hipGetErrorString(e));
    if (!rc) {
        (math_mode == SLAMGPU_MATH_FAST ? kernels_fast() : kernels_strict())->kat(nullptr, op, din, n, dout);
        if ((e = hipGetLastError()) != hipSuccess) rc = fail(SLAMGPU_ERR_HIP, "launch: %s", hipGetErrorString(e));
    }
    if (!rc && (e = hipMemcpy(out, dout, sizeof(float) * n, hipMemcpyDeviceToHost)) != hipSuccess) rc = fail(SLAMGPU_ERR_HIP, "D2H: %s", hipGetErrorString(e));
    (void) hipFree(din);
    (void) hipFree(dout);
    return rc;
}

}  // extern "C"
