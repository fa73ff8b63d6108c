// slam-backend — headless C++ host driver with the reference's command-line surface
//   slam-backend -m <map.mat> [-n name] -mode waypoints -method EKF1|FASTSLAM1|FASTSLAM2 [-KEY value ...]
// (SLAMBackendApplication.cpp:40-89; anything but FASTSLAM1/FASTSLAM2 selects the EKF, :26-29), plus
//   -rng parity|philox   parity = feed the libc rand() tape in the reference's draw order (default philox)
//   -math strict|fast    kernel build (default fast; strict replays the reference's float operations one by one)
//   -log <file.csv>      per control step: iteration, true pose, estimated pose, loop time [us]
//   -maxsteps <n>        stop after n control steps
//   -assoc known|gated   FastSLAM data association: known = dataAssociationKnown (core.cpp:91-120), what the reference's
//                        FastSLAM wrappers always use (default); gated = per-particle gated nearest neighbour
//                        (EKFSLAM::dataAssociate, ekfslam.cpp:151-189, applied to every particle with GATE_REJECT /
//                        GATE_AUGMENT) reduced to one association per step by weighted vote (slamgpu_associate)
//   -assoc particle      the same gates, but every particle ACTS on its own decisions, on a map of its own (slamgpu_update_particle;
//                        Particle.cpp:61-73 lets a particle's map grow by itself).  -PARTICLE_SLOTS k (slot capacity = k x the map's
//                        landmarks, default 4), -PARTICLE_NEW_SHARE (0.02), -PARTICLE_P_NEW (default: the Gaussian at the reject
//                        gate), -PARTICLE_EXCL_BASE (2.0 m) / -PARTICLE_EXCL_PER_M (0.05) / -PARTICLE_UNIQUE_RATIO (2): the exclusion
//                        rule of include/slamgpu.h: slamgpu_particle_assoc.  The map reported at the end is the best particle's.
//   -plot <sinks>        the per-step output the reference sends to slam-gui (plotting/NetworkPlot.cpp), byte for byte:
//                        tcp://127.0.0.1:4242 (the existing slam-gui) | file:<frames> | gather:<dir> (the GUI's DataGatherer
//                        files, headless) | none (default); several separated by ','
//   -plotstride <k>      particles / feature particles sent per step are decimated to every k-th particle (default: as
//                        many as keep a frame below ~2 000 particles; the reference sends all of them: N = 10^5 would be
//                        1.6 MB of poses and 56 MB of feature points per control step)
//   -gpus <k>            FastSLAM over k GPUs from this one process (slamgpu_dist_group_*): shard g = particles
//                        [g N/k, (g+1) N/k) on device g, one launch + one RCCL all-gather per observation step, the set is
//                        never moved; results do not depend on k.  k above the number of devices: logical shards on device 0
//                        (rehearsal).  Needs -rng philox, -assoc known, NPARTICLES a multiple of 256 k.
// It restates the wrapper loops (wrappers/fastslam2wrapper.cpp:31-122, fastslam1wrapper.cpp:32-113,
// ekfslamwrapper.cpp:33-109) minus the ZeroMQ plotting: the FastSLAM hot path runs on the GPU through the
// slamgpu C ABI (the seam AcceleratorHandler occupied), EKF-SLAM runs on the host CPU.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/slamgpu.h"
#include "ekfslam.h"
#include "frontend.h"
#include "gated.h"
#include "plotwire.h"

using namespace slamhost;

// slamgpu_num_landmarks returns a negative slamgpu_status on failure (e.g. the device front end reported a capacity overflow):
// never hand that to the association as a count
static int landmark_count(slamgpu_ctx *ctx, int &rc) {
    const int nf = slamgpu_num_landmarks(ctx);
    if (nf < 0) {
        rc = nf;
        return 0;
    }
    return nf;
}

static void usage(const char *a0) {
    printf("%s\n", a0);
    printf("    -m                  [s] input map file name\n");
    printf("    -n                  [s] experiment name\n");
    printf("    -mode               [s] running mode (waypoints)\n");
    printf("    -method             [s] SLAM method: EKF1 | FASTSLAM1 | FASTSLAM2\n");
    printf("    -rng parity|philox  -math strict|fast  -log file.csv  -maxsteps n\n");
    printf("    -plot tcp://127.0.0.1:4242|file:<path>|gather:<dir>|none   -plotstride k\n");
    printf("    -assoc known|gated|particle   FastSLAM data association: the reference's table (default), the per-particle gates reduced to a vote,\n");
    printf("                        or the per-particle gates acted on by every particle on a map of its own\n");
    printf("    -gpus k             FastSLAM particle set distributed over k GPUs (k > devices: logical shards on device 0)\n");
    printf("    -observe host|device  where the observation of a step is made: host (default) or on the GPU (the packet never leaves\n");
    printf("                        device memory: slamgpu_step_observe; -rng philox, known association, no -plot)\n");
    printf("    -loop step|batched  step: the wrapper's loop call by call (one predict per control step, estimate every iteration);\n");
    printf("                        batched (default without -plot, -rng parity, -assoc gated): one slamgpu_step per observation,\n");
    printf("                        estimates fetched 4096 at a time\n");
    printf("    -KEY value          any ini key, e.g. -NPARTICLES 100000 -NEFFECTIVE 75000 -SWITCH_SEED_RANDOM 7\n");
    printf("    -h  (print usage)\n\n");
}

// -gpus k: the wrapper loop with the particle set distributed over k contexts.  The queued controls ride inside the next
// observation step's launch; estimates are recorded on the device per observation step and fetched in batches.
static int run_distributed(Simulator &sim, int k, long maxsteps, FILE *log, Plot &plot) {
    const Conf &c = sim.conf;
    const int N = c.NPARTICLES;
    if (c.s("rng") == "parity" || c.s("assoc") == "gated" || c.s("assoc") == "particle") {
        fprintf(stderr, "-gpus %d needs -rng philox and -assoc known\n", k);
        return EXIT_FAILURE;
    }
    if (k < 1) {
        fprintf(stderr, "-gpus needs a positive number of GPUs\n");
        return EXIT_FAILURE;
    }
    if (N % (256 * k) != 0) {
        fprintf(stderr, "-gpus %d: NPARTICLES must be a multiple of %d (e.g. %d)\n", k, 256 * k, (N + 256 * k - 1) / (256 * k) * (256 * k));
        return EXIT_FAILURE;
    }
    const int ndev = slamgpu_device_count();
    if (ndev < 1) {
        fprintf(stderr, "slamgpu: no GPU (libslamgpu has no CPU fallback)\n");
        return EXIT_FAILURE;
    }
    const bool logical = k > ndev;
    printf("%s, %d particles over %d %s\n\n", c.method == 2 ? "FastSLAM 2" : "FastSLAM 1", N, k,
           logical ? "logical shards on device 0" : "GPUs");
    std::vector<slamgpu_ctx *> ctx((size_t) k, nullptr);
    slamgpu_dist_group *grp = nullptr;
    int rc = 0;
    for (int g = 0; g < k && !rc; g++) {
        slamgpu_config q{};
        q.struct_size = sizeof q;
        q.device = logical ? 0 : g;
        q.method = c.method;
        q.n_particles = N / k;
        q.n_particles_global = N;
        q.first_particle = (int64_t) g * (N / k);
        q.max_landmarks = sim.map.nlm;
        q.use_heading = c.SWITCH_HEADING_KNOWN == 1;
        q.add_predict_noise = c.method == 1 ? 1 : (c.SWITCH_PREDICT_NOISE == 1);
        q.resample = c.SWITCH_RESAMPLE == 1;
        q.n_effective = c.NEFFECTIVE;
        q.wheel_base = c.WHEELBASE;
        q.sigma_phi = c.sigmaT;
        q.rng_mode = SLAMGPU_RNG_PHILOX;
        q.math_mode = c.s("math") == "strict" ? SLAMGPU_MATH_STRICT : SLAMGPU_MATH_FAST;
        q.seed = (uint64_t) c.SWITCH_SEED_RANDOM;
        q.external_stream = (logical && g > 0) ? (uint64_t) (uintptr_t) slamgpu_stream(ctx[0]) : 0;  // logical shards share one stream
        rc = slamgpu_create(&q, &ctx[(size_t) g]);
    }
    if (!rc) rc = slamgpu_dist_group_create(ctx.data(), k, &grp);
    if (rc) {
        fprintf(stderr, "slamgpu: %s\n", slamgpu_last_error());
        for (slamgpu_ctx *x : ctx)
            if (x) slamgpu_destroy(x);
        return EXIT_FAILURE;
    }
    sim.seed();  // (after the HIP runtime has initialised: see main)

    struct ObsRow {
        long iter;
        float xt[3];
        double us;
    };
    std::vector<ObsRow> rows;       // observation steps whose estimate has not been fetched yet
    std::vector<float> controls, zf, zn;
    std::vector<int32_t> idf;
    std::vector<double> xyt(3 * 4096);
    long iter = 0, nobs = 0;
    double sum_us = 0, sq_err = 0, est[3] = {0, 0, 0};
    auto fetch = [&]() -> int {
        int32_t got = 0;
        if (int r = slamgpu_dist_group_history(grp, xyt.data(), nullptr, nullptr, nullptr, 4096, &got)) return r;
        for (int t = 0; t < got && t < (int) rows.size(); t++) {
            const ObsRow &o = rows[(size_t) t];
            for (int q = 0; q < 3; q++) est[q] = xyt[3 * (size_t) t + q];
            sq_err += (est[0] - o.xt[0]) * (est[0] - o.xt[0]) + (est[1] - o.xt[1]) * (est[1] - o.xt[1]);
            if (log) fprintf(log, "%ld,%.6f,%.6f,%.6f,%.6f,%.6f,%.6f,%.1f\n", o.iter, o.xt[0], o.xt[1], o.xt[2], est[0], est[1], est[2], o.us);
            if (plot.active()) {
                plot.setCurrentIteration((uint32_t) o.iter);
                plot.addTruePosition(o.xt[0], o.xt[1]);
                plot.addEstimatedPosition(est[0], est[1]);
                plot.setCarTruePosition(o.xt[0], o.xt[1], o.xt[2]);
                plot.setCarEstimatedPosition(est[0], est[1], est[2]);
                plot.plot();
            }
        }
        rows.clear();
        return 0;
    };
    auto t_obs = std::chrono::steady_clock::now();
    while ((maxsteps < 0 || iter < maxsteps) && !rc) {
        const int r = sim.control();
        if (r < 0) break;
        controls.push_back(sim.Vnoisy);
        controls.push_back(sim.Gnoisy);
        controls.push_back(sim.xTrue[2]);
        iter++;
        if (r != 1) continue;
        sim.observe();
        const int nf_now = landmark_count(ctx[0], rc);
        if (rc) break;
        sim.associate_known(nf_now, zf, idf, zn);
        rc = slamgpu_dist_group_step(grp, controls.data(), (int) (controls.size() / 3), sim.Qe, sim.dt, zf.data(), idf.data(), (int) idf.size(),
                                     zn.data(), (int) (zn.size() / 2), sim.Re, 1);
        controls.clear();
        nobs++;
        const auto now = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(now - t_obs).count();
        t_obs = now;
        sum_us += us;
        rows.push_back(ObsRow{iter, {sim.xTrue[0], sim.xTrue[1], sim.xTrue[2]}, us});
        if (!rc && rows.size() == 4096) rc = fetch();
    }
    if (!rc) rc = fetch();
    if (rc) fprintf(stderr, "slamgpu: %s\n", slamgpu_last_error());
    printf("control steps %ld, observation steps %ld, mean observation-step time %.1f us, rms position error %.4f m, final estimate (%.4f, %.4f, %.4f)\n",
           iter, nobs, nobs ? sum_us / nobs : 0.0, nobs ? std::sqrt(sq_err / nobs) : 0.0, est[0], est[1], est[2]);
    printf("landmarks in map: %d\n", slamgpu_num_landmarks(ctx[0]));
    slamgpu_dist_group_destroy(grp);
    for (slamgpu_ctx *x : ctx) slamgpu_destroy(x);
    return rc ? EXIT_FAILURE : 0;
}

// The wrapper's loop (fastslam2wrapper.cpp:51-117) for a headless run, batched: what the per-iteration form asks of the GPU
// between two observations -- eight predict calls and eight synchronous pose estimates, only the last of which anything
// but a plot consumes -- is ONE slamgpu_step per observation (controls + observation + update + recorded estimate, one
// launch), and the estimates come back 4 096 at a time.  -observe device: the observation itself is made on the GPU
// (slamgpu_step_observe): the host sends the controls and the true pose.
static int run_batched(Simulator &sim, slamgpu_ctx *ctx, bool observe_dev, long maxsteps, FILE *log, bool gpubusy) {
    const Conf &c = sim.conf;
    struct ObsRow {
        long iter;
        float xt[3];
        double us;
    };
    std::vector<ObsRow> rows;
    std::vector<float> controls, zf, zn;
    std::vector<int32_t> idf;
    std::vector<double> xyt(3 * 4096);
    long iter = 0, nobs = 0;
    double sq_err = 0, est[3] = {0, 0, 0};
    int rc = 0;
    size_t run_first = 0;  // -observe device: first row of `rows` that belongs to the chunk not yet handed over
    if (observe_dev) rc = slamgpu_set_map(ctx, sim.map.lm.data(), sim.map.nlm);
    if (!rc && gpubusy) rc = slamgpu_profile(ctx, 1);
    auto fetch = [&]() -> int {
        int32_t got = 0;
        if (int r = slamgpu_history_fetch(ctx, xyt.data(), nullptr, nullptr, nullptr, 4096, &got)) return r;
        for (int t = 0; t < got && t < (int) rows.size(); t++) {
            const ObsRow &o = rows[(size_t) t];
            for (int q = 0; q < 3; q++) est[q] = xyt[3 * (size_t) t + q];
            sq_err += (est[0] - o.xt[0]) * (est[0] - o.xt[0]) + (est[1] - o.xt[1]) * (est[1] - o.xt[1]);
            if (log) fprintf(log, "%ld,%.6f,%.6f,%.6f,%.6f,%.6f,%.6f,%.1f\n", o.iter, o.xt[0], o.xt[1], o.xt[2], est[0], est[1], est[2], o.us);
        }
        rows.clear();
        run_first = 0;
        return 0;
    };
    const auto t_begin = std::chrono::steady_clock::now();
    auto t_obs = t_begin;
    // (the hand-over does not wait for the GPU: round 5 found slamgpu_run_observe's queue upload blocking behind the launch before it,
    // and took the upload out: slamgpu.cpp: run_observe_persist)
    constexpr int kRunChunk = 256;
    std::vector<int32_t> run_counts;
    std::vector<float> run_xt;
    size_t run_rows = 0;
    double us_handover = 0, us_final = 0;  // where the host's time goes (printed with the result)
    auto flush_run = [&]() -> int {
        if (run_counts.empty()) return 0;
        const auto t_call = std::chrono::steady_clock::now();
        const int r = slamgpu_run_observe(ctx, (int32_t) run_counts.size(), run_counts.data(), controls.data(), sim.Qe, sim.dt, run_xt.data(), c.MAX_RANGE,
                                          sim.Re, c.SWITCH_SENSOR_NOISE ? 2 : 0);
        const auto now = std::chrono::steady_clock::now();
        us_handover += std::chrono::duration<double, std::micro>(now - t_call).count();
        const double us = std::chrono::duration<double, std::micro>(now - t_obs).count() / (double) run_counts.size();
        for (size_t t = run_first; t < rows.size(); t++) rows[t].us = us;  // (per iteration: the chunk's enqueue time, evenly)
        t_obs = now;
        run_first = rows.size();
        run_counts.clear();
        run_xt.clear();
        controls.clear();
        run_rows = 0;
        return r;
    };
    while ((maxsteps < 0 || iter < maxsteps) && !rc) {
        const int r = sim.control();
        if (r < 0) break;
        controls.push_back(sim.Vnoisy);
        controls.push_back(sim.Gnoisy);
        controls.push_back(sim.xTrue[2]);
        iter++;
        if (r != 1) continue;
        if (observe_dev) {
            // the true poses and controls do not depend on the filter: kRunChunk iterations are collected and handed over in ONE
            // call (slamgpu_run_observe: the same launches as one slamgpu_step_observe per iteration, bit-identical results)
            run_counts.push_back((int32_t) (controls.size() / 3 - run_rows));
            run_rows = controls.size() / 3;
            for (int q = 0; q < 3; q++) run_xt.push_back(sim.xTrue[q]);
            nobs++;
            rows.push_back(ObsRow{iter, {sim.xTrue[0], sim.xTrue[1], sim.xTrue[2]}, 0.0});
            if ((int) run_counts.size() == kRunChunk || rows.size() == 4096) {
                rc = flush_run();
                if (!rc && rows.size() == 4096) rc = fetch();
            }
            continue;
        } else {
            sim.observe();
            const int nf_now = landmark_count(ctx, rc);
            if (rc) break;
            sim.associate_known(nf_now, zf, idf, zn);
            rc = slamgpu_step(ctx, controls.data(), (int) (controls.size() / 3), sim.Qe, sim.dt, zf.data(), idf.data(), (int) idf.size(), zn.data(),
                              (int) (zn.size() / 2), sim.Re, nullptr, nullptr, 1);
        }
        controls.clear();
        nobs++;
        const auto now = std::chrono::steady_clock::now();
        rows.push_back(ObsRow{iter, {sim.xTrue[0], sim.xTrue[1], sim.xTrue[2]}, std::chrono::duration<double, std::micro>(now - t_obs).count()});
        t_obs = now;
        if (!rc && rows.size() == 4096) rc = fetch();
    }
    if (!rc && observe_dev) rc = flush_run();
    const auto t_final = std::chrono::steady_clock::now();
    if (!rc) rc = fetch();  // (synchronises: everything enqueued has finished)
    us_final = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_final).count();
    const double wall_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count();
    if (rc) fprintf(stderr, "slamgpu: %s\n", slamgpu_last_error());
    printf("control steps %ld, observation steps %ld, wall time per observation step %.2f us (whole loop, host front end included), "
           "rms position error %.4f m, final estimate (%.4f, %.4f, %.4f)\n",
           iter, nobs, nobs ? wall_us / nobs : 0.0, nobs ? std::sqrt(sq_err / nobs) : 0.0, est[0], est[1], est[2]);
    if (observe_dev && nobs)
        printf("host side of that, per observation step: %.2f us inside slamgpu_run_observe (hand-over), %.2f us waiting for the GPU at the end "
               "(the last fetch), the rest simulating the vehicle\n", us_handover / nobs, us_final / nobs);
    if (!rc && gpubusy) {
        double ms = 0, tot = 0;
        int64_t n = 0;
        for (const char *k : {"fs2_update", "fs1_update", "persist_loop", "resample", "scan", "observe", "finish", "gather", "predict", "estimate"})
            if (slamgpu_kernel_time(ctx, k, &ms, &n) == 0) tot += ms;
        printf("GPU busy (sum of kernel times between event pairs) %.2f us per observation step = %.0f %% of the wall time\n",
               nobs ? 1e3 * tot / nobs : 0.0, wall_us > 0 ? 100.0 * 1e3 * tot / wall_us : 0.0);
    }
    printf("landmarks in map: %d\n", slamgpu_num_landmarks(ctx));
    return rc ? EXIT_FAILURE : 0;
}

int main(int argc, char **argv) {
    for (int i = 1; i < argc; i++)
        if (strcmp(argv[i], "-h") == 0) {
            usage(argv[0]);
            return 0;
        }
    Simulator sim;
    std::string err;
    if (!sim.init(argc, argv, &err)) {
        fprintf(stderr, "%s\n", err.c_str());
        return EXIT_FAILURE;
    }
    const Conf &c = sim.conf;
    printf("map: %s\n", c.map_path.c_str());
    c.print(stdout);
    const std::string rng = c.s("rng"), math = c.s("math"), logf = c.s("log");
    const long maxsteps = c.s("maxsteps").empty() ? -1 : atol(c.s("maxsteps").c_str());
    FILE *log = logf.empty() ? nullptr : fopen(logf.c_str(), "wt");
    if (log) fprintf(log, "iteration,true_x,true_y,true_t,est_x,est_y,est_t,loop_us\n");

    // per-step output (SLAMBackendApplication.cpp:18-20: NetworkPlot created first, then named)
    Plot plot;
    {
        std::string perr;
        if (!plot.open(c.s("plot").empty() ? "none" : c.s("plot"), &perr)) {
            fprintf(stderr, "%s\n", perr.c_str());
            return EXIT_FAILURE;
        }
        if (plot.active()) plot.setSimulationName(c.simulation_name);
    }
    if (c.method != 0 && !c.s("gpus").empty() && atoi(c.s("gpus").c_str()) != 1) {
        if (plot.active()) {
            plot.setCarSize(c.WHEELBASE, 0);
            plot.setCarSize(c.WHEELBASE, 1);
        }
        const int rcd = run_distributed(sim, atoi(c.s("gpus").c_str()), maxsteps, log, plot);
        if (plot.active()) {
            plot.endPlot();
            plot.close();
        }
        if (log) fclose(log);
        return rcd;
    }
    slamgpu_ctx *ctx = nullptr;
    EkfSlam ekf;
    std::vector<float> ekf_table((size_t) sim.map.nlm, -1.0f);
    const int N = c.NPARTICLES;
    const bool parity = rng == "parity";
    const bool gated = c.s("assoc") == "gated";
    const bool particle = c.s("assoc") == "particle";
    const bool observe_dev = c.s("observe") == "device";
    const bool batched = c.method != 0 && !plot.active() && !parity && !gated && !particle && c.s("loop") != "step";
    auto numkey = [&](const char *key, double dflt) { return c.s(key).empty() ? dflt : atof(c.s(key).c_str()); };
    slamgpu_particle_assoc popt{};
    long pp_opened = 0, pp_reused = 0, pp_dropped = 0;
    int pp_most = 0;
    if (observe_dev && !batched) {
        fprintf(stderr, "-observe device needs a FastSLAM method, -rng philox, known association and no -plot / -loop step\n");
        return EXIT_FAILURE;
    }
    if (c.method != 0) {
        printf("%s\n\n", c.method == 2 ? "FastSLAM 2" : "FastSLAM 1");
        slamgpu_config g{};
        g.struct_size = sizeof g;
        g.method = c.method;
        g.n_particles = N;
        g.max_landmarks = gated ? 2 * sim.map.nlm : sim.map.nlm;  // unknown association may open spurious landmarks
        if (particle) g.max_landmarks = std::max(1, (int) numkey("PARTICLE_SLOTS", 4)) * sim.map.nlm;  // slots: hypotheses of all particles together
        g.use_heading = c.SWITCH_HEADING_KNOWN == 1;
        g.add_predict_noise = c.method == 1 ? 1 : (c.SWITCH_PREDICT_NOISE == 1);
        g.resample = c.SWITCH_RESAMPLE == 1;
        g.n_effective = c.NEFFECTIVE;
        g.wheel_base = c.WHEELBASE;
        g.sigma_phi = c.sigmaT;
        g.rng_mode = parity ? SLAMGPU_RNG_TAPE : SLAMGPU_RNG_PHILOX;
        g.math_mode = math == "strict" ? SLAMGPU_MATH_STRICT : SLAMGPU_MATH_FAST;
        g.seed = (uint64_t) c.SWITCH_SEED_RANDOM;
        g.flags = (observe_dev ? SLAMGPU_FLAG_DEVICE_OBSERVE : 0) | (particle ? SLAMGPU_FLAG_PARTICLE_MAPS : 0);
        popt.gate_reject = c.GATE_REJECT;
        popt.gate_augment = c.GATE_AUGMENT;
        popt.mode = SLAMGPU_ASSOC_AUTO;
        popt.new_share = (float) numkey("PARTICLE_NEW_SHARE", 0.02);
        popt.p_new = (float) numkey("PARTICLE_P_NEW", std::exp(-0.5 * c.GATE_REJECT) / (2.0 * 3.14159265358979323846 * std::sqrt(std::max(1e-30, (double) sim.Re[0] * sim.Re[3] - (double) sim.Re[1] * sim.Re[2]))));
        popt.census_every = (int32_t) numkey("PARTICLE_CENSUS", 1);
        popt.excl_base = (float) numkey("PARTICLE_EXCL_BASE", 2.0);
        popt.excl_per_m = (float) numkey("PARTICLE_EXCL_PER_M", 0.05);
        popt.unique_ratio = (float) numkey("PARTICLE_UNIQUE_RATIO", 2.0);
        if (slamgpu_create(&g, &ctx) != 0) {
            fprintf(stderr, "slamgpu_create: %s\n", slamgpu_last_error());
            return EXIT_FAILURE;
        }
        // the reference creates its accelerator object before the wrapper seeds rand() (SLAMBackendApplication.cpp:22-24,
        // slamwrapper.cpp:48-52); HIP runtime initialisation draws from libc rand(), so seed (again) only now
        sim.seed();
    } else {
        printf("EKFSLAM\n\n");
        ekf.enableBatchUpdate = c.SWITCH_BATCH_UPDATE == 1;
        ekf.useHeading = c.SWITCH_HEADING_KNOWN == 1;
        ekf.wheelBase = c.WHEELBASE;
        ekf.gateReject = c.GATE_REJECT;
        ekf.gateAugment = c.GATE_AUGMENT;
        ekf.associationKnown = c.SWITCH_ASSOCIATION_KNOWN;
        ekf.sigmaPhi = c.sigmaT;
    }

    if (batched) {
        const int rcb = run_batched(sim, ctx, observe_dev, maxsteps, log, c.s("gpubusy") == "1");
        if (log) fclose(log);
        slamgpu_destroy(ctx);
        return rcb;
    }
    int stride = c.s("plotstride").empty() ? std::max(1, N / 2000) : std::max(1, atoi(c.s("plotstride").c_str()));
    if (plot.active()) {
        // SLAMWrapper::configurePlot (slamwrapper.cpp:94-110) + addWaypointsAndLandmarks (:112-139) + setPlotRange (:141-172)
        plot.setCarSize(c.WHEELBASE, 0);
        plot.setCarSize(c.WHEELBASE, 1);
        std::vector<double> wx, wy, lx, ly;
        double xMin = 1e30, xMax = -1e30, yMin = 1e30, yMax = -1e30;
        auto grow = [&](double x, double y) {
            if (x > xMax) xMax = x;
            if (x < xMin) xMin = x;
            if (y > yMax) yMax = y;
            if (y < yMin) yMin = y;
        };
        for (int i = 0; i < sim.map.nwp; i++) {
            wx.push_back(sim.map.wp[i]);
            wy.push_back(sim.map.wp[(size_t) sim.map.nwp + i]);
            grow(wx.back(), wy.back());
        }
        plot.setWaypoints(wx, wy);
        for (int i = 0; i < sim.map.nlm; i++) {
            lx.push_back(sim.map.lm[i]);
            ly.push_back(sim.map.lm[(size_t) sim.map.nlm + i]);
            grow(lx.back(), ly.back());
        }
        plot.setLandmarks(lx, ly);
        plot.setPlotRange(xMin - (xMax - xMin) * 0.05, xMax + (xMax - xMin) * 0.05, yMin - (yMax - yMin) * 0.05, yMax + (yMax - yMin) * 0.05);
        plot.addTruePosition(sim.xTrue[0], sim.xTrue[1]);
        plot.setCarTruePosition(sim.xTrue[0], sim.xTrue[1], sim.xTrue[2]);
        plot.addEstimatedPosition(sim.xTrue[0], sim.xTrue[1]);
        plot.setCarEstimatedPosition(sim.xTrue[0], sim.xTrue[1], sim.xTrue[2]);
        plot.plot();
    }
    std::vector<float> plines;  // 4 x len, row-major: makeLaserLines (core.cpp:330-355)
    uint32_t plines_cols = 0;
    auto mark = std::chrono::steady_clock::now();
    std::vector<float> dxv, dxf;
    std::vector<double> px, py, fx, fy;

    auto laser_lines = [&]() {  // makeLaserLines(landmarksRangeBearing, xTrue) + transform_to_global (core.cpp:330-355, 827-843)
        if (!plot.active()) return;
        plines_cols = (uint32_t) (sim.z.size() / 2);
        plines.assign(4 * (size_t) plines_cols, 0.0f);
        const float cs = std::cos(sim.xTrue[2]), sn = std::sin(sim.xTrue[2]);
        for (uint32_t q = 0; q < plines_cols; q++) {
            const float gx = sim.z[2 * q] * std::cos(sim.z[2 * q + 1]), gy = sim.z[2 * q] * std::sin(sim.z[2 * q + 1]);
            plines[0 * plines_cols + q] = sim.xTrue[0];
            plines[1 * plines_cols + q] = sim.xTrue[1];
            plines[2 * plines_cols + q] = (cs * gx + -sn * gy) + sim.xTrue[0];
            plines[3 * plines_cols + q] = (sn * gx + cs * gy) + sim.xTrue[1];
        }
    };

    std::vector<float> zf, zn, normals, strata, noise2, g_xf;
    std::vector<int32_t> idf;
    slamhost::GatedPolicy policy;
    FILE *alog = (gated && !c.s("assoclog").empty()) ? fopen(c.s("assoclog").c_str(), "w") : nullptr;
    std::vector<int> truth_of;
    {
        auto num = [&](const char *key, double dflt) { return c.s(key).empty() ? dflt : atof(c.s(key).c_str()); };
        policy.enabled = num("ASSOC_POLICY", 1) != 0;
        policy.new_share = (float) num("ASSOC_NEW_SHARE", policy.new_share);
        policy.match_share = (float) num("ASSOC_MATCH_SHARE", policy.match_share);
        policy.credit_start = (int) num("ASSOC_CREDIT_START", policy.credit_start);
        policy.credit_max = (int) num("ASSOC_CREDIT_MAX", policy.credit_max);
        policy.retire_below = (int) num("ASSOC_RETIRE_BELOW", policy.retire_below);
        policy.rescue = num("ASSOC_RESCUE", 1) != 0;
        policy.rescue_base = (float) num("ASSOC_RESCUE_BASE", policy.rescue_base);
        policy.rescue_per_m = (float) num("ASSOC_RESCUE_PER_M", policy.rescue_per_m);
        policy.unique_ratio = (float) num("ASSOC_UNIQUE_RATIO", policy.unique_ratio);
        policy.new_factor = (float) num("ASSOC_NEW_FACTOR", policy.new_factor);
    }
    long iter = 0, nobs = 0;
    double sum_us = 0, sq_err = 0;
    double est[3] = {0, 0, 0};
    int rc = 0;
    while (maxsteps < 0 || iter < maxsteps) {
        const auto t0 = std::chrono::steady_clock::now();
        const int r = sim.control();
        if (r < 0) break;
        if (ctx) {
            const float *n2 = nullptr;
            if (parity && (c.method == 1 || c.SWITCH_PREDICT_NOISE == 1)) {
                noise2.resize(2 * (size_t) N);
                for (int i = 0; i < N; i++) randn(2, 1, &noise2[2 * (size_t) i]);
                n2 = noise2.data();
            }
            rc = slamgpu_predict(ctx, sim.Vnoisy, sim.Gnoisy, sim.Qe, sim.dt, sim.xTrue[2], n2);
            if (!rc && r == 1) {
                sim.observe();
                laser_lines();
                if (particle) {
                    // unknown association, per particle all the way: every particle gates the observations against its own map and
                    // acts on its own decisions (slamgpu_update_particle); nothing of the association visits the host
                    const int nz = (int) (sim.z.size() / 2);
                    const float *nm = nullptr, *st = nullptr;
                    if (parity) {
                        if (c.method == 2 && nz > 0) {
                            normals.resize(3 * (size_t) N);
                            for (int i = 0; i < N; i++) randn(3, 1, &normals[3 * (size_t) i]);
                            nm = normals.data();
                        }
                        strata.resize((size_t) N);
                        stratified_random(N, strata.data());
                        st = strata.data();
                    }
                    int32_t rep[8] = {0};
                    rc = slamgpu_update_particle(ctx, sim.z.data(), nz, sim.Re, &popt, nm, st, rep);
                    pp_opened += rep[1];
                    pp_reused += rep[2];
                    pp_dropped += rep[3];
                    pp_most = std::max(pp_most, (int) rep[0]);
                    nobs++;
                } else {
                if (gated) {
                    // unknown association: every particle gates the observations against its own map; the weighted vote
                    // becomes this step's association (slamgpu_update's association is per step)
                    // (the policy that turns the vote into the step's packet: host/gated.h)
                    const int nz = (int) (sim.z.size() / 2);
                    std::vector<int32_t> cons((size_t) std::max(nz, 1));
                    std::vector<float> supp((size_t) std::max(nz, 1));
                    if (nz > 0) rc = slamgpu_associate(ctx, sim.z.data(), nz, sim.Re, c.GATE_REJECT, c.GATE_AUGMENT, nullptr, cons.data(), supp.data());
                    const int nf_now = rc ? 0 : slamgpu_num_landmarks(ctx);
                    if (!rc && nf_now < 0) rc = nf_now;
                    float xv0[3] = {0, 0, 0};
                    g_xf.resize(2 * (size_t) std::max(nf_now, 1));
                    // the map the credits are kept against: particle 0's (one strided read, nothing rewritten)
                    if (!rc && policy.enabled) rc = slamgpu_peek(ctx, 0, 1, 1, xv0, nullptr, nullptr, nf_now ? g_xf.data() : nullptr, nullptr);
                    std::vector<int32_t> retire;
                    if (!rc) {
                        policy.step(sim.z.data(), nz, cons.data(), supp.data(), xv0, g_xf.data(), nf_now, c.MAX_RANGE, sim.map.nlm * 2 - nf_now, zf, idf, zn, retire);
                        if (!retire.empty()) rc = slamgpu_retire_landmarks(ctx, retire.data(), (int32_t) retire.size());
                        if (alog) {
                            // diagnostic (-assoclog file): every decision beside the truth the simulator knows (sim.vis: the TRUE landmark
                            // of each observation; truth_of[k]: the true landmark map entry k was opened for)
                            for (int q = 0; q < nz; q++) {
                                const int t = sim.vis[(size_t) q];
                                const int dcs = policy.decision[(size_t) q];
                                const char *what = "unused";
                                int k = -1;
                                if (dcs >= 0) {
                                    k = dcs % 1000000;
                                    what = truth_of[(size_t) k] == t ? (dcs >= 1000000 ? "match2" : "match") : (dcs >= 1000000 ? "MISMATCH2" : "MISMATCH");
                                } else if (dcs == -1) {
                                    bool dup = false;
                                    for (int e = 0; e < (int) truth_of.size(); e++) dup = dup || (truth_of[(size_t) e] == t && !policy.retired[(size_t) e]);
                                    what = dup ? "DUPLICATE" : "new";
                                    truth_of.push_back(t);
                                }
                                const float ex = sim.xTrue[0] - (float) est[0], ey = sim.xTrue[1] - (float) est[1];
                                fprintf(alog, "%ld obs %d true %d label %d share %.3f -> %s %d  (range %.2f, pose error %.3f m)\n", nobs, q, t, cons[q], supp[q], what, k,
                                        sim.z[2 * q], std::sqrt(ex * ex + ey * ey));
                            }
                            for (int j : retire) fprintf(alog, "%ld retire %d (true %d)\n", nobs, j, truth_of[(size_t) j]);
                        }
                    }
                    if (rc) {
                        fprintf(stderr, "slamgpu: %s\n", slamgpu_last_error());
                        break;
                    }
                } else {
                    const int nf_now = landmark_count(ctx, rc);
                    if (rc) {
                        fprintf(stderr, "slamgpu: %s\n", slamgpu_last_error());
                        break;
                    }
                    sim.associate_known(nf_now, zf, idf, zn);
                }
                const float *nm = nullptr, *st = nullptr;
                if (parity) {
                    if (c.method == 2 && (!idf.empty() || !zn.empty())) {
                        normals.resize(3 * (size_t) N);
                        for (int i = 0; i < N; i++) randn(3, 1, &normals[3 * (size_t) i]);
                        nm = normals.data();
                    }
                    strata.resize((size_t) N);
                    stratified_random(N, strata.data());
                    st = strata.data();
                }
                rc = slamgpu_update(ctx, zf.data(), idf.data(), (int) idf.size(), zn.data(), (int) (zn.size() / 2), sim.Re, nm, st);
                nobs++;
                }
            }
            if (!rc) rc = slamgpu_estimate(ctx, est);
            if (rc) {
                fprintf(stderr, "slamgpu: %s\n", slamgpu_last_error());
                break;
            }
        } else {
            if (r == 1) {
                sim.observe();
                laser_lines();
                nobs++;
            }
            const float phi = (float) (sim.xTrue[2] + c.sigmaT * unif_rand());  // ekfslamwrapper.cpp:82
            ekf.sim(sim.Vnoisy, sim.Gnoisy, sim.Qe, sim.dt, phi, sim.z, sim.vis, sim.Re, r == 1, sim.R, ekf_table);
            est[0] = ekf.x[0];
            est[1] = ekf.x[1];
            est[2] = ekf.x[2];
        }
        iter++;
        if (plot.active()) {
            // the tail of the wrappers' loop body (fastslam2wrapper.cpp:92-117, ekfslamwrapper.cpp:86-105)
            const auto now = std::chrono::steady_clock::now();
            plot.loopTime((uint32_t) std::chrono::duration_cast<std::chrono::microseconds>(now - mark).count());
            mark = now;
            plot.setCurrentIteration((uint32_t) iter);
            if (ctx) {  // drawParticles / drawFeatureParticles (ParticleSLAMWrapper.cpp:34-54), decimated
                const int nfl = std::max(0, slamgpu_num_landmarks(ctx));
                px.clear(); py.clear(); fx.clear(); fy.clear();
                // one strided read-only view per iteration (slamgpu_peek: one kernel, through the genealogy, nothing rewritten)
                const int cnt = (N + stride - 1) / stride;
                dxv.resize(3 * (size_t) cnt);
                dxf.resize(2 * (size_t) std::max(nfl, 1) * (size_t) cnt);
                rc = slamgpu_peek(ctx, 0, stride, cnt, dxv.data(), nullptr, nullptr, nfl ? dxf.data() : nullptr, nullptr);
                for (int i = 0; i < cnt && !rc; i++) {
                    px.push_back(dxv[3 * (size_t) i]);
                    py.push_back(dxv[3 * (size_t) i + 1]);
                    for (int j = 0; j < nfl; j++) {
                        const float lx = dxf[2 * ((size_t) i * nfl + j)];
                        if (lx != lx) continue;  // (-assoc particle: a landmark this particle does not hold)
                        fx.push_back(lx);
                        fy.push_back(dxf[2 * ((size_t) i * nfl + j) + 1]);
                    }
                }
                plot.setParticles(px, py);
                plot.setFeatureParticles(fx, fy);
            }
            plot.addTruePosition(sim.xTrue[0], sim.xTrue[1]);
            plot.addEstimatedPosition(est[0], est[1]);
            plot.setCarTruePosition(sim.xTrue[0], sim.xTrue[1], sim.xTrue[2]);
            plot.setCarEstimatedPosition(est[0], est[1], est[2]);
            plot.setLaserLines(plines_cols ? 4 : 0, plines_cols, plines.data());
            plot.plot();
            if (rc) {
                fprintf(stderr, "slamgpu: %s\n", slamgpu_last_error());
                break;
            }
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        sum_us += us;
        sq_err += (est[0] - sim.xTrue[0]) * (est[0] - sim.xTrue[0]) + (est[1] - sim.xTrue[1]) * (est[1] - sim.xTrue[1]);
        if (log) fprintf(log, "%ld,%.6f,%.6f,%.6f,%.6f,%.6f,%.6f,%.1f\n", iter, sim.xTrue[0], sim.xTrue[1], sim.xTrue[2], est[0], est[1], est[2], us);
    }
    printf("control steps %ld, observation steps %ld, mean loop time %.1f us, rms position error %.4f m, final estimate (%.4f, %.4f, %.4f)\n",
           iter, nobs, iter ? sum_us / iter : 0.0, iter ? std::sqrt(sq_err / iter) : 0.0, est[0], est[1], est[2]);
    if (ctx && gated) {
        const int nfl = slamgpu_num_landmarks(ctx);
        printf("landmarks in map: %d (%d opened, %d retired by the association policy, %d in use; %d observations matched by the second stage, %d "
               "left unused, %d refused as new next to a mapped landmark)\n",
               nfl - policy.n_retired, policy.n_opened, policy.n_retired, nfl - policy.n_retired, policy.n_rescued, policy.n_discarded_votes, policy.n_new_refused);
    } else if (ctx && particle) {
        // the map of the best (largest-weight) particle: what a FastSLAM with per-particle association reports
        const int slots = slamgpu_num_landmarks(ctx);
        std::vector<float> w((size_t) N);
        int held = 0, covered = 0, best = 0;
        if (slots >= 0 && slamgpu_download_range(ctx, 0, N, nullptr, nullptr, w.data(), nullptr, nullptr) == 0) {
            for (int i = 1; i < N; i++)
                if (w[(size_t) i] > w[(size_t) best]) best = i;
            std::vector<float> xf(2 * (size_t) std::max(slots, 1));
            if (slots > 0 && slamgpu_download_range(ctx, best, 1, nullptr, nullptr, nullptr, xf.data(), nullptr) == 0) {
                std::vector<char> hit((size_t) sim.map.nlm, 0);
                for (int j = 0; j < slots; j++) {
                    if (xf[2 * (size_t) j] != xf[2 * (size_t) j]) continue;  // absent
                    held++;
                    for (int t = 0; t < sim.map.nlm; t++) {
                        const float dx = xf[2 * (size_t) j] - sim.map.lm[(size_t) t], dy = xf[2 * (size_t) j + 1] - sim.map.lm[(size_t) sim.map.nlm + t];
                        if (dx * dx + dy * dy < 1.0f) hit[(size_t) t] = 1;
                    }
                }
                for (char h : hit) covered += h;
            }
        }
        printf("landmarks in map: %d (the best particle's, number %d; %d of the %d true landmarks within 1 m of one of them; %d slots in use by all particles "
               "together, %ld opened, %ld of them dead slots reused, %ld observations dropped for want of a slot, at most %d slots rewritten in a step)\n",
               held, best, covered, sim.map.nlm, slots, pp_opened, pp_reused, pp_dropped, pp_most);
    } else if (ctx) printf("landmarks in map: %d\n", slamgpu_num_landmarks(ctx));
    else printf("landmarks in map: %d\n", ekf.num_features());
    if (plot.active()) {
        plot.endPlot();
        plot.close();
    }
    if (log) fclose(log);
    if (alog) fclose(alog);
    if (ctx) slamgpu_destroy(ctx);
    return rc ? EXIT_FAILURE : 0;
}
