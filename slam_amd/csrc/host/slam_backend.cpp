// slam-backend — headless C++ host driver with the reference's command-line surface
//   slam-backend -m <map.mat> [-n name] -mode waypoints -method EKF1|FASTSLAM1|FASTSLAM2 [-KEY value ...]
// (SLAMBackendApplication.cpp:40-89; anything but FASTSLAM1/FASTSLAM2 selects the EKF, :26-29), plus
//   -rng parity|philox   parity = feed the libc rand() tape in the reference's draw order (default philox)
//   -math strict|fast    kernel build (default fast; strict replays the reference's float operations one by one)
//   -log <file.csv>      per control step: iteration, true pose, estimated pose, loop time [us]
//   -maxsteps <n>        stop after n control steps
// It restates the wrapper loops (wrappers/fastslam2wrapper.cpp:31-122, fastslam1wrapper.cpp:32-113,
// ekfslamwrapper.cpp:33-109) minus the ZeroMQ plotting: the FastSLAM hot path runs on the GPU through the
// slamgpu C ABI (the seam AcceleratorHandler occupied), EKF-SLAM runs on the host CPU.
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

using namespace slamhost;

static void usage(const char *a0) {
    printf("%s\n", a0);
    printf("    -m                  [s] input map file name\n");
    printf("    -n                  [s] experiment name\n");
    printf("    -mode               [s] running mode (waypoints)\n");
    printf("    -method             [s] SLAM method: EKF1 | FASTSLAM1 | FASTSLAM2\n");
    printf("    -rng parity|philox  -math strict|fast  -log file.csv  -maxsteps n\n");
    printf("    -KEY value          any ini key, e.g. -NPARTICLES 100000 -NEFFECTIVE 75000 -SWITCH_SEED_RANDOM 7\n");
    printf("    -h  (print usage)\n\n");
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

    slamgpu_ctx *ctx = nullptr;
    EkfSlam ekf;
    std::vector<float> ekf_table((size_t) sim.map.nlm, -1.0f);
    const int N = c.NPARTICLES;
    const bool parity = rng == "parity";
    if (c.method != 0) {
        printf("%s\n\n", c.method == 2 ? "FastSLAM 2" : "FastSLAM 1");
        slamgpu_config g{};
        g.struct_size = sizeof g;
        g.method = c.method;
        g.n_particles = N;
        g.max_landmarks = sim.map.nlm;
        g.use_heading = c.SWITCH_HEADING_KNOWN == 1;
        g.add_predict_noise = c.method == 1 ? 1 : (c.SWITCH_PREDICT_NOISE == 1);
        g.resample = c.SWITCH_RESAMPLE == 1;
        g.n_effective = c.NEFFECTIVE;
        g.wheel_base = c.WHEELBASE;
        g.sigma_phi = c.sigmaT;
        g.rng_mode = parity ? SLAMGPU_RNG_TAPE : SLAMGPU_RNG_PHILOX;
        g.math_mode = math == "strict" ? SLAMGPU_MATH_STRICT : SLAMGPU_MATH_FAST;
        g.seed = (uint64_t) c.SWITCH_SEED_RANDOM;
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

    std::vector<float> zf, zn, normals, strata, noise2;
    std::vector<int32_t> idf;
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
                sim.associate_known(slamgpu_num_landmarks(ctx), zf, idf, zn);
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
            if (!rc) rc = slamgpu_estimate(ctx, est);
            if (rc) {
                fprintf(stderr, "slamgpu: %s\n", slamgpu_last_error());
                break;
            }
        } else {
            if (r == 1) {
                sim.observe();
                nobs++;
            }
            const float phi = (float) (sim.xTrue[2] + c.sigmaT * unif_rand());  // ekfslamwrapper.cpp:82
            ekf.sim(sim.Vnoisy, sim.Gnoisy, sim.Qe, sim.dt, phi, sim.z, sim.vis, sim.Re, r == 1, sim.R, ekf_table);
            est[0] = ekf.x[0];
            est[1] = ekf.x[1];
            est[2] = ekf.x[2];
        }
        iter++;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        sum_us += us;
        sq_err += (est[0] - sim.xTrue[0]) * (est[0] - sim.xTrue[0]) + (est[1] - sim.xTrue[1]) * (est[1] - sim.xTrue[1]);
        if (log) fprintf(log, "%ld,%.6f,%.6f,%.6f,%.6f,%.6f,%.6f,%.1f\n", iter, sim.xTrue[0], sim.xTrue[1], sim.xTrue[2], est[0], est[1], est[2], us);
    }
    printf("control steps %ld, observation steps %ld, mean loop time %.1f us, rms position error %.4f m, final estimate (%.4f, %.4f, %.4f)\n",
           iter, nobs, iter ? sum_us / iter : 0.0, iter ? std::sqrt(sq_err / iter) : 0.0, est[0], est[1], est[2]);
    if (ctx) printf("landmarks in map: %d\n", slamgpu_num_landmarks(ctx));
    else printf("landmarks in map: %d\n", ekf.num_features());
    if (log) fclose(log);
    if (ctx) slamgpu_destroy(ctx);
    return rc ? EXIT_FAILURE : 0;
}
