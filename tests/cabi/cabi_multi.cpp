// TEST HARNESS for the MULTIPARTICLE_ACCELERATOR form of seam 1 (AcceleratorHandler.h:17-21): accel_shim.h compiled with that
// macro, driven by the window traffic of FastSLAM2::precomputeAllLikelihoodGivenXv (algorithms/fastslam2.cpp:172-286): one
// self-describing record per particle and re-observed landmark -- n = 1, xv, R, the landmark's xf and Pf in Eigen's linear
// order, sixteen output slots --, setParticlesCount, start, spin on isDone, then zp / Hf / Hv / Sf read back record by
// record.  (Upstream's read loop steps over one float too many per record and its caller works on copies of the particles:
// the function is unfinished there; the WINDOW, which is what the accelerator sees, is restated here as written.)
// Its own shared library: the class has another interface under this macro than in cabi_driver.cpp.
#define MULTIPARTICLE_ACCELERATOR 1
#include <string>

#include "accel_shim.h"

namespace {
thread_local std::string g_err;
}

extern "C" {
const char *cabi_multi_last_error() { return g_err.c_str(); }

// P particles: xv[P][3]; R[4], xf[P][nf][2], Pf[P][nf][4] ROW-major on this C boundary; idf[k] = the re-observed landmarks.
// Outputs per particle and observation, row-major: zp[P][k][2], Hv[P][k][6], Hf[P][k][4], Sf[P][k][4].
int cabi_multi_window(const float *xv, const float *R4, const float *xf, const float *Pf4, int P, int nf, const int *idf, int k, float *zp,
                      float *Hv, float *Hf, float *Sf) {
    try {
        static AcceleratorHandler *acc = new AcceleratorHandler();
        float *win = (float *) acc->getMemoryPointer();
        unsigned wr = 0;
        uint32_t records = 0;
        for (int p = 0; p < P; p++) {
            for (int i = 0; i < k; i++) {
                const int j = idf[i];
                win[wr++] = 1.0f;                                             // (float) idfi.size()  (:186)
                for (int a = 0; a < 3; a++) win[wr++] = xv[3 * p + a];        // xv(i)                (:188-190)
                win[wr++] = R4[0]; win[wr++] = R4[2]; win[wr++] = R4[1]; win[wr++] = R4[3];   // R(i): Eigen linear = column-major (:192-194)
                const float *x = xf + ((size_t) p * nf + j) * 2, *Pm = Pf4 + ((size_t) p * nf + j) * 4;
                win[wr++] = x[0]; win[wr++] = x[1];                           // landmarkXs[idfi[i]](j) (:201-203)
                win[wr++] = Pm[0]; win[wr++] = Pm[2]; win[wr++] = Pm[1]; win[wr++] = Pm[3];   // landmarkPs[..](j), column-major (:205-207)
                records++;
                wr += (2 + 4 + 6 + 4) * 1;                                    // the output slots (:213)
            }
        }
        if (records == 0) return 0;
        acc->setParticlesCount(records);
        acc->start();
        while (!acc->isDone()) {
        }
        unsigned rd = 0;
        for (int p = 0; p < P; p++) {
            for (int i = 0; i < k; i++) {
                const unsigned n = (unsigned) win[rd++];                      // (:240)
                rd += 3 + 4 + (2 + 4) * n;                                    // to the outputs of this record
                const size_t o = (size_t) p * k + i;
                zp[2 * o] = win[rd++]; zp[2 * o + 1] = win[rd++];             // zpVec (:249-250)
                for (int a = 0; a < 4; a++) Hf[4 * o + a] = win[rd++];        // HfMat << ... row by row (:252-255)
                for (int a = 0; a < 6; a++) Hv[6 * o + a] = win[rd++];        // HvMat (:258-263)
                for (int a = 0; a < 4; a++) Sf[4 * o + a] = win[rd++];        // SfMat (:265-268)
                if (n != 1) throw std::runtime_error("record count changed in the window");
            }
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}
}
