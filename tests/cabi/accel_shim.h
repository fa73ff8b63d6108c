// Seam 1, reference side: the class a maintainer of matzipan/slam drops in for src/backend/AcceleratorHandler.{h,cpp}
// (AcceleratorHandler.h:10-23) when building with -DJACOBIAN_ACCELERATOR=on.  Same public methods; the 256 KB on-chip
// memory window (AcceleratorHandler.cpp:13-14) becomes a host staging buffer, start() runs the batch on the GPU through
// slamgpu_jacobians, isDone() lets the spin loop of core.cpp:622 fall through.  computeJacobians itself (core.cpp:586-664)
// is unchanged: it packs xv, R, xf, Pf into the window, calls setN / start, spins, and reads zp, Hf, Hv, Sf back.
// With -DMULTIPARTICLE_ACCELERATOR the class takes the other form of the interface (setParticlesCount: AcceleratorHandler.h:17-21).
// This header is what INTEGRATION.md documents; tests/cabi/cabi_driver.cpp compiles it and tests/test_gpu_cabi.py runs it.
#ifndef SLAMGPU_ACCEL_SHIM_H
#define SLAMGPU_ACCEL_SHIM_H

#include <slamgpu.h>

#include <stdexcept>
#include <stdint.h>
#include <vector>

class AcceleratorHandler {
public:
    AcceleratorHandler() : window(256 * 1024 / sizeof(float)) {
        if (slamgpu_device_count() < 1) throw std::runtime_error("no MI355X visible");   // was: /dev/mem open failure
    }
    void *getMemoryPointer() { return window.data(); }
#ifdef MULTIPARTICLE_ACCELERATOR
    // AcceleratorHandler.h:17-21: the window holds `particles_count` self-describing records (fastslam2.cpp:172-216)
    void setParticlesCount(uint32_t particles_count) { n = particles_count; }
    void start() {
        done = slamgpu_jacobians_multi(window.data(), n, window.size()) == 0;
        if (!done) throw std::runtime_error(slamgpu_last_error());
    }
#else
    void setN(uint32_t n_) { n = n_; }
    void start() {                                   // core.cpp:620: runs the batch, synchronously
        done = false;
        if (7 + (6 + 16) * (size_t) n > window.size()) throw std::runtime_error("accelerator window overflow");
        done = slamgpu_jacobians(window.data(), n, window.data() + 7 + 6 * n) == 0;
        if (!done) throw std::runtime_error(slamgpu_last_error());
    }
#endif
    unsigned isDone() { return done; }               // core.cpp:622: the spin loop exits at once
private:
    std::vector<float> window;
    uint32_t n = 0;
    bool done = false;
};

#endif
