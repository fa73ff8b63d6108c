// EKF-SLAM on the host CPU (BASELINE config 1: "-method EKF1", plumbing path, no GPU kernel: one joint filter of
// dimension 3 + 2*Nf <= 73).  Restates matzipan/slam src/backend/algorithms/ekfslam.cpp:17-323 and the pieces of
// core.cpp it calls (choleskyUpdate :275-291, josephUpdate :294-317) with plain float32 loops, no Eigen.
#pragma once
#include <vector>

#include "frontend.h"

namespace slamhost {

struct Mat {  // dense row-major float matrix
    int r = 0, c = 0;
    std::vector<float> a;
    Mat() {}
    Mat(int r_, int c_) : r(r_), c(c_), a((size_t) r_ * c_, 0.0f) {}
    float &operator()(int i, int j) { return a[(size_t) i * c + j]; }
    float operator()(int i, int j) const { return a[(size_t) i * c + j]; }
};

class EkfSlam {
   public:
    // tunables copied by EKFSLAMWrapper's ctor (wrappers/ekfslamwrapper.cpp:17-27)
    bool enableBatchUpdate = true, useHeading = false;
    float wheelBase = 4, gateReject = 4, gateAugment = 25, sigmaPhi = 0;
    int associationKnown = 0;

    std::vector<float> x;  // 3 + 2*Nf
    Mat P;

    EkfSlam();
    // EKFSLAM::sim (ekfslam.cpp:17-43).  z / ids = this step's observations when `observe`.
    void sim(float Vn, float Gn, const float Qe[4], float dt, float phi, const std::vector<float> &z,
             const std::vector<int32_t> &ids, const float Re[4], bool observe, const float R[4], std::vector<float> &table);
    int num_features() const { return ((int) x.size() - 3) / 2; }

   private:
    void predict(float V, float G, const float Q[4], float dt);                                   // :46-77
    void observe_heading(float phi);                                                              // :86-95
    void observe_model(int idf, float zp[2], float H5[10]) const;                                 // :97-132
    void associate(const std::vector<float> &z, const float R[4], std::vector<float> &zf, std::vector<int> &idf,
                   std::vector<float> &zn) const;                                                 // :151-189
    void associate_known(const std::vector<float> &z, const std::vector<int32_t> &ids, std::vector<float> &zf,
                         std::vector<int> &idf, std::vector<float> &zn, std::vector<float> &table) const;  // :200-236
    void batch_update(const std::vector<float> &zf, const float R[4], const std::vector<int> &idf);  // :238-267
    void augment(const std::vector<float> &zn, const float Re[4]);                                // :269-323
};

}  // namespace slamhost
