// Host-side front end of the simulator loop (C++; product code, independent of oracle/).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../../include/slamhost.h"

namespace slamhost {

// ---- Conf: `<map stem>.ini` + `-KEY value` overrides, typed fields with the reference defaults -------
class Conf : public slamhost_conf {
   public:
    std::map<std::string, std::string> kv;  // every raw string (utils.cpp set_s)
    std::string map_path, simulation_name = "simulation", method_name, mode;
    bool load_ini(const std::string &path);        // utils.cpp:504-565
    void set_args(int argc, char **argv);          // utils.cpp:1032-1046
    void parse();                                  // core.cpp:971-1073
    std::string s(const std::string &k) const;
    void print(FILE *f) const;
};

struct Map {
    std::vector<float> lm, wp;  // 2 x n row-major
    int nlm = 0, nwp = 0;
};
bool read_map(const std::string &path, Map *out, std::string *err);  // core.cpp:855-962

float trig_offset(float ang);                                      // core.cpp:460-477
void randn(int m, int n, float *out);                              // core.cpp:383-419
void multivariate_gauss2(const float x[2], const float P[4], const float g[2], float out[2]);  // core.cpp:452-458
int stratified_random(int N, float *sel);                          // core.cpp:751-769
double unif_rand();                                                // core.cpp:775

// ---- the vehicle / sensor simulator that SLAMWrapper::control() and the wrapper loops implement ------
class Simulator {
   public:
    Conf conf;
    Map map;
    float Q[4] = {0, 0, 0, 0}, R[4] = {0, 0, 0, 0}, Qe[4] = {0, 0, 0, 0}, Re[4] = {0, 0, 0, 0};
    float Vtrue = 0, Gtrue = 0, Vnoisy = 0, Gnoisy = 0, dt = 0, dtSum = 0;
    int nLoop = 0, iwp = 0;
    float xTrue[3] = {0, 0, 0};
    std::vector<float> table;  // dataAssociationTable (float, -1 = never seen)
    std::vector<float> z;      // last observation (range, bearing) pairs
    std::vector<int32_t> vis;  // ids of the visible landmarks
    int64_t control_steps = 0;

    bool init(int argc, char **argv, std::string *err);  // SLAMBackendApplication.cpp:59-89 + slamwrapper.cpp:8-53
    void seed();                                         // slamwrapper.cpp:48-52 (call again after the GPU context exists)
    int control();                                       // slamwrapper.cpp:174-238 (+ dtSum bookkeeping)
    void observe();                                      // core.cpp:185-273, 438-449
    void associate_known(int nf, std::vector<float> &zf, std::vector<int32_t> &idf, std::vector<float> &zn);  // core.cpp:91-120

   private:
    void update_steering();      // core.cpp:41-78
    void predict_true();         // core.cpp:35-39
};

}  // namespace slamhost

struct slamhost_sim {
    slamhost::Simulator sim;
};

#include "gated.h"
struct slamhost_gated {
    slamhost::GatedPolicy policy;
};
