// Host-side front end: config, map, vehicle/sensor simulation, known data association, libc-rand tape.
// float32 arithmetic in the reference's order (the observation tape must be reproducible draw for draw).
#include "frontend.h"
#include "ekfslam.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <sstream>

namespace slamhost {

namespace {
thread_local std::string g_err;

std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char) s[a])) a++;
    while (b > a && std::isspace((unsigned char) s[b - 1])) b--;
    return s.substr(a, b - a);
}
}  // namespace

// ---------------------------------------------------------------------------------------------------
// Conf
// ---------------------------------------------------------------------------------------------------
bool Conf::load_ini(const std::string &path) {
    std::ifstream in(path);
    if (!in) return false;
    std::string line;
    while (std::getline(in, line)) {
        std::string b = trim(line);
        if (b.empty() || b[0] == '#' || b[0] == ':') continue;
        size_t eq = b.find('=');
        if (eq == std::string::npos) continue;
        kv[trim(b.substr(0, eq))] = trim(b.substr(eq + 1));
    }
    return true;
}

void Conf::set_args(int argc, char **argv) {
    for (int i = 1; i < argc; i++)
        if (argv[i][0] == '-' && i + 1 < argc) {
            kv[argv[i] + 1] = argv[i + 1];
            i++;
        }
}

std::string Conf::s(const std::string &k) const {
    auto it = kv.find(k);
    return it == kv.end() ? std::string() : it->second;
}

void Conf::parse() {
    auto f = [&](const char *k, float &v) {
        auto it = kv.find(k);
        if (it != kv.end()) v = (float) atof(it->second.c_str());
    };
    auto i = [&](const char *k, int32_t &v) {
        auto it = kv.find(k);
        if (it != kv.end()) v = atoi(it->second.c_str());
    };
    V = 3.0;
    MAXG = (float) (30 * M_PI / 180);
    RATEG = (float) (20 * M_PI / 180);
    WHEELBASE = 4;
    DT_CONTROLS = 0.025;
    sigmaV = 0.3;
    sigmaG = (float) (3.0 * M_PI / 180);
    MAX_RANGE = 30.0;
    DT_OBSERVE = 8 * DT_CONTROLS;
    sigmaR = 0.1;
    sigmaB = (float) (1.0 * M_PI / 180);
    sigmaT = (float) (1.0 * M_PI / 180);
    GATE_REJECT = 4.0;
    GATE_AUGMENT = 25.0;
    AT_WAYPOINT = 1.0;
    NUMBER_LOOPS = 2;
    NPARTICLES = 100;
    NEFFECTIVE = (int32_t) (0.75 * NPARTICLES);
    SWITCH_CONTROL_NOISE = 1;
    SWITCH_SENSOR_NOISE = 1;
    SWITCH_INFLATE_NOISE = 0;
    SWITCH_PREDICT_NOISE = 0;
    SWITCH_SAMPLE_PROPOSAL = 1;
    SWITCH_HEADING_KNOWN = 1;
    SWITCH_RESAMPLE = 1;
    SWITCH_PROFILE = 1;
    SWITCH_SEED_RANDOM = 0;
    SWITCH_ASSOCIATION_KNOWN = 0;
    SWITCH_BATCH_UPDATE = 1;
    SWITCH_USE_IEKF = 0;
    f("Vtrue", V);  // NB the ini key for the speed is Vtrue (core.cpp:1033)
    f("MAXG", MAXG);
    f("RATEG", RATEG);
    f("WHEELBASE", WHEELBASE);
    f("DT_CONTROLS", DT_CONTROLS);
    f("sigmaV", sigmaV);
    f("sigmaG", sigmaG);
    f("MAX_RANGE", MAX_RANGE);
    f("DT_OBSERVE", DT_OBSERVE);
    f("sigmaR", sigmaR);
    f("sigmaB", sigmaB);
    f("sigmaT", sigmaT);
    f("GATE_REJECT", GATE_REJECT);
    f("GATE_AUGMENT", GATE_AUGMENT);
    f("AT_WAYPOINT", AT_WAYPOINT);
    i("NUMBER_LOOPS", NUMBER_LOOPS);
    i("NPARTICLES", NPARTICLES);
    i("NEFFECTIVE", NEFFECTIVE);
    i("SWITCH_CONTROL_NOISE", SWITCH_CONTROL_NOISE);
    i("SWITCH_SENSOR_NOISE", SWITCH_SENSOR_NOISE);
    i("SWITCH_INFLATE_NOISE", SWITCH_INFLATE_NOISE);
    i("SWITCH_PREDICT_NOISE", SWITCH_PREDICT_NOISE);
    i("SWITCH_SAMPLE_PROPOSAL", SWITCH_SAMPLE_PROPOSAL);
    i("SWITCH_HEADING_KNOWN", SWITCH_HEADING_KNOWN);
    i("SWITCH_RESAMPLE", SWITCH_RESAMPLE);
    i("SWITCH_PROFILE", SWITCH_PROFILE);
    i("SWITCH_SEED_RANDOM", SWITCH_SEED_RANDOM);
    i("SWITCH_ASSOCIATION_KNOWN", SWITCH_ASSOCIATION_KNOWN);
    i("SWITCH_BATCH_UPDATE", SWITCH_BATCH_UPDATE);
    i("SWITCH_USE_IEKF", SWITCH_USE_IEKF);
    method_name = s("method");
    mode = s("mode");
    method = method_name == "FASTSLAM1" ? 1 : (method_name == "FASTSLAM2" ? 2 : 0);
}

void Conf::print(FILE *fp) const {
    for (auto &e : kv) fprintf(fp, "%-28s = %s\n", e.first.c_str(), e.second.c_str());
}

// ---------------------------------------------------------------------------------------------------
// map reader: "lm <rows> <cols>" / "wp <rows> <cols>" headers, then one column (point) per line
// ---------------------------------------------------------------------------------------------------
bool read_map(const std::string &path, Map *out, std::string *err) {
    std::ifstream in(path);
    if (!in) {
        if (err) *err = "Unable to read input file " + path;
        return false;
    }
    std::string line;
    int lineno = 0, lm_rows = 0;
    while (std::getline(in, line)) {
        lineno++;
        std::istringstream ls(line);
        std::vector<std::string> tok;
        for (std::string t; ls >> t;) tok.push_back(t);
        if (tok.empty() || tok[0][0] == '#') continue;
        const bool is_lm = tok[0] == "lm", is_wp = tok[0] == "wp";
        if (!is_lm && !is_wp) {
            if (err) *err = "Unknown command " + tok[0] + " on line " + std::to_string(lineno);
            return false;
        }
        if (tok.size() != 3) {
            if (err) *err = "Wrong args for " + tok[0] + " on line " + std::to_string(lineno);
            return false;
        }
        const int rows = (int) strtof(tok[1].c_str(), nullptr), cols = (int) strtof(tok[2].c_str(), nullptr);
        if (is_lm) lm_rows = rows;
        std::vector<float> dst((size_t) std::max(rows, 2) * cols, 0.0f);
        const int nread = is_lm ? rows : lm_rows;  // upstream reads lm_rows values per waypoint line (core.cpp:950)
        for (int c = 0; c < cols; c++) {
            if (!std::getline(in, line)) {
                if (err) *err = "EOF after reading";
                return false;
            }
            lineno++;
            std::istringstream cs(line);
            std::vector<std::string> ct;
            for (std::string t; cs >> t;) ct.push_back(t);
            if ((int) ct.size() < rows) {
                if (err) *err = "invalid line for coordinate on line " + std::to_string(lineno);
                return false;
            }
            for (int r = 0; r < nread && r < rows; r++) dst[(size_t) r * cols + c] = strtof(ct[r].c_str(), nullptr);
        }
        if (is_lm) {
            out->lm = dst;
            out->nlm = cols;
        } else {
            out->wp = dst;
            out->nwp = cols;
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------------------
float trig_offset(float ang) {
    if ((ang < -2 * M_PI) || (ang > 2 * M_PI)) {
        int n = (int) floor(ang / (2 * M_PI));
        ang = (float) (ang - n * (2 * M_PI));
    }
    if (ang > M_PI) ang = (float) (ang - (2 * M_PI));
    if (ang < -M_PI) ang = (float) (ang + (2 * M_PI));
    return ang;
}

void randn(int m, int n, float *out) {
    const int cnt = m * n;
    std::vector<float> u((size_t) cnt + 1);
    for (int r = 0; r <= cnt; r++) u[r] = (float) (std::rand() * 1.0 / RAND_MAX);
    float square, amp = 0, angle = 0;
    for (int k = 0; k < cnt; k++) {
        if (k % 2 == 0) {
            square = (float) (-2. * std::log(u[k]));
            if (square < 0.) square = 0.;
            amp = std::sqrt(square);
            angle = (float) (2. * M_PI * u[k + 1]);
            out[k] = amp * std::sin(angle);
        } else {
            out[k] = amp * std::cos(angle);
        }
    }
}

void multivariate_gauss2(const float x[2], const float P[4], const float g[2], float out[2]) {
    // S = P.llt().matrixL(); S*g + x   (Eigen LLT: sub-diagonal scaled by the reciprocal pivot)
    float l00 = P[0], l10 = P[2], l11 = P[3];
    if (l00 > 0.0f) {
        l00 = std::sqrt(l00);
        l10 = l10 * (1.0f / l00);
        float t = l11 - l10 * l10;
        if (t > 0.0f) l11 = std::sqrt(t);
    }
    out[0] = (l00 * g[0] + 0.0f * g[1]) + x[0];
    out[1] = (l10 * g[0] + l11 * g[1]) + x[1];
}

double unif_rand() { return std::rand() / double(RAND_MAX); }

int stratified_random(int N, float *sel) {
    float k = (float) (1.0 / (float) N);
    float temp = k / 2;
    int cnt = 0;
    while (temp < (1 - k / 2)) {
        if (cnt < N) sel[cnt] = temp;
        cnt++;
        temp = temp + k;
    }
    if (cnt == N) {
        for (int i = 0; i < N; i++) sel[i] = (float) (sel[i] + unif_rand() * k - (k / 2));
    } else {
        // the reference asserts for such N (core.cpp:762); well-defined strata instead
        for (int i = 0; i < N; i++) sel[i] = (float) (((double) i + unif_rand()) / (double) N);
    }
    return cnt;
}

// ---------------------------------------------------------------------------------------------------
// Simulator
// ---------------------------------------------------------------------------------------------------
bool Simulator::init(int argc, char **argv, std::string *err) {
    std::string mapFilename = "example_webmap.mat";
    for (int i = 1; i < argc; i++) {
        if (strcmp(argv[i], "-m") == 0 && i + 1 < argc) mapFilename = argv[i + 1];
        if (strcmp(argv[i], "-n") == 0 && i + 1 < argc) conf.simulation_name = argv[i + 1];
    }
    conf.map_path = mapFilename;
    const size_t dot = mapFilename.find_last_of('.');
    const std::string ini = (dot == std::string::npos ? mapFilename : mapFilename.substr(0, dot)) + ".ini";
    conf.load_ini(ini);  // a missing ini is not an error upstream either (defaults apply)
    conf.set_args(argc, argv);
    conf.parse();
    if (!read_map(mapFilename, &map, err)) return false;
    if (map.nlm <= 0 || map.nwp <= 0) {
        if (err) *err = "map has no landmarks or no waypoints: " + mapFilename;
        return false;
    }
    conf.n_landmarks = map.nlm;
    conf.n_waypoints = map.nwp;
    Vtrue = conf.V;
    Gtrue = 0;
    Q[0] = (float) std::pow((double) conf.sigmaV, 2);
    Q[3] = (float) std::pow((double) conf.sigmaG, 2);
    R[0] = (float) std::pow((double) conf.sigmaR, 2);
    R[3] = (float) std::pow((double) conf.sigmaB, 2);
    if (conf.SWITCH_INFLATE_NOISE == 1) {
        for (int i = 0; i < 4; i++) {
            Q[i] = 2 * Q[i];
            R[i] = 2 * R[i];
        }
        // upstream leaves Qe/Re unset here (slamwrapper.cpp:31-37); use the inflated values
        memcpy(Qe, Q, sizeof Q);
        memcpy(Re, R, sizeof R);
    } else {
        memcpy(Qe, Q, sizeof Q);
        memcpy(Re, R, sizeof R);
    }
    memcpy(conf.Q, Q, sizeof Q);
    memcpy(conf.R, R, sizeof R);
    memcpy(conf.Qe, Qe, sizeof Q);
    memcpy(conf.Re, Re, sizeof R);
    nLoop = conf.NUMBER_LOOPS;
    dt = conf.DT_CONTROLS;
    iwp = 0;
    table.assign((size_t) map.nlm, -1.0f);
    seed();
    return true;
}

// slamwrapper.cpp:48-52.  NB the reference constructs its accelerator object BEFORE the wrapper seeds libc rand()
// (SLAMBackendApplication.cpp:22-24); a drop-in must keep that order, or seed again after creating the GPU context:
// the first HIP call of a process initialises the runtime, which draws from libc rand() itself and would shift the tape.
void Simulator::seed() {
    if (conf.SWITCH_SEED_RANDOM != 0)
        srand((unsigned) conf.SWITCH_SEED_RANDOM);
    else
        srand((unsigned) time(nullptr));
}

void Simulator::update_steering() {
    const int nw = map.nwp;
    double cw0 = map.wp[iwp], cw1 = map.wp[(size_t) nw + iwp];
    float d2 = (float) (std::pow(cw0 - xTrue[0], 2) + std::pow(cw1 - xTrue[1], 2));
    if (d2 < conf.AT_WAYPOINT * conf.AT_WAYPOINT) {
        iwp++;
        if (iwp >= nw) {
            iwp = -1;
            return;
        }
        cw0 = map.wp[iwp];
        cw1 = map.wp[(size_t) nw + iwp];
    }
    float deltaG = (float) (std::atan2(cw1 - xTrue[1], cw0 - xTrue[0]) - xTrue[2] - Gtrue);
    deltaG = trig_offset(deltaG);
    const float maxDelta = conf.RATEG * dt;
    if (std::fabs(deltaG) > maxDelta) {
        int sign = (deltaG > 0) ? 1 : ((deltaG < 0) ? -1 : 0);
        deltaG = sign * maxDelta;
    }
    Gtrue = Gtrue + deltaG;
    if (std::fabs(Gtrue) > conf.MAXG) {
        int sign2 = (Gtrue > 0) ? 1 : ((Gtrue < 0) ? -1 : 0);
        Gtrue = sign2 * conf.MAXG;
    }
}

void Simulator::predict_true() {
    float *x = xTrue;
    const float V = Vtrue, G = Gtrue;
    x[0] = x[0] + V * dt * std::cos(G + x[2]);
    x[1] = x[1] + V * dt * std::sin(G + x[2]);
    x[2] = trig_offset(x[2] + V * dt * std::sin(G) / conf.WHEELBASE);
}

int Simulator::control() {
    if (iwp == -1) return -1;
    update_steering();
    if (iwp == -1 && nLoop > 1) {
        iwp = 0;
        nLoop--;
    }
    if (iwp == -1 && nLoop == 1) return -1;
    predict_true();
    if (conf.SWITCH_CONTROL_NOISE) {
        float g[2], A[2] = {Vtrue, Gtrue}, C[2];
        randn(2, 1, g);
        multivariate_gauss2(A, Q, g, C);
        Vnoisy = C[0];
        Gnoisy = C[1];
    }
    control_steps++;
    dtSum += dt;
    if (dtSum >= conf.DT_OBSERVE) {
        dtSum = 0;
        return 1;
    }
    return 0;
}

void Simulator::observe() {
    const float *x = xTrue;
    const float range = conf.MAX_RANGE, phi = x[2];
    const int nl = map.nlm;
    z.clear();
    vis.clear();
    for (int j = 0; j < nl; j++) {
        const float dx = map.lm[j] - x[0];
        const float dy = map.lm[(size_t) nl + j] - x[1];
        if ((std::fabs(dx) < range) && (std::fabs(dy) < range) && ((dx * std::cos(phi) + dy * std::sin(phi)) > 0.0) &&
            ((std::pow((double) dx, 2) + std::pow((double) dy, 2)) < std::pow((double) range, 2))) {
            vis.push_back(j);
            z.push_back((float) std::sqrt(std::pow((double) dx, 2) + std::pow((double) dy, 2)));
            z.push_back(std::atan2(dy, dx) - phi);
        }
    }
    const int len = (int) vis.size();
    if (conf.SWITCH_SENSOR_NOISE && len > 0) {
        std::vector<float> r1((size_t) len), r2((size_t) len);
        randn(1, len, r1.data());
        randn(1, len, r2.data());
        for (int c = 0; c < len; c++) {
            z[2 * c] = z[2 * c] + r1[c] * std::sqrt(R[0]);
            z[2 * c + 1] = z[2 * c + 1] + r2[c] * std::sqrt(R[3]);
        }
    }
}

void Simulator::associate_known(int nf, std::vector<float> &zf, std::vector<int32_t> &idf, std::vector<float> &zn) {
    zf.clear();
    idf.clear();
    zn.clear();
    std::vector<int> idn;
    for (size_t i = 0; i < vis.size(); i++) {
        const int ii = vis[i];
        if (table[ii] == -1) {
            zn.push_back(z[2 * i]);
            zn.push_back(z[2 * i + 1]);
            idn.push_back(ii);
        } else {
            zf.push_back(z[2 * i]);
            zf.push_back(z[2 * i + 1]);
            idf.push_back((int32_t) table[ii]);
        }
    }
    for (size_t i = 0; i < idn.size(); i++) table[idn[i]] = (float) (nf + (int) i);
}

const char *last_error() { return g_err.c_str(); }
void set_error(const std::string &e) { g_err = e; }

}  // namespace slamhost

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
extern "C" {

const char *slamhost_last_error(void) { return slamhost::last_error(); }

slamhost_sim *slamhost_sim_create(int argc, char **argv) {
    slamhost_sim *s = new slamhost_sim();
    std::string err;
    if (!s->sim.init(argc, argv, &err)) {
        slamhost::set_error(err);
        delete s;
        return nullptr;
    }
    return s;
}

void slamhost_sim_destroy(slamhost_sim *s) { delete s; }

slamhost_gated *slamhost_gated_create(void) { return new slamhost_gated(); }
void slamhost_gated_destroy(slamhost_gated *g) { delete g; }
int slamhost_gated_set(slamhost_gated *g, const char *name, double v) {
    if (!g || !name) return -1;
    slamhost::GatedPolicy &p = g->policy;
    const std::string k = name;
    if (k == "enabled") p.enabled = v != 0;
    else if (k == "new_share") p.new_share = (float) v;
    else if (k == "match_share") p.match_share = (float) v;
    else if (k == "credit_start") p.credit_start = (int) v;
    else if (k == "credit_max") p.credit_max = (int) v;
    else if (k == "retire_below") p.retire_below = (int) v;
    else if (k == "rescue") p.rescue = v != 0;
    else if (k == "rescue_base") p.rescue_base = (float) v;
    else if (k == "rescue_per_m") p.rescue_per_m = (float) v;
    else if (k == "unique_ratio") p.unique_ratio = (float) v;
    else if (k == "new_factor") p.new_factor = (float) v;
    else if (k == "grid") p.grid = v != 0;
    else return -1;
    return 0;
}
int slamhost_gated_step(slamhost_gated *g, const float *z, int32_t nz, const int32_t *consensus, const float *support, const float xv[3],
                        const float *xf, int32_t nf, float max_range, int32_t room, float *zf, int32_t *idf, int32_t *m, float *zn, int32_t *n,
                        int32_t *retire, int32_t *n_retire) {
    if (!g || nz < 0 || nf < 0 || (nz > 0 && (!z || !consensus)) || !xv || (nf > 0 && !xf) || !m || !n) return -1;
    std::vector<float> vzf, vzn;
    std::vector<int32_t> vidf, vret;
    g->policy.step(z, nz, consensus, support, xv, xf, nf, max_range, room, vzf, vidf, vzn, vret);
    if (zf) memcpy(zf, vzf.data(), sizeof(float) * vzf.size());
    if (idf) memcpy(idf, vidf.data(), sizeof(int32_t) * vidf.size());
    if (zn) memcpy(zn, vzn.data(), sizeof(float) * vzn.size());
    *m = (int32_t) vidf.size();
    *n = (int32_t) (vzn.size() / 2);
    if (retire) memcpy(retire, vret.data(), sizeof(int32_t) * vret.size());
    if (n_retire) *n_retire = (int32_t) vret.size();
    return 0;
}
void slamhost_gated_counts(const slamhost_gated *g, int32_t c[6]) {
    const slamhost::GatedPolicy &p = g->policy;
    c[0] = p.n_opened;
    c[1] = p.n_retired;
    c[2] = p.n_rescued;
    c[3] = p.n_discarded_votes;
    c[4] = p.n_new_refused;
    c[5] = p.active();
}

int slamhost_sim_conf(const slamhost_sim *s, slamhost_conf *out) {
    if (!s || !out) return -1;
    *out = static_cast<const slamhost_conf &>(s->sim.conf);
    return 0;
}

int slamhost_sim_map(const slamhost_sim *s, float *lm, float *wp) {
    if (!s) return -1;
    if (lm) memcpy(lm, s->sim.map.lm.data(), sizeof(float) * 2 * (size_t) s->sim.map.nlm);
    if (wp) memcpy(wp, s->sim.map.wp.data(), sizeof(float) * 2 * (size_t) s->sim.map.nwp);
    return 0;
}

int slamhost_sim_control(slamhost_sim *s, float *Vn, float *Gn, float *phi_true) {
    if (!s) return -1;
    const int r = s->sim.control();
    if (Vn) *Vn = s->sim.Vnoisy;
    if (Gn) *Gn = s->sim.Gnoisy;
    if (phi_true) *phi_true = s->sim.xTrue[2];
    return r;
}

int slamhost_sim_observe(slamhost_sim *s, int32_t nf_known, float *zf, int32_t *idf, int32_t *m, float *zn, int32_t *n) {
    if (!s || !m || !n) return -1;
    std::vector<float> vzf, vzn;
    std::vector<int32_t> vidf;
    s->sim.observe();
    s->sim.associate_known(nf_known, vzf, vidf, vzn);
    *m = (int32_t) vidf.size();
    *n = (int32_t) (vzn.size() / 2);
    if (zf && !vzf.empty()) memcpy(zf, vzf.data(), sizeof(float) * vzf.size());
    if (idf && !vidf.empty()) memcpy(idf, vidf.data(), sizeof(int32_t) * vidf.size());
    if (zn && !vzn.empty()) memcpy(zn, vzn.data(), sizeof(float) * vzn.size());
    return 0;
}

int slamhost_sim_last_z(const slamhost_sim *s, float *z, int32_t *vis, int32_t *nz) {
    if (!s) return -1;
    if (nz) *nz = (int32_t) s->sim.vis.size();
    if (z && !s->sim.z.empty()) memcpy(z, s->sim.z.data(), sizeof(float) * s->sim.z.size());
    if (vis && !s->sim.vis.empty()) memcpy(vis, s->sim.vis.data(), sizeof(int32_t) * s->sim.vis.size());
    return 0;
}

void slamhost_sim_true(const slamhost_sim *s, float x[3]) { memcpy(x, s->sim.xTrue, 3 * sizeof(float)); }
int64_t slamhost_sim_control_steps(const slamhost_sim *s) { return s->sim.control_steps; }

struct slamhost_ekf {
    slamhost::EkfSlam ekf;
    std::vector<float> table;
};

slamhost_ekf *slamhost_ekf_create(const slamhost_sim *s) {
    if (!s) return nullptr;
    slamhost_ekf *e = new slamhost_ekf();
    const slamhost::Conf &c = s->sim.conf;
    e->ekf.enableBatchUpdate = c.SWITCH_BATCH_UPDATE == 1;
    e->ekf.useHeading = c.SWITCH_HEADING_KNOWN == 1;
    e->ekf.wheelBase = c.WHEELBASE;
    e->ekf.gateReject = c.GATE_REJECT;
    e->ekf.gateAugment = c.GATE_AUGMENT;
    e->ekf.associationKnown = c.SWITCH_ASSOCIATION_KNOWN;
    e->ekf.sigmaPhi = c.sigmaT;
    e->table.assign((size_t) s->sim.map.nlm, -1.0f);
    return e;
}

void slamhost_ekf_destroy(slamhost_ekf *e) { delete e; }

int slamhost_ekf_step(slamhost_ekf *e, slamhost_sim *s) {
    if (!e || !s) return -1;
    slamhost::Simulator &sim = s->sim;
    const int r = sim.control();
    if (r < 0) return r;
    if (r == 1) sim.observe();
    const float phi = (float) (sim.xTrue[2] + sim.conf.sigmaT * slamhost::unif_rand());
    e->ekf.sim(sim.Vnoisy, sim.Gnoisy, sim.Qe, sim.dt, phi, sim.z, sim.vis, sim.Re, r == 1, sim.R, e->table);
    return r;
}

int slamhost_ekf_state(const slamhost_ekf *e, float *x, float *P, int32_t cap) {
    if (!e) return -1;
    const int d = (int) e->ekf.x.size();
    for (int i = 0; i < d && i < cap; i++) {
        if (x) x[i] = e->ekf.x[i];
        if (P)
            for (int j = 0; j < d && j < cap; j++) P[(size_t) i * cap + j] = e->ekf.P(i, j);
    }
    return d;
}

void slamhost_draw_normals(int32_t count, int32_t dim, float *out) {
    for (int32_t i = 0; i < count; i++) slamhost::randn(dim, 1, out + (size_t) i * dim);
}

int32_t slamhost_draw_strata(int32_t N, float *out) { return slamhost::stratified_random(N, out); }
double slamhost_unif_rand(void) { return slamhost::unif_rand(); }

void slamhost_synthetic_landmarks(uint64_t seed, int32_t n, float x0, float x1, float y0, float y1, float *lm) {
    // SplitMix64 -> 53-bit uniforms; x then y per landmark
    uint64_t st = seed;
    auto next = [&]() {
        uint64_t zz = (st += 0x9E3779B97F4A7C15ull);
        zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ull;
        zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBull;
        zz = zz ^ (zz >> 31);
        return (double) (zz >> 11) * (1.0 / 9007199254740992.0);
    };
    for (int32_t i = 0; i < n; i++) {
        lm[i] = (float) (x0 + (x1 - x0) * next());
        lm[(size_t) n + i] = (float) (y0 + (y1 - y0) * next());
    }
}

int slamhost_write_map(const char *path, const float *lm, int32_t nlm, const float *wp, int32_t nwp) {
    FILE *f = fopen(path, "wt");
    if (!f) return -1;
    fprintf(f, "#type rows cols\nlm 2 %d\n", nlm);
    for (int i = 0; i < nlm; i++) fprintf(f, "%.6f %.6f\n", lm[i], lm[(size_t) nlm + i]);
    fprintf(f, "wp 2 %d\n", nwp);
    for (int i = 0; i < nwp; i++) fprintf(f, "%.6f %.6f\n", wp[i], wp[(size_t) nwp + i]);
    fclose(f);
    return 0;
}

}  // extern "C"
