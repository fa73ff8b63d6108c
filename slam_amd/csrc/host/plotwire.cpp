// See plotwire.h.  Plain C++ + POSIX sockets; no ZeroMQ dependency.
#include "plotwire.h"

#include <arpa/inet.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <numeric>
#include <sstream>
#include <thread>

namespace slamhost {

// ---- zmqpp's serialisation (libs/zmqpp/message.cpp:225-328): network byte order -----------------------
static std::string be32(uint32_t v) {
    char b[4] = {(char) (v >> 24), (char) (v >> 16), (char) (v >> 8), (char) v};
    return std::string(b, 4);
}
static std::string be_f32(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return be32(u);
}
static std::string be_f64(double d) {
    uint64_t u;
    memcpy(&u, &d, 8);
    char b[8];
    for (int i = 0; i < 8; i++) b[i] = (char) (u >> (56 - 8 * i));
    return std::string(b, 8);
}

PlotMessage PlotEncoder::xy(const char *cmd, const double *xs, int32_t nx, const double *ys, int32_t ny) {
    PlotMessage m;
    m.reserve(3 + (size_t) nx + (size_t) ny);
    m.emplace_back(cmd);
    m.push_back(be32((uint32_t) nx));
    for (int32_t i = 0; i < nx; i++) m.push_back(be_f64(xs[i]));
    m.push_back(be32((uint32_t) ny));
    for (int32_t i = 0; i < ny; i++) m.push_back(be_f64(ys[i]));
    return m;
}

PlotMessage PlotEncoder::matrix(const char *cmd, uint32_t rows, uint32_t cols, const float *a, bool with_idx, int32_t idx) {
    PlotMessage m;
    m.emplace_back(cmd);
    m.push_back(be32(rows));
    m.push_back(be32(cols));
    for (uint32_t i = 0; i < rows; i++)
        for (uint32_t j = 0; j < cols; j++) m.push_back(be_f32(a[(size_t) i * cols + j]));
    if (with_idx) m.push_back(be32((uint32_t) idx));
    return m;
}

PlotMessage PlotEncoder::doubles(const char *cmd, const double *v, int n) {
    PlotMessage m;
    m.emplace_back(cmd);
    for (int i = 0; i < n; i++) m.push_back(be_f64(v[i]));
    return m;
}

PlotMessage PlotEncoder::car_size(double s, uint32_t id) { return PlotMessage{"setCarSize", be_f64(s), be32(id)}; }
PlotMessage PlotEncoder::u32(const char *cmd, uint32_t v) { return PlotMessage{cmd, be32(v)}; }
PlotMessage PlotEncoder::bare(const char *cmd) { return PlotMessage{cmd}; }
PlotMessage PlotEncoder::name(const std::string &n) { return PlotMessage{"setSimulationName", n}; }

// ---- frame file -------------------------------------------------------------------------------------------
FileSink::FileSink(const std::string &path) {
    f_ = fopen(path.c_str(), "wb");
    if (f_) {
        uint32_t zero = 0;
        fwrite(&zero, 4, 1, f_);
    }
}

bool FileSink::send(const PlotMessage &m, std::string *err) {
    if (!f_) {
        if (err) *err = "frame file is not open";
        return false;
    }
    const uint32_t nf = (uint32_t) m.size();
    fwrite(&nf, 4, 1, f_);
    for (const auto &fr : m) {
        const uint32_t l = (uint32_t) fr.size();
        fwrite(&l, 4, 1, f_);
        if (l) fwrite(fr.data(), 1, l, f_);
    }
    n_++;
    return true;
}

void FileSink::close() {
    if (!f_) return;
    fseek(f_, 0, SEEK_SET);
    fwrite(&n_, 4, 1, f_);
    fclose(f_);
    f_ = nullptr;
}

// ---- ZMTP 3.0, PAIR, NULL mechanism -----------------------------------------------------------------------
bool ZmtpPairClient::write_all(const void *p, size_t n, std::string *err) {
    const char *c = (const char *) p;
    while (n) {
        const ssize_t k = ::send(fd_, c, n, MSG_NOSIGNAL);
        if (k < 0) {
            if (errno == EINTR) continue;
            if (err) *err = std::string("plot socket write: ") + strerror(errno);
            return false;
        }
        c += k;
        n -= (size_t) k;
    }
    return true;
}

bool ZmtpPairClient::read_all(void *p, size_t n, std::string *err) {
    char *c = (char *) p;
    while (n) {
        struct pollfd pf = {fd_, POLLIN, 0};
        const int pr = poll(&pf, 1, 5000);
        if (pr <= 0) {
            if (err) *err = "plot socket: peer did not answer the ZMTP handshake";
            return false;
        }
        const ssize_t k = ::recv(fd_, c, n, 0);
        if (k <= 0) {
            if (k < 0 && errno == EINTR) continue;
            if (err) *err = "plot socket closed during the ZMTP handshake";
            return false;
        }
        c += k;
        n -= (size_t) k;
    }
    return true;
}

bool ZmtpPairClient::connect(const std::string &host, int port, double timeout_s, std::string *err) {
    struct addrinfo hints {}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    const std::string ps = std::to_string(port);
    if (getaddrinfo(host.c_str(), ps.c_str(), &hints, &res) != 0 || !res) {
        if (err) *err = "cannot resolve plot server " + host;
        return false;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {  // the reference blocks until slam-gui is there (README.md:28); here: retry until the timeout
        fd_ = socket(res->ai_family, res->ai_socktype, res->ai_protocol);
        if (fd_ >= 0 && ::connect(fd_, res->ai_addr, res->ai_addrlen) == 0) break;
        if (fd_ >= 0) ::close(fd_);
        fd_ = -1;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
            freeaddrinfo(res);
            if (err) *err = "no plot server at tcp://" + host + ":" + ps + " (start slam-gui first, or use -plot file:... / gather:...)";
            return false;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    freeaddrinfo(res);
    int one = 1;
    setsockopt(fd_, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    // greeting: signature (FF, 8 padding bytes, 7F), version 3.0, mechanism "NULL" (20 bytes), as-server 0, 31 filler bytes
    unsigned char g[64];
    memset(g, 0, sizeof g);
    g[0] = 0xFF;
    g[8] = 0x01;
    g[9] = 0x7F;
    g[10] = 3;
    g[11] = 0;
    memcpy(g + 12, "NULL", 4);
    unsigned char peer[64];
    if (!write_all(g, 64, err) || !read_all(peer, 64, err)) return false;
    if (peer[0] != 0xFF || peer[9] != 0x7F || peer[10] < 3) {
        if (err) *err = "plot server does not speak ZMTP 3";
        return false;
    }
    // READY command: 0x04 (command, short), size, [5]"READY" [11]"Socket-Type" [00 00 00 04]"PAIR"
    static const char body[] = "\x05READY\x0bSocket-Type\x00\x00\x00\x04PAIR";
    const unsigned char hdr[2] = {0x04, (unsigned char) (sizeof body - 1)};
    if (!write_all(hdr, 2, err) || !write_all(body, sizeof body - 1, err)) return false;
    unsigned char fh[2];
    if (!read_all(fh, 2, err)) return false;
    uint64_t len = fh[1];
    if (fh[0] & 0x02) {  // long command
        unsigned char rest[7];
        if (!read_all(rest, 7, err)) return false;
        len = fh[1];
        for (int i = 0; i < 7; i++) len = (len << 8) | rest[i];
    }
    if (len > 4096) {  // (a READY command is a few dozen bytes: the length is the peer's word, never an allocation size)
        if (err) *err = "plot server sent an oversized ZMTP command";
        return false;
    }
    std::vector<char> rb((size_t) len);
    if (len && !read_all(rb.data(), (size_t) len, err)) return false;
    if (!(fh[0] & 0x04) || len < 6 || memcmp(rb.data() + 1, "READY", 5) != 0) {
        if (err) *err = "plot server refused the ZMTP handshake";
        return false;
    }
    return true;
}

bool ZmtpPairClient::send(const PlotMessage &m, std::string *err) {
    if (fd_ < 0) {
        if (err) *err = "plot socket is not connected";
        return false;
    }
    std::string out;
    for (size_t i = 0; i < m.size(); i++) {
        const bool more = i + 1 < m.size();
        const size_t l = m[i].size();
        if (l <= 255) {
            out.push_back((char) (more ? 0x01 : 0x00));
            out.push_back((char) l);
        } else {
            out.push_back((char) ((more ? 0x01 : 0x00) | 0x02));
            for (int b = 7; b >= 0; b--) out.push_back((char) ((uint64_t) l >> (8 * b)));
        }
        out.append(m[i]);
    }
    return write_all(out.data(), out.size(), err);
}

void ZmtpPairClient::close() {
    if (fd_ >= 0) ::close(fd_);
    fd_ = -1;
}

// ---- DataGatherer behind the wire (Controller.cpp:60-230 -> DataGatherer.cpp) -----------------------------------
static uint32_t rd32(const std::string &s) {
    return ((uint32_t) (unsigned char) s[0] << 24) | ((uint32_t) (unsigned char) s[1] << 16) | ((uint32_t) (unsigned char) s[2] << 8) |
           (uint32_t) (unsigned char) s[3];
}
static float rd_f32(const std::string &s) {
    const uint32_t u = rd32(s);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static double rd_f64(const std::string &s) {
    uint64_t u = 0;
    for (int i = 0; i < 8; i++) u = (u << 8) | (unsigned char) s[i];
    double d;
    memcpy(&d, &u, 8);
    return d;
}

void GathererSink::cleanup() {
    errors_.clear();
    times_.clear();
    epx_.clear();
    epy_.clear();
    tpx_.clear();
    tpy_.clear();
    counts_.clear();
    avglen_.clear();
}

template <class T>
static void stats_line(const char *title, const std::vector<T> &v, std::ostream &out) {
    // DataGatherer::outputErrorsStats / outputTimesStats (:22-48), default ostream formatting
    const double sum = std::accumulate(v.begin(), v.end(), 0.0);
    const double mean = sum / v.size();
    const double sq_sum = std::inner_product(v.begin(), v.end(), v.begin(), 0.0);
    const double stdev = std::sqrt(sq_sum / v.size() - mean * mean);
    const double minimum = *std::min_element(v.begin(), v.end());
    const double maximum = *std::max_element(v.begin(), v.end());
    out << title << ":\nMean: " << mean << " Std: " << stdev << " Min: " << minimum << " Max: " << maximum << "\n";
}

void GathererSink::save_data() {
    const std::string dir = base_.empty() ? name_ : base_ + "/" + name_;
    if (!base_.empty()) mkdir(base_.c_str(), S_IRWXU | S_IRWXG | S_IROTH | S_IXOTH);
    mkdir(dir.c_str(), S_IRWXU | S_IRWXG | S_IROTH | S_IXOTH);
    std::ofstream results(dir + "/results.txt"), errs(dir + "/errors.txt"), times(dir + "/times.txt"), pos(dir + "/positions.txt"),
        counts(dir + "/observedCounts.txt"), avg(dir + "/averageLengthLandmark.txt");
    if (!errors_.empty()) stats_line("Errors", errors_, results);
    if (!times_.empty()) stats_line("Times", times_, results);
    for (double x : errors_) errs << std::setprecision(10) << x << "\n";
    for (uint32_t x : times_) times << std::setprecision(10) << x << "\n";
    for (long x : counts_) counts << x << "\n";
    for (float x : avglen_) avg << x << "\n";
    for (size_t i = 0; i < epx_.size(); i++)
        pos << std::setprecision(10) << tpx_[i] << ", " << tpy_[i] << ", " << epx_[i] << ", " << epy_[i] << "\n";
}

bool GathererSink::send(const PlotMessage &m, std::string *) {
    if (m.empty()) return true;
    const std::string &c = m[0];
    if (c == "setLaserLines" && m.size() >= 3) {
        const uint32_t rows = rd32(m[1]), cols = rd32(m[2]);
        float sumlens = 0;
        if (rows >= 4 && m.size() >= 3 + (size_t) rows * cols)
            for (uint32_t i = 0; i < cols; i++) {
                const float xlen = rd_f32(m[3 + 0 * cols + i]) - rd_f32(m[3 + 2 * cols + i]);
                const float ylen = rd_f32(m[3 + 1 * cols + i]) - rd_f32(m[3 + 3 * cols + i]);
                sumlens += std::sqrt(xlen * xlen + ylen * ylen);
            }
        avglen_.push_back(sumlens / cols);  // cols == 0: NaN, as upstream (Controller.cpp:135)
        counts_.push_back((long) cols);
    } else if (c == "setCarTruePosition" && m.size() >= 4) {
        tx_ = rd_f64(m[1]);
        ty_ = rd_f64(m[2]);
    } else if (c == "setCarEstimatedPosition" && m.size() >= 4) {
        ex_ = rd_f64(m[1]);
        ey_ = rd_f64(m[2]);
    } else if (c == "plot") {  // DataGatherer::nextTurn (:103-115)
        errors_.push_back(std::sqrt(std::pow(tx_ - ex_, 2) + std::pow(ty_ - ey_, 2)));
        epx_.push_back(ex_);
        epy_.push_back(ey_);
        tpx_.push_back(tx_);
        tpy_.push_back(ty_);
        turn_++;
        if (turn_ % 100 == 0) save_data();
    } else if (c == "setSimulationName" && m.size() >= 2) {
        name_ = m[1];
        cleanup();
    } else if (c == "endPlot") {
        save_data();
        cleanup();
    } else if (c == "loopTime" && m.size() >= 2) {
        times_.push_back(rd32(m[1]));
    }
    return true;
}

// ---- Plot ---------------------------------------------------------------------------------------------------------
bool Plot::open(const std::string &spec, std::string *err) {
    size_t at = 0;
    while (at <= spec.size()) {
        size_t comma = spec.find(',', at);
        if (comma == std::string::npos) comma = spec.size();
        const std::string one = spec.substr(at, comma - at);
        at = comma + 1;
        if (one.empty() || one == "none") continue;
        if (one.rfind("tcp://", 0) == 0) {
            const std::string hp = one.substr(6);
            const size_t colon = hp.rfind(':');
            if (colon == std::string::npos) {
                if (err) *err = "plot endpoint needs a port: " + one;
                return false;
            }
            std::unique_ptr<ZmtpPairClient> c(new ZmtpPairClient());
            if (!c->connect(hp.substr(0, colon), atoi(hp.c_str() + colon + 1), 10.0, err)) return false;
            sinks_.push_back(std::move(c));
        } else if (one.rfind("file:", 0) == 0) {
            std::unique_ptr<FileSink> f(new FileSink(one.substr(5)));
            if (!f->ok()) {
                if (err) *err = "cannot create " + one.substr(5);
                return false;
            }
            sinks_.push_back(std::move(f));
        } else if (one.rfind("gather:", 0) == 0) {
            sinks_.push_back(std::unique_ptr<PlotSink>(new GathererSink(one.substr(7))));
        } else {
            if (err) *err = "unknown plot sink '" + one + "' (tcp://host:port | file:<path> | gather:<dir> | none)";
            return false;
        }
    }
    return true;
}

void Plot::close() {
    for (auto &s : sinks_) s->close();
    sinks_.clear();
}

bool Plot::emit(const PlotMessage &m) {
    bool ok = true;
    for (auto &s : sinks_)
        if (!s->send(m, &err_)) ok = false;
    return ok;
}

}  // namespace slamhost

// ---- C ABI (include/slamhost.h) ---------------------------------------------------------------------------------------
namespace slamhost {
void set_error(const std::string &e);
}

extern "C" {
#include "../../../include/slamhost.h"

slamhost_plot *slamhost_plot_open(const char *spec) {
    slamhost_plot *p = new slamhost_plot();
    std::string err;
    if (!p->plot.open(spec ? spec : "none", &err)) {
        slamhost::set_error(err);
        delete p;
        return nullptr;
    }
    return p;
}

void slamhost_plot_close(slamhost_plot *p) {
    if (!p) return;
    p->plot.close();
    delete p;
}

static int done(slamhost_plot *p, bool ok) {
    if (!ok) slamhost::set_error(p->plot.error());
    return ok ? 0 : -1;
}

int slamhost_plot_xy(slamhost_plot *p, const char *cmd, const double *xs, int32_t nx, const double *ys, int32_t ny) {
    if (!p || !cmd || nx < 0 || ny < 0) return -1;
    return done(p, p->plot.emit(slamhost::PlotEncoder::xy(cmd, xs, nx, ys, ny)));
}

int slamhost_plot_matrix(slamhost_plot *p, const char *cmd, uint32_t rows, uint32_t cols, const float *a, int32_t idx) {
    if (!p || !cmd) return -1;
    return done(p, p->plot.emit(slamhost::PlotEncoder::matrix(cmd, rows, cols, a, strcmp(cmd, "setCovEllipse") == 0, idx)));
}

int slamhost_plot_doubles(slamhost_plot *p, const char *cmd, const double *v, int32_t n) {
    if (!p || !cmd || n < 0) return -1;
    return done(p, p->plot.emit(slamhost::PlotEncoder::doubles(cmd, v, n)));
}

int slamhost_plot_car_size(slamhost_plot *p, double s, uint32_t id) {
    if (!p) return -1;
    return done(p, p->plot.emit(slamhost::PlotEncoder::car_size(s, id)));
}

int slamhost_plot_u32(slamhost_plot *p, const char *cmd, uint32_t v) {
    if (!p || !cmd) return -1;
    if (strcmp(cmd, "setCurrentIteration") == 0) return 0;  // disabled upstream: nothing is sent
    return done(p, p->plot.emit(slamhost::PlotEncoder::u32(cmd, v)));
}

int slamhost_plot_cmd(slamhost_plot *p, const char *cmd) {
    if (!p || !cmd) return -1;
    return done(p, p->plot.emit(slamhost::PlotEncoder::bare(cmd)));
}

int slamhost_plot_name(slamhost_plot *p, const char *name) {
    if (!p || !name) return -1;
    return done(p, p->plot.emit(slamhost::PlotEncoder::name(name)));
}
}
