// Plot wire format of matzipan/slam's backend -> GUI link, and a headless sink for it (product code, host C++).
//
// The reference's NetworkPlot (src/backend/plotting/NetworkPlot.cpp:22-218) sends one ZeroMQ multipart message per
// command over a PAIR socket connected to tcp://127.0.0.1:4242: frame 0 = the command name (raw bytes), then ONE FRAME
// PER VALUE in network byte order, exactly as the vendored zmqpp serialises `message << value`
// (libs/zmqpp/message.cpp:225-328: int32/uint32 = 4 bytes big-endian, float = 4, double = 8 big-endian, string = raw).
//   setLandmarks / setWaypoints / setParticles / setFeatureParticles : int32 nx, nx doubles, int32 ny, ny doubles
//   setLaserLines   : uint32 rows, uint32 cols, rows*cols floats (row-major walk of the Eigen matrix)
//   setCovEllipse   : same, then int32 idx
//   addTruePosition / addEstimatedPosition : 2 doubles;  setCarTruePosition / setCarEstimatedPosition : 3 doubles
//   setCarSize : double, uint32;  setPlotRange : 4 doubles;  loopTime / covEllipseAdd : uint32
//   setSimulationName : string;  clear / plot / endPlot : no payload;  setCurrentIteration : sends nothing upstream (:176-186)
// PlotEncoder produces those frames byte for byte (golden: tests/golden/networkplot_frames.bin, captured from the
// reference's own NetworkPlot); sinks carry them: a ZMTP 3.0 PAIR client (the existing slam-gui listens with libzmq), a
// frame file, and GathererSink = the GUI side's DataGatherer (src/gui/plotting/DataGatherer.cpp:50-138, fed the way
// Controller.cpp:60-230 feeds it), so that slam-backend can write results/errors/times/positions files with no GUI.
#pragma once
#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

namespace slamhost {

using PlotMessage = std::vector<std::string>;  // frames

struct PlotEncoder {
    static PlotMessage xy(const char *cmd, const double *xs, int32_t nx, const double *ys, int32_t ny);
    static PlotMessage matrix(const char *cmd, uint32_t rows, uint32_t cols, const float *row_major, bool with_idx, int32_t idx);
    static PlotMessage doubles(const char *cmd, const double *v, int n);
    static PlotMessage car_size(double s, uint32_t id);
    static PlotMessage u32(const char *cmd, uint32_t v);
    static PlotMessage bare(const char *cmd);
    static PlotMessage name(const std::string &n);
};

class PlotSink {
   public:
    virtual ~PlotSink() {}
    virtual bool send(const PlotMessage &m, std::string *err) = 0;
    virtual void close() {}
};

// u32 n_messages (patched at close), then per message: u32 n_frames, per frame: u32 length + bytes (little-endian)
class FileSink : public PlotSink {
   public:
    explicit FileSink(const std::string &path);
    ~FileSink() override { close(); }
    bool ok() const { return f_ != nullptr; }
    bool send(const PlotMessage &m, std::string *err) override;
    void close() override;

   private:
    FILE *f_ = nullptr;
    uint32_t n_ = 0;
};

// Minimal ZMTP 3.0 client, PAIR socket type, NULL security: enough to talk to the libzmq PAIR socket slam-gui binds.
class ZmtpPairClient : public PlotSink {
   public:
    ZmtpPairClient() {}
    ~ZmtpPairClient() override { close(); }
    bool connect(const std::string &host, int port, double timeout_s, std::string *err);
    bool send(const PlotMessage &m, std::string *err) override;
    void close() override;

   private:
    int fd_ = -1;
    bool write_all(const void *p, size_t n, std::string *err);
    bool read_all(void *p, size_t n, std::string *err);
};

// The GUI side's DataGatherer driven by the wire messages as Controller.cpp drives it.
class GathererSink : public PlotSink {
   public:
    explicit GathererSink(const std::string &base_dir) : base_(base_dir) {}
    bool send(const PlotMessage &m, std::string *err) override;
    void save_data();  // DataGatherer::saveData

   private:
    std::string base_, name_ = "simulation";
    std::vector<double> errors_, epx_, epy_, tpx_, tpy_;
    std::vector<uint32_t> times_;
    std::vector<long> counts_;
    std::vector<float> avglen_;
    double tx_ = 0, ty_ = 0, ex_ = 0, ey_ = 0;
    long turn_ = 0;
    void cleanup();
};

// NetworkPlot's method surface over any number of sinks.
class Plot {
   public:
    // spec: "tcp://host:port" | "file:<path>" | "gather:<dir>" ; several separated by ','
    bool open(const std::string &spec, std::string *err);
    void close();
    bool active() const { return !sinks_.empty(); }
    const std::string &error() const { return err_; }
    bool emit(const PlotMessage &m);
    bool setLandmarks(const std::vector<double> &x, const std::vector<double> &y) { return emit(PlotEncoder::xy("setLandmarks", x.data(), (int32_t) x.size(), y.data(), (int32_t) y.size())); }
    bool setWaypoints(const std::vector<double> &x, const std::vector<double> &y) { return emit(PlotEncoder::xy("setWaypoints", x.data(), (int32_t) x.size(), y.data(), (int32_t) y.size())); }
    bool setParticles(const std::vector<double> &x, const std::vector<double> &y) { return emit(PlotEncoder::xy("setParticles", x.data(), (int32_t) x.size(), y.data(), (int32_t) y.size())); }
    bool setFeatureParticles(const std::vector<double> &x, const std::vector<double> &y) { return emit(PlotEncoder::xy("setFeatureParticles", x.data(), (int32_t) x.size(), y.data(), (int32_t) y.size())); }
    bool setLaserLines(uint32_t rows, uint32_t cols, const float *row_major) { return emit(PlotEncoder::matrix("setLaserLines", rows, cols, row_major, false, 0)); }
    bool setCovEllipse(uint32_t rows, uint32_t cols, const float *row_major, int32_t idx) { return emit(PlotEncoder::matrix("setCovEllipse", rows, cols, row_major, true, idx)); }
    bool addTruePosition(double x, double y) { const double v[2] = {x, y}; return emit(PlotEncoder::doubles("addTruePosition", v, 2)); }
    bool addEstimatedPosition(double x, double y) { const double v[2] = {x, y}; return emit(PlotEncoder::doubles("addEstimatedPosition", v, 2)); }
    bool setCarSize(double s, uint32_t id) { return emit(PlotEncoder::car_size(s, id)); }
    bool setCarTruePosition(double x, double y, double t) { const double v[3] = {x, y, t}; return emit(PlotEncoder::doubles("setCarTruePosition", v, 3)); }
    bool setCarEstimatedPosition(double x, double y, double t) { const double v[3] = {x, y, t}; return emit(PlotEncoder::doubles("setCarEstimatedPosition", v, 3)); }
    bool setPlotRange(double a, double b, double c, double d) { const double v[4] = {a, b, c, d}; return emit(PlotEncoder::doubles("setPlotRange", v, 4)); }
    bool clear() { return emit(PlotEncoder::bare("clear")); }
    bool setSimulationName(const std::string &n) { return emit(PlotEncoder::name(n)); }
    bool plot() { return emit(PlotEncoder::bare("plot")); }
    bool endPlot() { return emit(PlotEncoder::bare("endPlot")); }
    bool covEllipseAdd(uint32_t n) { return emit(PlotEncoder::u32("covEllipseAdd", n)); }
    bool loopTime(uint32_t us) { return emit(PlotEncoder::u32("loopTime", us)); }
    bool setCurrentIteration(uint32_t) { return true; }  // disabled upstream (NetworkPlot.cpp:176-186): nothing is sent

   private:
    std::vector<std::unique_ptr<PlotSink>> sinks_;
    std::string err_;
};

}  // namespace slamhost

struct slamhost_plot {
    slamhost::Plot plot;
};
