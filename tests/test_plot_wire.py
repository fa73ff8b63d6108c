"""Plot wire format + headless sink: the product's encoder (slam_amd/csrc/host/plotwire.cpp,
C ABI slamhost_plot_*) against

  * tests/golden/networkplot_frames.bin -- every frame of every message the REFERENCE'S OWN NetworkPlot
    (src/backend/plotting/NetworkPlot.cpp over the vendored libs/zmqpp) sent to a ZeroMQ PAIR server for a fixed call
    sequence (oracle/plot_driver.cpp, built by `make -C oracle plotgolden`): byte-identical;
  * a real libzmq PAIR socket (the image's /opt/conda/lib/libzmq.so.5, through ctypes) on the other side of the
    product's ZMTP 3.0 client: what slam-gui would receive;
  * the GUI's DataGatherer file formats (src/gui/plotting/DataGatherer.cpp:50-138)."""
import ctypes as C
import os
import struct
import threading

import numpy as np
import pytest

from conftest import GOLDEN


def read_container(path):
    b = open(path, "rb").read()
    off = 0
    (n,) = struct.unpack_from("<I", b, off)
    off += 4
    msgs = []
    for _ in range(n):
        (nf,) = struct.unpack_from("<I", b, off)
        off += 4
        fr = []
        for _ in range(nf):
            (ln,) = struct.unpack_from("<I", b, off)
            off += 4
            fr.append(b[off:off + ln])
            off += ln
        msgs.append(fr)
    assert off == len(b)
    return msgs


def golden_sequence(p):
    """oracle/plot_driver.cpp's call sequence, argument for argument"""
    f32 = np.float32
    p.name("golden run")
    p.cmd("clear")
    p.car_size(4.0, 0)
    p.car_size(4.0, 1)
    p.xy("setWaypoints", [0.0, 10.5, -3.25], [1.0, -2.0, 7.125])
    p.xy("setLandmarks", [2.9922, -15.5, 1e-3, 100.0, -130.0], [-25.7009, 20.25, -1e5, 90.0, 3.0])
    p.doubles("setPlotRange", -136.5, 106.5, -109.5, 99.5)
    p.doubles("addTruePosition", 0.0, 0.0)
    p.doubles("setCarTruePosition", 0.0, 0.0, 0.0)
    p.doubles("addEstimatedPosition", 0.0, 0.0)
    p.doubles("setCarEstimatedPosition", 0.0, 0.0, 0.0)
    p.cmd("plot")
    p.u32("loopTime", 1234)
    p.u32("setCurrentIteration", 7)  # sends nothing, as upstream
    p.xy("setParticles", [0.61, 0.62, 0.63, 0.64], [-0.02, -0.03, -0.01, 0.0])
    p.xy("setFeatureParticles", [], [])
    p.xy("setFeatureParticles", [3.19, 2.85, -1.5], [-25.56, -26.0, 12.75])
    p.doubles("addTruePosition", 0.6154, -0.0248)
    p.doubles("addEstimatedPosition", 0.61504266, -0.02534972)
    p.doubles("setCarTruePosition", 0.6154, -0.0248, -0.00613)
    p.doubles("setCarEstimatedPosition", 0.61504266, -0.02534972, -0.00570246)
    lines = np.array([[0.6154, 0.6154, 0.6154], [-0.0248, -0.0248, -0.0248], [3.19, 2.85, -1.5], [-25.56, -26.0, 12.75]], f32)
    p.matrix("setLaserLines", lines)
    p.matrix("setLaserLines", np.zeros((0, 0), f32))
    ell = np.array([[1, 2, 3, 4], [-1, -2, -3, -4]], f32)
    p.u32("covEllipseAdd", 2)
    p.matrix("setCovEllipse", ell, 0)
    p.matrix("setCovEllipse", ell, 5)
    p.u32("loopTime", 4000000000)
    p.cmd("plot")
    p.cmd("endPlot")


def test_frames_are_byte_identical_to_the_reference_networkplot(tmp_path):
    from slam_amd import host
    path = str(tmp_path / "frames.bin")
    p = host.Plot("file:" + path)
    golden_sequence(p)
    p.close()
    got, exp = read_container(path), read_container(os.path.join(GOLDEN, "networkplot_frames.bin"))
    assert len(got) == len(exp) == 28
    for k, (g, e) in enumerate(zip(got, exp)):
        assert g == e, (k, e[0], [x.hex() for x in g][:6], [x.hex() for x in e][:6])


def test_unknown_sink_and_missing_server_fail_loudly():
    from slam_amd import host
    with pytest.raises(RuntimeError, match="unknown plot sink"):
        host.Plot("udp://nowhere")


ZMQ_PAIR, ZMQ_RCVMORE, ZMQ_RCVTIMEO, ZMQ_LINGER = 0, 13, 27, 17


def load_libzmq():
    for cand in ("/opt/conda/lib/libzmq.so.5", "libzmq.so.5"):
        try:
            return C.CDLL(cand)
        except OSError:
            continue
    return None


def test_zmtp_client_talks_to_a_real_libzmq_pair_socket(tmp_path):
    """What the reference's slam-gui (zmqpp over libzmq, PAIR, bind tcp://*:4242) receives from the product's client:
    the same multipart messages as the frame file holds."""
    Z = load_libzmq()
    if Z is None:
        pytest.skip("no libzmq in this image")
    from slam_amd import host
    Z.zmq_ctx_new.restype = C.c_void_p
    Z.zmq_socket.restype = C.c_void_p
    Z.zmq_socket.argtypes = [C.c_void_p, C.c_int]
    Z.zmq_bind.argtypes = [C.c_void_p, C.c_char_p]
    Z.zmq_setsockopt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    Z.zmq_getsockopt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
    Z.zmq_recv.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    Z.zmq_close.argtypes = [C.c_void_p]
    Z.zmq_ctx_term.argtypes = [C.c_void_p]
    ctx = Z.zmq_ctx_new()
    sock = Z.zmq_socket(ctx, ZMQ_PAIR)
    port = 42000 + os.getpid() % 2000
    assert Z.zmq_bind(sock, ("tcp://127.0.0.1:%d" % port).encode()) == 0
    tmo = C.c_int(10000)
    Z.zmq_setsockopt(sock, ZMQ_RCVTIMEO, C.byref(tmo), C.sizeof(tmo))
    received = []

    def server():
        buf = C.create_string_buffer(1 << 16)
        while True:
            frames = []
            while True:
                n = Z.zmq_recv(sock, buf, len(buf), 0)
                if n < 0:
                    return
                frames.append(buf.raw[:n])
                more = C.c_int(0)
                sz = C.c_size_t(C.sizeof(more))
                Z.zmq_getsockopt(sock, ZMQ_RCVMORE, C.byref(more), C.byref(sz))
                if not more.value:
                    break
            received.append(frames)
            if frames[0] == b"endPlot":
                return
    th = threading.Thread(target=server)
    th.start()
    path = str(tmp_path / "tee.bin")
    p = host.Plot("tcp://127.0.0.1:%d,file:%s" % (port, path))
    golden_sequence(p)
    # a frame longer than 255 bytes exercises the long-frame encoding... after endPlot the server thread is gone, so before:
    th.join(timeout=15)
    p.close()
    lin = C.c_int(0)
    Z.zmq_setsockopt(sock, ZMQ_LINGER, C.byref(lin), C.sizeof(lin))
    Z.zmq_close(sock)
    Z.zmq_ctx_term(ctx)
    exp = read_container(os.path.join(GOLDEN, "networkplot_frames.bin"))
    assert not th.is_alive() and len(received) == len(exp)
    for g, e in zip(received, exp):
        assert g == e, (e[0], g[:3])
    assert read_container(path) == exp


def test_headless_gatherer_writes_the_gui_side_files(tmp_path):
    """gather:<dir> = DataGatherer fed as Controller.cpp feeds it: error per `plot`, loop times, observed counts and mean
    laser-line length per setLaserLines; files written at endPlot (and every 100 turns) with the reference's formatting
    (setprecision(10) for errors / times / positions, default for the rest)."""
    from slam_amd import host
    base = str(tmp_path / "out")
    p = host.Plot("gather:" + base)
    p.name("runA")
    rng = np.random.default_rng(5)
    true = np.cumsum(rng.normal(size=(7, 2)), axis=0)
    est = true + rng.normal(size=(7, 2)) * 0.1
    times = [1000, 1500, 70000, 1200, 999, 4000000000, 1]
    for k in range(7):
        p.u32("loopTime", times[k])
        p.doubles("setCarTruePosition", true[k, 0], true[k, 1], 0.1 * k)
        p.doubles("setCarEstimatedPosition", est[k, 0], est[k, 1], 0.1 * k)
        lines = np.array([[true[k, 0]] * 2, [true[k, 1]] * 2, [true[k, 0] + 3, true[k, 0]], [true[k, 1], true[k, 1] - 4]], np.float32)
        p.matrix("setLaserLines", lines)
        p.cmd("plot")
    p.cmd("endPlot")
    p.close()
    d = os.path.join(base, "runA")
    errs = np.loadtxt(os.path.join(d, "errors.txt"))
    np.testing.assert_allclose(errs, np.hypot(*(true - est).T), rtol=1e-9)
    assert [int(x) for x in open(os.path.join(d, "times.txt")).read().split()] == times
    pos = np.loadtxt(os.path.join(d, "positions.txt"), delimiter=",")
    np.testing.assert_allclose(pos, np.hstack([true, est]), rtol=1e-9)
    assert open(os.path.join(d, "observedCounts.txt")).read().split() == ["2"] * 7
    np.testing.assert_allclose(np.loadtxt(os.path.join(d, "averageLengthLandmark.txt")), 3.5, rtol=1e-5)
    res = open(os.path.join(d, "results.txt")).read().splitlines()
    assert res[0] == "Errors:" and res[1].startswith("Mean: ") and res[2] == "Times:" and " Max: 4e+09" in res[3]
    mean = float(res[1].split()[1])
    assert abs(mean - errs.mean()) <= 1e-5 * errs.mean() + 1e-6  # 6 significant digits (default ostream precision)


def test_slam_backend_plot_sinks_headless_ekf(tmp_path):
    """slam-backend -plot gather:<dir>,file:<frames> (EKF1 runs on the host: no GPU needed): the message stream has the
    reference's shape -- configurePlot preamble (slamwrapper.cpp:94-110), then per control step loopTime, positions, laser
    lines, plot (ekfslamwrapper.cpp:86-105), endPlot -- and the gatherer files agree with the CSV log of the same run."""
    import subprocess
    from conftest import DATA
    root = os.path.dirname(DATA)
    exe = os.path.join(root, "slam_amd", "bin", "slam-backend")
    frames, log = str(tmp_path / "frames.bin"), str(tmp_path / "log.csv")
    r = subprocess.run([exe, "-m", os.path.join(DATA, "example_loop1.mat"), "-method", "EKF1", "-SWITCH_SEED_RANDOM", "3", "-maxsteps", "230",
                        "-n", "ekfrun", "-plot", "gather:%s,file:%s" % (tmp_path / "g", frames), "-log", log], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr
    msgs = read_container(frames)
    names = [m[0].decode() for m in msgs]
    assert names[:11] == ["setSimulationName", "setCarSize", "setCarSize", "setWaypoints", "setLandmarks", "setPlotRange", "addTruePosition",
                          "setCarTruePosition", "addEstimatedPosition", "setCarEstimatedPosition", "plot"]
    assert msgs[0][1] == b"ekfrun" and names[-1] == "endPlot"
    per_step = ["loopTime", "addTruePosition", "addEstimatedPosition", "setCarTruePosition", "setCarEstimatedPosition", "setLaserLines", "plot"]
    body = names[11:-1]
    assert len(body) == 230 * len(per_step) and body[:len(per_step)] == per_step and body[-len(per_step):] == per_step
    rows = np.loadtxt(log, delimiter=",", skiprows=1)
    pos = np.loadtxt(str(tmp_path / "g" / "ekfrun" / "positions.txt"), delimiter=",")
    assert pos.shape == (231, 4)  # the configurePlot turn + one per control step
    np.testing.assert_allclose(pos[1:], rows[:, [1, 2, 4, 5]], atol=2e-6)
    counts = [int(x) for x in open(str(tmp_path / "g" / "ekfrun" / "observedCounts.txt")).read().split()]
    assert len(counts) == 230 and max(counts) >= 1
