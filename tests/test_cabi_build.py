"""CPU side of the executable boundary (tests/cabi): the two reference-side bindings INTEGRATION.md documents compile with
plain g++ against include/slamgpu.h, the harness library links, and -- where the reference tree is present (authoring
container) -- they compile against the reference's own headers and Eigen, the adapter instantiated with the reference's
Particle / VectorXf / MatrixXf and called the way fastslam2wrapper.cpp:64,88 calls its algorithm object, the handler in the
place of AcceleratorHandler.h inside the reference's -DJACOBIAN_ACCELERATOR branch of core.cpp."""
import ctypes as C
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"


def test_harness_builds_and_exports():
    import slam_amd
    slam_amd.load_library()   # libslamgpu.so must exist for the link (built by __graft_entry__.build())
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cabi")])
    L = C.CDLL(os.path.join(HERE, "cabi", "libcabi_driver.so"))
    for sym in ("cabi_compute_jacobians", "cabi_algo_create", "cabi_algo_predict", "cabi_algo_update", "cabi_algo_estimate",
                "cabi_algo_landmarks", "cabi_algo_fetch", "cabi_algo_destroy", "cabi_last_error"):
        assert hasattr(L, sym), sym


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "backend")), reason="reference tree absent (GPU box)")
def test_bindings_compile_against_the_reference_headers(tmp_path):
    flags = ["g++", "-std=c++11", "-fsyntax-only", "-w", "-msse", "-D__STDC_CONSTANT_MACROS", "-I" + os.path.join(ROOT, "include"),
             "-I" + os.path.join(HERE, "cabi"), "-I" + REF, "-I" + os.path.join(REF, "src", "backend"), "-I" + os.path.join(REF, "libs", "eigen3")]
    # seam 1: the reference's core.cpp, accelerator branch, with the drop-in class where AcceleratorHandler.h would be
    subprocess.check_call(flags + ["-DJACOBIAN_ACCELERATOR", "-DSLAM_OCMHANDLER_H", "-include", os.path.join(HERE, "cabi", "accel_shim.h"),
                                   os.path.join(REF, "src", "backend", "core.cpp")])
    # seam 2: the adapter over the reference's types, called with the wrapper's argument lists
    tu = tmp_path / "adapter_tu.cpp"
    tu.write_text('''
#include "core.h"
#include "fastslam2gpu_adapter.h"
template class FastSLAMGpuT<Particle, VectorXf, MatrixXf>;
typedef FastSLAMGpuT<Particle, VectorXf, MatrixXf> FastSLAM2Gpu;
void wrapper_calls(FastSLAM2Gpu *algorithm, vector<Particle> &particles, VectorXf &xTrue, float Vn, float Gn, MatrixXf &Qe, float dt,
                   vector<VectorXf> &zf, vector<VectorXf> &zn, vector<int> &idf, vector<VectorXf> &z, VectorXf &table, MatrixXf &Re) {
    algorithm->predict(particles, xTrue, Vn, Gn, Qe, dt);
    algorithm->update(particles, zf, zn, idf, z, table, Re);
}
''')
    subprocess.check_call(flags + [str(tu)])


def test_integration_md_shows_the_compiled_bindings():
    """The code blocks of INTEGRATION.md are the compiled files, not a paraphrase of them."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    shim = open(os.path.join(HERE, "cabi", "accel_shim.h")).read()
    adapter = open(os.path.join(HERE, "cabi", "fastslam2gpu_adapter.h")).read()
    assert shim[shim.index("class AcceleratorHandler {"):shim.rindex("#endif")].rstrip() in doc
    assert adapter[adapter.index("template <class Particle, class VectorXf, class MatrixXf>"):adapter.index("#endif")].rstrip() in doc


def test_integration_md_per_particle_snippet_compiles(tmp_path):
    """The per-particle association call INTEGRATION.md shows (slamgpu_particle_assoc + slamgpu_update_particle) is cut out of the
    document and compiled against the header: the field names and the argument list a maintainer copies are the real ones."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    a = doc.index("slamgpu_particle_assoc opt = {};")
    b = doc.index("```", a)
    body = doc[a:b]
    assert "slamgpu_update_particle(ctx, z, nz, R, &opt, normals, strata, report)" in body
    tu = tmp_path / "pp_snippet.cpp"
    tu.write_text("#include <cmath>\n#include <cstdint>\n#include <slamgpu.h>\nstatic int check(int rc) { return rc; }\n"
                  "int run(slamgpu_ctx *ctx, const float *z, int nz, const float R[4], float GATE_REJECT, float GATE_AUGMENT, const float *normals,\n"
                  "        const float *strata) {\n" + body + "  return report[0];\n}\n")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-c", str(tu), "-o", str(tmp_path / "pp_snippet.o")])


def test_headers_are_plain_c(tmp_path):
    """The boundary is a C ABI: include/slamgpu.h (stable part and experimental block) and include/slamhost.h compile as C99 with
    -pedantic, and a C program links against the library's version symbol."""
    src = tmp_path / "abi.c"
    src.write_text("#define SLAMGPU_EXPERIMENTAL 1\n#include <slamgpu.h>\n#include <slamhost.h>\n"
                   "int main(void) { return slamgpu_abi_version() == SLAMGPU_ABI_VERSION ? 0 : 1; }\n")
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L" + os.path.join(ROOT, "slam_amd"), "-lslamgpu", "-Wl,-rpath," + os.path.join(ROOT, "slam_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.call([str(exe)]) == 0
