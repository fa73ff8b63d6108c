#!/usr/bin/env python3
"""Static instruction table of the strict build's building blocks (tools/microbench/strict_costs.hip), with and without the
compiler's SLP vectoriser (the pass that forms v_pk_mul_f32 / v_pk_add_f32): prints the table of profiles/strict_instruction_table_r06.txt.
CPU only (hipcc -S cross-compiles gfx950)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "microbench", "strict_costs.hip")
FLAGS = ["-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "--cuda-device-only", "-S", "-Wno-unused-value", "-Wno-unused-command-line-argument"]


def counts(extra):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "a.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", out, SRC])
        cur, res = None, collections.OrderedDict()
        for ln in open(out):
            m = re.match(r"^(k_[A-Za-z0-9_]+):", ln)
            if m:
                cur = m.group(1)
                res[cur] = collections.Counter()
                continue
            if cur and (ln.startswith("\t.end_amdhsa_kernel") or ln.startswith(".Lfunc_end")):
                cur = None
            m = re.match(r"^\s+(v_[a-z0-9_]+)", ln)
            if cur and m:
                res[cur][m.group(1)] += 1
        return res


def row(c):
    valu = sum(c.values())
    pk = sum(v for k, v in c.items() if k.startswith("v_pk_"))
    mov = sum(v for k, v in c.items() if k.startswith("v_mov") or k.startswith("v_accvgpr"))
    div = c["v_div_scale_f32"] + c["v_div_fmas_f32"] + c["v_div_fixup_f32"] + c["v_rcp_f32_e32"]
    f64 = sum(v for k, v in c.items() if "f64" in k)
    return valu, pk, mov, div, f64


def main():
    slp, noslp = counts([]), counts(["-fno-slp-vectorize"])
    base = row(slp["k_baseline_16_in_8_out"])[0]
    print("%-28s | %26s | %26s" % ("", "as the build compiles it", "-fno-slp-vectorize"))
    print("%-28s | %5s %4s %4s %9s %4s | %5s %4s %4s %9s %4s" % ("kernel (one call, in -> out)", "VALU", "pk", "mov", "div/rcp*", "f64", "VALU", "pk", "mov", "div/rcp*", "f64"))
    for k in slp:
        a, b = row(slp[k]), row(noslp[k])
        print("%-28s | %5d %4d %4d %9d %4d | %5d %4d %4d %9d %4d" % ((k[2:],) + a + b))
    print("(* v_div_scale + v_div_fmas + v_div_fixup + v_rcp: the non-packable spine of the IEEE divisions; VALU includes ~%d address / "
          "conversion instructions of the wrapper: compare rows, not absolute values)" % 10)
    return 0


if __name__ == "__main__":
    sys.exit(main())
