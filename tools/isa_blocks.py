#!/usr/bin/env python3
"""Static per-basic-block view of one kernel's ISA (asm built with -gline-tables-only): VALU count, transcendental
count, branch targets and the kernels.hip line range each block covers -- to size the loop bodies.
usage: tools/isa_blocks.py <asm.s> <kernel-substring>"""
import re, sys, collections
asm, key = sys.argv[1], sys.argv[2]
files, on = {}, False
blocks, cur = [], None
loc = None
for ln in open(asm, errors="replace"):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]; continue
    if re.match(r'^_Z.*' + key + r'.*:', ln):
        on = True; cur = dict(name="entry", valu=0, trans=0, salu=0, vmem=0, lds=0, lines=collections.Counter(), br=[]); blocks.append(cur); continue
    if not on: continue
    if "s_endpgm" in ln: on = False; continue
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m:
        cur = dict(name=m.group(1), valu=0, trans=0, salu=0, vmem=0, lds=0, lines=collections.Counter(), br=[]); blocks.append(cur); continue
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', ln)
    if m:
        loc = (files.get(int(m.group(1)), "?"), int(m.group(2))); continue
    m = re.match(r'\s+([a-z_0-9]+)\s*(.*)', ln)
    if not m or ln.lstrip().startswith((".", ";")): continue
    op = m.group(1)
    if op.startswith("v_"):
        cur["valu"] += 1
        if re.match(r'v_(rcp|sqrt|rsq|exp|log|sin|cos)_', op): cur["trans"] += 1
        cur["lines"][loc] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
        if op.startswith("s_cbranch") or op == "s_branch": cur["br"].append(op[2:] + "->" + m.group(2).strip())
    elif op.startswith(("global_", "buffer_", "flat_")): cur["vmem"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
tot = sum(b["valu"] for b in blocks)
print("blocks %d, VALU %d" % (len(blocks), tot))
for b in blocks:
    if b["valu"] < 8 and not b["br"]: continue
    kl = sorted(l for (f, l), c in b["lines"].items() if f == "kernels.hip" and l)
    top = ", ".join("%s:%d x%d" % (f[:12], l, c) for (f, l), c in b["lines"].most_common(3))
    print("%-10s valu %4d trans %3d salu %3d vmem %2d lds %2d  hip[%s]  %s  | %s" % (b["name"], b["valu"], b["trans"], b["salu"], b["vmem"], b["lds"],
          ("%d-%d" % (kl[0], kl[-1])) if kl else "", " ".join(b["br"]), top))
