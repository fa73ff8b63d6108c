#!/usr/bin/env python3
"""Static attribution of one kernel's ISA to source lines (needs an asm listing built with -gline-tables-only).
usage: tools/isa_lines.py <asm.s> <kernel-substring> [top]"""
import re, sys, collections
asm, key = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
files, cur, on = {}, None, False
by_line, by_file = collections.Counter(), collections.Counter()
for ln in open(asm, errors="replace"):
    m = re.match(r'\s+\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if re.match(r'^_Z.*' + key + r'.*:', ln):
        on = True; continue
    if on and "s_endpgm" in ln:
        on = False
    if not on:
        continue
    m = re.match(r'\s+\.loc\s+(\d+)\s+(\d+)', ln)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
    if re.match(r'\s+v_', ln):
        by_line[cur] += 1; by_file[cur[0] if cur else None] += 1
print("VALU by file:", dict(by_file))
for (k, v) in by_line.most_common(top):
    print("%5d  %s:%s" % (v, k[0], k[1]))
