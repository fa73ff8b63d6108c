#!/bin/bash
# usage: tools/isa_mix.sh <asm.s> <kernel-name-substring>  — static instruction mix of one kernel
f=$1; k=$2
awk -v k="$k" '$0 ~ "^_ZN.*"k".*:" {p=1} p && /s_endpgm/ {p=0} p' "$f" > /tmp/_kern.s
echo "total lines: $(grep -cE '^\s+[a-z]' /tmp/_kern.s)  valu: $(grep -cE '^\s+v_' /tmp/_kern.s)  salu: $(grep -cE '^\s+s_' /tmp/_kern.s) vmem: $(grep -cE '^\s+(global|buffer|flat)_' /tmp/_kern.s)"
grep -E '^\s+[a-z]' /tmp/_kern.s | awk '{print $1}' | sed 's/_e32//;s/_e64//' | sort | uniq -c | sort -rn | head -${3:-30}
