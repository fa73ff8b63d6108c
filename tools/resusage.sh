#!/bin/bash
# usage: resusage.sh fast|strict [extra flags]  -> table of kernel resource usage
B=$1; shift
if [ "$B" = fast ]; then F="-ffp-contract=fast -DSLAM_FAST_MATH=1 -DSLAM_KNS=slam_fast -DSLAM_TABLE=fast"; else F="-ffp-contract=off -DSLAM_KNS=slam_strict -DSLAM_TABLE=strict"; fi
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -I/root/repo/slam_amd/csrc -I/root/repo/include $F "$@" -c /root/repo/slam_amd/csrc/kernels.hip -o /tmp/k_$B.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows=[]
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    for k in ('TotalSGPRs','VGPRs','AGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]','SGPRs Spill','VGPRs Spill','LDS Size \[bytes/block\]'):
        m=re.search(k+r': (\d+)',ln)
        if m and cur is not None: cur[k.split(' ')[0]+('Spill' if 'Spill' in k else '')]=m.group(1)
import subprocess
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.split('(')[0]
    print('%-60s sgpr %s vgpr %s scratch %s occ %s sspill %s vspill %s lds %s'%(n[-60:],r.get('TotalSGPRs'),r.get('VGPRs'),r.get('ScratchSize'),r.get('Occupancy'),r.get('SGPRsSpill'),r.get('VGPRsSpill'),r.get('LDS')))
"
