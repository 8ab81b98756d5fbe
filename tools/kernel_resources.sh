#!/bin/bash
# Compact per-kernel resource table (VGPR / AGPR / spills / scratch / LDS / occupancy) from hipcc's remarks.
#   tools/kernel_resources.sh [file.hip ...] [-- extra hipcc flags]
cd "$(dirname "$0")/.."
files=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; extra=("$@"); break; fi; files+=("$1"); shift; done
[ ${#files[@]} -eq 0 ] && files=(render_sdfnet render_sampler render_colour)
for f in "${files[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage \
     "${extra[@]}" -c nicer_slam_amd/csrc/${f%.hip}.hip -o /tmp/kr_$$.o 2>&1 | python3 -c '
import sys,re
cur={}
def flush():
    if cur: print("%-46s vgpr %3s agpr %3s spill %3s scratch %4s lds %6s occ %s" % (cur.get("name","?")[:46],cur.get("VGPRs"),cur.get("AGPRs"),cur.get("VGPRs Spill"),cur.get("ScratchSize [bytes/lane]"),cur.get("LDS Size [bytes/block]"),cur.get("Occupancy [waves/SIMD]")))
for line in sys.stdin:
    m=re.search(r"remark:\s+(Function Name|[A-Za-z \[\]/]+):\s+(\S+)",line)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2)
    if k=="Function Name":
        flush(); cur.clear()
        import subprocess
        cur["name"]=subprocess.run(["c++filt",v],capture_output=True,text=True).stdout.strip().replace("nsa::","").split("(")[0].replace("void ","")
    else: cur[k]=v
flush()'
done
rm -f /tmp/kr_$$.o
