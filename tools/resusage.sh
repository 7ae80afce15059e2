#!/bin/bash
# kernel resource usage of the device code (VGPRs, SGPRs, spills, scratch, LDS, occupancy), one line per kernel
# usage: tools/resusage.sh [pattern]
cd "$(dirname "$0")/../sigmod-2018_amd"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I../include -Icsrc -S --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage ${RHJ_DEFS} csrc/rhj_device.hip -o /tmp/rhj_device.s 2>&1 |
python3 -c '
import sys, re
cur = None; rows = {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":",1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":",1); rows[cur][k.strip()] = v.strip()
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for k, r in rows.items():
    if pat in k:
        print("%-70s VGPR %-4s AGPR %-3s SGPR %-4s spillS %-4s spillV %-4s scratch %-5s LDS %-6s occ %s" % (k[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
' "$1"
