#!/bin/bash
# Register / scratch / occupancy table of every kernel of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage):
#   tools/kernel_resources.sh inconsistencymasks_amd/csrc/imk_gemm.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /dev/null 2>&1 | python3 -c '
import re, subprocess, sys
name, vals = None, {}
def flush():
    if not name: return
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    d = d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print("%-60s VGPR %3s SGPR %3s scratch %4s B occ %s" % (d[:60], vals.get("VGPRs"), vals.get("TotalSGPRs"),
          vals.get("ScratchSize [bytes/lane]"), vals.get("Occupancy [waves/SIMD]")))
for line in sys.stdin:
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        flush(); name = m.group(1); vals = {}
        continue
    m = re.search(r"remark:\s+([\w /\[\]]+): (\w+) \[-Rpass", line)
    if m and name: vals[m.group(1).strip()] = m.group(2)
flush()
'
