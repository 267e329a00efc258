#!/usr/bin/env python3
"""VGPRs / spills / scratch / LDS / occupancy of every kernel in csrc/rt_kernels.hip, from hipcc's resource remarks.
  python3 tools/kernel_resources.py [filter] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cuda-raytracer_amd")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
       "-c", os.path.join(PKG, "csrc", "rt_kernels.hip"), "-o", "/tmp/rt_kernels_res.o"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z ]+?)(?: \[[^\]]*\])?: (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, r in rows.items():
    if flt in name:
        print("%-44s VGPR %-4s AGPR %-3s spillV %-3s spillS %-3s scratch %-4s LDS %-6s occ %s" % (
            name[:44], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize"), r.get("LDS Size"), r.get("Occupancy")))
