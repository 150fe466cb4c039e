#!/usr/bin/env python3
"""Table of VGPRs / SGPRs / LDS / scratch / occupancy per kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
`python scripts/kernel_resources.py [file.hip] [name-filter]`"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else "draw_heatmap.hip"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cs = os.path.join(ROOT, "accv-lab_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{cs}",
       "-fno-gpu-rdc", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(cs, src), "-o", "/dev/null"] + \
      (["-DACCV_TUNE_BUILD"] if os.environ.get("TUNE") else []) + os.environ.get("EXTRA_DEFS", "").split()
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: (.+?): (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = cur.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'LDS':>7s} {'scratch':>7s} {'waves/SIMD':>10s}")
for k, r in rows.items():
    if flt in k:
        print(f"{k[:58]:58s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} "
              f"{r.get('LDS Size [bytes/block]','?'):>7s} {r.get('ScratchSize [bytes/lane]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>10s}")
