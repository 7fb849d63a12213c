#!/usr/bin/env python3
"""One line per kernel of hipcc's -Rpass-analysis=kernel-resource-usage (development aid)."""
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "raytracer_amd/csrc/rt_api.hip"
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)\s*\[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{name:45s} vgpr {r.get('VGPRs','?'):>4s} sgpr {r.get('TotalSGPRs','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} "
          f"occ {r.get('Occupancy [waves/SIMD]','?'):>2s} sspill {r.get('SGPRs Spill','?'):>4s} vspill {r.get('VGPRs Spill','?'):>4s} lds {r.get('LDS Size [bytes/block]','?')}")
