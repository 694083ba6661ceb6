#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/occupancy per kernel of libkpeg_hip (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "libkpeg_amd", "csrc", "kpeg_hip.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-slp-vectorize",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_kpeg_usage.so", src]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        cur = re.sub(r"^_ZN8kpeg_dev\d+", "", cur)
        cur = re.sub(r"E(NS_|PK|P).*$", "", cur)
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
print("%-24s %6s %6s %6s %8s %8s %6s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
for k, r in rows.items():
    print("%-24s %6s %6s %6s %8s %8s %6s" % (k, r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"),
                                     r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
