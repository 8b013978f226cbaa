#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of libvdet_hip.so's kernels (hipcc -Rpass-analysis=kernel-resource-usage).
    python devtools/kernel_resources.py [substring ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "vdetlib_amd", "csrc", "vdet_capi.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                      "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_kr.so", src] + [],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
pats = sys.argv[1:]
print("%-60s %5s %5s %6s %6s %4s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "spillV", "occ", "LDS"))
for k, r in rows.items():
    if pats and not any(p in k for p in pats):
        continue
    print("%-60s %5d %5d %6d %6d %4d %7d" % (k[:60], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1),
                                             r.get("VGPRs Spill", -1), r.get("Occupancy [waves/SIMD]", -1),
                                             r.get("LDS Size [bytes/block]", -1)))
