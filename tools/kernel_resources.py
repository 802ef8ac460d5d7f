#!/usr/bin/env python3
"""Register / spill / LDS table of the persistent kernel's variants, from hipcc's resource remarks.
Usage: tools/kernel_resources.py [waves per SIMD (default: the Makefile's)] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
waves = sys.argv[1] if len(sys.argv) > 1 else "5"
cmd = ["/opt/rocm/bin/hipcc", f"-DRTG_PHASED_WAVES={waves}", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
       "-ffp-contract=off", "-fno-fast-math", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage", "-c",
       "--cuda-device-only", "-o", "/dev/null", "rtg_phased.hip"] + sys.argv[2:]
text = subprocess.run(cmd, cwd=os.path.join(root, "raytracer_amd", "csrc"), capture_output=True, text=True).stderr
rows, cur = [], None
for line in text.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
print(f"waves per SIMD requested: {waves}")
print(f"{'source':22s} {'COUNT PATCH PRE KEEP STATS':27s} {'VGPR':>5s} {'spill':>5s} {'scratch':>7s} {'LDS(static)':>11s} {'occ':>3s}")
for r in rows:
    m = re.match(r"_ZN3rtg14k_trace_phasedINS_\d+(\w+?)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", r["name"])
    if not m:
        continue
    flags = "     ".join(m.group(i) for i in range(2, 7))
    print(f"{m.group(1):22s} {flags:27s} {r.get('vgpr', 0):5d} {r.get('spill', 0):5d} {r.get('scratch', 0):7d} {r.get('lds', 0):11d} {r.get('occ', 0):3d}")
