#!/usr/bin/env python
"""Register / LDS / spill figures of the gfx950 kernels (compiles the device side to assembly and reads the kernel
metadata):  python tools/kernel_regs.py [name filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/singa_hip_gfx950.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                       os.path.join(ROOT, "singa_amd", "csrc", "singa_hip.hip")], stderr=subprocess.DEVNULL)
s = open(out).read()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
for m in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size", s, re.S):
    body = m.group(0)
    name = re.search(r"\.name:\s+(\S+)", body).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).replace("void ", "")
    dem = dem[:dem.index("(")] if "(" in dem else dem
    if flt not in dem:
        continue
    g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", body).group(1))
    rows.append((dem, g("vgpr_count"), g("agpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("group_segment_fixed_size"),
                 g("private_segment_fixed_size")))
print(f"{'kernel':70s} vgpr agpr spill sgpr   lds scratch")
for r in sorted(rows):
    print(f"{r[0][:70]:70s} {r[1]:4d} {r[2]:4d} {r[3]:5d} {r[4]:4d} {r[5]:5d} {r[6]:7d}")
