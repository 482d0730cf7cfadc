#!/usr/bin/env python3
"""Registers / scratch / LDS of every trace_image_kernel instantiation as the compiler reports them: python tools/kres.py [extra hipcc flags]"""
import re, subprocess, sys
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
       *sys.argv[1:], "-c", "all_is_cubes_amd/csrc/aic_trace.hip", "-o", "/dev/null"]
r = subprocess.run(cmd, capture_output=True, text=True)
txt = subprocess.run(["c++filt"], input=r.stderr, capture_output=True, text=True).stdout
if "error" in txt: print(txt[:3000])
for b in txt.split("Function Name: ")[1:]:
    name = b.split("\n")[0]
    if "trace_image_kernel" not in name: continue
    g = lambda k: re.search(k + r": (\S+)", b).group(1)
    print(name[name.index("<"):name.index(">") + 1], "VGPR", g("VGPRs"), "AGPR", g("AGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"), "occ", g(r"Occupancy \[waves/SIMD\]"),
          "sgpr-spill", g("SGPRs Spill"), "vgpr-spill", g("VGPRs Spill"), "LDS", g(r"LDS Size \[bytes/block\]"))
