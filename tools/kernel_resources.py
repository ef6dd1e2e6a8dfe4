#!/usr/bin/env python3
"""Print VGPR / SGPR / LDS / scratch / occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
srcs = sys.argv[1:] or [os.path.join(REPO, "dynamicfusion_amd", "csrc", f) for f in
                        ("dfusion_volume.hip", "dfusion_warp.hip", "dfusion_raycast.hip")]
for src in srcs:
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
           "-I", os.path.join(REPO, "include"), "-I", os.path.join(REPO, "dynamicfusion_amd", "csrc"),
           "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: (?:.*?:\d+:\d+: )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        else:
            cur[k.split(" ")[0]] = v
            if k.startswith("LDS"):
                print("%-55s vgpr %-4s sgpr %-4s scratch %-4s lds %-6s occ %s" % (
                    cur["name"][:55], cur.get("VGPRs"), cur.get("TotalSGPRs"), cur.get("ScratchSize"), cur.get("LDS"), cur.get("Occupancy")))
