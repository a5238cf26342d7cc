#!/usr/bin/env python3
"""Prints VGPR/SGPR/scratch/LDS/occupancy per kernel of a source under csrc/ (default kernels.hip; hipcc remarks)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "seqkit_amd", "csrc", sys.argv[1] if len(sys.argv) > 1 else "kernels.hip")
p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c",
                    "-I", os.path.join(ROOT, "include"), "-I", os.path.dirname(src),
                    "-Rpass-analysis=kernel-resource-usage", src, "-o", "/tmp/_kr.o"],
                   stderr=subprocess.PIPE, text=True)
cur = {}
rows = []
for line in p.stderr.splitlines():
    m = re.search(r"remark: (.*?): (.*?)( \[-R.*)?$", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    else:
        cur[k] = v
print("%-44s %5s %5s %7s %6s %4s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS", "occ"))
for r in rows:
    print("%-44s %5s %5s %7s %6s %4s" % (r["name"][-44:], r.get("VGPRs"), r.get("TotalSGPRs"),
          r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
