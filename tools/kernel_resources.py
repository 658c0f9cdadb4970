#!/usr/bin/env python3
"""Register / LDS / scratch table of every gfx950 kernel in libgrnet_hip.so, from the code-object metadata hipcc emits
(-save-temps into a scratch directory; nothing is written into the source tree but profiles/rNN_kernel_resources.md)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "video-based-gait-analysis-for-dementia_amd", "csrc")
rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
rows = []
with tempfile.TemporaryDirectory() as tmp:
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", CSRC, "-c", os.path.join(CSRC, src),
                        "-save-temps", "-o", "x.o"], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        s = open(os.path.join(tmp, src.replace(".hip", "") + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
        for b in s.split("  - .agpr_count:")[1:]:
            g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", b).group(1))
            name = re.search(r"\.name:\s+(\S+)", b).group(1)
            rows.append((src, name, g("vgpr_count"), int(b.split("\n")[0]), g("sgpr_count"), g("group_segment_fixed_size"),
                         g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
names = subprocess.run(["c++filt"] + [r[1] for r in rows], capture_output=True, text=True).stdout.split("\n")
out = [f"# Kernel resources ({rnd}): hipcc -O3 --offload-arch=gfx950, code-object metadata", "",
       "`vgpr` is the total per lane (architectural + accumulation registers; a SIMD holds 512), waves/SIMD = floor(512 / vgpr rounded up to 8), "
       "capped at 8; static LDS only (the convolution kernels take their tiles as dynamic LDS, sized per launch).", "",
       "| file | kernel | vgpr (of which agpr) | waves/SIMD | sgpr | static LDS B | scratch B | spilled vgpr / sgpr |", "|---|---|---|---|---|---|---|---|"]
for r, d in zip(rows, names):
    d = re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "").replace("grk::", "").replace("void ", ""))
    v = max(r[2], 1)
    occ = min(8, 512 // ((v + 7) // 8 * 8))
    out.append(f"| {r[0]} | `{d}` | {r[2]} ({r[3]}) | {occ} | {r[4]} | {r[5]} | {r[6]} | {r[7]} / {r[8]} |")
p = os.path.join(ROOT, "profiles", f"{rnd}_kernel_resources.md")
open(p, "w").write("\n".join(out) + "\n")
print("wrote", p, len(rows), "kernels;", sum(1 for r in rows if r[6]), "with scratch")
