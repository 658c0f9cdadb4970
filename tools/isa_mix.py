"""Instruction mix of the MFMA loops of a HIP source: hipcc -S for gfx950, then per basic block with >= 8 MFMAs the counts of
MFMA / other vector-ALU / LDS / vector-memory / scalar instructions.  On gfx950 an fp32 MFMA shares the SIMD's FP32 lanes with the
vector ALU (tools/micro/mfma_interleave.hip): every vector-ALU instruction in such a loop is matrix-pipe time lost, so the ratio is the
figure of merit.  Usage: python tools/isa_mix.py <file.hip> [substring of the demangled kernel name ...]"""
import re, subprocess, sys, os
from collections import Counter

def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op in ("s_waitcnt", "s_nop", "s_barrier"): return op
    if op.startswith("s_"): return "salu"
    return op

def main():
    src, pats = sys.argv[1], sys.argv[2:]
    csrc = os.path.dirname(os.path.abspath(src))
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", "-", os.path.abspath(src)],
                         capture_output=True, text=True, cwd=csrc).stdout
    for f in re.split(r"\n(?=_Z\S+:\s)", asm):
        m = re.match(r"(_Z\S+):", f)
        if not m: continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if pats and not any(p in name for p in pats): continue
        body = f[:f.index(".Lfunc_end")] if ".Lfunc_end" in f else f
        blocks, cur, lab = [], [], "entry"
        for l in body.split("\n"):
            t = l.strip()
            if re.match(r"^\.LBB\d+_\d+:", t):
                blocks.append((lab, cur)); cur, lab = [], t.split(":")[0]
            elif t and not t.startswith((";", ".")):
                cur.append(t.split()[0])
        blocks.append((lab, cur))
        shown = False
        for lab, b in blocks:
            c = Counter(classify(x) for x in b)
            if c["mfma"] >= 8:
                if not shown: print(name); shown = True
                print(f"   {lab:10s} {len(b):5d} instr: " + "  ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])) +
                      f"   | valu/mfma {c['valu'] / c['mfma']:.2f}  lds/mfma {c['lds'] / c['mfma']:.2f}")

if __name__ == "__main__":
    main()
