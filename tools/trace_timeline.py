"""Concurrency profile and long kernels of ONE step from a rocprofv3 kernel trace (gpurun_out/prof/bench_kernel_trace.csv):
how long 0, 1, 2, ... kernels were running at once, and every kernel longer than a threshold with its start offset.
usage: python tools/trace_timeline.py [trace.csv] [min_us]"""
import csv, glob, collections, sys
f = sys.argv[1] if len(sys.argv) > 1 else glob.glob("gpurun_out/prof/*kernel_trace.csv")[0]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
def short(n):
    return n.replace("void ", "").replace("grk::", "").replace("(anonymous namespace)::", "").split("(")[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in csv.DictReader(open(f)))
marks = [e for e in ev if "smpl_chain" in e[2]]                      # once per step, near its end
t0, t1 = marks[-2][1], marks[-1][1]
step = [e for e in ev if e[0] >= t0 and e[1] <= t1 + 1]
print(f"{len(marks)} steps in the trace; last step: {len(step)} kernels, {(t1 - t0) / 1e3:.0f} us")
pts = sorted([(s, 1) for s, e, n, q in step] + [(e, -1) for s, e, n, q in step])
cur, last, hist = 0, t0, collections.Counter()
for t, d in pts:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
for k in sorted(hist): print(f"  {k} kernels running: {hist[k] / 1e3:7.1f} us  {100 * hist[k] / tot:5.1f} %")
print("per queue (= lane of the plan): kernels, busy us, first start, last end")
for q in sorted({e[3] for e in step}):
    k = [e for e in step if e[3] == q]
    print(f"  queue {q}: {len(k):4d} kernels, busy {sum(e[1] - e[0] for e in k) / 1e3:7.1f} us, +{(k[0][0] - t0) / 1e3:6.0f} .. +{(max(e[1] for e in k) - t0) / 1e3:6.0f} us")
print("by kernel: launches, mean us, total us")
agg = collections.defaultdict(list)
for s, e, n, q in step: agg[n].append(e - s)
for n, d in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  {len(d):4d}  {sum(d) / len(d) / 1e3:7.1f}  {sum(d) / 1e3:8.1f}  {n[:80]}")
print(f"kernels longer than {min_us:.0f} us (start offset, duration, queue):")
for s, e, n, q in step:
    if e - s > min_us * 1e3: print(f"  +{(s - t0) / 1e3:6.0f} us  {(e - s) / 1e3:6.0f} us  q{q}  {n[:90]}")
