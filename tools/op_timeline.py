"""Un-traced timeline of one forward (grnet_op_timeline: HIP timing events around every op on the lane streams, no profiler).
Prints how long 0, 1, 2, ... ops were running at once, the time per section of the network, per-lane busy time, and (with --dump
A B) every op that starts between A and B us.
usage: python tools/op_timeline.py [--frames 16] [--dtype f32] [--dump A B] [--out file]"""
import argparse, collections, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--dump", type=float, nargs=2, default=None)
ap.add_argument("--out", default=None)
args = ap.parse_args()
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.build_synthetic_model(max_frames=args.frames, with_gru=False, dtype=args.dtype)
x = torch.from_numpy(pkg.synth.make_frames(args.frames)).cuda()
m(x.unsqueeze(0)); torch.cuda.synchronize()
m.tune(args.frames)
rows = m.op_timeline(x)
out = open(args.out, "w") if args.out else sys.stdout


def section(label):
    for key, name in (("conv1.weight", None), ("stage2", "stage2"), ("stage3", "stage3"), ("stage4", "stage4"), ("transition", "transitions"),
                      ("layer1", "stem+layer1"), ("upsample_stage", "upsample heads"), ("bilinear", "upsample heads"), ("head.", "PARE head"),
                      ("attn_pool", "tail"), ("head_tail", "tail"), ("smpl", "tail")):
        if key in label and name:
            return name
    return "stem+layer1" if "backbone.conv" in label else "other"


t_end = max(r[3] for r in rows)
print(f"# {len(rows)} ops, step {t_end:.0f} us (events included), {args.frames} frames", file=out)
pts = sorted([(r[2], 1) for r in rows] + [(r[3], -1) for r in rows])
cur, last, hist = 0, 0.0, collections.Counter()
for t, d in pts:
    hist[cur] += t - last; last = t; cur += d
for k in sorted(hist):
    print(f"  {k} ops running: {hist[k]:8.1f} us  {100 * hist[k] / t_end:5.1f} %", file=out)
sec = collections.OrderedDict()
for i, lane, a, b, label in rows:
    s = sec.setdefault(section(label), [1e30, 0.0, 0.0, 0])
    s[0] = min(s[0], a); s[1] = max(s[1], b); s[2] += b - a; s[3] += 1
print("section: ops, first start .. last end (span), sum of op durations", file=out)
for name, (a, b, tot, cnt) in sec.items():
    print(f"  {name:16s} {cnt:4d} ops  +{a:7.0f} .. +{b:7.0f}  ({b - a:6.0f} us)  sum {tot:7.0f} us", file=out)
for lane in sorted({r[1] for r in rows}):
    k = [r for r in rows if r[1] == lane]
    print(f"  lane {lane}: {len(k):4d} ops, busy {sum(r[3] - r[2] for r in k):7.0f} us", file=out)
kinds = collections.defaultdict(list)
for i, lane, a, b, label in rows:
    key = " ".join(label.split(" ")[:5]) if label.startswith("conv") else label.split(" ")[0]
    kinds[key].append(b - a)
print("by op shape: count, mean us, total us", file=out)
for k, d in sorted(kinds.items(), key=lambda kv: -sum(kv[1]))[:24]:
    print(f"  {len(d):4d} {sum(d) / len(d):7.1f} {sum(d):8.1f}  {k}", file=out)
if args.dump:
    for i, lane, a, b, label in rows:
        if args.dump[0] <= a <= args.dump[1]:
            print(f"+{a:7.1f} {b - a:6.1f} L{lane} {'    ' * lane}{label[:90]}", file=out)
m.close()
