"""Which kernels slow which: every given convolution runs as a chain of launches on its OWN stream, all streams at once; prints us per
launch of every stream (the time the launches of an HR module step take in company).
usage: python tools/conv_mix.py "c,hw,hint c,hw,hint ..." ["..." more mixes]"""
import importlib, os, sys, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
os.environ["GRNET_CONV_REPS"] = "400"
N = 16
import io, contextlib, re
for mix in sys.argv[1:]:
    specs = [tuple(int(v) for v in t.split(",")) for t in mix.split()]
    models = [pkg.GRNet(max_frames=1) for _ in specs]
    streams = [torch.cuda.Stream() for _ in specs]
    data = []
    for c, hw, hint in specs:
        data.append((torch.randn(N, c, hw, hw, device="cuda"), torch.randn(N, c, hw, hw, device="cuda"), (np.random.randn(c, c, 3, 3) * 0.05).astype(np.float32), np.zeros(c, np.float32)))
    torch.cuda.synchronize()
    def run(i):
        x, r, w, b = data[i]
        with torch.cuda.stream(streams[i]):
            models[i].op_conv2d(x, w, b, stride=1, relu=True, add=r, tile_hint=specs[i][2])
    print(f"=== mix: {mix}", file=sys.stderr, flush=True)
    th = [threading.Thread(target=run, args=(i,)) for i in range(len(specs))]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    for m in models: m.close()
