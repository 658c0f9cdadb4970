mkdir -p gpurun_out/r3e
timeout 600 python -m pytest tests/test_gpu_harness.py tests/test_gpu_round3.py -m gpu -q -x --timeout 600 2>&1 | tail -3
python - <<'PY'
import importlib, os, sys, torch, numpy as np
os.environ["GRNET_CONV_REPS"]="50"
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.build_synthetic_model(max_frames=2, with_gru=False)
for c, hw, hints in ((128, 14, (2024, 2028, 2034)), (256, 7, (2024, 2028, 2034)), (256, 14, (2024, 2028))):
    x = torch.randn(16, c, hw, hw, device="cuda"); r = torch.randn(16, c, hw, hw, device="cuda")
    w = (np.random.randn(c, c, 3, 3) * 0.05).astype(np.float32); b = np.zeros(c, np.float32)
    ref = None
    for hint in hints:
        y = m.op_conv2d(x, w, b, relu=True, add=r, tile_hint=hint)
        if ref is None: ref = y
        print(c, hw, hint, "max diff vs ks4:", float((y - ref).abs().max()), file=sys.stderr)
PY
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 > gpurun_out/r3e/bench_$tag.json
  python -c "
import json;d=json.loads(open('gpurun_out/r3e/bench_$tag.json').read());print('$tag:',d['value'],d['ms_per_step'])"; }
run base A=1
run ks8 GRNET_WINO4S_KS=8
run ks14 GRNET_WINO4S_KS=14
