export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_bf16_roll.py tests/test_gpu_bf16.py -m gpu -q -x --timeout 900 -k "roll or production" 2>&1 | tail -1; done
python - <<'PY'
# stress: the Bottleneck / stem launches under memory contention from a copy kernel on another stream, bits compared with an undisturbed run
import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
n = 256
m = pkg.build_synthetic_model(max_frames=n, with_gru=False, dtype="bf16")
frames = torch.from_numpy(np.tile(pkg.synth.make_frames(8), (n // 8, 1, 1, 1))).cuda()
ref = m(frames)[-1]
torch.cuda.synchronize()
ref = {k: v.clone() for k, v in ref.items()}
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); big2 = torch.empty_like(big)
side = torch.cuda.Stream()
bad = 0
for it in range(40):
    with torch.cuda.stream(side):
        for _ in range(6): big2.copy_(big)
    out = m(frames)[-1]
    torch.cuda.synchronize()
    for k in ("theta", "verts", "kp_3d"):
        if not torch.equal(out[k], ref[k]): bad += 1
print("stress under HBM contention: mismatching outputs in 40 forwards:", bad)
m.close()
PY
