export TMPDIR=/tmp
mkdir -p gpurun_out
cat > /tmp/fw.py <<'PY'
import importlib, sys, os, time, numpy as np, torch, hashlib
sys.path.insert(0, '.')
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
n = 256
m = pkg.build_synthetic_model(max_frames=n, with_gru=False, dtype="bf16")
frames = torch.from_numpy(np.tile(pkg.synth.make_frames(8), (n // 8, 1, 1, 1))).cuda()
out = m(frames)[-1]; torch.cuda.synchronize()
h = hashlib.sha1(out["theta"].cpu().numpy().tobytes() + out["verts"].cpu().numpy().tobytes()).hexdigest()[:16]
print("bf16 256 frames sha", h, "(before this change: 40b3acc5d935a963)")
m.close()
PY
python3 /tmp/fw.py
timeout 1500 python -m pytest tests/test_gpu_conv_bf16.py -m gpu -q --timeout 900 -k "fuse" > gpurun_out/pytest_c.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_c.log | tail -2
GRNET_MULTI_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fs -o b -- python3 bench.py --dtype bf16 --frames 256 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-table --no-graph --tune-level 0 > /dev/null 2>&1
f=$(find gpurun_out/fs -name "b_kernel_stats.csv" | head -1)
grep "fuse_sum\|bilinear" "$f" | cut -c1-200
rm -rf gpurun_out/fs
for i in 1 2; do python bench.py --dtype bf16 --frames 256 --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-kernel-table 2>/dev/null | tail -1 > gpurun_out/b.json; python - <<PY
import json; j=json.loads(open("gpurun_out/b.json").read()); print("product: bf16 256:", j["value"], j["ms_per_step"], j["roofline"]["frac"])
PY
done
