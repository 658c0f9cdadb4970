export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_temporal.py -m gpu -q -x --timeout 900 2>&1 | tail -3
cat > /tmp/ov.py <<'PY'
import importlib, sys, os, time, numpy as np, torch, hashlib
sys.path.insert(0, '.')
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.build_synthetic_model(max_frames=256, use_gait_feat=True)
for (b, n) in ((1, 10000), (2, 3000)):
    x, cp = pkg.synth.make_featcorr_inputs(b, n)
    bb = np.zeros((b, n, 4), np.float32); bb[..., 2:] = 224.0
    args = (torch.from_numpy(x).reshape(b * n, 128, 24).cuda(), torch.zeros(b * n, 64, 24).cuda(), torch.from_numpy(cp).reshape(b * n, 3).cuda(),
            torch.from_numpy(bb).cuda(), torch.zeros(b, n, 2).cuda(), b, n)
    r = m.gait_correct(*args); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        t = time.perf_counter(); r = m.gait_correct(*args); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    h = hashlib.sha1(r["point_local_feat"].cpu().numpy().tobytes() + r["theta"].cpu().numpy().tobytes()).hexdigest()[:16]
    print("gait_correct mfma32", os.environ.get("GRNET_GEMM_MFMA32"), (b, n), "ms", [round(t, 2) for t in ts], "sha", h)
m.close()
PY
export GRNET_LIB_PATH=$PWD/video-based-gait-analysis-for-dementia_amd/libgrnet_hip_abl.so
for mf in 1 0 1; do GRNET_GEMM_MFMA32=$mf GRNET_GEMM_BIG_ROWS=1024 python3 /tmp/ov.py; done
GRNET_GEMM_BIG_ROWS=1024 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ovtrace -- python3 /tmp/ov.py > /dev/null 2>&1
f=$(find gpurun_out/ovtrace -name "*kernel_stats.csv" | head -1)
head -6 "$f" | cut -c1-230
rm -rf gpurun_out/ovtrace
