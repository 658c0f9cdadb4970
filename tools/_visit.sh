export TMPDIR=/tmp
cat > /tmp/fw.py <<'PY'
import importlib, sys, os, time, numpy as np, torch, hashlib
sys.path.insert(0, '.')
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
n = 256
m = pkg.build_synthetic_model(max_frames=n, with_gru=False, dtype="bf16")
frames = torch.from_numpy(np.tile(pkg.synth.make_frames(8), (n // 8, 1, 1, 1))).cuda()
out = m(frames)[-1]; torch.cuda.synchronize()
h = hashlib.sha1(out["theta"].cpu().numpy().tobytes() + out["verts"].cpu().numpy().tobytes()).hexdigest()[:16]
print("bf16 256 frames pipe", os.environ.get("GRNET_BF16_PIPE"), "sha", h)
m.close()
PY
export GRNET_LIB_PATH=$PWD/video-based-gait-analysis-for-dementia_amd/libgrnet_hip_abl.so
for pv in 1 2 3; do GRNET_BF16_PIPE=$pv python3 /tmp/fw.py 2>/dev/null; GRNET_BF16_PIPE=$pv python3 tools/chain_micro.py 20 2>&1 | grep "chain<32"; done
