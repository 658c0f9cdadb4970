"""The two entry points end to end on the GPU with synthetic frames / weights: flags, model loop,
output schema (.pkl of demo.py:211-222 and the joblib db of batch_generation.py:265-267)."""
import importlib
import os
import sys

import joblib
import numpy as np
import pytest
import torch

from .conftest import CALL_SIZE_NOISE, ROOT, elem_ratio, rel_err

pytestmark = pytest.mark.gpu


def _write_frames(folder, frames):
    os.makedirs(folder, exist_ok=True)
    for i, f in enumerate(frames):
        np.save(os.path.join(folder, f"{i:06d}.npy"), f)


def test_demo_entry_point(pkg, tmp_path):
    sys.path.insert(0, ROOT)
    demo = importlib.import_module("demo")
    frames = pkg.synth.make_frames(30)
    img_dir = str(tmp_path / "vid")
    _write_frames(img_dir, frames)
    bbox = np.tile(np.array([[112.0, 112.0, 224.0, 224.0]], np.float32), (30, 1))
    tracking = {1: {"bbox": bbox.copy(), "frames": np.arange(30)},
                2: {"bbox": bbox[:10].copy(), "frames": np.arange(10)}}         # < 25 frames: dropped (demo.py:101-103)
    tp = str(tmp_path / "tracking.pkl")
    joblib.dump(tracking, tp)
    args = demo.parser().parse_args(["--img_folder", img_dir, "--tracking_path", tp, "--output_folder", str(tmp_path / "out"),
                                     "--synthetic_weights", "--grnet_batch_size", "16", "--max_frames", "16"])
    out = demo.main(args)
    assert out.endswith("synthetic.pkl") and os.path.isfile(out)
    assert os.path.basename(os.path.dirname(out)).startswith("normal-")
    res = joblib.load(out)
    assert list(res.keys()) == [1]
    r = res[1]
    want = {"pred_cam": (30, 3), "orig_cam": (30, 4), "verts": (30, 6890, 3), "pose": (30, 72), "betas": (30, 10),
            "joints3d": (30, 29, 3), "joints2d": (30, 29, 2), "bboxes": (30, 4), "frame_ids": (30,)}
    assert set(r) == set(want)
    for k, shp in want.items():
        assert r[k].shape == shp, k
    # same numbers as a direct call of the model on the same frames
    m = pkg.build_synthetic_model(max_frames=30, with_gru=False)
    direct = m(torch.from_numpy(frames).cuda())[-1]
    torch.cuda.synchronize()
    assert rel_err(r["joints3d"], direct["kp_3d"][0].cpu().numpy()) < CALL_SIZE_NOISE
    assert rel_err(r["pose"], direct["theta"][0, :, 3:75].cpu().numpy()) < 2e-5
    m.close()
    # a second run must not overwrite the first (demo.py:258-266)
    out2 = demo.main(args)
    assert out2.endswith("synthetic1.pkl")
    # --joint_type (demo.py:224-229): joints re-ordered into the requested skeleton; an unknown one is reported and left as is
    args_k = demo.parser().parse_args(["--img_folder", img_dir, "--tracking_path", tp, "--output_folder", str(tmp_path / "out"),
                                       "--synthetic_weights", "--grnet_batch_size", "16", "--max_frames", "16", "--joint_type", "kinectv2"])
    rk = joblib.load(demo.main(args_k))[1]
    assert rk["joints3d"].shape == (30, 25, 3) and rk["joints2d"].shape == (30, 25, 2)
    assert np.array_equal(rk["joints3d"], pkg.pipeline.spin2_to_kinectv2(r["joints3d"]))
    args_u = demo.parser().parse_args(["--img_folder", img_dir, "--tracking_path", tp, "--output_folder", str(tmp_path / "out"),
                                       "--synthetic_weights", "--grnet_batch_size", "16", "--max_frames", "16", "--joint_type", "nonsense"])
    assert joblib.load(demo.main(args_u))[1]["joints3d"].shape == (30, 29, 3)


def test_batch_generation_entry_point(pkg, tmp_path):
    sys.path.insert(0, ROOT)
    bg = importlib.import_module("batch_generation")
    names = ["S001C001P001R001A002", "S001C001P001R001A001"]
    annos = {}
    for vi, name in enumerate(names):
        n = 5 + vi
        _write_frames(str(tmp_path / "vids" / name), pkg.synth.make_frames(n, start=100 * vi))
        annos[name] = np.tile(np.array([[112.0, 112.0, 200.0, 200.0]], np.float32), (n, 1))
    bp = str(tmp_path / "bbox.pkl")
    joblib.dump(annos, bp)
    written = bg.prepare_data(fv=bp, vid_folder=str(tmp_path / "vids"), outpath=str(tmp_path / "db.json"),
                              synthetic_weights=True, max_frames=8)
    assert [os.path.basename(w) for w in written] == ["db_0.json"]
    db = joblib.load(written[0])
    assert db["vid_name"].shape == (11,) and db["bbox"].shape == (11, 4) and db["joints3D"].shape == (11, 25, 3)
    assert list(db["vid_name"][:6]) == ["S001C001P001R001A001"] * 6            # sorted by the digits of the name
    assert db["joints3D"].dtype == np.float32 and np.isfinite(db["joints3D"]).all()
    # the reference stores the boxes AFTER Inference scaled w,h by 1.1 in place (inference.py:48, batch_generation.py:265)
    assert db["bbox"].dtype == np.float32
    assert np.array_equal(db["bbox"], np.tile(np.array([[112.0, 112.0, 200.0, 200.0]], np.float32) * np.array([1, 1, 1.1, 1.1], np.float32), (11, 1)))
    assert np.array_equal(annos[names[0]][:, 2], np.full(5, 200.0, np.float32))          # the caller's annotations are not touched
    # kinectv2 joint 0 is spin2 joint 0 (pelvis), joint 20 is the thorax (index 28)
    m = pkg.build_synthetic_model(max_frames=8, with_gru=False)
    f = torch.from_numpy(pkg.synth.make_frames(6, start=100)).cuda()
    kp = m(f)[-1]["kp_3d"][0].cpu().numpy()
    assert rel_err(db["joints3D"][:6, 20], kp[:, 28]) < 2e-5 and rel_err(db["joints3D"][:6, 1], kp[:, 6]) < 2e-5
    m.close()


def test_smooth_pose_row_f3(pkg, oracle, synth_smpl):
    """--smooth (smooth_pose.py:28-116): filtered pose -> SMPL re-evaluated with the betas of frame 0, one LBS launch."""
    pipe = pkg.pipeline
    m = pkg.build_synthetic_model(max_frames=8, with_gru=False)
    g = np.random.Generator(np.random.Philox(key=[17, 17]))
    T = 20                                                    # > max_frames: smpl_forward chunks
    pose = np.cumsum(g.standard_normal((T, 72)) * 0.03, axis=0).astype(np.float32)
    betas = (g.standard_normal((T, 10)) * 0.5).astype(np.float32)
    verts, pose_hat, j49 = pipe.smooth_pose(m, pose, betas, smpl_tables=synth_smpl)
    assert verts.shape == (T, 6890, 3) and pose_hat.shape == (T, 72) and j49.shape == (T, 49, 3)
    assert np.array_equal(pose_hat[0], pose[0])               # the filter starts at the first pose
    # CPU check of the same composition: filter -> rodrigues -> oracle LBS with betas[0]
    R = pipe.rodrigues(pose_hat.reshape(-1, 3)).reshape(T, 24, 3, 3)
    v_ref, j24 = oracle.smpl_lbs(np.repeat(betas[:1], T, 0), R, synth_smpl)
    assert rel_err(verts, v_ref) < 1e-4
    assert rel_err(j49[:, 8], j24[:, 0]) < 1e-4               # SPIN joint 8 'OP MidHip' is SMPL joint 0
    _, _, k25 = pipe.smooth_pose(m, pose, betas, kinectv2=True)
    assert k25.shape == (T, 25, 3)
    m.close()


def test_clip_runner_sequence_feeds_gru(pkg, oracle):
    """The reassembled per-frame records (what the all-gather delivers) are the GRU's input: point_local_feat
    (T,128,24) -> (1,T,3072) in the c*24+j layout of grnet.py:163, cparams from the predicted camera."""
    h = pkg.harness
    m = pkg.build_synthetic_model(max_frames=12, with_gru=True)
    frames = torch.from_numpy(pkg.synth.make_frames(12)).cuda()
    runner = h.ClipRunner(m, frames, use_graph=True, tune_level=0)
    runner.step()
    runner.step()                                             # second step replays the captured graph
    torch.cuda.synchronize()
    seq = runner.sequence()
    assert seq["theta"].shape == (12, 85) and seq["kp_3d"].shape == (12, 29, 3) and seq["point_local_feat"].shape == (12, 128, 24)
    direct = m(frames, extras=("point_local_feat",))[-1]
    assert rel_err(seq["theta"].cpu().numpy(), direct["theta"][0].cpu().numpy()) < 2e-5
    assert rel_err(seq["point_local_feat"].cpu().numpy(), direct["point_local_feat"].cpu().numpy()) < 2e-5
    x = seq["point_local_feat"].reshape(1, 12, 3072).contiguous()
    cp = seq["theta"][:, :3].reshape(1, 12, 3).contiguous()
    y, phase, _ = m.gru_forward(x, cp)
    ry, rph, _ = oracle.gru_forward(x.cpu().numpy(), cp.cpu().numpy(), pkg.synth.make_gru_state_dict())
    assert rel_err(y.cpu().numpy(), ry) < 1e-4 and rel_err(phase.cpu().numpy(), rph) < 1e-4
    m.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_multi_track_overlapped_preprocess_stream(pkg, dtype):
    """BASELINE configs[4] in miniature on one GPU: 4 person tracks x 24 frames in batches of 12, crop + normalise of the next
    batch on a side stream while the (graph-replayed) forward of the current batch runs; same numbers as the sequential loop."""
    p = pkg.pipeline
    m = pkg.build_synthetic_model(max_frames=12, with_gru=False, dtype=dtype)
    m.set_option(pkg._lib.OPT_USE_GRAPH, 1)
    rng = np.random.default_rng(11)
    tracks = []
    for t in range(4):
        batches = []
        for b in range(2):
            raw = rng.integers(0, 256, size=(12, 180, 240, 3), dtype=np.uint8)
            bb = np.stack([np.array([120 + 3 * t + i, 90 - 2 * b + i, 140 + 2 * i, 140 + 2 * i], np.float32) for i in range(12)])
            batches.append((raw, bb))
        tracks.append(batches)
    got = p.run_tracks_overlapped(m, tracks, batch_size=12)
    torch.cuda.synchronize()
    assert len(got) == 4 and got[0]["verts"].shape == (24, 6890, 3) and got[3]["joints3d"].shape == (24, 29, 3)
    for t in (0, 3):
        for b in range(2):
            raw, bb = tracks[t][b]
            x = m.crop_normalise(torch.from_numpy(raw).cuda(), torch.from_numpy(bb), scale=1.1)
            ref = m(x.unsqueeze(0))[-1]
            torch.cuda.synchronize()
            assert np.array_equal(got[t]["pose"][12 * b:12 * b + 12], ref["theta"][0, :, 3:75].cpu().numpy())
            assert np.array_equal(got[t]["verts"][12 * b:12 * b + 12], ref["verts"][0].cpu().numpy())
    m.close()


# ---- checkpoint-file round trips (round-4 review: every test built its model from synth in memory; a real MAX-GRNet checkpoint would have been the
# first thing ever to walk demo.py's --ckpt branch, batch_generation's pretrained_file branch and load_pare_dict) -----------------------------------
def _reference_format_checkpoint(pkg, path):
    """A file as the reference writes it: torch.save({'gen_state_dict': model.state_dict()}) with torch tensors, int64 num_batches_tracked, the unused
    backbone.final_layer.*, head.temperature / init_*, and the SMPL buffers under regressor.smpl.smpl.* including the ones the path never reads
    (faces_tensor, betas, global_orient, body_pose, vertex_joint_selector.extra_joints_idxs: smplx registers them)."""
    sd = {}
    for k, v in pkg.synth.make_state_dict().items():
        t = torch.as_tensor(np.asarray(v))                        # 0-d stays 0-d (head.temperature, num_batches_tracked), as in a torch state_dict
        sd[k] = t.to(torch.int64) if k.endswith("num_batches_tracked") else t.to(torch.float32)
    assert any("final_layer" in k for k in sd) and sd["head.temperature"].ndim == 0 and sd["backbone.bn1.num_batches_tracked"].dtype == torch.int64
    pre = "regressor.smpl.smpl."
    for k, v in pkg.synth.make_smpl_tables().items():
        sd[pre + k] = torch.from_numpy(np.ascontiguousarray(v)).to(torch.int64 if k == "parents" else torch.float32)
    sd[pre + "faces_tensor"] = torch.zeros(13776, 3, dtype=torch.int64)
    sd[pre + "betas"] = torch.zeros(1, 10)
    sd[pre + "global_orient"] = torch.zeros(1, 3)
    sd[pre + "body_pose"] = torch.zeros(1, 69)
    sd[pre + "vertex_joint_selector.extra_joints_idxs"] = torch.arange(21)
    torch.save({"gen_state_dict": sd, "epoch": 7, "performance": 0.0}, path)
    return sd


def test_demo_from_checkpoint_file_equals_synthetic_weights(pkg, tmp_path):
    """demo.py --ckpt (demo.py:116-122: torch.load(f)['gen_state_dict'] -> load_state_dict(strict=False)) on a reference-format file holding the
    synthetic weights: the .pkl must be bit-identical to the --synthetic_weights run, and carry the checkpoint's stem as its name (demo.py:254-267)."""
    sys.path.insert(0, ROOT)
    demo = importlib.import_module("demo")
    ck = str(tmp_path / "grnet_epoch7.pth.tar")
    _reference_format_checkpoint(pkg, ck)
    frames = pkg.synth.make_frames(26)
    img_dir = str(tmp_path / "vid")
    _write_frames(img_dir, frames)
    tp = str(tmp_path / "tracking.pkl")
    joblib.dump({3: {"bbox": np.tile(np.array([[112.0, 112.0, 224.0, 224.0]], np.float32), (26, 1)), "frames": np.arange(26)}}, tp)
    common = ["--img_folder", img_dir, "--tracking_path", tp, "--grnet_batch_size", "16", "--max_frames", "16"]
    out_ck = demo.main(demo.parser().parse_args(common + ["--output_folder", str(tmp_path / "a"), "--ckpt", ck]))
    out_sy = demo.main(demo.parser().parse_args(common + ["--output_folder", str(tmp_path / "b"), "--synthetic_weights"]))
    assert os.path.basename(out_ck).startswith("grnet_epoch7") and out_ck.endswith(".pkl")
    a, b = joblib.load(out_ck)[3], joblib.load(out_sy)[3]
    assert set(a) == set(b)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_batch_generation_from_checkpoint_file_strict(pkg, tmp_path):
    """batch_generation.py:214-218: load_state_dict(torch.load(f)['gen_state_dict'], strict=True) -- the full key set of a reference checkpoint must
    load strictly (unused and tolerated keys included), and the db must equal the synthetic-weights run bit for bit."""
    sys.path.insert(0, ROOT)
    bg = importlib.import_module("batch_generation")
    ck = str(tmp_path / "model_best.pth.tar")
    _reference_format_checkpoint(pkg, ck)
    name = "S001C001P001R001A003"
    _write_frames(str(tmp_path / "vids" / name), pkg.synth.make_frames(7, start=40))
    bp = str(tmp_path / "bbox.pkl")
    joblib.dump({name: np.tile(np.array([[112.0, 112.0, 210.0, 190.0]], np.float32), (7, 1))}, bp)
    w_ck = bg.prepare_data(fv=bp, vid_folder=str(tmp_path / "vids"), outpath=str(tmp_path / "ck.json"), pretrained_file=ck, max_frames=8)
    w_sy = bg.prepare_data(fv=bp, vid_folder=str(tmp_path / "vids"), outpath=str(tmp_path / "sy.json"), synthetic_weights=True, max_frames=8)
    a, b = joblib.load(w_ck[0]), joblib.load(w_sy[0])
    for k in ("vid_name", "bbox", "joints3D"):
        assert np.array_equal(a[k], b[k]), k
    # strict means strict: one key too many, or one missing, is refused like torch refuses it
    sd = torch.load(ck, map_location="cpu")["gen_state_dict"]
    sd["backbone.not_a_layer.weight"] = torch.zeros(3)
    torch.save({"gen_state_dict": sd}, ck)
    with pytest.raises(RuntimeError, match="unexpected"):
        bg.prepare_data(fv=bp, vid_folder=str(tmp_path / "vids"), outpath=str(tmp_path / "x.json"), pretrained_file=ck, max_frames=8)
    del sd["backbone.not_a_layer.weight"], sd["backbone.stage3.2.branches.1.3.bn2.running_var"]
    torch.save({"gen_state_dict": sd}, ck)
    with pytest.raises(RuntimeError, match="missing"):
        bg.prepare_data(fv=bp, vid_folder=str(tmp_path / "vids"), outpath=str(tmp_path / "y.json"), pretrained_file=ck, max_frames=8)


def test_load_pare_dict_and_size_mismatch(pkg, tmp_path):
    """lib/models/grnet.py:93-109: the PARE checkpoint's 'model.head.*' tensors re-keyed to 'head.*'; a file without init_pose / init_shape is refused;
    a tensor of the wrong size raises like torch's load_state_dict ('size mismatch')."""
    full = pkg.synth.make_state_dict()
    pare = {"model.head." + k[len("head."):]: torch.as_tensor(np.asarray(v)) for k, v in full.items() if k.startswith("head.")}
    pare["model.backbone.conv1.weight"] = torch.zeros(64, 3, 3, 3)          # not a head tensor: ignored
    pf = str(tmp_path / "pare_w_3dpw_checkpoint.ckpt")
    torch.save({"state_dict": pare}, pf)
    frames = torch.from_numpy(pkg.synth.make_frames(3)).cuda()
    ref_model = pkg.build_synthetic_model(max_frames=4, with_gru=False)
    ref = ref_model(frames)[-1]
    m = pkg.GRNet(max_frames=4, pretrained_pare=pf)                           # the constructor argument of the reference (grnet.py:87)
    rest = {k: v for k, v in full.items() if not k.startswith("head.")}
    res = m.load_state_dict(rest, strict=False)
    assert not res.missing_keys, res.missing_keys[:3]                          # head.* came from the PARE file
    m.load_smpl(pkg.synth.make_smpl_tables())
    got = m.finalize()(frames)[-1]
    torch.cuda.synchronize()
    for k in ("theta", "kp_3d", "verts"):
        assert torch.equal(got[k], ref[k]), k
    m.close()
    ref_model.close()
    bad = dict(pare)
    del bad["model.head.init_pose"]
    torch.save({"state_dict": bad}, pf)
    with pytest.raises(KeyError, match="VPARE"):
        pkg.GRNet(max_frames=1, pretrained_pare=pf)
    m2 = pkg.GRNet(max_frames=1)
    wrong = dict(full)
    wrong["backbone.layer1.0.conv1.weight"] = np.zeros((64, 32, 1, 1), np.float32)
    with pytest.raises(RuntimeError, match="size mismatch for backbone.layer1.0.conv1.weight"):
        m2.load_state_dict(wrong, strict=False)
    m2.close()


# ---- regrouped from the per-round files in round 6 (unchanged): BASELINE configs[3] / [4] at their per-GPU size, the crop kernel's two-warp branch, the entry
# points on PNG frames, the C ABI's exchange with one rank, the overlapped track runner
@pytest.fixture(scope="module")
def model(pkg):
    m = pkg.build_synthetic_model(max_frames=16, with_gru=True)
    yield m
    m.close()


def test_config4_per_gpu_share_1250_frames(pkg, oracle, synth_weights, synth_smpl):
    """BASELINE configs[3] at the size ONE GPU sees: its 1 250-frame share of the 10 000-frame directory, in calls of <= 128 frames
    (SURVEY 8d), then the temporal GRU over the reassembled sequence.  Checked against the oracle on a strided subset of the
    frames, by size-independent properties on all of them, and for the GRU against the oracle on the full 1 250-step sequence."""
    n, chunk = 1250, 128
    h = pkg.harness
    lo, hi = h.shard_range(10000, 8, 3)
    assert hi - lo == n
    m = pkg.build_synthetic_model(max_frames=chunk, with_gru=True)
    # frames of this rank's shard: the counter-based generator addresses frames by their global index (no 7.5 GB host array)
    theta, kp3d, plf = [], [], []
    for s in range(0, n, chunk):
        c = min(chunk, n - s)
        x = torch.from_numpy(pkg.synth.make_frames(c, start=lo + s)).cuda()
        o = m(x, extras=("point_local_feat",))[-1]
        theta.append(o["theta"][0]); kp3d.append(o["kp_3d"][0]); plf.append(o["point_local_feat"])
    theta, kp3d, plf = torch.cat(theta), torch.cat(kp3d), torch.cat(plf)
    torch.cuda.synchronize()
    assert theta.shape == (n, 85) and kp3d.shape == (n, 29, 3) and plf.shape == (n, 128, 24)
    assert torch.isfinite(theta).all() and torch.isfinite(kp3d).all()
    pick = np.arange(0, n, 139)                                           # 9 frames across all 10 calls, incl. the short last one
    sub = np.concatenate([pkg.synth.make_frames(1, start=lo + int(i)) for i in pick])
    ref = oracle.grnet_forward(sub, synth_weights, synth_smpl)
    assert rel_err(theta[pick].cpu().numpy(), np.asarray(ref["theta"]).reshape(len(pick), 85)) < 1e-3
    assert rel_err(kp3d[pick].cpu().numpy(), np.asarray(ref["kp_3d"]).reshape(len(pick), 29, 3)) < 1e-3
    # position in a call does not matter: frame 700 alone equals frame 700 inside its 128-frame call
    one = m(torch.from_numpy(pkg.synth.make_frames(1, start=lo + 700)).cuda())[-1]
    assert rel_err(one["theta"][0, 0].cpu().numpy(), theta[700].cpu().numpy()) < CALL_SIZE_NOISE
    # the temporal encoder over the whole share (on 8 GPUs: after the all-gather, over all 10 000)
    x = plf.reshape(1, n, 3072).contiguous()
    cp = theta[:, :3].reshape(1, n, 3).contiguous()
    y, phase, _ = m.gru_forward(x, cp)
    ry, rph, _ = oracle.gru_forward(x.cpu().numpy(), cp.cpu().numpy(), pkg.synth.make_gru_state_dict())
    assert rel_err(y.cpu().numpy(), ry) < 1e-3 and rel_err(phase.cpu().numpy(), rph) < 1e-3
    m.close()


def test_config5_four_tracks_of_64_frames_bf16(pkg, oracle, synth_weights, synth_smpl):
    """BASELINE configs[4] at full per-node size on one GPU: 4 person tracks x 64 frames, bf16, crop + normalise of the next
    track's frames on the side stream under the graph-replayed forward of the current one.  Equal to the sequential loop bit
    for bit; against the fp32 oracle on the same crops within the bf16 storage noise (features-level bound of test_gpu_bf16)."""
    p = pkg.pipeline
    m = pkg.build_synthetic_model(max_frames=64, with_gru=False, dtype="bf16")
    m.set_option(pkg._lib.OPT_USE_GRAPH, 1)
    rng = np.random.default_rng(23)
    tracks = []
    for t in range(4):
        raw = rng.integers(0, 256, size=(64, 240, 320, 3), dtype=np.uint8)
        bb = np.stack([np.array([150 + 2 * t + 0.5 * i, 120 - t + 0.25 * i, 180 + i, 180 + i], np.float32) for i in range(64)])
        tracks.append([(raw, bb)])
    got = p.run_tracks_overlapped(m, tracks, batch_size=64)
    again = p.run_tracks_overlapped(m, tracks, batch_size=64)             # second pass: the captured graphs are replayed
    torch.cuda.synchronize()
    assert len(got) == 4 and got[0]["verts"].shape == (64, 6890, 3) and got[3]["joints3d"].shape == (64, 29, 3)
    for t in range(4):
        for k in ("pose", "verts", "joints3d", "pred_cam"):
            assert np.array_equal(got[t][k], again[t][k]), (t, k)
    raw, bb = tracks[2][0]
    x = m.crop_normalise(torch.from_numpy(raw).cuda(), torch.from_numpy(bb), scale=1.1)
    seq = m(x.unsqueeze(0))[-1]
    torch.cuda.synchronize()
    assert np.array_equal(got[2]["pose"], seq["theta"][0, :, 3:75].cpu().numpy())
    assert np.array_equal(got[2]["verts"], seq["verts"][0].cpu().numpy())
    pick = [0, 21, 42, 63]
    ref = oracle.grnet_forward(x[pick].cpu().numpy(), synth_weights, synth_smpl)
    d = got[2]["joints3d"][pick] - np.asarray(ref["kp_3d"]).reshape(4, 29, 3)
    assert np.linalg.norm(d, axis=-1).mean() < 0.02                       # MPJPE vs fp32 in metres: bf16 storage noise
    m.close()


# ----------------------------------------------------------------------------- row f1: the crop of ANY box, and the real-data path
def _u8_image(h, w, seed):
    """A smooth, structured 8-bit RGB image (sums of a few sinusoids + noise): bilinear resampling of it is not degenerate."""
    g = np.random.Generator(np.random.Philox(key=[57, seed]))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        for _ in range(4):
            fx, fy, ph = g.uniform(0.01, 0.12), g.uniform(0.01, 0.12), g.uniform(0, 6.28)
            img[..., c] += np.sin(fx * xx + fy * yy + ph)
    img = (img - img.min()) / (img.max() - img.min()) * 235 + g.uniform(0, 20, (h, w, 3))
    return img.astype(np.uint8)


def test_crop_kernel_two_warp_boxes_bit_exact(model, pkg, oracle):
    """grnet_crop_normalise_cv_maps on NON-SQUARE boxes (the reference's two-warp branch, img_utils.py:97-106) and square ones mixed in
    one call, boxes hanging over every border, an odd intermediate width (half-pixel second warp), float32 / float64 boxes,
    per-frame images and one shared image, RGB / BGR: bit-identical to the oracle's patch_image_cv + normalisation."""
    imgs = np.stack([_u8_image(260, 340, s) for s in range(6)])
    boxes = np.array([[170.0, 130.0, 300.0, 150.0], [30.25, 240.5, 101.0, 224.0], [320.0, 20.0, 90.0, 160.0], [100.0, 100.0, 180.0, 180.0],
                      [5.5, 250.0, 260.0, 130.0], [200.0, 128.0, 223.0, 111.0]], np.float32)
    for bb in (boxes, boxes.astype(np.float64)):
        for scale in (1.0, 1.1):
            got = model.crop_normalise(torch.from_numpy(imgs).cuda(), torch.from_numpy(bb), scale=scale).cpu().numpy()
            for i in range(len(bb)):
                assert np.array_equal(got[i], oracle.crop_normalise_box_cv(imgs[i], bb[i], scale)), (i, scale)
    one = model.crop_normalise(torch.from_numpy(imgs[1]).cuda(), torch.from_numpy(boxes), scale=1.1).cpu().numpy()
    for i in range(len(boxes)):
        assert np.array_equal(one[i], oracle.crop_normalise_box_cv(imgs[1], boxes[i], 1.1))
    bgr = model.crop_normalise(torch.from_numpy(imgs[:, :, :, ::-1].copy()).cuda(), torch.from_numpy(boxes), scale=1.1, bgr=True).cpu().numpy()
    assert np.array_equal(bgr[0], oracle.crop_normalise_box_cv(imgs[0], boxes[0], 1.1))
    # the letterbox of the 2:1 box is exactly the normalised zero
    zero = ((0.0 - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)).astype(np.float32)
    wide = model.crop_normalise(torch.from_numpy(imgs[:1]).cuda(), torch.from_numpy(boxes[:1]), scale=1.0).cpu().numpy()[0]
    assert np.array_equal(wide[:, :56], np.broadcast_to(zero[:, None, None], (3, 56, 224))) and np.array_equal(wide[:, 168:], np.broadcast_to(zero[:, None, None], (3, 56, 224)))


def _write_png(folder, images):
    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    for i, im in enumerate(images):
        Image.fromarray(im).save(os.path.join(folder, f"{i:06d}.png"))


def test_demo_on_png_frames_matches_oracle_crops(pkg, oracle, tmp_path):
    """demo.py's real-data path (BASELINE configs[0] in its image form): 8-bit PNG frames are decoded, uploaded, cropped + normalised by
    the HIP kernel (InferenceFrames.batches -> GRNet.crop_normalise) and run through the model in batches of 16; compared frame by
    frame with the model fed the ORACLE's crops of the same frames (bit-exact crop => only the call-size bound remains).  Square
    tracker boxes that hang over the image border; two tracks."""
    sys.path.insert(0, ROOT)
    demo = importlib.import_module("demo")
    n = 28
    imgs = [_u8_image(240, 320, 100 + i) for i in range(n)]
    img_dir = str(tmp_path / "clip")
    _write_png(img_dir, imgs)
    t = np.arange(n, dtype=np.float32)
    box1 = np.stack([40 + 8 * t, 60 + 5 * t, 150 + 2 * t, 150 + 2 * t], 1).astype(np.float32)        # drifts from the top-left corner outwards
    box2 = np.stack([300 - 2 * t, 200 + t, np.full(n, 180.0), np.full(n, 180.0)], 1).astype(np.float32)   # hangs over the right / bottom border
    tp = str(tmp_path / "tracking.pkl")
    joblib.dump({7: {"bbox": box1.copy(), "frames": np.arange(n)}, 9: {"bbox": box2[2:].copy(), "frames": np.arange(2, n)}}, tp)
    args = demo.parser().parse_args(["--img_folder", img_dir, "--tracking_path", tp, "--output_folder", str(tmp_path / "out"),
                                     "--synthetic_weights", "--grnet_batch_size", "16", "--max_frames", "16"])
    res = joblib.load(demo.main(args))
    assert sorted(res) == [7, 9]
    m = pkg.build_synthetic_model(max_frames=32, with_gru=False)
    for pid, bb, fr in ((7, box1, np.arange(n)), (9, box2[2:], np.arange(2, n))):
        crops = np.stack([oracle.crop_normalise_box_cv(imgs[f], b, 1.0) for f, b in zip(fr, bb)])
        direct = m(torch.from_numpy(crops).cuda())[-1]
        torch.cuda.synchronize()
        r = res[pid]
        assert r["joints3d"].shape == (len(fr), 29, 3) and np.array_equal(r["frame_ids"], fr) and np.array_equal(r["bboxes"], bb)
        assert rel_err(r["joints3d"], direct["kp_3d"][0].cpu().numpy()) < CALL_SIZE_NOISE
        assert rel_err(r["pose"], direct["theta"][0, :, 3:75].cpu().numpy()) < CALL_SIZE_NOISE
        assert rel_err(r["verts"], direct["verts"][0].cpu().numpy()) < CALL_SIZE_NOISE
    m.close()


def test_batch_generation_on_png_frames_incl_non_square_annotations(pkg, oracle, tmp_path):
    """batch_generation.prepare_data on image files (run_on_frames -> GPU crop): two videos with frames of DIFFERENT sizes; the
    second video's precomputed annotations are NON-SQUARE boxes (batch_generation.py:39-93 produces such boxes from 2D joints), which
    take the reference's aspect-preserving two-warp crop.  joints3D vs the model on the oracle's crops, boxes scaled by 1.1 in place
    and by 1.1 again in the crop (inference.py:48,80), kinectv2 order."""
    sys.path.insert(0, ROOT)
    bg = importlib.import_module("batch_generation")
    vids = {"S001C001P001R001A001": ([_u8_image(200, 300, 200 + i) for i in range(7)],
                                     np.tile(np.array([[150.0, 100.0, 170.0, 170.0]], np.float32), (7, 1)) + np.arange(7, dtype=np.float32)[:, None] * np.array([3, 2, 1, 1], np.float32)),
            "S001C001P001R001A002": ([_u8_image(260, 180, 300 + i) for i in range(5)],
                                     np.tile(np.array([[90.0, 130.0, 100.0, 210.0]], np.float32), (5, 1)) + np.arange(5, dtype=np.float32)[:, None] * np.array([2, -3, 1, 2], np.float32))}
    annos = {}
    for name, (imgs, bb) in vids.items():
        _write_png(str(tmp_path / "vids" / name), imgs)
        annos[name] = bb.copy()
    bp = str(tmp_path / "bbox.pkl")
    joblib.dump(annos, bp)
    written = bg.prepare_data(fv=bp, vid_folder=str(tmp_path / "vids"), outpath=str(tmp_path / "db.json"), synthetic_weights=True, max_frames=8)
    db = joblib.load(written[0])
    assert db["joints3D"].shape == (12, 25, 3) and list(db["vid_name"][:7]) == ["S001C001P001R001A001"] * 7
    m = pkg.build_synthetic_model(max_frames=8, with_gru=False)
    row = 0
    for name, (imgs, bb) in vids.items():
        scaled = bb.copy()
        scaled[:, 2:] *= np.float32(1.1)                                        # Inference.__init__, in place, float32
        assert np.array_equal(db["bbox"][row:row + len(bb)], scaled)
        crops = np.stack([oracle.crop_normalise_box_cv(im, b, 1.1) for im, b in zip(imgs, scaled)])
        kp = m(torch.from_numpy(crops).cuda())[-1]["kp_3d"][0].cpu().numpy()
        assert rel_err(db["joints3D"][row:row + len(bb)], pkg.pipeline.spin2_to_kinectv2(kp)) < CALL_SIZE_NOISE, name
        row += len(bb)
    m.close()


def test_c_abi_exchange_world1(pkg):
    """grnet_comm_create + grnet_allgather: RCCL bound at run time from the process's librccl.so.1, a communicator of ONE rank on this
    box's one GPU; the all-gather of one rank is the identity, on the caller's stream, for the packed pose-record block of a 16-frame clip and for an
    odd byte count.  (N > 1 needs one GPU per rank: the 8-GPU node of the round-end driver run, `bench.py --gpus N --exchange capi`.)"""
    import ctypes as C
    harness = pkg.harness
    dev = torch.device("cuda", 0)
    comm = harness.RcclComm(1, 0, dev)
    w, r = C.c_int(), C.c_int()
    assert comm._lib.grnet_comm_info(comm._h, C.byref(w), C.byref(r)) == 0 and (w.value, r.value) == (1, 0)
    assert comm.info() == (1, 0)                                   # what bench.py's exchange proof cross-checks config.exchange_ranks against
    _, block = harness.pack_layout(16, harness.POSE_RECORD_GAIT)
    g = torch.Generator(device="cpu").manual_seed(5)
    send = torch.randn(block, generator=g).to(dev)
    recv = torch.zeros_like(send)
    out = harness.gather_pose_records(send, 16, 1, None, out=recv)       # world 1: plain copy, no collective
    assert torch.equal(out.view(-1), send)
    recv.zero_()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        comm.all_gather(recv, send)
    st.synchronize()
    assert torch.equal(recv, send)
    odd = torch.arange(1001, dtype=torch.uint8, device=dev)
    got = torch.zeros_like(odd)
    comm.all_gather(got, odd)
    torch.cuda.synchronize()
    assert torch.equal(got, odd)
    with pytest.raises(AssertionError):
        comm.all_gather(torch.zeros(3, device=dev), torch.zeros(2, device=dev))
    comm.close()
    comm.close()                                                         # idempotent


@pytest.mark.parametrize("dtype,use_graph,call_frames", [("f32", False, None), ("f32", True, 8), ("f32", True, 12), ("bf16", True, None)])
def test_overlapped_track_runner_equals_sequential_calls(pkg, dtype, use_graph, call_frames):
    """BASELINE configs[4]'s loop (harness.OverlappedTrackRunner: crops on a side stream into two alternating buffers, forwards -- graph replay or lane
    streams -- on the caller's stream, tracks packed into calls of <= call_frames frames, the next step's first crop staged under this step's last forward,
    no allocation per step) gives what crop_normalise + forward per track give through the allocating host API (bit for bit when a call is one track; to the
    call-size noise of the kernels' different tilings when tracks share a call), for tracks of different lengths, boxes over the border, repeated steps (buffer reuse + captured graphs), and with the side stream switched off."""
    harness = pkg.harness
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    video = torch.randint(0, 256, (7, 360, 480, 3), dtype=torch.uint8, device=dev, generator=g)
    lens = [7, 4, 7]
    raws = [video[:t] for t in lens]
    boxes = [np.stack([np.linspace(40 + 150 * k, 120 + 150 * k, t), np.linspace(60, 330, t), np.full(t, 180.0), np.full(t, 180.0)], 1).astype(np.float32)
             for k, t in enumerate(lens)]
    model = pkg.build_synthetic_model(max_frames=18, device_id=0, with_gru=False, dtype=dtype)
    ref = []
    for raw, box in zip(raws, boxes):
        out = model(model.crop_normalise(raw, torch.as_tensor(box), scale=1.1).unsqueeze(0))[-1]
        ref.append({k: out[k].clone() for k in ("theta", "kp_3d", "kp_2d", "verts")})
    for overlap in (True, False):
        runner = harness.OverlappedTrackRunner(model, raws, boxes, use_graph=use_graph, tune_level=0, overlap=overlap, call_frames=call_frames)
        assert [[k for k, _ in c] for c in runner.calls] == {None: [[0, 1, 2]], 8: [[0], [1], [2]], 12: [[0, 1], [2]]}[call_frames]
        for _ in range(3):
            res = runner.step()
        torch.cuda.synchronize()
        for k in range(len(lens)):
            for name in ref[k]:
                a, b = res[k][name].reshape(-1), ref[k][name].reshape(-1)
                if call_frames == 8:
                    assert torch.equal(a, b), (overlap, k, name)
                else:
                    a, b = a.cpu().numpy(), b.cpu().numpy()
                    assert rel_err(a, b) <= (CALL_SIZE_NOISE if dtype == "f32" else 2e-2), (overlap, k, name, rel_err(a, b))
    with pytest.raises(ValueError):
        harness.OverlappedTrackRunner(model, raws, boxes, call_frames=5)
    model.close()
