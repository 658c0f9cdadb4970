"""The two entry points end to end on the GPU with synthetic frames / weights: flags, model loop,
output schema (.pkl of demo.py:211-222 and the joblib db of batch_generation.py:265-267)."""
import importlib
import os
import sys

import joblib
import numpy as np
import pytest
import torch

from .conftest import CALL_SIZE_NOISE, ROOT, rel_err

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
