"""Round-4 GPU parity tests: the fuse layer of every HR module (hrnet.py:189-244, 258-265) as the fp32 path launches it since round 4 --
the stride-2 chains run early as plain convolutions on the streams of the branches they start from, ONE grouped launch
(csrc/hr_fuse.hip) finishes outputs 0 .. nb-2 (all 1x1 "up" terms, the identity, the finished chains, ReLU) and one stride-2
convolution finishes the last output -- against the oracle's fuse layer fed the SAME branch outputs the GPU produced."""
import importlib
import os
import sys

import joblib
import numpy as np
import pytest
import torch

from .conftest import CALL_SIZE_NOISE, ROOT, rel_err

pytestmark = pytest.mark.gpu

MODULES = [("stage2", 0, 2)] + [("stage3", m, 3) for m in range(4)] + [("stage4", m, 4) for m in range(3)]


@pytest.fixture(scope="module")
def model(pkg):
    m = pkg.build_synthetic_model(max_frames=16, with_gru=False)
    yield m
    m.close()


@pytest.mark.parametrize("n", [1, 3, 16])
def test_fuse_layer_of_every_hr_module_matches_oracle(model, pkg, oracle, synth_weights, n):
    """For each of the 8 HR modules: the module's branch outputs x_b and outputs y_i are read back from the HIP forward
    (grnet_debug_tensor), the oracle's hr_fuse runs on those x_b with the same weights, and every y_i must agree to 2e-5 of its scale
    (fp32 sums re-associated: the grouped launch adds identity, chains, bias and up terms in its own order).  Covers stage 2 (one 1x1 term, one stride-2 convolution), stage 3 (3 + 3 incl. a two-convolution chain)
    and stage 4 (6 terms, chains of one / two / three stride-2 convolutions, the merged first convolutions of the chains that start at one branch, linear and ReLU'd segments in one launch);
    1 / 3 / 16 frames."""
    frames = pkg.synth.make_frames(n)
    model(torch.from_numpy(frames).cuda().unsqueeze(0))
    torch.cuda.synchronize()
    for stage, m, nb in MODULES:
        tag = f"{stage}.{m}."
        xs = [model.debug_tensor(tag + f"x{b}", n).cpu() for b in range(nb)]
        for b in range(nb):
            assert xs[b].shape == (n, 32 << b, 56 >> b, 56 >> b)
        ref = oracle.hr_fuse(xs, synth_weights, f"backbone.{tag}")
        for i in range(nb):
            got = model.debug_tensor(tag + f"y{i}", n).cpu().numpy()
            e = rel_err(got, ref[i].numpy())
            assert got.shape == tuple(ref[i].shape) and e < 2e-5, (tag, i, e)
            assert got.min() >= 0.0                                     # the ReLU is applied exactly once, by the finishing launch


def test_fuse_layer_launch_count_and_macs(pkg):
    """The grouped fuse launch replaces 31 1x1 convolution launches and 8 elementwise sums; the merged first convolution of the
    stage-4 chains (2,0) / (3,0) three more launches: 272 convolutions + 8 grouped launches, and the MACs still add up to SURVEY
    8(d)'s 15 441 563 648 per frame (the 1x1 terms are computed, not dropped)."""
    m = pkg.build_synthetic_model(max_frames=2, with_gru=False)
    convs = m.describe_convs()
    grouped = [c for c in convs if c["cin"] == 0]
    assert len(grouped) == 8 and all(c["name"].endswith("fuse_layers(up)") for c in grouped)
    assert len(convs) == m.num_conv_launches() == 280
    assert sum(c["macs"] for c in convs) == 15441563648
    assert not any(c["ks"] == 1 and "fuse_layers" in c["name"] and c["cin"] for c in convs)      # no separate 1x1 fuse launch is left
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


def test_attention_block_large_lds_branch_17000_frames(pkg, oracle):
    """The temporal attention keeps one softmax row over the clip's frames in LDS; beyond ~15 800 frames that is more than 64 KB and the
    launcher raises the kernel's dynamic-LDS limit (up to the 160 KB of gfx950, from which the 32 768-frame limit follows; both read from
    the device).  A 17 000-frame clip made of a 50-frame pattern repeated 340 times: every distinct key appears 340 times with the same
    logit, so each frame's attention output -- and the clip means of the gate -- equal those of the 50-frame clip, which the CPU oracle
    computes in a moment.  Covers the large-LDS launch, which no test ran before (round-3 advisor)."""
    m = pkg.build_synthetic_model(max_frames=2, with_gru=True, with_tsattn=True, use_gait_feat=False)
    tsd = pkg.synth.make_tsattn_state_dict()
    x, xs = pkg.synth.make_tsattn_inputs(1, 50)
    reps = 340
    xl, xsl = np.tile(x, (1, reps, 1, 1)), np.tile(xs, (1, reps, 1, 1))
    y = m.tsattn_forward(torch.from_numpy(xl).cuda(), torch.from_numpy(xsl).cuda()).cpu().numpy()
    ref = oracle.ts_attn_block(x, xs, tsd)
    assert y.shape == (1, 50 * reps) + ref.shape[2:]
    assert rel_err(y[:, :50], ref) < 5e-5 and rel_err(y[:, -50:], ref) < 5e-5
    assert rel_err(y.reshape(reps, 50, -1), np.broadcast_to(ref.reshape(1, 50, -1), (reps, 50, ref[0, 0].size))) < 5e-5
    m.close()


@pytest.mark.parametrize("case", [(16, 480, 256, 56), (3, 256, 256, 56), (1, 128, 128, 56), (5, 32, 256, 56), (4, 48, 128, 56), (2, 64, 384, 56),
                                  (4, 64, 64, 56), (16, 64, 64, 28), (5, 256, 256, 28)],
                         ids=lambda c: "x".join(map(str, c)))
def test_wide_winograd_kernel_eight_waves(model, oracle, case):
    """conv_wino4w_f32 (F(4x4,3x3), eight waves per workgroup sharing one transformed chunk, 16-channel chunks, the two waves of a SIMD in
    opposite phases) vs the oracle's direct convolution.  128 output channels per workgroup: the three 56x56 layer shapes of the heads
    (480 -> 256 at 16 frames = 448 workgroups, 256 -> 256, 128 -> 128), two chunks (32 channels), an odd chunk count (48), three channel
    blocks (384).  The last three shapes run the eight-wave kernel only under GRNET_WINO_WIDE bits 1-3 (64 channels per workgroup, 28x28
    maps: measured and left off, conv_wino4_wide) -- by default they repeat the 4-wave kernel's check on the same data.  Bias + ReLU, + residual, the linear form with the borders looked at separately, bit-identical repeats, and agreement with
    the 4-wave kernel (hint 2003) to re-association noise.  Bound 1e-4 of the output scale as for the other F(4x4,3x3) kernels."""
    n, cin, cout, hw = case
    g = np.random.Generator(np.random.Philox(key=[94, n * 100000 + cin * 1000 + cout + hw]))
    x = g.standard_normal((n, cin, hw, hw)).astype(np.float32)
    w = (g.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    b = (g.standard_normal((cout,)) * 0.1).astype(np.float32)
    r = g.standard_normal((n, cout, hw, hw)).astype(np.float32)
    conv = oracle.conv2d(x, w, stride=1, bias=b)
    xd, rd = torch.from_numpy(x).cuda(), torch.from_numpy(r).cuda()
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, tile_hint=2001).cpu().numpy()
    assert got.shape == conv.shape
    assert rel_err(got, torch.relu(conv).numpy()) < 1e-4, rel_err(got, torch.relu(conv).numpy())
    got_r = model.op_conv2d(xd, w, b, stride=1, relu=True, add=rd, tile_hint=2001).cpu().numpy()
    assert rel_err(got_r, torch.relu(conv + torch.from_numpy(r)).numpy()) < 1e-4
    old_r = model.op_conv2d(xd, w, b, stride=1, relu=True, add=rd, tile_hint=2003).cpu().numpy()
    assert rel_err(got_r, old_r) < 2e-6                                       # same transform arithmetic, the k order of the sums differs
    lin = oracle.conv2d(x, w, stride=1).numpy()
    got = model.op_conv2d(xd, w, None, stride=1, relu=False, tile_hint=2001).cpu().numpy()
    assert rel_err(got, lin) < 1e-4
    assert rel_err(got[:, :, [0, hw - 1]], lin[:, :, [0, hw - 1]]) < 1e-4 and rel_err(got[..., [0, hw - 1]], lin[..., [0, hw - 1]]) < 1e-4
    again = model.op_conv2d(xd, w, None, stride=1, relu=False, tile_hint=2001).cpu().numpy()
    assert np.array_equal(got, again)


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
