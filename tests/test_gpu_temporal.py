"""The temporal branch (grnet.py:154-173): FeatCorrector (feature_correction.py:104-157) and the gait branch against goldens made by the reference's own code,
frame shards gathered before the temporal branch, the attention block (attention_utils.py:261-270) on clips of 4 200 / 17 000 / 2 x 1 100 frames (per-query and
blocked kernels, the > 64 KB LDS launch), and every form of the GRU recurrence (gait_feat_encoder.py:79-104; GRNET_OPT_GRU_MODE) on long sequences.
Regrouped by component in round 6; the tests themselves are unchanged."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from .conftest import CALL_SIZE_NOISE, ROOT, elem_ratio, rel_err

pytestmark = pytest.mark.gpu

def test_feature_corrector_and_gait_branch_match_reference_golden(pkg, oracle, synth_weights, synth_smpl):
    """Row f2 end to end on the GPU: GRNet(use_gait_feat=True) -- first head pass, cparams (grnet.py:156-160), FeatCorrector
    (feature_correction.py:104-157), second head pass, regressor -- against the outputs of the reference's own code run with its
    undefined names bound (tests/golden/featcorr.npz), and the corrector alone against its module golden and the oracle."""
    import os
    from .conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "featcorr.npz"))
    m = pkg.build_synthetic_model(max_frames=3, use_gait_feat=True)          # 4 frames > max_frames: both passes chunk
    frames = torch.from_numpy(pkg.synth.make_frames(4)).cuda().reshape(1, 4, 3, 224, 224)
    bbox, cimg = pkg.synth.make_gait_boxes(1, 4)
    out = m(frames, bbox=torch.from_numpy(bbox).cuda(), cimg=torch.from_numpy(cimg).cuda())[-1]
    torch.cuda.synchronize()
    assert rel_err(out["pred_cparam"].cpu().numpy(), g["gait_pred_cparam"]) < 1e-5
    assert rel_err(out["pred_avg"].cpu().numpy(), g["gait_pred_avg"]) < 1e-4
    assert rel_err(out["pred_phase"].cpu().numpy(), g["gait_pred_phase"]) < 1e-4
    for k in ("theta", "kp_3d", "kp_2d", "rotmat"):
        assert out[k].shape == g["gait_" + k].shape, k
        assert rel_err(out[k].cpu().numpy(), g["gait_" + k]) < 1e-4, (k, rel_err(out[k].cpu().numpy(), g["gait_" + k]))
    assert rel_err(out["verts"].cpu().numpy()[:, :, ::5], g["gait_verts_s5"]) < 1e-4
    # the corrector alone on the module golden's inputs (features in, corrected features out)
    sd = pkg.synth.make_featcorr_state_dict()
    for (b, n) in ((2, 8), (1, 16), (1, 1)):
        x, cp = pkg.synth.make_featcorr_inputs(b, n)
        # gait_correct derives cparams from (cam, bbox, cimg): choose them so that cparams == cp exactly
        # (bbox w = 224 -> bs = 1, cam = [s, tx, ty] = cp, bbox centre == cimg -> no translation term)
        bb = np.zeros((b, n, 4), np.float32); bb[..., 2:] = 224.0
        ci = np.zeros((b, n, 2), np.float32)
        csf = np.zeros((b * n, 64, 24), np.float32)
        r = m.gait_correct(torch.from_numpy(x).reshape(b * n, 128, 24), torch.from_numpy(csf), torch.from_numpy(cp).reshape(b * n, 3),
                           torch.from_numpy(bb), torch.from_numpy(ci), b, n)
        torch.cuda.synchronize()
        assert np.array_equal(r["pred_cparam"].cpu().numpy(), cp.reshape(-1, 3))
        ry, ravg, rph = oracle.feat_corrector(x, cp, sd)
        assert rel_err(r["point_local_feat"].cpu().numpy(), ry) < 3e-5, (b, n)
        assert rel_err(r["pred_avg"].cpu().numpy(), ravg) < 1e-4 and rel_err(r["pred_phase"].cpu().numpy(), rph) < 1e-4
        if f"y_{b}_{n}" in g.files:
            assert rel_err(r["point_local_feat"].cpu().numpy(), g[f"y_{b}_{n}"]) < 3e-5, (b, n)
    m.close()
    m2 = pkg.build_synthetic_model(max_frames=2, with_gru=True)               # corrector weights absent: loud
    with pytest.raises(pkg._lib.GrnetError):
        m2.gait_correct(torch.zeros(2, 128, 24), torch.zeros(2, 64, 24), torch.ones(2, 3), torch.ones(1, 2, 4), torch.zeros(1, 2, 2), 1, 2)
    m2.close()
    with pytest.raises(ValueError):
        pkg.GRNet(max_frames=1, use_gait_feat=True, featcorr=dict(AVG_DIM=3, ESTIM_PHASE=True, NUM_LAYERS=2, H_SIZE=1024, NUM_HEADS=4, USE_JWFF=True))

def test_frame_shards_gather_then_temporal_branch_equals_one_process(pkg):
    """BASELINE configs[3]'s data flow on one GPU: 3 'ranks' run the per-frame path on their shard_range() of a 2-clip batch, the
    packed records (theta, kp, point_local_feat, cam_shape_feats) are reassembled exactly as the all-gather delivers them, and the
    temporal branch runs on the whole sequence -- same result as GRNet(use_gait_feat=True) in one process."""
    h = pkg.harness
    b, t, world = 1, 14, 3
    n_total = b * t
    m = pkg.build_synthetic_model(max_frames=8, use_gait_feat=True)
    frames = torch.from_numpy(pkg.synth.make_frames(n_total)).cuda()
    bbox, cimg = pkg.synth.make_gait_boxes(b, t)
    bbox, cimg = torch.from_numpy(bbox).cuda(), torch.from_numpy(cimg).cuda()
    whole = m(frames.reshape(b, t, 3, 224, 224), bbox=bbox, cimg=cimg)[-1]
    per = -(-n_total // world)
    blocks = []
    m.use_gait_feat = False                                   # the ranks run the per-frame path only
    for rank in range(world):
        lo, hi = h.shard_range(n_total, world, rank)
        pad = torch.zeros(per, 3, 224, 224, device="cuda")
        pad[:hi - lo] = frames[lo:hi]
        runner = h.ClipRunner(m, pad, use_graph=False, tune_level=0, record=h.POSE_RECORD_GAIT)
        runner.step()
        torch.cuda.synchronize()
        blocks.append(runner.packed.clone())
    m.use_gait_feat = True
    gathered = torch.stack(blocks)                            # what all_gather_into_tensor(...).view(world, block) holds on every rank
    seq = h.unpack_sequence(gathered, per, n_total, h.POSE_RECORD_GAIT)
    assert seq["cam_shape_feats"].shape == (n_total, 64, 24)
    got = h.temporal_after_gather(m, seq, bbox, cimg, b, t)
    torch.cuda.synchronize()
    for k in ("theta", "kp_3d", "verts", "rotmat"):
        assert rel_err(got[k].cpu().numpy().reshape(whole[k].shape), whole[k].cpu().numpy()) < CALL_SIZE_NOISE, k
    assert rel_err(got["pred_phase"].cpu().numpy(), whole["pred_phase"].cpu().numpy()) < CALL_SIZE_NOISE
    m.close()

def test_attention_block_takes_clips_longer_than_4096_frames(pkg, oracle):
    """The attention block of the temporal branch on ONE clip of 4 200 frames (round 2 refused n > 4096 deep inside launch_tsattn,
    after the GRU had been enqueued): same kernels, the softmax row over the clip's frames is 4 200 floats of LDS.  Checked against the
    oracle; the limit that remains (32 768 frames per clip: 128 KB of LDS) is refused UP FRONT with a message that says what to do, by
    grnet_tsattn_forward and by grnet_gait_correct alike."""
    m = pkg.build_synthetic_model(max_frames=2, with_gru=True, with_tsattn=True, use_gait_feat=False)
    tsd = pkg.synth.make_tsattn_state_dict()
    x, xs = pkg.synth.make_tsattn_inputs(1, 4200)
    y = m.tsattn_forward(torch.from_numpy(x).cuda(), torch.from_numpy(xs).cuda()).cpu().numpy()
    ref = oracle.ts_attn_block(x, xs, tsd)
    assert y.shape == ref.shape and rel_err(y, ref) < 5e-5, rel_err(y, ref)
    with pytest.raises(pkg._lib.GrnetError, match="split the sequence into clips"):
        m.tsattn_forward(torch.zeros(1, 32769, 128, 24).cuda(), torch.zeros(1, 32769, 128, 25).cuda())
    m.close()
    mg = pkg.build_synthetic_model(max_frames=2, with_gru=True, use_gait_feat=True)
    t = 32769
    with pytest.raises(pkg._lib.GrnetError, match="split the sequence into clips"):
        mg.gait_correct(torch.zeros(t, 128, 24).cuda(), torch.zeros(t, 64, 24).cuda(), torch.zeros(t, 3).cuda(), torch.zeros(1, t, 4).cuda(),
                        torch.zeros(1, t, 2).cuda(), 1, t)
    mg.close()

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

@pytest.mark.parametrize("b,n", [(2, 1100), (3, 500), (1, 385)], ids=["2x1100", "3x500", "1x385"])
def test_blocked_temporal_attention_ragged_blocks_and_key_parts(pkg, oracle, b, n):
    """temporal_attn_flash_kernel (clips >= 384 frames: 128 queries per workgroup as S^T = K Q^T / O^T += V^T P^T, keys / values in blocks of 32 through two LDS
    stages, keys split over up to 8 workgroups whose shares temporal_attn_combine_kernel merges) against the oracle: 1 100 frames -- a last query tile of 76 rows
    (4.75 waves), a last key block of 12, 35 key blocks over 7 parts -- 500 frames (3 parts, last block of 20 keys) and 385 (one key block beyond 12 full
    ones, 3 parts); the two-stage frame mean of the gate; each clip also alone (clips must not see each other, and the part count follows from n alone)."""
    from .conftest import rel_err
    m = pkg.build_synthetic_model(max_frames=2, with_gru=True, with_tsattn=True, use_gait_feat=False)
    tsd = pkg.synth.make_tsattn_state_dict()
    x, xs = pkg.synth.make_tsattn_inputs(b, n)
    xd, xsd = torch.from_numpy(x).cuda(), torch.from_numpy(xs).cuda()
    y = m.tsattn_forward(xd, xsd).cpu().numpy()
    ref = oracle.ts_attn_block(x, xs, tsd)
    assert y.shape == ref.shape and rel_err(y, ref) < 5e-5, rel_err(y, ref)
    y1 = m.tsattn_forward(xd[b - 1:b], xsd[b - 1:b]).cpu().numpy()
    assert np.array_equal(y1[0], y[b - 1])
    m.close()

# ---- GRU recurrence, round 5: rows-per-wave kernel, hand-off inside the XCD's L2 (gru_kernels.hip) --------------------------------------
@pytest.mark.parametrize("mode", [3, 3 + 16, 2, 1, 0], ids=["default", "agent_scope_stores", "libm_gates", "column_slices", "unsplit"])
def test_gru_recurrence_variants_long_sequences(pkg, oracle, mode):
    """Every form of the recurrence (GRNET_OPT_GRU_MODE) against the oracle (gait_feat_encoder.py:79-104) on sequences long enough for an error of the gate
    functions or a missed hand-off to show: 1 x 2000 steps (each direction 2 layers x 2000 dependent steps), 3 x 257 (six groups of eight
    workgroups), 16 x 9 (the largest batch the split form takes).  The default takes the v_exp / v_rcp gate functions and, where the eight
    slices of a group share an XCD, workgroup-scope granule stores; + 16 is the path of a group that spans XCDs (and the handle's own fall-back
    after a hand-off timeout)."""
    m = pkg.build_synthetic_model(max_frames=4, with_gru=True)
    try:
        m.set_option(pkg._lib.OPT_GRU_MODE, mode)
        sd = pkg.synth.make_gru_state_dict()
        for (b, t) in [(1, 2000), (3, 257), (16, 9)]:
            x, cp = pkg.synth.make_gru_inputs(b, t)
            y, ph, _ = m.gru_forward(torch.from_numpy(x).cuda(), torch.from_numpy(cp).cuda())
            torch.cuda.synchronize()
            ry, rph, _ = oracle.gru_forward(x, cp, sd)
            e = lambda a, r: float(np.abs(a.cpu().numpy().astype(np.float64) - r).max() / np.abs(r).max())
            assert bool(torch.isfinite(y).all() and torch.isfinite(ph).all()), (b, t, mode)
            assert e(y, ry) < 1e-4 and e(ph, rph) < 1e-4, (b, t, mode, e(y, ry), e(ph, rph))
        with pytest.raises(pkg._lib.GrnetError):
            m.set_option(pkg._lib.OPT_GRU_MODE, 7)
    finally:
        m.close()
