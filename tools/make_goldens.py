#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (this container only).

The reference (/root/reference, Python) is imported read-only with five stub
modules for dependencies that are absent offline:
``turtle``, ``yacs``, ``torchvision``, ``timm`` and ``smplx``.  The first four
are import shims with no arithmetic.  ``smplx`` is third-party arithmetic that
is neither installed nor vendored (pinned smplx==0.1.26, requirements.txt:13):
the stand-in below restates the published SMPL linear-blend-skinning (SURVEY
A.7), so parity at that boundary is "unpinned" -- it is pinned only to this
stand-in, never to smplx itself.

Weights, SMPL tables and frames come from ``synth.py`` (seed-defined), so the
fixtures hold only reference OUTPUTS; inputs are regenerated on the GPU box.
Nothing from /root/reference is written anywhere.

    python tools/make_goldens.py            # writes tests/golden/grnet_n4.npz, gru_b2_t6.npz
"""
import importlib
import os
import sys
import tempfile
import textwrap
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
netspec, synth = pkg.netspec, pkg.synth

STUBS = {
    "turtle.py": "def forward(*a, **k):\n    pass\n",
    "yacs/__init__.py": "",
    "yacs/config.py": textwrap.dedent('''
        import copy, yaml
        class CfgNode(dict):
            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError:
                    raise AttributeError(k)
            def __setattr__(self, k, v):
                self[k] = v
            def clone(self):
                return copy.deepcopy(self)
            def merge_from_file(self, f):
                def merge(a, b):
                    for k, v in b.items():
                        if isinstance(v, dict) and isinstance(a.get(k), dict):
                            merge(a[k], v)
                        else:
                            a[k] = v
                with open(f) as fh:
                    merge(self, yaml.safe_load(fh))
    '''),
    "torchvision/__init__.py": "",
    "torchvision/models/__init__.py": "",
    "torchvision/models/resnet.py": "",
    "torchvision/transforms.py": "",
    # plumbing-only shims so that lib/utils/demo_utils.py (cam / coordinate conversion) can be imported
    "cv2.py": "",
    "pytube.py": "class YouTube: pass\n",
    "skimage/__init__.py": "",
    "skimage/util/__init__.py": "",
    "skimage/util/shape.py": "def view_as_windows(*a, **k):\n    raise NotImplementedError\n",
    "timm/__init__.py": "",
    "timm/models/__init__.py": "",
    "timm/models/layers.py": "from torch.nn.init import trunc_normal_\n",
    "smplx/__init__.py": "from .body_models import SMPL\n",
    "smplx/utils.py": textwrap.dedent('''
        from dataclasses import dataclass
        from typing import Optional
        import torch
        @dataclass
        class ModelOutput:
            vertices: Optional[torch.Tensor] = None
            joints: Optional[torch.Tensor] = None
            full_pose: Optional[torch.Tensor] = None
            global_orient: Optional[torch.Tensor] = None
            transl: Optional[torch.Tensor] = None
        @dataclass
        class SMPLOutput(ModelOutput):
            betas: Optional[torch.Tensor] = None
            body_pose: Optional[torch.Tensor] = None
    '''),
    # Published SMPL LBS in batched-tensor form; stand-in for smplx.lbs
    "smplx/lbs.py": textwrap.dedent('''
        import torch
        def vertices2joints(J_regressor, vertices):
            return torch.einsum('bik,ji->bjk', [vertices, J_regressor])
        def blend_shapes(betas, shape_disps):
            return torch.einsum('bl,mkl->bmk', [betas, shape_disps])
        def lbs_rotmats(betas, rot_mats, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
            B = betas.shape[0]
            dtype = betas.dtype
            v_shaped = v_template[None] + blend_shapes(betas, shapedirs)
            J = vertices2joints(J_regressor, v_shaped)
            ident = torch.eye(3, dtype=dtype)
            pose_feature = (rot_mats[:, 1:] - ident).reshape(B, -1)
            v_posed = v_shaped + torch.matmul(pose_feature, posedirs).view(B, -1, 3)
            rel = J.clone()
            rel[:, 1:] = rel[:, 1:] - J[:, parents[1:]]
            T = torch.zeros(B, 24, 4, 4, dtype=dtype)
            T[:, :, :3, :3] = rot_mats
            T[:, :, :3, 3] = rel
            T[:, :, 3, 3] = 1
            chain = [T[:, 0]]
            for i in range(1, 24):
                chain.append(torch.matmul(chain[int(parents[i])], T[:, i]))
            G = torch.stack(chain, dim=1)
            posed_joints = G[:, :, :3, 3]
            Jh = torch.cat([J, torch.zeros(B, 24, 1, dtype=dtype)], dim=2).unsqueeze(-1)
            corr = torch.matmul(G, Jh)
            A = G.clone()
            A[:, :, :, 3] = A[:, :, :, 3] - corr[..., 0]
            Tv = torch.matmul(lbs_weights[None].expand(B, -1, -1), A.view(B, 24, 16)).view(B, -1, 4, 4)
            vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=dtype)], dim=2)
            verts = torch.matmul(Tv, vh.unsqueeze(-1))[:, :, :3, 0]
            return verts, posed_joints
    '''),
    "smplx/body_models.py": textwrap.dedent('''
        import numpy as np, torch, torch.nn as nn
        from .lbs import lbs_rotmats
        from .utils import SMPLOutput
        class SMPL(nn.Module):
            """Stand-in for smplx.SMPL: tables come from a synthetic npz in model_path."""
            def __init__(self, model_path, batch_size=1, create_transl=True, **kw):
                super().__init__()
                d = np.load(model_path + '/SMPL_SYNTH.npz')
                for k in ('v_template', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights'):
                    self.register_buffer(k, torch.tensor(d[k], dtype=torch.float32))
                self.register_buffer('parents', torch.tensor(d['parents'], dtype=torch.long))
                self.register_buffer('extra_joints_idxs', torch.tensor(d['extra_ids'], dtype=torch.long))
                self.faces = np.zeros((1, 3), np.int64)
            def forward(self, betas=None, body_pose=None, global_orient=None, pose2rot=True, **kw):
                assert not pose2rot
                full = torch.cat([global_orient, body_pose], dim=1)
                verts, joints = lbs_rotmats(betas, full, self.v_template, self.shapedirs, self.posedirs,
                                            self.J_regressor, self.parents, self.lbs_weights)
                joints = torch.cat([joints, verts[:, self.extra_joints_idxs]], dim=1)
                return SMPLOutput(vertices=verts, joints=joints, global_orient=global_orient,
                                  body_pose=body_pose, betas=betas, full_pose=full)
    '''),
}


def setup_workdir():
    tmp = tempfile.mkdtemp(prefix="grnet_goldens_")
    stubs = os.path.join(tmp, "stubs")
    for rel, src in STUBS.items():
        p = os.path.join(stubs, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(src)
    sd = synth.make_state_dict()
    smpl = synth.make_smpl_tables()
    os.makedirs(os.path.join(tmp, "data/smpl_data"))
    os.makedirs(os.path.join(tmp, "data/grnet_data"))
    np.savez(os.path.join(tmp, "data/smpl_data/smpl_mean_params.npz"),
             pose=sd["head.init_pose"][0], shape=sd["head.init_shape"][0], cam=sd["head.init_cam"][0])
    np.save(os.path.join(tmp, "data/smpl_data/J_regressor_extra.npy"), smpl["J_regressor_extra"])
    np.savez(os.path.join(tmp, "data/smpl_data/SMPL_SYNTH.npz"),
             extra_ids=np.asarray(netspec.SMPL_EXTRA_VERT_IDS), **smpl)
    return tmp, stubs, sd, smpl


def sample(t, stride):
    return np.ascontiguousarray(t[..., ::stride, ::stride])


def main():
    import torch
    torch.manual_seed(0)
    tmp, stubs, sd, smpl = setup_workdir()
    os.chdir(tmp)
    sys.path[:0] = [stubs, REF]

    # the PARE checkpoint the constructor insists on (grnet.py:87,99-108): same head weights, re-keyed
    pare_sd = {"model." + k: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if k.startswith("head.")}
    torch.save({"state_dict": pare_sd}, "data/grnet_data/pare_w_3dpw_checkpoint.ckpt")

    from lib.models.grnet import GRNet
    from lib.models.layers.gait_feat_encoder import BidirectionalModel
    GRNet.is_demo = True
    model = GRNet(writer=None, seqlen=100, featcorr=None).eval()

    # --- the spec must describe exactly the reference's tensors -----------------------------
    ref_sd = model.state_dict()
    spec = netspec.grnet_spec()
    ref_keys = [k for k in ref_sd if not k.startswith("regressor.")]
    for i, (a, b) in enumerate(zip(ref_keys, spec.keys())):
        if a != b:
            print("first order mismatch at", i, a, b); break
    assert set(ref_keys) == set(spec.keys()), (set(ref_keys) ^ set(spec.keys()))
    for k, (shape, _) in spec.items():
        assert tuple(ref_sd[k].shape) == tuple(shape), (k, ref_sd[k].shape, shape)
    n_backbone = sum(k.startswith("backbone.") for k in spec)
    n_head = sum(k.startswith("head.") for k in spec)
    print(f"state_dict keys match netspec: backbone {n_backbone}, head {n_head}")
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.startswith("regressor.") for k in missing), (missing, unexpected)

    N = 4
    frames = torch.from_numpy(synth.make_frames(N)).reshape(2, 2, 3, 224, 224)

    cap = {}

    def hook(name):
        def f(mod, inp, out):
            cap[name] = out
        return f

    bb = model.backbone
    bb.relu.register_forward_hook(lambda m, i, o: cap.setdefault("stem_relu", []).append(o.detach().clone()))
    bb.layer1.register_forward_hook(hook("layer1"))
    bb.stage2.register_forward_hook(hook("stage2"))
    bb.stage3.register_forward_hook(hook("stage3"))
    bb.stage4.register_forward_hook(hook("stage4"))
    bb.register_forward_hook(hook("features"))
    with torch.no_grad():
        t0 = time.time()
        out = model(frames)[-1]
        print(f"reference forward N={N}: {time.time() - t0:.2f}s")
        feats = cap["features"]
        plf, csf, hout = model.head.feature_extractor(features=feats)
        patt = model.head(plf, csf, dict(hout))

    def stat(name, t):
        t = t.float()
        print(f"  {name:24s} shape {tuple(t.shape)!s:22s} mean {t.mean():+.4f} std {t.std():.4f} absmax {t.abs().max():.3f}")

    stat("stem conv2", cap["stem_relu"][1])
    stat("layer1", cap["layer1"])
    for s in ("stage2", "stage3", "stage4"):
        for i, t in enumerate(cap[s]):
            stat(f"{s}[{i}]", t)
    stat("features", feats)
    stat("part_attn", hout["part_attn"])
    sm = torch.softmax(hout["part_attn"].reshape(N, 24, -1), -1)
    print(f"  softmax max prob: mean {sm.max(-1).values.mean():.4f} (uniform = {1 / 3136:.5f})")
    stat("smpl_feats", hout["smpl_feats"])
    stat("point_local_feat", plf)
    stat("cam_shape_feats", csf)
    stat("pred_rot6d", patt["pred_rot6d"])
    stat("pred_shape", patt["pred_shape"])
    stat("pred_cam", patt["pred_cam"])
    stat("kp_3d", out["kp_3d"])
    stat("kp_2d", out["kp_2d"])
    stat("verts", out["verts"])
    stat("theta", out["theta"])

    g = {
        "n_frames": np.int64(N),
        "weight_seed": np.int64(synth.WEIGHT_SEED), "smpl_seed": np.int64(synth.SMPL_SEED),
        "frame_seed": np.int64(synth.FRAME_SEED),
        "stem_conv1_s4": sample(cap["stem_relu"][0].numpy(), 4),
        "stem_conv2_s4": sample(cap["stem_relu"][1].numpy(), 4),
        "layer1_s4": sample(cap["layer1"].numpy(), 4),
        "stage2_0_s4": sample(cap["stage2"][0].numpy(), 4), "stage2_1_s2": sample(cap["stage2"][1].numpy(), 2),
        "stage3_0_s4": sample(cap["stage3"][0].numpy(), 4), "stage3_1_s2": sample(cap["stage3"][1].numpy(), 2),
        "stage3_2": cap["stage3"][2].numpy(),
        "stage4_0_s4": sample(cap["stage4"][0].numpy(), 4), "stage4_1_s2": sample(cap["stage4"][1].numpy(), 2),
        "stage4_2": cap["stage4"][2].numpy(), "stage4_3": cap["stage4"][3].numpy(),
        "features_s4": sample(feats.numpy(), 4),
        "features_chan_absmean": feats.abs().mean((2, 3)).numpy(),
        "part_attn_s2": sample(hout["part_attn"].numpy(), 2),
        "smpl_feats_s4": sample(hout["smpl_feats"].numpy(), 4),
        "part_feats_s4": sample(hout["part_feats"].numpy(), 4),
        "point_local_feat": plf.numpy(), "cam_shape_feats": csf.numpy(),
        "pred_rot6d": patt["pred_rot6d"].numpy(), "pred_shape": patt["pred_shape"].numpy(),
        "pred_cam": patt["pred_cam"].numpy(), "pred_rotmat": patt["pred_rotmat"].numpy(),
        "theta": out["theta"].numpy(), "kp_3d": out["kp_3d"].numpy(), "kp_2d": out["kp_2d"].numpy(),
        "rotmat": out["rotmat"].numpy(),
        "verts_s5": np.ascontiguousarray(out["verts"].numpy()[:, :, ::5]),
        "verts_frame0": out["verts"].numpy()[0, 0],
    }
    os.makedirs(os.path.join(ROOT, "tests/golden"), exist_ok=True)
    p = os.path.join(ROOT, "tests/golden/grnet_n4.npz")
    np.savez_compressed(p, **g)
    print(f"wrote {p} ({os.path.getsize(p) / 1e6:.2f} MB)")

    # --- geometry edge cases straight from the reference functions ---------------------------
    from lib.utils.geometry import rot6d_to_rotmat, rotation_matrix_to_angle_axis
    gg = np.random.Generator(np.random.Philox(key=[7, 7]))
    r6 = torch.from_numpy(gg.standard_normal((512, 6)).astype(np.float32))
    r6[0] = 0.0                                 # degenerate: zero vectors
    r6[1] = torch.tensor([1., 1., 0., 0., 0., 0.])  # a1 == a2 (b2 collapses)
    r6[2] = torch.tensor([1., 0., 0., 1., 0., 0.])  # identity
    rm = rot6d_to_rotmat(r6)
    # near-pi and branch-boundary rotations for the quaternion path
    ang = torch.tensor([0.0, 1e-4, 0.5, 1.5707964, 3.1, 3.1415925, 3.1415927, 2.5])
    extra = []
    for a in ang:
        for ax in ([1., 0, 0], [0, 1., 0], [0, 0, 1.], [0.577, 0.577, 0.577], [-0.6, 0.64, 0.48]):
            ax_t = torch.tensor(ax)
            ax_t = ax_t / ax_t.norm()
            K = torch.tensor([[0, -ax_t[2], ax_t[1]], [ax_t[2], 0, -ax_t[0]], [-ax_t[1], ax_t[0], 0]])
            extra.append(torch.eye(3) + torch.sin(a) * K + (1 - torch.cos(a)) * (K @ K))
    rm_all = torch.cat([rm, torch.stack(extra)], 0)
    aa = rotation_matrix_to_angle_axis(rm_all.clone())
    p = os.path.join(ROOT, "tests/golden/geometry.npz")
    np.savez_compressed(p, rot6d=r6.numpy(), rotmat=rm.numpy(), rotmat_all=rm_all.numpy(), aa=aa.numpy())
    print(f"wrote {p}")

    # --- GRU gait encoder standalone (gait_feat_encoder.py:10-104) ---------------------------
    gsd = synth.make_gru_state_dict()
    gru = BidirectionalModel(seqlen=6, input_size=128, num_joints=24, num_outputs=3,
                             estime_phase=True, use_pareFeat=True).eval()
    gspec = netspec.gru_spec()
    assert list(gru.state_dict().keys()) == list(gspec.keys()), set(gru.state_dict()) ^ set(gspec)
    for k, (shape, _) in gspec.items():
        assert tuple(gru.state_dict()[k].shape) == tuple(shape), k
    gru.load_state_dict({k: torch.from_numpy(v) for k, v in gsd.items()}, strict=True)
    outs = {}
    for (b, t) in ((2, 6), (1, 16)):
        x, cp = synth.make_gru_inputs(b, t)
        with torch.no_grad():
            y, ph, xc = gru(torch.from_numpy(x), torch.from_numpy(cp))
        outs[f"y_{b}_{t}"] = y.numpy()
        outs[f"phase_{b}_{t}"] = ph.numpy()
        outs[f"xc_{b}_{t}"] = xc.numpy()
        stat(f"gru y b{b} t{t}", y)
        stat(f"gru phase b{b} t{t}", ph)
    p = os.path.join(ROOT, "tests/golden/gru.npz")
    np.savez_compressed(p, **outs)
    print(f"wrote {p}")

    # --- temporal/spatial attention block standalone (attention_utils.py:219-270), row f2 -----------------
    from lib.models.layers.attention_utils import TSAttnBlock
    tsd = synth.make_tsattn_state_dict()
    blk = TSAttnBlock(use_jwff=True, **netspec.TSATTN).eval()
    tspec = netspec.tsattn_spec()
    assert list(blk.state_dict().keys()) == list(tspec.keys()), set(blk.state_dict()) ^ set(tspec)
    for k, (shape, _) in tspec.items():
        assert tuple(blk.state_dict()[k].shape) == tuple(shape), k
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in tsd.items()}, strict=True)
    outs = {}
    for (b, t) in ((2, 8), (1, 16)):
        x, xs = synth.make_tsattn_inputs(b, t)
        with torch.no_grad():
            y = blk(torch.from_numpy(x), torch.from_numpy(xs))
            a = blk.mulattn(x=torch.from_numpy(x), xs=torch.from_numpy(xs))
        outs[f"y_{b}_{t}"] = y.numpy()
        outs[f"attn_{b}_{t}"] = a.numpy()[:, :, ::8]
        stat(f"tsattn y b{b} t{t}", y)
        stat(f"tsattn mulattn b{b} t{t}", a)
    p = os.path.join(ROOT, "tests/golden/tsattn.npz")
    np.savez_compressed(p, **outs)
    print(f"wrote {p}")

    # --- output-side conversions of the two entry points (demo_utils.py:176-209, kp_utils.py:26-36) -----
    try:
        from lib.data_utils.kp_utils import convert_kps
        from lib.utils.demo_utils import convert_crop_cam_to_orig_img, convert_crop_coords_to_orig_img
        hg = np.random.Generator(np.random.Philox(key=[11, 11]))
        F_ = 7
        j3d = hg.standard_normal((F_, 29, 3)).astype(np.float32)
        bbox = np.stack([hg.uniform(200, 800, F_), hg.uniform(150, 500, F_), hg.uniform(120, 400, F_)], 1).astype(np.float32)
        bbox = np.concatenate([bbox, bbox[:, 2:3]], 1)
        cam = np.stack([hg.uniform(0.6, 1.2, F_), hg.standard_normal(F_) * 0.1, hg.standard_normal(F_) * 0.1], 1).astype(np.float32)
        kp2d = hg.uniform(-1, 1, (F_, 29, 2)).astype(np.float32)
        p = os.path.join(ROOT, "tests/golden/harness.npz")
        np.savez_compressed(p, j3d=j3d, bbox=bbox, cam=cam, kp2d=kp2d,
                            kinectv2=convert_kps(j3d, src="spin2", dst="kinectv2"),
                            orig_cam=convert_crop_cam_to_orig_img(cam=cam, bbox=bbox, img_width=1920, img_height=1080),
                            joints2d_img=convert_crop_coords_to_orig_img(bbox=bbox, keypoints=kp2d.copy(), crop_size=224))
        print(f"wrote {p}")
    except Exception as e:                                   # pragma: no cover
        print("harness goldens skipped:", repr(e))

    # --- One-Euro filter exactly as smooth_pose.py:47-52,84-88 drives it ----------------------------------------
    from lib.utils.one_euro_filter import OneEuroFilter
    og = np.random.Generator(np.random.Philox(key=[13, 13]))
    seq = np.cumsum(og.standard_normal((40, 24, 3)) * 0.05, axis=0).astype(np.float32)
    filt = OneEuroFilter(np.zeros_like(seq[0]), seq[0], min_cutoff=0.004, beta=0.7)
    hat = np.zeros_like(seq)
    hat[0] = seq[0]
    for idx in range(1, seq.shape[0]):
        hat[idx] = filt(np.ones_like(seq[idx]) * idx, seq[idx])
    p = os.path.join(ROOT, "tests/golden/one_euro.npz")
    np.savez_compressed(p, seq=seq, hat=hat)
    print(f"wrote {p}")

    if "--time" in sys.argv:
        x16 = torch.from_numpy(synth.make_frames(16)).reshape(1, 16, 3, 224, 224)
        with torch.no_grad():
            model(x16)
            best = min((lambda t0: (model(x16), time.time() - t0)[1])(time.time()) for _ in range(3))
        print(f"reference CPU path N=16: {best:.3f}s -> {16 / best:.2f} frames/s on {torch.get_num_threads()} threads")


if __name__ == "__main__":
    main()
