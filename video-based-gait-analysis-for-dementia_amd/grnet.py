"""Host-side mirror of the reference's model interface for the per-frame path.

``GRNet`` keeps the constructor signature, ``load_state_dict`` / ``eval`` / ``to`` surface and the
``forward(features, bbox=None, cimg=None, J_regressor=None) -> [dict]`` contract of
``lib/models/grnet.py:25-175`` (output keys per ``lib/models/pare.py:78-84``), but every FLOP runs
in libgrnet_hip.so on the MI355X.  PyTorch is used only as the tensor container (``data_ptr()``),
for the current HIP stream and for reading checkpoints.

Differences from the reference that are deliberate: no ``sys.exit`` on a missing PARE
checkpoint (weights arrive through ``load_state_dict`` / ``load_pare_dict``) and ``torch.load`` uses
``map_location='cpu'``.  ``use_gait_feat=True`` runs the temporal branch of ``grnet.py:154-173`` with the
names the reference's FeatCorrector leaves undefined bound as DESIGN.md records (feature_correction.py:40-62,144);
``featcorr`` must describe the one configuration the class can run in (configs/config_grnet.yaml).
"""
import ctypes as C
import logging
from collections import namedtuple

import numpy as np
import torch

from . import _lib, netspec

logger = logging.getLogger(__name__)
_IncompatibleKeys = namedtuple("_IncompatibleKeys", ["missing_keys", "unexpected_keys"])

SMPL_PREFIX = "regressor.smpl.smpl."
_SMPL_KEYS = ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "parents", "J_regressor_extra")
_TOLERATED_SMPL_KEYS = ("faces_tensor", "vertex_joint_selector.extra_joints_idxs", "betas", "global_orient",
                        "body_pose", "transl")


def _np32(t):
    if torch.is_tensor(t):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(t), dtype=np.float32)


class GRNet:
    is_demo = False

    def __init__(self, num_joints=24, num_input_features=480, num_features_pare=128, num_features_smpl=64,
                 backbone='hrnet_w32', focal_length=5000., img_res=224, pretrained_pare=None, writer=None, seqlen=50,
                 pretrained_hrnet=None, use_gait_feat=False, featcorr=None, use_pose_encoder=False,
                 use_shpcam_encoder=False, max_frames=64, device_id=0, dtype="f32"):
        if (num_joints, num_input_features, num_features_pare, num_features_smpl) != (24, 480, 128, 64) \
                or backbone != 'hrnet_w32' or focal_length != 5000. or img_res != 224:
            raise ValueError("the HIP path implements the reference's fixed configuration "
                             "(24 joints, hrnet_w32 -> 480 features, PARE 128/64, f=5000, 224 px)")
        self.use_gait_feat = bool(use_gait_feat)
        if use_gait_feat and featcorr is not None:
            want = dict(AVG_DIM=3, ESTIM_PHASE=True, NUM_LAYERS=1, H_SIZE=1024, NUM_HEADS=4, USE_JWFF=True)
            get = (lambda k: featcorr[k]) if isinstance(featcorr, dict) else (lambda k: getattr(featcorr, k))
            got = {k: get(k) for k in want}
            if got != want:
                raise ValueError(f"MODEL.FEAT_CORR {got}: the HIP path implements configs/config_grnet.yaml's {want} -- the only "
                                 "configuration in which FeatCorrector's reshapes are consistent (feature_correction.py:92,144)")
        self._lib = _lib.load()
        self.max_frames = int(max_frames)
        self.device = torch.device("cuda", device_id)
        h = C.c_void_p()
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' (the reference's precision) or 'bf16' (bf16 storage, fp32 accumulation; BASELINE configs 3/5)")
        self.dtype = dtype
        rc = self._lib.grnet_create(C.byref(h), device_id, 1 if dtype == "bf16" else 0, self.max_frames)
        if rc != 0:
            raise _lib.GrnetError(f"grnet_create failed with code {rc} (is a GPU visible?)")
        self._h = h
        self._finalized = False
        self._smpl_loaded = False
        self._loaded = set()
        self.seqlen = seqlen
        self.training = False
        self.focal_length = focal_length
        if pretrained_pare:
            self.load_pare_dict(pretrained_pare)

    # ------------------------------------------------------------------ weights
    def _load_tensor(self, key, value):
        if torch.is_tensor(value):
            value = value.detach().cpu().numpy()
        value = np.asarray(value)
        if value.dtype.kind in "iu":
            dtype, arr = _lib.DTYPE_I64, np.ascontiguousarray(value, dtype=np.int64)
        else:
            dtype, arr = _lib.DTYPE_F32, np.ascontiguousarray(value, dtype=np.float32)
        shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
        rc = self._lib.grnet_load_tensor(self._h, key.encode(), arr.ctypes.data_as(C.c_void_p), shape, arr.ndim, dtype)
        _lib.check(self._lib, self._h, rc, f"grnet_load_tensor({key})")
        self._loaded.add(key)

    def load_smpl(self, tables):
        """SMPL buffers (smplx.SMPL tables + J_regressor_extra, lib/models/smpl.py:97-106)."""
        arrs = [_np32(tables[k]) for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights")]
        parents = np.ascontiguousarray(np.asarray(tables["parents"]).astype(np.int32))
        parents[0] = -1
        extra = _np32(tables["J_regressor_extra"])
        want = [netspec.SMPL_TABLE_SHAPES[k] for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights")]
        for a, w, k in zip(arrs, want, _SMPL_KEYS):
            if int(np.prod(a.shape)) != int(np.prod(w)):
                raise ValueError(f"SMPL table {k} has shape {a.shape}, expected {w}")
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = self._lib.grnet_load_smpl(self._h, ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), ptr(arrs[3]), ptr(arrs[4]),
                                       ptr(parents), ptr(extra))
        _lib.check(self._lib, self._h, rc, "grnet_load_smpl")
        self._smpl_loaded = True

    def load_state_dict(self, state_dict, strict=True):
        """Reference key names (demo.py:116-122; batch_generation.py:214-218)."""
        if self._finalized:
            raise RuntimeError("weights are already finalized on the device")
        spec = netspec.grnet_spec()
        unexpected, smpl = [], {}
        for k, v in state_dict.items():
            if k.startswith(SMPL_PREFIX):
                name = k[len(SMPL_PREFIX):]
                if name in _SMPL_KEYS:
                    smpl[name] = v
                elif name not in _TOLERATED_SMPL_KEYS:
                    unexpected.append(k)
            elif k in spec or k.startswith(("pfeat_corrector.", "gru.", "tsattn.")):
                want = spec.get(k)
                if want is not None and tuple(np.shape(v)) != tuple(want[0]):
                    raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(np.shape(v))} vs model {tuple(want[0])}")
                self._load_tensor(k, v)
            else:
                unexpected.append(k)
        if len(smpl) == len(_SMPL_KEYS):
            self.load_smpl(smpl)
        missing = [k for k in spec if k not in self._loaded]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for GRNet: missing {missing[:5]}{'...' if len(missing) > 5 else ''}"
                               f" unexpected {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
        return _IncompatibleKeys(missing, unexpected)

    def load_pare_dict(self, pretrained_pare):
        """PARE head weights re-keyed from 'model.head.*' (grnet.py:93-109; utils.py:185-196)."""
        ckpt = torch.load(pretrained_pare, map_location="cpu")["state_dict"]
        if "model.head.init_pose" not in ckpt or "model.head.init_shape" not in ckpt:
            raise KeyError(f"Checkpoint at {pretrained_pare} does not match VPARE implementation.")
        sd = {"head." + k[len("model.head."):]: v for k, v in ckpt.items() if k.startswith("model.head.")}
        return self.load_state_dict(sd, strict=False)

    def finalize(self):
        if not self._finalized:
            if not self._smpl_loaded:
                raise RuntimeError("SMPL tables were not loaded (regressor.smpl.smpl.* keys or load_smpl())")
            _lib.check(self._lib, self._h, self._lib.grnet_finalize_weights(self._h), "grnet_finalize_weights")
            self._finalized = True
        return self

    # torch.nn.Module surface used by the entry points
    def to(self, device=None):
        return self

    def eval(self):
        return self

    def set_option(self, option, value):
        _lib.check(self._lib, self._h, self._lib.grnet_set_option(self._h, option, value), "grnet_set_option")

    # ------------------------------------------------------------------ forward
    def forward(self, features, bbox=None, cimg=None, J_regressor=None, extras=()):
        if features.dim() == 5:
            batch_size, seqlen, nc, h, w = features.shape
            features = features.reshape(-1, nc, h, w)
        elif features.dim() == 4:
            batch_size = 1
            seqlen, nc, h, w = features.shape
        else:
            raise ValueError(f"Wrong feature dimension: {features.dim()}.")
        if (nc, h, w) != (3, 224, 224):
            raise ValueError(f"expected frames of shape (3,224,224), got {(nc, h, w)}")
        if J_regressor is not None:
            raise NotImplementedError("J_regressor override (evaluation-only, pare.py:70-76) is outside the inference path")
        if self.use_gait_feat:
            assert (bbox is not None) and (cimg is not None)                     # grnet.py:133
            if bbox.dim() == 2:
                bbox = bbox.unsqueeze(0)
            if cimg.dim() == 2:
                cimg = cimg.unsqueeze(0)
            extras = tuple(extras) + tuple(k for k in ("point_local_feat", "cam_shape_feats") if k not in extras)
        if not features.is_cuda:
            raise RuntimeError("frames must live in HBM (features.to('cuda')); the HIP path has no CPU fallback")
        if features.device != self.device:
            raise RuntimeError(f"frames live on {features.device} but this model's handle is bound to {self.device}")
        self.finalize()
        x = features.to(torch.float32).contiguous()
        n = x.shape[0]
        dev = x.device
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        out = {"theta": new(n, 85), "verts": new(n, 6890, 3), "kp_2d": new(n, 29, 2), "kp_3d": new(n, 29, 3),
               "rotmat": new(n, 24, 3, 3)}
        shapes = {"point_local_feat": (128, 24), "cam_shape_feats": (64, 24), "pred_rot6d": (24, 6),
                  "features": (480, 56, 56), "part_attn": (25, 56, 56), "smpl_feats": (128, 56, 56)}
        for k in extras:
            out[k] = new(n, *shapes[k])
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for s in range(0, n, self.max_frames):
            m = min(self.max_frames, n - s)
            o = _lib.Outputs()
            for k, t in out.items():
                setattr(o, k, t[s:s + m].data_ptr())
            rc = self._lib.grnet_forward(self._h, C.c_void_p(x[s:s + m].data_ptr()), m, C.byref(o), stream)
            _lib.check(self._lib, self._h, rc, "grnet_forward")
        res = {"theta": out["theta"].reshape(batch_size, seqlen, 85),
               "verts": out["verts"].reshape(batch_size, seqlen, 6890, 3),
               "kp_2d": out["kp_2d"].reshape(batch_size, seqlen, 29, 2),
               "kp_3d": out["kp_3d"].reshape(batch_size, seqlen, 29, 3),
               "rotmat": out["rotmat"].reshape(batch_size, seqlen, 24, 3, 3)}
        for k in extras:
            res[k] = out[k]
        if self.use_gait_feat:                                 # grnet.py:154-173: FeatCorrector, second head pass, regressor
            g = self.gait_correct(out["point_local_feat"], out["cam_shape_feats"], out["theta"], bbox, cimg, batch_size, seqlen)
            for k in ("theta", "verts", "kp_2d", "kp_3d", "rotmat"):
                res[k] = g[k].reshape(res[k].shape)
            res["pred_avg"], res["pred_phase"] = g["pred_avg"], g["pred_phase"]
            res["pred_cparam"] = g["pred_cparam"]
            res["point_local_feat"] = g["point_local_feat"]
        return [res]

    __call__ = forward

    def gait_correct(self, point_local_feat, cam_shape_feats, theta_or_cam, bbox, cimg, b, t):
        """grnet.py:154-173 on the first pass's results for whole clips: (b*t,128,24), (b*t,64,24), theta (b*t,85) or pred_cam
        (b*t,3), bbox (b,t,4), cimg (b,t,2) -> dict(theta, verts, kp_2d, kp_3d, rotmat, pred_avg (b,3), pred_phase (b,t,4),
        pred_cparam (b*t,3), point_local_feat (b*t,128,24) corrected)."""
        self.finalize()
        dev, m = self.device, b * t
        f = lambda x, *shape: x.to(dev, torch.float32).reshape(*shape).contiguous()
        plf, csf = f(point_local_feat, m, 128, 24), f(cam_shape_feats, m, 64, 24)
        cam = f(theta_or_cam, m, -1)
        bb, ci = f(bbox, m, 4), f(cimg, m, 2)
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        out = {"theta": new(m, 85), "verts": new(m, 6890, 3), "kp_2d": new(m, 29, 2), "kp_3d": new(m, 29, 3), "rotmat": new(m, 24, 3, 3)}
        gait = {"pred_avg": new(b, 3), "pred_phase": new(b, t, 4), "pred_cparam": new(m, 3), "point_local_feat": new(m, 128, 24)}
        o, g = _lib.Outputs(), _lib.GaitOutputs()
        for k, v in out.items():
            setattr(o, k, v.data_ptr())
        for k, v in gait.items():
            setattr(g, k, v.data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = self._lib.grnet_gait_correct(self._h, plf.data_ptr(), csf.data_ptr(), cam.data_ptr(), cam.shape[1], bb.data_ptr(), ci.data_ptr(),
                                          b, t, C.byref(o), C.byref(g), stream)
        _lib.check(self._lib, self._h, rc, "grnet_gait_correct")
        out.update(gait)
        return out

    def tune(self, n_frames, level=1, cache=None):
        """Measure-and-pick launch configurations for calls of ``n_frames`` frames (see grnet_tune).

        ``cache``: path of a text file holding a previously measured table; applied if present, written after tuning."""
        self.finalize()
        import os
        if cache and os.path.isfile(cache):
            with open(cache) as f:
                rc = self._lib.grnet_set_tuning(self._h, int(n_frames), f.read().encode())
            if rc == 0:
                return self
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib, self._h, self._lib.grnet_tune(self._h, int(n_frames), stream, int(level)), "grnet_tune")
        if cache:
            buf = C.create_string_buffer(1 << 16)
            if self._lib.grnet_get_tuning(self._h, int(n_frames), buf, len(buf)) > 0:
                os.makedirs(os.path.dirname(os.path.abspath(cache)), exist_ok=True)
                with open(cache, "w") as f:
                    f.write(buf.value.decode())
        return self

    def tuned_mode(self, n_frames):
        """Schedule chosen by tune(): dict(measured_table, eager) or None if not tuned for n_frames."""
        buf = C.create_string_buffer(1 << 16)
        if self._lib.grnet_get_tuning(self._h, int(n_frames), buf, len(buf)) <= 0:
            return None
        mode = int(buf.value.decode().split("\n", 1)[0].split()[1])
        return {"measured_table": bool(mode & 1), "eager": bool(mode & 4)}

    # ------------------------------------------------------------------ introspection (bench / tests)
    def num_kernel_launches(self):
        return self._lib.grnet_num_kernel_launches(self._h)

    def num_conv_launches(self):
        return self._lib.grnet_num_conv_launches(self._h)

    def conv_flops_per_frame(self):
        return self._lib.grnet_conv_flops_per_frame(self._h)

    def conv_executed_flops_per_frame(self, n_frames=None):
        """Winograd F(4x4,3x3) layers counted at the 1/4 of their multiplies they execute (x the tile padding on 14x14 / 7x7 maps), for a
        call of n_frames frames (default: the handle's latest forward).  Reporting only."""
        self.finalize()
        if n_frames is None:
            return self._lib.grnet_conv_executed_flops_per_frame(self._h)
        return self._lib.grnet_conv_executed_flops_per_frame_n(self._h, int(n_frames))

    def kernel_table(self, n_frames, reps=20):
        """Per kernel (family<shape>) of the conv-class launches of one forward of n_frames frames: launches, time of each distinct layer
        shape measured ALONE (grnet_time_conv: HIP events around `reps` back-to-back launches), algorithmic and executed FLOPs.
        Returns a list of dicts sorted by total time: name, launches, total_us, avg_us, gflop (algorithmic, per step), executed_gflop."""
        self.finalize()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        convs = self.describe_convs()
        timed, rows = {}, {}
        for pos, c in enumerate(convs):
            name, ex = C.create_string_buffer(96), C.c_double()
            _lib.check(self._lib, self._h, self._lib.grnet_conv_kernel_info(self._h, pos, int(n_frames), name, 96, C.byref(ex)), "grnet_conv_kernel_info")
            key = (name.value, c["cin"], c["cout"], c["ks"], c["stride"], c["hin"], c["n_add"], c["macs"])
            if key not in timed:
                us = C.c_float()
                _lib.check(self._lib, self._h, self._lib.grnet_time_conv(self._h, pos, int(n_frames), int(reps), stream, C.byref(us)), "grnet_time_conv")
                timed[key] = us.value
            member = name.value.endswith(b"+")              # a convolution inside a conv_bf16_chain launch: its FLOPs count, it has no launch of its own
            row = name.value.decode().rstrip("+")
            r = rows.setdefault(row, {"name": row, "launches": 0, "total_us": 0.0, "gflop": 0.0, "executed_gflop": 0.0})
            r["launches"] += 0 if member else 1
            r["total_us"] += timed[key]
            r["gflop"] += 2.0 * c["macs"] * n_frames / 1e9
            r["executed_gflop"] += 2.0 * ex.value * n_frames / 1e9
        out = sorted(rows.values(), key=lambda r: -r["total_us"])
        for r in out:
            r["avg_us"] = r["total_us"] / r["launches"]
        return out

    def describe_convs(self):
        """The convolution launches of one forward in launch order: list of dicts (shape, fused addends, weight key, MACs per frame).
        An entry with cin == 0 is the grouped launch of an HR module's 1x1 fuse terms (csrc/hr_fuse.hip)."""
        self.finalize()
        keys = ("cin", "cout", "ks", "stride", "hin", "win", "hout", "wout", "n_add", "relu", "lane", "add_elems")
        out = []
        for pos in range(self.num_conv_launches()):
            info, name = (C.c_int32 * 12)(), C.create_string_buffer(160)
            _lib.check(self._lib, self._h, self._lib.grnet_describe_conv(self._h, pos, info, name, 160), "grnet_describe_conv")
            d = dict(zip(keys, list(info)))
            d["name"] = name.value.decode()
            d["macs"] = int(self._lib.grnet_describe_conv_macs(self._h, pos))
            out.append(d)
        return out

    def op_timeline(self, frames):
        """Diagnostic (grnet_op_timeline): [(index, lane, start_us, end_us, label)] of one eager forward on `frames` (n,3,224,224)."""
        self.finalize()
        x = frames.to(self.device, torch.float32).contiguous()
        buf = C.create_string_buffer(1 << 18)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = self._lib.grnet_op_timeline(self._h, x.data_ptr(), x.shape[0], stream, buf, len(buf))
        if rc < 0:
            _lib.check(self._lib, self._h, rc, "grnet_op_timeline")
        rows = []
        for line in buf.value.decode().splitlines():
            i, lane, a, b, label = line.split(" ", 4)
            rows.append((int(i), int(lane), float(a), float(b), label))
        return rows

    def time_convs(self, n_frames):
        ms = C.c_float()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib, self._h, self._lib.grnet_time_convs(self._h, n_frames, stream, C.byref(ms)), "grnet_time_convs")
        return ms.value

    def smpl_forward(self, betas, rotmat, cam=None):
        """SMPL LBS on the GPU: betas (n,10), rotmat (n,24,3,3) [, cam (n,3)] -> verts, kp_3d (29 spin2 joints) [, kp_2d]."""
        self.finalize()
        n = betas.shape[0]
        dev = self.device
        b = betas.to(dev, torch.float32).contiguous()
        r = rotmat.to(dev, torch.float32).reshape(n, 24, 9).contiguous()
        c = cam.to(dev, torch.float32).contiguous() if cam is not None else None
        verts = torch.empty(n, 6890, 3, dtype=torch.float32, device=dev)
        kp3d = torch.empty(n, 29, 3, dtype=torch.float32, device=dev)
        kp2d = torch.empty(n, 29, 2, dtype=torch.float32, device=dev) if cam is not None else None
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for s0 in range(0, n, self.max_frames):
            m = min(self.max_frames, n - s0)
            rc = self._lib.grnet_smpl_forward(self._h, b[s0:].data_ptr(), r[s0:].data_ptr(), c[s0:].data_ptr() if c is not None else None,
                                              m, verts[s0:].data_ptr(), kp3d[s0:].data_ptr(),
                                              kp2d[s0:].data_ptr() if kp2d is not None else None, stream)
            _lib.check(self._lib, self._h, rc, "grnet_smpl_forward")
        return verts, kp3d, kp2d

    def crop_normalise(self, images, bboxes, scale=1.0, bgr=False, mode="cv2"):
        """uint8 frames (n,H,W,3) [or one (H,W,3) frame] + boxes (n,4) -> (n,3,224,224) normalised crops, on the GPU.
        mode "cv2" (default): OpenCV's fixed-point warpAffine arithmetic (grnet_crop_normalise_cv_maps; the maps are computed on the
        host as the reference's gen_trans_from_patch_cv / getAffineTransform / warpAffine do -- pipeline.cv_crop_maps; a box with
        w != h takes the reference's two-warp, aspect-preserving branch, img_utils.py:97-106);
        mode "ideal": exact bilinear sampling in float (grnet_crop_normalise), kept for A/B."""
        shared = images.dim() == 3
        if images.dtype != torch.uint8 or images.shape[-1] != 3 or not images.is_cuda:
            raise ValueError("images must be a uint8 CUDA tensor (n,H,W,3) or (H,W,3)")
        images = images.contiguous()
        bb_host = bboxes.detach().cpu().numpy() if torch.is_tensor(bboxes) else np.asarray(bboxes)
        n = bb_host.shape[0]
        if bb_host.shape != (n, 4) or (not shared and images.shape[0] != n):
            raise ValueError("bboxes must be (n,4) and match the number of frames")
        hgt, wid = images.shape[-3], images.shape[-2]
        out = torch.empty(n, 3, 224, 224, dtype=torch.float32, device=images.device)
        stream = C.c_void_p(torch.cuda.current_stream(images.device).cuda_stream)
        if mode == "cv2":
            from .pipeline import cv_crop_maps
            maps = torch.from_numpy(cv_crop_maps(bb_host, scale)).to(images.device, non_blocking=True)
            rc = self._lib.grnet_crop_normalise_cv_maps(self._h, images.data_ptr(), n, hgt, wid, int(shared), maps.data_ptr(), int(bgr),
                                                        out.data_ptr(), stream)
            _lib.check(self._lib, self._h, rc, "grnet_crop_normalise_cv_maps")
            return out
        if mode != "ideal":
            raise ValueError("mode must be 'cv2' or 'ideal'")
        bb = torch.as_tensor(bb_host).to(device=images.device, dtype=torch.float32).contiguous()
        rc = self._lib.grnet_crop_normalise(self._h, images.data_ptr(), n, hgt, wid, int(shared), bb.data_ptr(), float(scale),
                                            int(bgr), out.data_ptr(), stream)
        _lib.check(self._lib, self._h, rc, "grnet_crop_normalise")
        return out

    def debug_tensor(self, name, n_frames):
        """Named intermediate of the last forward as an (n,C,H,W) tensor (see grnet_debug_tensor)."""
        shp = (C.c_int64 * 3)()
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib, self._h, self._lib.grnet_debug_tensor(self._h, name.encode(), n_frames, None, shp, stream),
                   "grnet_debug_tensor")
        out = torch.empty(n_frames, shp[0], shp[1], shp[2], dtype=torch.float32, device=self.device)
        _lib.check(self._lib, self._h, self._lib.grnet_debug_tensor(self._h, name.encode(), n_frames, out.data_ptr(), shp,
                                                                     stream), "grnet_debug_tensor")
        return out

    def gru_forward(self, x, cparams):
        """BidirectionalModel.forward on this handle's GRU weights (keys gru.* / pfeat_corrector.featnet.*)."""
        self.finalize()
        b, t, f = x.shape
        if f != 3072 or tuple(cparams.shape) != (b, t, 3):
            raise ValueError("x must be (b,T,3072) and cparams (b,T,3)")
        x = x.to(torch.float32).contiguous()
        cp = cparams.to(torch.float32).contiguous()
        y = torch.empty(b, 3, dtype=torch.float32, device=x.device)
        ph = torch.empty(b, t, 4, dtype=torch.float32, device=x.device)
        xc = torch.empty(b, t, 3072, dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        rc = self._lib.grnet_gru_forward(self._h, x.data_ptr(), cp.data_ptr(), b, t, y.data_ptr(), ph.data_ptr(),
                                         xc.data_ptr(), stream)
        _lib.check(self._lib, self._h, rc, "grnet_gru_forward")
        return y, ph, xc

    def tsattn_forward(self, x, xs):
        """TSAttnBlock.forward (attention_utils.py:261-270) on this handle's attention-block weights
        (keys tsattn.* / pfeat_corrector.featTencoder.0.*): x (b,n,128,24), xs (b,n,128,25) -> (b,n,3072)."""
        self.finalize()
        if x.dim() != 4 or tuple(x.shape[2:]) != (128, 24) or tuple(xs.shape) != (x.shape[0], x.shape[1], 128, 25):
            raise ValueError("x must be (b,n,128,24) and xs (b,n,128,25)")
        b, n = x.shape[:2]
        x = x.to(torch.float32).contiguous()
        xs = xs.to(torch.float32).contiguous()
        y = torch.empty(b, n, 3072, dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        rc = self._lib.grnet_tsattn_forward(self._h, x.data_ptr(), xs.data_ptr(), b, n, y.data_ptr(), stream)
        _lib.check(self._lib, self._h, rc, "grnet_tsattn_forward")
        return y

    def head_forward(self, point_local_feat, cam_shape_feats):
        """PareHead.forward + VPRegressor.forward from given pooled features (pare.py:271-303,52-91): (n,128,24), (n,64,24) ->
        dict(theta (n,85), verts, kp_2d, kp_3d, rotmat, pred_rot6d).  The second head pass of the use_gait_feat branch."""
        self.finalize()
        n = point_local_feat.shape[0]
        if tuple(point_local_feat.shape) != (n, 128, 24) or tuple(cam_shape_feats.shape) != (n, 64, 24):
            raise ValueError("point_local_feat must be (n,128,24) and cam_shape_feats (n,64,24)")
        dev = self.device
        plf = point_local_feat.to(dev, torch.float32).contiguous()
        csf = cam_shape_feats.to(dev, torch.float32).contiguous()
        new = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        out = {"theta": new(n, 85), "verts": new(n, 6890, 3), "kp_2d": new(n, 29, 2), "kp_3d": new(n, 29, 3),
               "rotmat": new(n, 24, 3, 3), "pred_rot6d": new(n, 24, 6)}
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for s0 in range(0, n, self.max_frames):
            m = min(self.max_frames, n - s0)
            o = _lib.Outputs()
            for k, t in out.items():
                setattr(o, k, t[s0:s0 + m].data_ptr())
            rc = self._lib.grnet_head_forward(self._h, plf[s0:].data_ptr(), csf[s0:].data_ptr(), m, C.byref(o), stream)
            _lib.check(self._lib, self._h, rc, "grnet_head_forward")
        return out

    def op_rot6d_to_rotmat(self, x):
        x = x.to(self.device, torch.float32).reshape(-1, 6).contiguous()
        out = torch.empty(x.shape[0], 3, 3, dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib, self._h, self._lib.grnet_op_rot6d_to_rotmat(self._h, x.data_ptr(), x.shape[0], out.data_ptr(), stream),
                   "grnet_op_rot6d_to_rotmat")
        return out

    def op_rotmat_to_aa(self, R):
        R = R.to(self.device, torch.float32).reshape(-1, 9).contiguous()
        out = torch.empty(R.shape[0], 3, dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self._lib, self._h, self._lib.grnet_op_rotmat_to_aa(self._h, R.data_ptr(), R.shape[0], out.data_ptr(), stream),
                   "grnet_op_rotmat_to_aa")
        return out

    # single-op hooks for kernel parity tests
    def op_conv2d(self, x, w, bias=None, stride=1, relu=False, add=None, tile_hint=0):
        n, cin, h, wd = x.shape
        cout, _, ks, _ = w.shape
        pad = ks // 2
        ho, wo = (h + 2 * pad - ks) // stride + 1, (wd + 2 * pad - ks) // stride + 1
        out = torch.empty(n, cout, ho, wo, dtype=torch.float32, device=x.device)
        wn = _np32(w)
        bn = _np32(bias) if bias is not None else None
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        rc = self._lib.grnet_op_conv2d(self._h, x.data_ptr(), n, cin, h, wd, wn.ctypes.data_as(C.c_void_p),
                                       bn.ctypes.data_as(C.c_void_p) if bn is not None else None, cout, ks, stride,
                                       int(relu), add.data_ptr() if add is not None else None, out.data_ptr(),
                                       tile_hint, stream)
        _lib.check(self._lib, self._h, rc, "grnet_op_conv2d")
        return out

    def op_conv_chain(self, x, ws, bs, reps=0):
        """bf16 handles: ONE conv_bf16_chain launch over the BasicBlock chain ws = [(c,c,3,3)] * nconv, bs = [(c,)] * nconv on x (n,c,w,w) f32.
        Returns the output (n,c,w,w) f32 (bf16 values), or (output, us per launch) with reps > 0."""
        n, c, h, wd = x.shape
        out = torch.empty_like(x, dtype=torch.float32)
        wn = np.ascontiguousarray(np.stack([_np32(w) for w in ws]))
        bn = np.ascontiguousarray(np.stack([_np32(b) for b in bs]))
        us = C.c_float(0.0)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        rc = self._lib.grnet_op_conv_chain(self._h, x.data_ptr(), n, c, wd, len(ws), wn.ctypes.data_as(C.c_void_p), bn.ctypes.data_as(C.c_void_p),
                                           out.data_ptr(), int(reps), C.byref(us), stream)
        _lib.check(self._lib, self._h, rc, "grnet_op_conv_chain")
        return (out, us.value) if reps else out

    def op_bilinear2x(self, x):
        n, c, h, w = x.shape
        out = torch.empty(n, c, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        rc = self._lib.grnet_op_bilinear2x(self._h, x.data_ptr(), n, c, h, w, out.data_ptr(), stream)
        _lib.check(self._lib, self._h, rc, "grnet_op_bilinear2x")
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.grnet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_synthetic_model(max_frames=64, device_id=0, with_gru=True, with_tsattn=False, dtype="f32", use_gait_feat=False):
    """GRNet with the seed-defined weights / SMPL tables of synth.py (no checkpoint exists offline).  use_gait_feat: the whole
    pose-feature corrector under its checkpoint keys (pfeat_corrector.*: GRU, gait-token MLPs, BatchNorm1d, attention block)."""
    from . import synth
    m = GRNet(max_frames=max_frames, device_id=device_id, dtype=dtype, use_gait_feat=use_gait_feat,
              featcorr=dict(AVG_DIM=3, ESTIM_PHASE=True, NUM_LAYERS=1, H_SIZE=1024, NUM_HEADS=4, USE_JWFF=True) if use_gait_feat else None)
    sd = synth.make_state_dict()
    if use_gait_feat:
        sd.update(synth.make_featcorr_state_dict())
        with_gru = with_tsattn = False
    if with_gru:
        sd.update({"gru." + k: v for k, v in synth.make_gru_state_dict().items()})
    if with_tsattn:
        sd.update({"tsattn." + k: v for k, v in synth.make_tsattn_state_dict().items()})
    m.load_state_dict(sd, strict=True)
    m.load_smpl(synth.make_smpl_tables())
    return m.finalize()
