"""Host pipeline around the per-frame path: the model loops of the two entry points and their
output formats, restated for the HIP model (numpy on the host, no kernels here).

Reference counterparts:
  * demo model loop + result dict     demo.py:126-231  -> run_tracklet(), make_demo_result()
  * batch 3D-joint generation         batch_generation.py:289-371, 222-284 -> run_on_frames(), BatchDb
  * crop-cam / crop-coords -> image   lib/utils/demo_utils.py:176-209
  * spin2 -> kinectv2 joints          lib/data_utils/kp_utils.py:26-36 with the tables :211-242, :904-931
  * crop + normalise of a frame       lib/dataset/inference.py:71-87, lib/data_utils/img_utils.py:252-285,355-363
    (SURVEY 8f-1 "next": OpenCV's warpAffine is third-party and absent offline -- parity UNPINNED; the
    PIL bilinear crop below follows the same geometry: box centre/size * scale -> 224x224, border 0)
"""
import os
import os.path as osp
from collections import defaultdict

import numpy as np
import torch

from . import netspec

IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32)
MAX_VID = 50          # batch_generation.py:36
MAX_SEQLEN = 400      # batch_generation.py:37 (MAX_seqlen)


# ----------------------------------------------------------------------------- output conversions
def convert_crop_cam_to_orig_img(cam, bbox, img_width, img_height):
    """(n,3) weak-perspective crop camera [s,tx,ty] -> (n,4) [sx,sy,tx,ty] in the full image (demo_utils.py:176-193)."""
    cx, cy, h = bbox[:, 0], bbox[:, 1], bbox[:, 2]
    hw, hh = img_width / 2.0, img_height / 2.0
    sx = cam[:, 0] * (1.0 / (img_width / h))
    sy = cam[:, 0] * (1.0 / (img_height / h))
    tx = ((cx - hw) / hw / sx) + cam[:, 1]
    ty = ((cy - hh) / hh / sy) + cam[:, 2]
    return np.stack([sx, sy, tx, ty]).T


def convert_crop_coords_to_orig_img(bbox, keypoints, crop_size=224):
    """normalised crop keypoints (n,J,2) in [-1,1] -> image pixels (demo_utils.py:196-209)."""
    cx, cy, h = bbox[:, 0], bbox[:, 1], bbox[:, 2]
    kp = 0.5 * crop_size * (np.asarray(keypoints, np.float32) + 1.0)
    kp = kp * (h[..., None, None] / crop_size)
    kp[:, :, 0] = (cx - h / 2)[..., None] + kp[:, :, 0]
    kp[:, :, 1] = (cy - h / 2)[..., None] + kp[:, :, 1]
    return kp


_KPS = None


def _kps_tables():
    global _KPS
    if _KPS is None:
        import json
        with open(osp.join(osp.dirname(osp.abspath(__file__)), "kps_tables.json")) as f:
            _KPS = json.load(f)
    return _KPS


def convert_kps(joints, src, dst):
    """convert_kps (kp_utils.py:26-36): (n,J_src,3) joints of skeleton ``src`` -> (n,J_dst,3) float64 in skeleton ``dst``; joints
    ``dst`` names that ``src`` lacks stay 0.  ``src`` is 'spin' (49) or 'spin2' (29) -- the two layouts the path emits; the index
    tables are derived by running the reference's joint-name functions (tools/make_kps_tables.py).  An unknown ``dst`` raises
    NameError, as the reference's eval() does (demo.py:227 catches exactly that)."""
    t = _kps_tables()
    if src not in ("spin", "spin2"):
        raise NameError(f"name 'get_{src}_joint_names' is not defined")
    if dst not in t["sizes"]:
        raise NameError(f"name 'get_{dst}_joint_names' is not defined")
    joints = np.asarray(joints)
    idx = t["from_" + src][dst]
    out = np.zeros((joints.shape[0], len(idx), 3))
    for k, i in enumerate(idx):
        if i >= 0:
            out[:, k] = joints[:, i]                          # IndexError / shape error on a wrong-sized input, like the reference
    return out


def spin2_to_kinectv2(joints):
    """(n,29,3) spin2 joints -> (n,25,3) kinectv2 joints (convert_kps(src='spin2', dst='kinectv2'))."""
    joints = np.asarray(joints)
    return joints[:, netspec.SPIN2_TO_KINECTV2].astype(np.float64)      # the reference returns float64 zeros-based arrays


# ----------------------------------------------------------------------------- --smooth (row f3)
def one_euro_filter(x, min_cutoff=0.004, beta=0.7, d_cutoff=1.0):
    """The One-Euro filter as smooth_pose drives it (one_euro_filter.py:5-46, smooth_pose.py:47-52,84-88): unit time
    steps, state initialised with x[0] and dx = 0; x (T, ...) -> filtered (T, ...), element-wise."""
    x = np.asarray(x)
    out = np.zeros_like(x)
    out[0] = x[0]
    x_prev, dx_prev = x[0], np.zeros_like(x[0])

    def alpha(cutoff):                                       # smoothing_factor with t_e = 1
        r = 2 * np.pi * cutoff
        return r / (r + 1)

    a_d = alpha(d_cutoff)
    for t in range(1, x.shape[0]):
        dx = x[t] - x_prev
        dx_hat = a_d * dx + (1 - a_d) * dx_prev
        a = alpha(min_cutoff + beta * np.abs(dx_hat))
        x_hat = a * x[t] + (1 - a) * x_prev
        out[t] = x_hat
        x_prev, dx_prev = x_hat, dx_hat
    return out


def rodrigues(aa):
    """Axis-angle (n,3) -> rotation matrices (n,3,3): smplx's batch_rodrigues (angle = |aa + 1e-8|, R = I + sin K +
    (1 - cos) K^2), the conversion smplx.SMPL applies when smooth_pose passes axis-angle poses.  Third-party: unpinned."""
    aa = np.asarray(aa, np.float32)
    angle = np.linalg.norm(aa + np.float32(1e-8), axis=1, keepdims=True)
    d = aa / angle
    K = np.zeros((aa.shape[0], 3, 3), np.float32)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -d[:, 2], d[:, 1], d[:, 2], -d[:, 0], -d[:, 1], d[:, 0]
    s, c = np.sin(angle)[..., None], np.cos(angle)[..., None]
    return (np.eye(3, dtype=np.float32)[None] + s * K + (1 - c) * (K @ K)).astype(np.float32)


def smooth_pose(model, pred_pose, pred_betas, min_cutoff=0.004, beta=0.7, kinectv2=False, smpl_tables=None):
    """lib/utils/smooth_pose.py:28-116: One-Euro filter over the axis-angle pose, then SMPL re-evaluated per frame with
    the betas of frame 0 (smooth_pose.py:97) -- batched into one LBS launch here.  Returns (verts, pose_hat, joints3d)
    with joints3d in the 49-joint SPIN order (kinectv2=False, what demo.py gets) or 25 kinectv2 joints."""
    T = pred_betas.shape[0]
    pose = np.asarray(pred_pose, np.float32).reshape(T, 24, 3)
    pose_hat = one_euro_filter(pose, min_cutoff, beta).astype(np.float32)
    rot = rodrigues(pose_hat.reshape(-1, 3)).reshape(T, 24, 3, 3)
    betas0 = np.repeat(np.asarray(pred_betas, np.float32)[:1], T, axis=0)
    verts, kp29, _ = model.smpl_forward(torch.from_numpy(betas0), torch.from_numpy(rot))
    verts, kp29 = verts.cpu().numpy(), kp29.cpu().numpy()
    if kinectv2:
        joints = spin2_to_kinectv2(kp29)
    else:
        if smpl_tables is None:
            raise ValueError("the 49-joint SPIN output needs J_regressor_extra (smpl_tables)")
        j45 = np.concatenate([kp29[:, :24], verts[:, netspec.SMPL_EXTRA_VERT_IDS]], 1)
        extra = np.einsum("jv,nvk->njk", np.asarray(smpl_tables["J_regressor_extra"], np.float32), verts)
        joints = np.concatenate([j45, extra], 1)[:, netspec.SPIN49_FROM_54]
    return verts, pose_hat.reshape(T, 72), joints


# ----------------------------------------------------------------------------- the crop's affine map, as the reference forms it
def _affine_from_box(cx, cy, w, h, dst_w, dst_h, scale):
    """gen_trans_from_patch_cv (img_utils.py:54-88, rot = 0) + cv2.getAffineTransform, then the inversion cv2.warpAffine applies:
      * src_w = w*scale, src_h = h*scale in double (numpy 1.18 promotes float32-scalar * Python float to float64);
      * the two triangles as FLOAT32 points: centre, centre + (0, src_h/2), centre + (src_w/2, 0) and (dst_w/2, dst_h/2), + (0, dst_h/2), + (dst_w/2, 0);
      * getAffineTransform: the 6x6 system of the three point pairs solved in double (LU with partial pivoting);
      * warpAffine (no WARP_INVERSE_MAP) inverts the 2x3 matrix in double: D = 1/(M0*M4 - M1*M3), ...
    The LU's last-bit rounding can differ between LAPACK and OpenCV; it matters only where cvRound(x*1024) sits on an exact tie."""
    src_w, src_h = float(w) * float(scale), float(h) * float(scale)
    centre = np.array([cx, cy], np.float64)
    src = np.zeros((3, 2), np.float32)
    src[0] = centre
    src[1] = centre + np.array([0, src_h * 0.5], np.float32)
    src[2] = centre + np.array([src_w * 0.5, 0], np.float32)
    dc = np.array([dst_w * 0.5, dst_h * 0.5], np.float32)
    dst = np.zeros((3, 2), np.float32)
    dst[0] = dc
    dst[1] = dc + np.array([0, dst_h * 0.5], np.float32)
    dst[2] = dc + np.array([dst_w * 0.5, 0], np.float32)
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for k in range(3):
        a[2 * k, 0:2], a[2 * k, 2] = src[k], 1.0
        a[2 * k + 1, 3:5], a[2 * k + 1, 5] = src[k], 1.0
        b[2 * k], b[2 * k + 1] = dst[k]
    m = np.linalg.solve(a, b)                              # [M0 M1 M2 M3 M4 M5]
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m0, m1, m3, m4 = a11, m[1] * -d, m[3] * -d, a22
    return (m0, m1, -m0 * m[2] - m1 * m[5], m3, m4, -m3 * m[2] - m4 * m[5])


def cv_crop_maps(bboxes, scale=1.0, crop_size=224):
    """(n,4) boxes [cx,cy,w,h] -> (n,10) float64 records for grnet_crop_normalise_cv_maps: what generate_patch_image_cv
    (img_utils.py:90-113, called by get_single_image_crop_demo :252-285 with do_flip = False, rot = 0) makes cv2.warpAffine evaluate.
      * w == h (what the tracker path of demo.py produces): ONE warp into the patch -- record = [inverse map (6), 0, 0, 0, 0];
      * w != h (:97-106; precomputed annotations may hold such boxes): TWO warps -- the scaled box resized, aspect kept, to
        (iw, ih) = (int(s*w), int(s*h)) with s = crop/max(w, h), whose uint8 result is then moved by (crop/2 - iw/2, crop/2 - ih/2)
        into the patch (the letterbox stays 0; a half-pixel offset blends neighbours) -- record = [inverse of the first map (6), iw, ih,
        tx, ty] with (tx, ty) the inverse translation.  The comparison is exact, as the reference's `bb_width != bb_height` is."""
    bboxes = np.asarray(bboxes)
    out = np.zeros((bboxes.shape[0], 10), np.float64)
    for i, (cx, cy, w, h) in enumerate(bboxes):
        if float(w) != float(h):
            s = crop_size / max(float(h), float(w))
            iw, ih = int(s * float(w)), int(s * float(h))
            if iw < 1 or ih < 1:
                raise ValueError(f"box {i} is {float(w)} x {float(h)}: its aspect-preserving resize has an empty side ({iw} x {ih}); the reference's cv2.warpAffine refuses that size too")
            out[i, :6] = _affine_from_box(cx, cy, w, h, iw, ih, scale)
            dx, dy = crop_size / 2 - iw / 2, crop_size / 2 - ih / 2
            out[i, 6:] = (iw, ih, -1.0 * dx - (-0.0) * dy, -(-0.0) * dx - 1.0 * dy)      # warpAffine's inversion of [[1,0,dx],[0,1,dy]]
        else:
            out[i, :6] = _affine_from_box(cx, cy, w, h, crop_size, crop_size, scale)
    return out


def cv_inverse_affine(bboxes, scale=1.0, crop_size=224):
    """(n,4) SQUARE boxes -> (n,6) float64: the inverse affine map of the single-warp crop (the first six entries of cv_crop_maps).
    A box with w != h takes the reference's two-warp branch, which one map cannot describe: ValueError -- use cv_crop_maps."""
    maps = cv_crop_maps(bboxes, scale, crop_size)
    if np.any(maps[:, 6] != 0):
        raise ValueError(f"box {int(np.nonzero(maps[:, 6])[0][0])} is not square: the reference crops it in two warps (img_utils.py:97-106); cv_crop_maps describes both")
    return np.ascontiguousarray(maps[:, :6])


# ----------------------------------------------------------------------------- frame sources
def crop_and_normalise(img_rgb_u8, bbox, scale=1.0, crop_size=224):
    """One frame: uint8 HxWx3 RGB + [cx,cy,w,h] -> float32 (3,224,224), ImageNet-normalised.

    Geometry of get_single_image_crop_demo / generate_patch_image_cv without rotation: the square box
    of side max(w,h)*scale ... the reference passes w == h boxes; pixels outside the image are 0.
    """
    from PIL import Image
    cx, cy, w, h = [float(v) for v in bbox]
    a, e = w * scale / crop_size, h * scale / crop_size
    # cv2.warpAffine geometry (integer coordinates are pixel centres): x = (u - 112) * a + cx.  PIL samples at pixel
    # centres u + 0.5 and expects source coordinates in the same half-pixel convention, hence the +-0.5 terms.
    c0 = cx - 0.5 * crop_size * a + 0.5 - 0.5 * a
    f0 = cy - 0.5 * crop_size * e + 0.5 - 0.5 * e
    img = Image.fromarray(img_rgb_u8)
    crop = img.transform((crop_size, crop_size), Image.AFFINE, (a, 0.0, c0, 0.0, e, f0), resample=Image.BILINEAR, fillcolor=0)
    x = np.asarray(crop, np.float32) / 255.0
    x = (x - IMAGENET_MEAN) / IMAGENET_STD
    return np.ascontiguousarray(x.transpose(2, 0, 1))


class InferenceFrames:
    """Counterpart of lib/dataset/inference.py:Inference for a folder of extracted frames.

    Accepts .png/.jpg (cropped + normalised on the fly) or .npy files holding already-normalised
    (3,224,224) crops (the synthetic-frame configs).  As in the reference, ``bboxes[:, 2:]`` is multiplied
    by ``scale`` in place at construction AND ``scale`` is applied again in the crop (inference.py:48,80).
    """

    def __init__(self, image_folder, frames, bboxes, scale=1.0, crop_size=224):
        names = sorted(x for x in os.listdir(image_folder) if x.endswith((".png", ".jpg", ".npy")))
        self.files = np.array([osp.join(image_folder, x) for x in names])[frames]
        self.bboxes = bboxes
        self.bboxes[:, 2:] *= scale
        self.frames = frames
        self.scale, self.crop_size = scale, crop_size

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        f = self.files[idx]
        if f.endswith(".npy"):
            return np.load(f).astype(np.float32)
        from PIL import Image
        img = np.asarray(Image.open(f).convert("RGB"))
        return crop_and_normalise(img, self.bboxes[idx], self.scale, self.crop_size)

    def image_size(self):
        f = self.files[0]
        if f.endswith(".npy"):
            return 224, 224
        from PIL import Image
        with Image.open(f) as im:
            return im.size

    def batches(self, batch_size, model=None):
        """(<=batch,3,224,224) crops.  With ``model`` (a GRNet) and image files, the raw uint8 frames are uploaded and
        cropped + normalised by the HIP kernel (grnet_crop_normalise) instead of on the host."""
        from PIL import Image
        device_crop = model is not None and len(self) and not self.files[0].endswith(".npy")
        for s in range(0, len(self), batch_size):
            idx = range(s, min(s + batch_size, len(self)))
            if not device_crop:
                yield np.stack([self[i] for i in idx])
                continue
            raw = np.stack([np.asarray(Image.open(self.files[i]).convert("RGB")) for i in idx])
            yield model.crop_normalise(torch.from_numpy(raw).cuda(), torch.from_numpy(np.ascontiguousarray(self.bboxes[list(idx)])),
                                       scale=self.scale)


# ----------------------------------------------------------------------------- model loops
def run_tracklet(model, batches, device="cuda"):
    """demo.py:151-188: feed (<=batch,3,224,224) batches, slice theta, concatenate, to numpy."""
    acc = defaultdict(list)
    for batch in batches:
        x = torch.as_tensor(batch, dtype=torch.float32).unsqueeze(0).to(device)
        bs, t = x.shape[:2]
        out = model(x)[-1]
        acc["pred_cam"].append(out["theta"][:, :, :3].reshape(bs * t, -1))
        acc["verts"].append(out["verts"].reshape(bs * t, -1, 3))
        acc["pose"].append(out["theta"][:, :, 3:75].reshape(bs * t, -1))
        acc["betas"].append(out["theta"][:, :, 75:].reshape(bs * t, -1))
        acc["joints3d"].append(out["kp_3d"].reshape(bs * t, -1, 3))
        acc["smpl_joints2d"].append(out["kp_2d"].reshape(bs * t, -1, 2))
    return {k: torch.cat(v, 0).cpu().numpy() for k, v in acc.items()}


def run_tracks_overlapped(model, tracks, batch_size=64, scale=1.1):
    """BASELINE configs[4] on one GPU: several person tracks, each a list of (raw uint8 frames (n,H,W,3), boxes (n,4)) batches.
    Upload + crop/normalise (grnet_crop_normalise, row f1) of batch k+1 run on a side HIP stream while the forward of batch k
    (a replayed hipGraph when GRNET_OPT_USE_GRAPH is set: fixed crop buffers make the pointers repeat) runs on the current
    stream; two crop buffers alternate and events order the hand-over.  Returns one demo-style dict per track
    (demo.py:151-188 slicing), identical to calling crop_normalise + model() batch after batch."""
    dev = model.device
    main = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)
    work = [(ti, raw, bb) for ti, batches in enumerate(tracks) for (raw, bb) in batches]
    n_max = max((len(bb) for _, _, bb in work), default=0)
    if n_max > batch_size:
        raise ValueError("a batch is larger than batch_size")
    bufs = [torch.empty(n_max, 3, 224, 224, dtype=torch.float32, device=dev) for _ in range(2)]
    ready = [torch.cuda.Event() for _ in range(2)]         # crop k has landed in bufs[k % 2]
    free = [torch.cuda.Event() for _ in range(2)]          # the forward that read bufs[k % 2] is done
    for e in free:
        e.record(main)

    def stage(k):
        _, raw, bb = work[k]
        with torch.cuda.stream(side):
            side.wait_event(free[k % 2])
            crop = model.crop_normalise(torch.as_tensor(raw).to(dev, non_blocking=True), torch.as_tensor(bb), scale=scale)
            bufs[k % 2][:len(bb)].copy_(crop)
            ready[k % 2].record(side)

    acc = [defaultdict(list) for _ in tracks]
    if work:
        stage(0)
    for k, (ti, _, bb) in enumerate(work):
        if k + 1 < len(work):
            stage(k + 1)                                    # overlaps the forward below
        main.wait_event(ready[k % 2])
        out = model(bufs[k % 2][:len(bb)].unsqueeze(0))[-1]
        free[k % 2].record(main)
        t = len(bb)
        a = acc[ti]
        a["pred_cam"].append(out["theta"][:, :, :3].reshape(t, -1))
        a["verts"].append(out["verts"].reshape(t, -1, 3))
        a["pose"].append(out["theta"][:, :, 3:75].reshape(t, -1))
        a["betas"].append(out["theta"][:, :, 75:].reshape(t, -1))
        a["joints3d"].append(out["kp_3d"].reshape(t, -1, 3))
        a["smpl_joints2d"].append(out["kp_2d"].reshape(t, -1, 2))
    return [{k: torch.cat(v, 0).cpu().numpy() for k, v in a.items()} for a in acc]


def make_demo_result(pred, bboxes, frames, orig_width, orig_height):
    """The per-person dict the demo pickles (demo.py:198-222)."""
    return {
        "pred_cam": pred["pred_cam"],
        "orig_cam": convert_crop_cam_to_orig_img(pred["pred_cam"], bboxes, orig_width, orig_height),
        "verts": pred["verts"],
        "pose": pred["pose"],
        "betas": pred["betas"],
        "joints3d": pred["joints3d"],
        "joints2d": convert_crop_coords_to_orig_img(bboxes, pred["smpl_joints2d"], crop_size=224),
        "bboxes": bboxes,
        "frame_ids": frames,
    }


def run_on_frames(model, image_folder, frames, bboxes, device="cuda", batch_size=None, on_device=False):
    """batch_generation.py:289-371: one batch per video (batch_size = max(n_frames, 400)), kp_3d -> kinectv2.  ``bboxes`` is scaled
    by 1.1 IN PLACE, as the reference's Inference.__init__ does to the caller's array (inference.py:48).  Image files are cropped
    and normalised by the HIP kernel (grnet_crop_normalise, row f1); .npy files hold ready crops.
    on_device: return {"kp_3d": (n,25,3) float32 tensor on ``device``} and never synchronise with the host (the multi-GPU driver
    keeps every work item's joints on the device until its single all-gather); default: the reference's numpy array."""
    ds = InferenceFrames(image_folder, frames, bboxes, scale=1.1)
    joints = []
    sel = torch.as_tensor(np.asarray(netspec.SPIN2_TO_KINECTV2), dtype=torch.long, device=device) if on_device else None
    for batch in ds.batches(batch_size or max(len(frames), MAX_SEQLEN), model=model):
        x = torch.as_tensor(batch, dtype=torch.float32).unsqueeze(0).to(device)
        out = model(x)[-1]
        if on_device:
            joints.append(out["kp_3d"].detach().squeeze(0).index_select(1, sel).to(torch.float32))
            continue
        j = out["kp_3d"].detach().cpu().squeeze(0).numpy()
        joints.append(spin2_to_kinectv2(j).astype(np.float32))
    if on_device:
        return {"kp_3d": torch.cat(joints, 0) if joints else torch.zeros(0, 25, 3, dtype=torch.float32, device=device)}
    return {"kp_3d": np.concatenate(joints, 0) if joints else np.zeros((0, 25, 3), np.float32)}


class BatchDb:
    """The joblib 'json' database of batch_generation.py:226-243,265-284: flushed every 50 videos."""

    def __init__(self, outpath):
        if not outpath.endswith(".json"):
            raise AssertionError("outpath must end with .json (batch_generation.py:236)")
        self.outpath, self.out_ind, self.db, self.written = outpath, 0, defaultdict(list), []

    def add(self, vid_name, bboxes, joints3d):
        n = bboxes.shape[0]
        self.db["vid_name"].extend([vid_name] * n)
        self.db["bbox"].append(np.asarray(bboxes).reshape(n, 4))
        self.db["joints3D"].append(np.asarray(joints3d).reshape(n, 25, 3))

    def flush(self):
        import joblib
        if not len(self.db):
            return None
        db = {k: (np.concatenate(v, 0).astype(np.float32) if isinstance(v[0], np.ndarray) else np.array(v))
              for k, v in self.db.items()}
        outfp = self.outpath[:-5] + f"_{self.out_ind}.json"
        joblib.dump(db, outfp)
        self.written.append(outfp)
        self.out_ind += 1
        self.db = defaultdict(list)
        return outfp
