#!/usr/bin/env python3
"""batch_generation.py of the MI355X-native path: 3D-joint generation over a folder of videos with
precomputed bounding boxes, writing the reference's joblib "json" database
{'vid_name': (F,), 'bbox': (F,4) f32, 'joints3D': (F,25,3) f32 kinectv2} every 50 videos
(reference: batch_generation.py:180-287 prepare_data, :289-371 run_grnet_on_frame, argparse :373-387).

Video decoding (ffmpeg) is out of scope: --vid_folder holds one sub-folder of extracted frames per video.
With WORLD_SIZE > 1 (torch.distributed.run) each rank takes a contiguous share of every video's frames
and the per-frame joints are all-gathered over RCCL before rank 0 appends them to the database.
"""
import argparse
import importlib
import os
import os.path as osp
import sys
import time

import numpy as np

ROOT = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "video-based-gait-analysis-for-dementia_amd"
MIN_FDIFF = 10            # batch_generation.py:35


def vid_sort_key(x):
    try:                                                      # "SxxxCxxxPxxxRxxxAxxx" names (batch_generation.py:195)
        return (0, int(x[1:4] + x[6:9] + x[11:14] + x[16:19]))
    except ValueError:
        return (1, x)


def prepare_data(fv, vid_folder, outpath, pretrained_file=None, synthetic_weights=False, max_frames=128, dtype="f32"):
    import joblib
    import torch
    pkg = importlib.import_module(PKG)
    pipe = importlib.import_module(PKG + ".pipeline")
    harness = pkg.harness
    assert osp.isfile(fv), fv
    annos = joblib.load(fv)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if synthetic_weights:
        model = pkg.build_synthetic_model(max_frames=max_frames, device_id=local_rank, with_gru=False, dtype=dtype)
    else:
        model = pkg.GRNet(writer=None, seqlen=100, featcorr=None, max_frames=max_frames, device_id=local_rank, dtype=dtype)
        ckpt = torch.load(pretrained_file, map_location="cpu")["gen_state_dict"]
        model.load_state_dict(ckpt, strict=True)              # batch_generation.py:218
        model.finalize()
    db = pipe.BatchDb(outpath) if rank == 0 else None
    vidnames = sorted(os.listdir(vid_folder), key=vid_sort_key)
    start, n_done = time.time(), 0
    for idx, vid_name in enumerate(vidnames):
        if rank == 0 and idx % pipe.MAX_VID == 0 and idx > 0 and (len(vidnames) - idx) > 10:
            print(f"Save database to {db.flush()}.")
        key = vid_name.split(".")[0]
        if key not in annos:
            print(f"Skip video {vid_name}, no precomputed 2D joints!")
            continue
        img_dir = osp.join(vid_folder, vid_name)
        files = sorted(x for x in os.listdir(img_dir) if x.endswith(("png", "jpg", "npy")))
        bboxes = np.asarray(annos[key], np.float32).copy()
        assert abs(len(files) - bboxes.shape[0]) < MIN_FDIFF
        if len(files) != bboxes.shape[0]:                      # align frame number (batch_generation.py:258-261)
            bboxes = np.repeat(bboxes[0, None, :], len(files), axis=0)
        n = len(files)
        lo, hi = harness.shard_range(n, world, rank)
        kp = np.zeros((0, 25, 3), np.float32)
        if hi > lo:
            kp = pipe.run_on_frames(model, img_dir, np.arange(lo, hi), bboxes[lo:hi].copy())["kp_3d"]
        if world > 1:                                          # reassemble the video's joints in frame order
            per = -(-n // world)
            buf = torch.zeros(per, 25, 3, device="cuda")
            buf[:hi - lo] = torch.from_numpy(kp).cuda()
            out = torch.empty(world * per, 25, 3, device="cuda")
            dist.all_gather_into_tensor(out, buf)
            kp = out[:n].cpu().numpy()
        if rank == 0:
            db.add(key, bboxes, kp)
        n_done += n
    if rank == 0:
        print(f"=====>>> Generation frame rate: {n_done / max(time.time() - start, 1e-9):.1f}")
        print(f"Save database to {db.flush()}.")
    model.close()
    if dist is not None:
        dist.destroy_process_group()
    return db.written if rank == 0 else []


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--vid_folder", type=str, default="", help="folder containing one frame folder per video.")
    p.add_argument("--bbox_path", type=str, default="", help="joblib file with the precomputed bbox per video.")
    p.add_argument("--outpath", type=str, default=f"data/{time.strftime('%Y%m%d-%H%M%S')}.json")
    p.add_argument("--pretrained_file", type=str, default="checkpoint/max-grnet.pth.tar")
    p.add_argument("--synthetic_weights", action="store_true")
    p.add_argument("--max_frames", type=int, default=128)
    p.add_argument("--dtype", choices=("f32", "bf16"), default="f32")
    a = p.parse_args()
    prepare_data(fv=a.bbox_path, vid_folder=a.vid_folder, outpath=a.outpath, pretrained_file=a.pretrained_file,
                 synthetic_weights=a.synthetic_weights, max_frames=a.max_frames, dtype=a.dtype)
