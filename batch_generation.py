#!/usr/bin/env python3
"""batch_generation.py of the MI355X-native path: 3D-joint generation over a folder of videos with
precomputed bounding boxes, writing the reference's joblib "json" database
{'vid_name': (F,), 'bbox': (F,4) f32, 'joints3D': (F,25,3) f32 kinectv2} every 50 videos
(reference: batch_generation.py:180-287 prepare_data, :289-371 run_grnet_on_frame, argparse :373-387).

Video decoding (ffmpeg) is out of scope: --vid_folder holds one sub-folder of extracted frames per video.
Image frames are cropped + normalised on the GPU (grnet_crop_normalise); .npy frames are ready crops.

Multi-GPU (torch.distributed.run, one process per GPU): the videos of one database window (<= 50 videos) are cut into
work items of <= --chunk consecutive frames (a short clip stays whole), the items are dealt to the ranks by load, every
rank runs its items in calls of >= the size at which the kernels are efficient, and the per-frame joints of the WHOLE
window are reassembled with ONE RCCL all-gather before rank 0 appends them to the database -- not one collective and one
host synchronisation per video.
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
BBOX_SCALE = 1.1          # batch_generation.py:296 (Inference(scale=1.1))


def vid_sort_key(x):
    try:                                                      # "SxxxCxxxPxxxRxxxAxxx" names (batch_generation.py:195)
        return (0, int(x[1:4] + x[6:9] + x[11:14] + x[16:19]))
    except ValueError:
        return (1, x)


def flush_windows(n_videos, max_vid):
    """Index ranges [a, b) of videos that end up in the same database file: the reference flushes at the top of iteration idx
    when idx % 50 == 0, idx > 0 and more than 10 videos remain (batch_generation.py:226), and once more at the end."""
    cuts = [0] + [i for i in range(1, n_videos) if i % max_vid == 0 and (n_videos - i) > 10] + [n_videos]
    return [(a, b) for a, b in zip(cuts, cuts[1:]) if b > a]


def prepare_data(fv, vid_folder, outpath, pretrained_file=None, synthetic_weights=False, max_frames=128, dtype="f32", chunk=None,
                 model_factory=None, backend="nccl", exchange="torch"):
    """model_factory(local_rank) -> model and backend="gloo" are the seam of the CPU tests (tests/test_host_cpu.py): the window / plan /
    run / gather / flush logic below then runs under two gloo ranks with a stand-in model and tensors on the CPU.
    exchange: "torch" = the window's all-gather through the launcher's process group; "capi" = through the C ABI's own RCCL communicator
    (harness.RcclComm: grnet_comm_create + grnet_allgather; needs one GPU per rank)."""
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
    on_gpu = backend == "nccl"
    if on_gpu:
        torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if on_gpu:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if model_factory is not None:
        model = model_factory(local_rank)
    elif synthetic_weights:
        model = pkg.build_synthetic_model(max_frames=max_frames, device_id=local_rank, with_gru=False, dtype=dtype)
    else:
        model = pkg.GRNet(writer=None, seqlen=100, featcorr=None, max_frames=max_frames, device_id=local_rank, dtype=dtype)
        ckpt = torch.load(pretrained_file, map_location="cpu")["gen_state_dict"]
        model.load_state_dict(ckpt, strict=True)              # batch_generation.py:218
        model.finalize()
    chunk = int(chunk or max_frames)
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    comm = None
    if exchange == "capi" and world > 1:
        if not on_gpu:
            raise ValueError("exchange='capi' needs the nccl backend (one GPU per rank)")
        comm = harness.RcclComm(world, rank, dev, dist=dist)
    db = pipe.BatchDb(outpath) if rank == 0 else None
    vidnames = sorted(os.listdir(vid_folder), key=vid_sort_key)
    start, n_done = time.time(), 0
    for (wa, wb) in flush_windows(len(vidnames), pipe.MAX_VID):
        vids = []                                              # (key, image folder, boxes as the reference stores them)
        for vid_name in vidnames[wa:wb]:
            key = vid_name.split(".")[0]
            if key not in annos:
                if rank == 0:
                    print(f"Skip video {vid_name}, no precomputed 2D joints!")
                continue
            img_dir = osp.join(vid_folder, vid_name)
            files = sorted(x for x in os.listdir(img_dir) if x.endswith(("png", "jpg", "npy")))
            bboxes = np.array(annos[key])                      # a copy in the annotation's own dtype
            assert abs(len(files) - bboxes.shape[0]) < MIN_FDIFF
            if len(files) != bboxes.shape[0]:                  # align frame number (batch_generation.py:258-261)
                bboxes = np.repeat(bboxes[0, None, :], len(files), axis=0)
            vids.append((key, img_dir, bboxes))
        items = harness.plan_work_items([v[2].shape[0] for v in vids], world, chunk)
        mine = []
        for vi, lo, hi, r in items:
            if r != rank:
                continue
            # run_on_frames scales the boxes it is given by 1.1 in place (as Inference.__init__ does, inference.py:48):
            # hand it a copy of the UNSCALED rows; the database rows are scaled once below, on every rank alike
            # the joints stay on the device: no host synchronisation per work item, one all-gather per window
            kp = pipe.run_on_frames(model, vids[vi][1], np.arange(lo, hi), vids[vi][2][lo:hi].copy(), device=dev, batch_size=chunk, on_device=True)["kp_3d"]
            mine.append(kp.reshape(hi - lo, 75))
        local = torch.cat(mine, 0) if mine else torch.zeros(0, 75, device=dev)
        per_video = harness.gather_work_items(items, local, 75, world, rank, dist, dev, comm=comm)
        if rank == 0:
            for vi, (key, _, bboxes) in enumerate(vids):
                # the reference's db holds the boxes AFTER Inference scaled w,h by 1.1 in place (batch_generation.py:263-266
                # appends the very array the dataset modified)
                bboxes[:, 2:] *= BBOX_SCALE
                db.add(key, bboxes, per_video[vi].cpu().numpy().reshape(-1, 25, 3))
                n_done += bboxes.shape[0]
            if wb < len(vidnames):
                print(f"Save database to {db.flush()}.")
    if rank == 0:
        print(f"=====>>> Generation frame rate: {n_done / max(time.time() - start, 1e-9):.1f}")
        print(f"Save database to {db.flush()}.")
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if hasattr(model, "close"):
        model.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return db.written if rank == 0 else []


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--vid_folder", type=str, default="", help="folder containing one frame folder per video.")
    p.add_argument("--bbox_path", type=str, default="", help="joblib file with the precomputed bbox per video.")
    p.add_argument("--outpath", type=str, default=f"data/{time.strftime('%Y%m%d-%H%M%S')}.json")
    p.add_argument("--pretrained_file", type=str, default="checkpoint/max-grnet.pth.tar")
    p.add_argument("--synthetic_weights", action="store_true")
    p.add_argument("--max_frames", type=int, default=400, help="frames per grnet_forward call (activation buffers are sized for it; 400 = the reference's MAX_seqlen: larger calls run the convolutions at a higher rate, 5 800 / 5 880 / 5 930 frames/s at 128 / 256 / 400)")
    p.add_argument("--chunk", type=int, default=None, help="frames per multi-GPU work item (default: --max_frames)")
    p.add_argument("--dtype", choices=("f32", "bf16"), default="f32")
    p.add_argument("--exchange", choices=("torch", "capi"), default="torch", help="multi-GPU: the all-gather through torch.distributed or through the C ABI's grnet_allgather")
    a = p.parse_args()
    prepare_data(fv=a.bbox_path, vid_folder=a.vid_folder, outpath=a.outpath, pretrained_file=a.pretrained_file,
                 synthetic_weights=a.synthetic_weights, max_frames=a.max_frames, dtype=a.dtype, chunk=a.chunk, exchange=a.exchange)
