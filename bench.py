#!/usr/bin/env python3
"""Headline benchmark: frames/s of the MAX-GRNet per-frame path (224x224, fp32) on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one clip of 16 synthetic frames per GPU (BASELINE.json
configs[1]: 1 clip x 16 x 224 x 224, fp32), frames already resident in HBM.  With N > 1 (one process
per GPU) every rank owns the 16-frame shard [16r, 16r+16) of a 16N-frame clip -- frames are independent
(grnet.py:136-152), so the shards need no collective -- and the step ends with the one exchange the north
star names: an RCCL all-gather of the per-frame pose results (theta, kp_3d, kp_2d, point_local_feat = the
GRU input).  Weak scaling.

Launching: under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (the driver's
command) every process is one rank (RANK / LOCAL_RANK / WORLD_SIZE from the environment).  A bare
`python bench.py --gpus N` with N > 1 starts that same launcher itself as a CHILD process -- before this
process has made any GPU call -- relays rank 0's JSON line and exits with the child's code.

Rank 0 prints ONE JSON line.  `roofline.achieved` follows SURVEY 8(d): frames/s per GPU x F_frame (the
algorithmic convolution FLOPs of one frame) -- the whole step's wall time is charged to the convolutions;
`roofline.conv_only_*` is the same FLOPs over the wall time of the conv launches alone (HIP events on the
launch stream).  `cpu_baseline` times the oracle (a port of the reference's CPU path; the reference itself
cannot travel to the GPU box) on the host cores; `parity` is BASELINE.json's second metric (MPJPE / max
relative error of the GPU outputs vs that oracle on the same 16 frames).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG = "video-based-gait-analysis-for-dementia_amd"
FRAMES_PER_GPU = 16
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA = vector peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA (the sparse headline figure is never used)
F_FRAME_FLOP = 2 * 15441563648     # algorithmic conv FLOPs of one 224x224 frame (pinned by tests/test_gpu_parity.py)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 300; 3 for --workload batchgen, whose step is a whole job)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default: 20; 1 for --workload batchgen)")
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU per step (default: configs[1])")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=1, help="clips in flight per GPU: 1 = strictly one clip at a time (headline); "
                    "k > 1 alternates k independent model instances on k streams so consecutive clips overlap (throughput mode)")
    ap.add_argument("--tune-level", type=int, default=1, help="0: cost model only, 1: per-shape measurement (~0.1 s), "
                    "2: + in-context greedy refinement (~20 s, untimed)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default=None, help="default f32 (bf16 for --workload tracks, as configs[4] names it). f32: the headline (BASELINE configs[1]); bf16: bf16 storage / "
                    "fp32 accumulation on the bf16 matrix cores (configs[2] with --frames 256), errors vs the fp32 oracle reported in `parity`")
    ap.add_argument("--tune-cache", default=None, help="tuning table file written by a previous run (default: none, grnet_tune measures)")
    ap.add_argument("--workload", choices=("clip", "batchgen", "tracks"), default="clip", help="tracks: BASELINE configs[4] -- --tracks person tracks x --track-frames raw "
                    "1080p uint8 frames per GPU per step, crop + normalise on a side HIP stream overlapped with the (hipGraph-replayed or lane-stream) forward of the previous "
                    "track, bf16; clip: the headline, one 16-frame clip per GPU per step (weak scaling); "
                    "batchgen: BASELINE configs[3] -- ONE job of --total-frames frames sharded over the GPUs in calls of <= --chunk frames, one all-gather of the "
                    "per-frame records, then the temporal branch (GRU + attention + second head pass) on the whole sequence; a step is the whole job (strong scaling)")
    ap.add_argument("--total-frames", type=int, default=10000, help="batchgen: frames of the whole job")
    ap.add_argument("--chunk", type=int, default=400, help="batchgen: frames per grnet_forward call (400 = the reference's MAX_seqlen, batch_generation.py:34,303: it feeds a video in calls of >= 400 frames)")
    ap.add_argument("--tracks", type=int, default=4, help="tracks: person tracks per GPU")
    ap.add_argument("--track-frames", type=int, default=64, help="tracks: frames per track")
    ap.add_argument("--call-frames", type=int, default=None, help="tracks: frames per forward call (default: all tracks of a step in ONE call; --track-frames = one call per track)")
    ap.add_argument("--no-overlap", action="store_true", help="tracks: crop and forward on ONE stream (the A/B of the side stream)")
    ap.add_argument("--exchange", choices=("auto", "capi", "torch"), default="torch", help="N > 1: who owns the all-gather. torch (default): the launcher's process group "
                    "(RCCL on the nccl backend); capi: the C ABI's own RCCL communicator (grnet_comm_create / grnet_allgather; a second communicator next to the launcher's, "
                    "verified on one rank only so far); auto: capi if every rank can bootstrap it, else torch")
    ap.add_argument("--no-kernel-table", action="store_true", help="skip roofline.dominant_kernel (its per-shape timing launches would sit in a profiler's dispatch list)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the legs of BASELINE configs[2] (bf16, 256 frames), [3] (batchgen at 1 GPU) and [4] (tracks) that the default 1-GPU fp32 run appends as the `secondary` list")
    a = ap.parse_args(argv)
    if a.dtype is None:
        a.dtype = "bf16" if a.workload == "tracks" else "f32"
    if a.steps is None:
        a.steps = {"batchgen": 3, "tracks": 30}.get(a.workload, 300)
    if a.warmup is None:
        a.warmup = {"batchgen": 1, "tracks": 5}.get(a.workload, 20)
    return a


# ------------------------------------------------------------------------------------------- launching
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def child_command(argv, n, port, script=None):
    """The driver's own command line for N ranks on one node (one process per GPU)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)


def self_launch(argv, n, run=subprocess.run, script=None):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks as a CHILD process tree (never exec: this must
    happen before any GPU call of this process, and it does -- nothing above has touched HIP), relay their output."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    proc = run(child_command(argv, n, free_port(), script), env=env)
    return proc.returncode


def rank_env():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_dist(world, rank, local_rank):
    """One process per GPU; backend "nccl" is RCCL on ROCm.  GRNET_BENCH_BACKEND=gloo rehearses the N > 1 path where the ranks
    share one GPU (or have none: CPU tests).  Returns (dist module or None, local_rank to use, device of the reduce tensor)."""
    import torch
    if world == 1:
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
        return None, local_rank, "cuda"
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("GRNET_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist, local_rank, ("cuda" if backend == "nccl" else "cpu")


def make_exchange(args, world, rank, local_rank, dist):
    """(RcclComm or None, label for config.exchange).  `auto` uses the C ABI's communicator only when EVERY rank created it.  The bootstrap inside
    harness.RcclComm is the same on every rank (local probe -> MIN over the ranks -> rank 0 always enters the id broadcast, with a status byte):
    a rank that cannot bind RCCL, or rank 0 failing to draw the id, makes ALL ranks raise before any rank enters grnet_comm_create, and the MIN
    below then sends every rank to torch's all-gather.  Both choices are RCCL on the nccl backend -- this is not a CPU fallback.  The default
    stays `torch` until a run on >= 2 GPUs has exercised `capi` (it has run with one rank only)."""
    backend = os.environ.get("GRNET_BENCH_BACKEND", "nccl")
    choice = getattr(args, "exchange", "torch")
    if world == 1:
        return None, "none (1 GPU)"
    if backend != "nccl":
        if choice == "capi":
            raise SystemExit("--exchange capi needs one GPU per rank (RCCL); the %s rehearsal shares a device" % backend)
        return None, backend + " (rehearsal backend, torch.distributed)"
    if choice == "torch":
        return None, "RCCL via torch.distributed.all_gather_into_tensor"
    import torch
    harness = importlib.import_module(PKG).harness
    comm, err = None, ""
    try:
        comm = harness.RcclComm(world, rank, torch.device("cuda", local_rank), dist=dist)
    except Exception as e:                                      # noqa: BLE001 -- reported below, on every rank alike
        err = f"{type(e).__name__}: {e}"
    ok = torch.tensor([1 if comm is not None else 0], device=f"cuda:{local_rank}")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        return comm, "RCCL via the C ABI (grnet_comm_create + grnet_allgather)"
    if comm is not None:
        comm.close()
    if choice == "capi":
        raise SystemExit("--exchange capi: grnet_comm_create failed on some rank" + (f" (this rank: {err})" if err else ""))
    return None, "RCCL via torch.distributed.all_gather_into_tensor (the C ABI's communicator could not be created on every rank" + (f"; rank {rank}: {err}" if err else "") + ")"


def timed_steps(do_step, device_sync, steps, warmup, dist, reduce_device):
    """W untimed steps, then EXACTLY K steps bracketed by barrier + device sync on both sides; MAX over ranks (seconds)."""
    import torch

    def sync_all():
        device_sync()
        if dist is not None:
            dist.barrier()
        device_sync()

    for _ in range(warmup):
        do_step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        do_step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def prove_exchange(state, world, rank, dist, comm=None):
    """What the all-gather ITSELF did, from its data (round-5 review: `n_gpus` only echoed the launcher's environment).  `state`: a workload's
    exchange_state() after a step -- {"local": this rank's send block, "gathered": the (world, block) receive buffer, "frames": (lo, hi) of this
    rank's shard}.  Every rank sends (rank, lo, hi, checksum of its send block) through the SAME collective path as the data (the C ABI's
    communicator when that owns the exchange, else the launcher's process group) and then checks, locally, that slot r of the data buffer has the
    checksum rank r reported and rank r's frame range follows rank r-1's.  The checksum is the int64 sum of the block's bit patterns: exact and
    order-independent.  Returns the fields of config: exchange_ranks = distinct ranks whose record arrived, frames_total = frames they hold."""
    import torch
    local, gathered, (lo, hi) = state["local"], state["gathered"], state["frames"]
    bits = lambda t: int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())
    mine = torch.tensor([rank, lo, hi, bits(local)], dtype=torch.int64, device=local.device)
    table = torch.empty(world * 4, dtype=torch.int64, device=local.device)
    if comm is not None:
        comm.all_gather(table, mine)
    else:
        dist.all_gather_into_tensor(table, mine)
    table = table.view(world, 4).cpu().tolist()
    ranks = sorted({int(r[0]) for r in table})
    assert [int(r[0]) for r in table] == list(range(world)), f"the all-gather's slots are not rank-major: {[r[0] for r in table]}"
    nxt = 0
    for r, (_, rlo, rhi, rsum) in enumerate(table):
        assert rlo == nxt and rhi >= rlo, f"rank {r} reports frames [{rlo}, {rhi}) but rank {r - 1}'s shard ended at {nxt}"
        nxt = rhi
        got = bits(gathered[r])
        assert got == rsum, f"slot {r} of the gathered block does not hold rank {r}'s send block (checksum {got} != {rsum})"
    if comm is not None:                                       # the C ABI's communicator also says how many ranks RCCL itself counts (ncclCommCount)
        nr, rk = comm.info()
        assert nr == len(ranks) and rk == rank, f"grnet_comm_info reports {nr} ranks / rank {rk}, the gathered records {len(ranks)} ranks / rank {rank}"
    return {"exchange_ranks": len(ranks), "frames_total": nxt,
            "exchange_check": f"slot r of the gathered block == rank r's send block for r = 0..{world - 1} (int64 bit-pattern checksums exchanged through the same collective, "
                              "checked on every rank after the timed region); frame ranges contiguous in rank order"}


# ------------------------------------------------------------------------------------------- the JSON line
def roofline_object(fps_per_gpu, dtype, conv_flops_per_frame, conv_ms, conv_ms_serial, n_conv, n, extra=None, executed_flops_per_frame=None):
    """`achieved` / `frac`: the multiplies the matrix cores EXECUTE per second over the dense peak of the dtype -- the whole step (pooling,
    tail, SMPL, launch gaps) charged to the convolutions.  `effective_*`: SURVEY 8(d)'s figure, frames/s per GPU x F_frame with F_frame the
    ALGORITHMIC (direct-convolution) count; on the fp32 path 92 % of F_frame runs as Winograd F(4x4,3x3) at a quarter of the multiplies,
    so the effective figure can exceed the peak of the fp32 matrix cores (1.1 at 256 frames per call) and is not a utilisation.  The
    conv-only figures (same FLOPs / wall time of the conv launches alone) and the serial per-launch average (what rocprofv3 --stats
    averages add up to) sit beside them."""
    peak = PEAK_BF16_MFMA_TFLOPS if dtype == "bf16" else PEAK_FP32_MFMA_TFLOPS
    executed = executed_flops_per_frame or conv_flops_per_frame
    effective = fps_per_gpu * conv_flops_per_frame / 1e12
    achieved = fps_per_gpu * executed / 1e12
    conv_flops = conv_flops_per_frame * n
    floor_ms = executed * n / (peak * 1e12) * 1e3
    r = {"bound": "mfma (executed multiplies)", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
         "definition": "frames/s per GPU x the FLOPs the matrix cores execute per frame (Winograd F(4x4,3x3) layers at 1/4 of their direct-convolution "
                       "count, x 256/196 on 14x14 and x 64/49 on 7x7 maps whose tiles are padded; every other layer at its direct count) / dense peak of "
                       "the dtype; the whole step -- pooling, tail, SMPL, launch gaps -- is charged to the convolutions",
         "executed_gflop_per_step": round(executed * n / 1e9, 3),
         "floor_ms": round(floor_ms, 4), "step_over_floor": round(n / fps_per_gpu * 1e3 / floor_ms, 3),
         "effective_achieved": round(effective, 3), "effective_frac": round(effective / peak, 4),
         "effective_is": "SURVEY 8d: frames/s per GPU x F_frame (30.883 GFLOP of convolutions per frame = the ALGORITHMIC, direct-convolution count). "
                         "An equivalent-work rate, NOT a share of the peak: the fp32 path executes 0.32 of F_frame, so this figure passes 1.0 at large calls",
         "traffic": None,
         "kernel": ("conv_wino4_f32 (Winograd F(4x4,3x3) on the fp32 matrix cores: the 3x3 stride-1 layers on 56x56 and 28x28 maps) + conv_wino4s_f32 "
                    "(the same on 14x14 / 7x7 maps, register-resident) + conv_mfma_f32 / conv_splitk_f32 (fp32 MFMA implicit-GEMM convolution: "
                    "1x1, stride-2 and stem layers) + hr_fuse_up_f32 (the 1x1 fuse terms of an HR module, grouped), all launches of a step" if dtype == "f32"
                    else "bf16 MFMA convolutions on NHWC activations: conv_bf16_wide_ring / _wide_band (wide 3x3), conv_bf16_chain / _block_frame (HR branches, frame or band resident in LDS), "
                         "conv_bf16_bneck / _bneck_dma / _stem_pair (layer1 Bottlenecks and the stem pair as row-walking launches), conv_bf16_nhwc / _s2_band / _pw_stream (stride-2, 1x1), all launches of a step"),
         "conv_launches_per_step": n_conv, "conv_gflop_per_step": round(conv_flops / 1e9, 3),
         "gflop_per_launch": round(conv_flops / 1e9 / max(n_conv, 1), 4)}
    if conv_ms:
        r.update(conv_only_ms_per_step=round(conv_ms, 4), conv_only_achieved=round(executed * n / (conv_ms * 1e-3) / 1e12, 3),
                 conv_only_frac=round(executed * n / (conv_ms * 1e-3) / 1e12 / peak, 4),
                 conv_only_effective_frac=round(conv_flops / (conv_ms * 1e-3) / 1e12 / peak, 4),
                 avg_launch_us=round(conv_ms * 1e3 / max(n_conv, 1), 3))
    if conv_ms_serial:
        r.update(conv_ms_per_step_serial=round(conv_ms_serial, 4), avg_launch_us_serial=round(conv_ms_serial * 1e3 / max(n_conv, 1), 3),
                 serial_note="the same launches one after another on ONE stream: sum of per-kernel durations, what rocprofv3 --kernel-trace "
                             "--stats averages of a GRNET_MULTI_LANE=0 run add up to (profiles/)")
    if extra:
        r.update(extra)
    return r


LAYER_TABLE_FILES = {"f32": "r06_layer_traffic.json", "bf16": "r06_bf16_n256_layer_traffic.json"}   # per-kernel counter / algorithmic bytes of this round (tools/layer_table.py), optional


def kernel_objects(table, dtype):
    """`dominant_kernel` (the kernel family whose launches add up to the most time when each runs alone) and the top of the table, from
    GRNet.kernel_table: durations measured live with HIP events (grnet_time_conv), FLOPs from the launch list."""
    peak = PEAK_BF16_MFMA_TFLOPS if dtype == "bf16" else PEAK_FP32_MFMA_TFLOPS
    table_file = LAYER_TABLE_FILES[dtype]
    ratios, why = {}, f"none: profiles/{table_file} is missing (the per-launch counter passes of this round were not taken)"
    try:
        with open(os.path.join(ROOT, "profiles", table_file)) as f:
            ratios = json.load(f).get("counter_over_algorithmic_by_kernel", {})
        why = f"profiles/{table_file}"
    except (OSError, ValueError):
        pass

    def obj(r):
        return {"name": r["name"], "launches": r["launches"], "us": round(r["avg_us"], 2), "total_us": round(r["total_us"], 1),
                "frac_executed": round(r["executed_gflop"] / r["total_us"] * 1e3 / peak, 4),
                "frac_effective": round(r["gflop"] / r["total_us"] * 1e3 / peak, 4),
                "counter_over_algorithmic": ratios.get(r["name"])}

    top = [obj(r) for r in table[:6]]
    dom = dict(top[0], traffic_source=why, measured="each distinct layer shape launched alone, 20 back-to-back launches between two HIP events "
               "on the launch stream (grnet_time_conv); frac_* = FLOPs of the family's launches / their summed durations / dense peak")
    return dom, top


ROUND = "r06"
TRAFFIC_FILES = {("f32", 16): f"{ROUND}_pmc_traffic.json",            # THIS round's counter passes (tools/gpu_profile.sh [f32|bf16] -> tools/summarize_profiles.py)
                 ("bf16", 256): f"{ROUND}_bf16_n256_pmc_traffic.json"}


def stored_traffic(n, dtype, algorithmic_bytes=None):
    """HBM bytes per step of the conv launches from the PMC passes (profiles/): a static, committed measurement of this same
    workload and these same kernels, not collected inside this run (counters need their own rocprofv3 passes).  Only the
    current round's file of this (dtype, frames per call) is read -- a missing or other-round file gives traffic = null with
    the reason, never an older number."""
    out = {"traffic": None, "algorithmic_bytes": algorithmic_bytes,
           "algorithmic_bytes_is": "per LAYER (accounting.step_algorithmic_bytes): input + fused addends + weights read once, output written once, at the "
                                   "path's storage size; launches that keep intermediates on chip (bf16 BasicBlock chains) move fewer bytes than this"}
    name = TRAFFIC_FILES.get((dtype, n))
    if name is None:
        out["traffic_source"] = f"none: counter passes exist for {sorted(TRAFFIC_FILES)} only"
        return out
    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            tj = json.load(f)
    except OSError:
        out["traffic_source"] = f"none: profiles/{name} is missing (the counter passes of this round were not taken)"
        return out
    out["traffic"] = tj.get("hbm_bytes_per_step_conv_kernels")
    out["traffic_over_algorithmic"] = round(out["traffic"] / algorithmic_bytes, 3) if out["traffic"] and algorithmic_bytes else None
    out["traffic_source"] = (f"profiles/{name}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this workload with this "
                             "round's kernels (FETCH x2 gfx950 correction per MI355X_MICROARCH.md, an upper bound for the narrow staging patterns), "
                             "bytes of all conv launches of one step; a stored measurement, not taken inside this run")
    return out


def cpu_baseline(pkg, frames_np, budget_s=20.0):
    """The oracle (port of the reference's CPU path) on the host cores, bounded sample.  oneDNN oversubscribes badly on
    a 128-thread host for 16-frame batches, so the thread count is probed first and the fastest one is used and reported.
    Returns (cpu_baseline object, the oracle's outputs for `frames_np` -- the checker for the `parity` object)."""
    import torch
    oracle = importlib.import_module("oracle.grnet_oracle")
    sd, smpl = pkg.synth.make_state_dict(), pkg.synth.make_smpl_tables()
    oracle.grnet_forward(frames_np[:2], sd, smpl)           # warm-up (oneDNN primitive caches)
    all_threads = torch.get_num_threads()
    best_t, best_dt = all_threads, None
    for t in sorted({8, 16, 32, 64, all_threads}):
        if t > all_threads:
            continue
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        oracle.grnet_forward(frames_np, sd, smpl)             # the probe runs the SAME clip the timed passes run
        dt = time.perf_counter() - t0
        if best_dt is None or dt < best_dt:
            best_t, best_dt = t, dt
    torch.set_num_threads(best_t)
    t_all, passes = 0.0, 0
    while t_all < budget_s and passes < 12:
        t0 = time.perf_counter()
        ref = oracle.grnet_forward(frames_np, sd, smpl)
        t_all += time.perf_counter() - t0
        passes += 1
    torch.set_num_threads(all_threads)
    n = frames_np.shape[0] * passes
    return {"value": round(n / t_all, 3), "unit": "frames/s", "cores": best_t, "host_cores": os.cpu_count(), "kind": "port",
            "sample": f"{passes} passes of the oracle (torch-CPU oneDNN convs + numpy tail) over the same {frames_np.shape[0]} frames, "
                      f"fp32, {best_t} threads used (fastest of a probe over 8..{all_threads} on the same clip) on a host with {os.cpu_count()} logical cores"}, ref


def parity_vs_oracle(got, ref):
    """BASELINE.json's second metric: MPJPE of kp_3d and the error of the GPU path's outputs against the CPU oracle on the same
    frames, in two forms: max|a-b| / max|b| per tensor (tensor scale) and element-wise |a-b| <= 1e-3 |b| + floor with the
    floor at 1e-3 of the tensor's RMS (a per-element relative bar that does not blow up on entries that are ~0 by cancellation)."""
    import numpy as np
    rel, med, elem = {}, {}, {}
    for k in ("theta", "kp_3d", "kp_2d", "verts", "rotmat"):
        a, b = np.asarray(got[k], np.float64), np.asarray(ref[k], np.float64).reshape(got[k].shape)
        rel[k] = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
        per_frame = np.abs(a - b).reshape(a.shape[0], -1).max(1) / max(np.abs(b).max(), 1e-30)
        med[k] = float(np.median(per_frame))
        floor = 1e-3 * float(np.sqrt(np.mean(b * b)))
        elem[k] = float((np.abs(a - b) / (1e-3 * np.abs(b) + floor)).max())      # <= 1 passes
    d = np.asarray(got["kp_3d"], np.float64) - np.asarray(ref["kp_3d"], np.float64).reshape(got["kp_3d"].shape)
    return {"mpjpe_m": float(np.linalg.norm(d, axis=-1).mean()), "max_rel_err": {k: float(f"{v:.3e}") for k, v in rel.items()},
            "median_frame_rel_err": {k: float(f"{v:.3e}") for k, v in med.items()},
            "elementwise_worst_ratio": {k: float(f"{v:.3e}") for k, v in elem.items()},
            "elementwise_form": "|a-b| <= 1e-3*|b| + 1e-3*rms(b) for every element (ratio <= 1 passes)",
            "tolerance": 1e-3, "ok": bool(max(rel.values()) < 1e-3 and max(elem.values()) <= 1.0),
            "vs": "oracle (CPU port of the reference path) on the same frames and weights; SMPL (smplx) and its tables are "
                  "pinned to the published algorithm only (no smplx offline)"}


def bf16_parity(pkg, got, frames_np):
    """The bf16 legs' gate (round-5 review: they carried ok = None).  bf16 has no reference mode, so the bar is the one tests/test_gpu_bf16.py states:
    per tensor, the GPU's distance from the fp32 oracle may not exceed twice the distance of the oracle's own bf16-storage emulation (oracle.bf16_storage:
    the same roundings at the same places, another summation order) + 1e-3, and the mean joint error stays below 1.5 cm.  `got`: outputs for `frames_np`
    (the first 8 frames of the call: frames are independent, so they stand for the call's kernels at the call's size)."""
    import numpy as np
    oracle = importlib.import_module("oracle.grnet_oracle")
    sd, smpl = pkg.synth.make_state_dict(), pkg.synth.make_smpl_tables()
    ref = oracle.grnet_forward(frames_np, sd, smpl)
    with oracle.bf16_storage():
        emu = oracle.grnet_forward(frames_np, sd, smpl)
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64).reshape(np.shape(a))).max() / max(np.abs(np.asarray(b)).max(), 1e-30))
    gpu, em, ok = {}, {}, True
    for k in ("theta", "kp_3d", "kp_2d", "verts", "rotmat"):
        gpu[k], em[k] = rel(got[k], ref[k]), rel(emu[k], ref[k])
        ok = ok and gpu[k] < 2.0 * em[k] + 1e-3
    d = np.asarray(got["kp_3d"], np.float64).reshape(-1, 29, 3) - np.asarray(ref["kp_3d"], np.float64).reshape(-1, 29, 3)
    mpjpe = float(np.linalg.norm(d, axis=-1).mean())
    return {"mpjpe_m": mpjpe, "max_rel_err": {k: float(f"{v:.3e}") for k, v in gpu.items()},
            "emulation_max_rel_err": {k: float(f"{v:.3e}") for k, v in em.items()},
            "tolerance": "per tensor: GPU vs fp32 oracle < 2 x (bf16-storage emulation of the oracle vs fp32 oracle) + 1e-3; MPJPE < 0.015 m",
            "ok": bool(ok and mpjpe < 0.015), "frames_checked": int(np.shape(frames_np)[0]),
            "vs": "oracle (CPU port of the reference path, fp32) and its bf16-storage emulation on the same frames and weights (tests/test_gpu_bf16.py states the same bound)"}


# ------------------------------------------------------------------------------------------- one rank
class GpuWorkload:
    """The real thing: the synthetic MAX-GRNet model of this rank + its resident 16-frame shard."""

    def __init__(self, args, world, rank, local_rank, dist):
        import torch
        self.torch, self.args, self.world, self.rank, self.dist = torch, args, world, rank, dist
        pkg = self.pkg = importlib.import_module(PKG)
        harness = pkg.harness
        n = self.n = args.frames
        self.model = pkg.build_synthetic_model(max_frames=n, device_id=local_rank, with_gru=False, dtype=args.dtype)
        self.frames_np = pkg.synth.make_frames(n, start=rank * n)
        frames = torch.from_numpy(self.frames_np).cuda()
        cache = self.cache = args.tune_cache                  # a table exported by grnet_get_tuning; default: measure (grnet_tune)
        self.comm, self.exchange = make_exchange(args, world, rank, local_rank, dist)
        mk = lambda m: harness.ClipRunner(m, frames, use_graph=not args.no_graph, world=world, rank=rank, dist=dist,
                                          tune_level=args.tune_level, tune_cache=cache, comm=self.comm)
        self.runner = mk(self.model)
        self.runners, self.streams = [self.runner], [torch.cuda.current_stream()]
        for _ in range(1, max(1, args.inflight)):             # extra clips in flight: own buffers, own stream
            m_k = pkg.build_synthetic_model(max_frames=n, device_id=local_rank, with_gru=False, dtype=args.dtype)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                self.runners.append(mk(m_k))
            self.streams.append(st)
        self.step_no = 0

    def step(self):
        k = self.step_no % len(self.runners)
        self.step_no += 1
        if len(self.runners) == 1:
            self.runner.step()
        else:
            with self.torch.cuda.stream(self.streams[k]):
                self.runners[k].step()

    def sync(self):
        self.torch.cuda.synchronize()

    def exchange_state(self):
        r = self.runner
        return {"local": r.packed, "gathered": r.gathered.view(self.world, -1), "frames": (self.rank * self.n, (self.rank + 1) * self.n), "comm": self.comm}

    def config(self):
        args, model, n = self.args, self.model, self.n
        tm = model.tuned_mode(n) or {}
        eager = args.no_graph or tm.get("eager", False)
        launch_desc = ("eager launches on 4 lane streams" if eager else "hipGraph replay") + \
                      ", one launch per convolution" + \
                      (" (schedule picked by grnet_tune)" if tm else "")
        return {"workload": f"1 clip x {n} frames x 3x224x224 per GPU, {'fp32' if args.dtype == 'f32' else 'bf16 storage / fp32 accumulation'}, "
                            "MAX-GRNet per-frame path (HRNet-W32 + PARE head + SMPL LBS), seed-defined synthetic weights",
                "frames_per_gpu": n, "clips_in_flight": len(self.runners), "launch": launch_desc,
                "kernel_launches_per_step": model.num_kernel_launches(),
                "launch_configs": ("stored table " + os.path.relpath(self.cache, ROOT)) if self.cache else f"grnet_tune level {args.tune_level}",
                "exchange": self.exchange + ("" if self.world == 1 else ": one all-gather of the per-frame pose records per step")}

    def roofline(self, fps_per_gpu):
        model, n, pkg = self.model, self.n, self.pkg
        conv_ms = min(model.time_convs(n) for _ in range(5))
        conv_ms_serial = None
        if not getattr(self.args, "light", False):
            model.set_option(pkg._lib.OPT_MULTI_LANE, 0)      # the same launches one after another on one stream
            conv_ms_serial = min(model.time_convs(n) for _ in range(3))
            model.set_option(pkg._lib.OPT_MULTI_LANE, 1)
        alg_bytes = pkg.accounting.step_algorithmic_bytes(model.describe_convs(), n, 4 if self.args.dtype == "f32" else 2)
        extra = stored_traffic(n, self.args.dtype, alg_bytes)
        if self.rank == 0 and not getattr(self.args, "light", False) and not getattr(self.args, "no_kernel_table", False):
            extra["dominant_kernel"], extra["kernels_by_time_alone"] = kernel_objects(model.kernel_table(n), self.args.dtype)
        return roofline_object(fps_per_gpu, self.args.dtype, model.conv_flops_per_frame(), conv_ms, conv_ms_serial,
                               model.num_conv_launches(), n, extra,
                               executed_flops_per_frame=model.conv_executed_flops_per_frame(n))

    def outputs_np(self, k=None):
        """One more step; the first k frames' outputs as numpy (all frames: k = None)."""
        self.runner.step()
        self.torch.cuda.synchronize()
        got = {name: v[:k].cpu().numpy() for name, v in self.runner.sequence().items() if name != "point_local_feat"}
        got.update(verts=self.runner.verts[:k].cpu().numpy(), rotmat=self.runner.rotmat[:k].cpu().numpy())
        return got

    def parity_bf16(self):
        k = min(8, self.n)
        return bf16_parity(self.pkg, self.outputs_np(k), self.frames_np[:k])

    def extras(self, line):
        """cpu_baseline + parity: rank 0 at N = 1 only, after the timed region."""
        if self.world != 1:
            return
        if self.args.dtype == "bf16":                          # the 1e-3 bar is the fp32 path's; the bf16 legs are gated on the emulation bound
            if not getattr(self.args, "no_parity", False):
                line["parity"] = self.parity_bf16()
            if self.args.no_cpu_baseline:
                return
        if self.args.no_cpu_baseline:
            return
        line["cpu_baseline"], ref = cpu_baseline(self.pkg, self.frames_np)
        if self.args.dtype != "bf16":
            line["parity"] = parity_vs_oracle(self.outputs_np(), ref)

    def close(self):
        if self.comm is not None:
            self.torch.cuda.synchronize()
            self.comm.close()
        for r in self.runners:
            r.model.close()


class BatchgenWorkload:
    """BASELINE configs[3] (`--workload batchgen`): harness.ShardedSequenceRunner over this rank's shard of a --total-frames job."""

    def __init__(self, args, world, rank, local_rank, dist):
        import torch
        self.torch, self.args, self.world, self.rank, self.dist = torch, args, world, rank, dist
        pkg = self.pkg = importlib.import_module(PKG)
        chunk = self.chunk = min(args.chunk, args.total_frames)
        self.model = pkg.build_synthetic_model(max_frames=chunk, device_id=local_rank, dtype=args.dtype, use_gait_feat=True)
        lo, hi = pkg.harness.shard_range(args.total_frames, world, rank)
        base = torch.from_numpy(pkg.synth.make_frames(min(chunk, max(hi - lo, 1)), start=lo)).cuda()
        reps = -(-(hi - lo) // base.shape[0])
        frames = base.repeat(reps, 1, 1, 1)[:hi - lo].contiguous()               # the shard resident in HBM: (hi-lo) x 602 KB
        self.model.finalize()                                                    # eager lane streams: every call has its own output pointers
        if args.tune_level:
            self.model.tune(chunk, level=args.tune_level)
            if (hi - lo) % chunk:
                self.model.tune((hi - lo) % chunk, level=args.tune_level)
        self.comm, self.exchange = make_exchange(args, world, rank, local_rank, dist)
        self.runner = pkg.harness.ShardedSequenceRunner(self.model, frames, args.total_frames, world, rank, dist, chunk=chunk, comm=self.comm)

    def step(self):
        self.runner.step()

    def sync(self):
        self.torch.cuda.synchronize()

    def exchange_state(self):
        r = self.runner
        lo, hi = self.pkg.harness.shard_range(self.args.total_frames, self.world, self.rank)
        return {"local": r.packed, "gathered": r.gathered.view(self.world, -1), "frames": (lo, hi), "comm": self.comm}

    def config(self):
        a = self.args
        return {"workload": f"batch_generation over ONE {a.total_frames}-frame synthetic video: per-frame path on ceil({a.total_frames}/{self.world}) frames per GPU in calls of <= "
                            f"{self.chunk} frames, one all-gather of the per-frame records (19.4 KB per frame), then cparams + GRU gait encoder + corrector / attention "
                            "block + second head pass on the whole sequence (replicated); seed-defined synthetic weights",
                "total_frames": a.total_frames, "frames_per_gpu": self.runner.n_local, "chunk": self.chunk, "calls_per_gpu": len(self.runner.calls),
                "exchange": self.exchange + ("" if self.world == 1 else f": one all-gather of {self.runner.packed.numel() * 4 / 1e6:.1f} MB per rank, once per job")}

    def roofline(self, fps_per_gpu):
        model, n = self.model, self.chunk
        return roofline_object(fps_per_gpu, self.args.dtype, model.conv_flops_per_frame(), None, None, model.num_conv_launches(), n,
                               {"note": "frames/s per GPU of the WHOLE job (per-frame path + exchange + replicated temporal branch) x the convolution FLOPs of a frame"},
                               executed_flops_per_frame=model.conv_executed_flops_per_frame(n))

    def extras(self, line):
        line["phases"] = self.runner.phases()                  # of the latest job on this rank: per-frame path / exchange / replicated temporal branch

    def close(self):
        if self.comm is not None:
            self.torch.cuda.synchronize()
            self.comm.close()
        self.model.close()


class TracksWorkload:
    """BASELINE configs[4] (`--workload tracks`): harness.OverlappedTrackRunner over --tracks person tracks of --track-frames raw 1080p frames, resident in HBM."""
    H, W = 1080, 1920

    def __init__(self, args, world, rank, local_rank, dist):
        import numpy as np
        import torch
        self.torch, self.args, self.world, self.rank = torch, args, world, rank
        pkg = self.pkg = importlib.import_module(PKG)
        nt, t = args.tracks, args.track_frames
        self.call_frames = args.call_frames or nt * t
        self.model = pkg.build_synthetic_model(max_frames=self.call_frames, device_id=local_rank, with_gru=False, dtype=args.dtype)
        dev = torch.device("cuda", local_rank)
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        video = torch.randint(0, 256, (t, self.H, self.W, 3), dtype=torch.uint8, device=dev, generator=g)        # ONE video, every track crops its own person from it
        rng = np.random.default_rng(77 + rank)
        self.boxes = []
        for k in range(nt):                                     # a person walking across the picture: centre drifts, box breathes, some hang over the border
            cx = np.linspace(150 + 400 * k, 500 + 400 * k, t) + rng.normal(0, 2, t)
            cy = 540 + 200 * np.sin(np.linspace(0, 3, t) + k) + rng.normal(0, 2, t)
            side = 380 + 60 * np.cos(np.linspace(0, 2, t) + k)
            self.boxes.append(np.stack([cx, cy, side, side], 1).astype(np.float32))
        self.raw = [video] * nt
        self.runner = pkg.harness.OverlappedTrackRunner(self.model, self.raw, self.boxes, use_graph=not args.no_graph, tune_level=args.tune_level,
                                                        overlap=not args.no_overlap, call_frames=self.call_frames)
        self.n = nt * t

    def step(self):
        self.runner.step()

    def sync(self):
        self.torch.cuda.synchronize()

    def config(self):
        a, model = self.args, self.model
        tm = model.tuned_mode(self.runner.call_n[0]) or {}
        eager = a.no_graph or tm.get("eager", False)
        return {"workload": f"{a.tracks} person tracks x {a.track_frames} frames per GPU per step, packed into {len(self.runner.calls)} forward call(s) of <= {self.call_frames} frames: raw {self.W}x{self.H} uint8 frames resident in HBM -> crop + normalise "
                            "(OpenCV arithmetic, HIP kernel) -> MAX-GRNet per-frame path, "
                            f"{'fp32' if a.dtype == 'f32' else 'bf16 storage / fp32 accumulation'}; video decode excluded (host, out of scope); seed-defined synthetic weights",
                "tracks": a.tracks, "track_frames": a.track_frames, "frames_per_gpu": self.n,
                "forward_calls_per_step": len(self.runner.calls),
                "preprocess": "same stream as the forward" if a.no_overlap else "side HIP stream, two crop buffers, events both ways (overlaps the previous call's forward, across steps too)",
                "launch": ("eager launches on 4 lane streams" if eager else "hipGraph replay (one captured forward per (crop buffer, track output block))") +
                          (" (schedule picked by grnet_tune)" if tm else ""),
                "kernel_launches_per_forward": model.num_kernel_launches(), "exchange": "none (tracks stay on their GPU)"}

    def roofline(self, fps_per_gpu):
        model, n = self.model, self.runner.call_n[0]
        return roofline_object(fps_per_gpu, self.args.dtype, model.conv_flops_per_frame(), None, None, model.num_conv_launches(), n,
                               {"note": "frames/s per GPU of the whole step (crops + forwards of all tracks) x the convolution FLOPs of a frame"},
                               executed_flops_per_frame=model.conv_executed_flops_per_frame(n))

    def extras(self, line):
        """parity: the overlapped, fixed-buffer loop against crop + forward of each track one after another through the allocating host API."""
        torch, model = self.torch, self.model
        res = self.runner.step()
        torch.cuda.synchronize()
        worst = 0.0
        for k, box in enumerate(self.boxes):
            crop = model.crop_normalise(self.raw[k], torch.as_tensor(box), scale=1.1)
            ref = model(crop.unsqueeze(0))[-1]
            for name in ("theta", "kp_3d", "verts"):
                a, b = res[k][name].reshape(-1), ref[name].reshape(-1)
                worst = max(worst, float((a - b).abs().max() / b.abs().max()))
        line["parity"] = {"max_rel_err_vs_sequential_calls": worst, "of": "theta, kp_3d, verts of every track: overlapped fixed-buffer loop vs crop_normalise + forward per track"}

    def close(self):
        self.model.close()


def run_rank(args, make_workload=GpuWorkload, out=None):
    """Everything one rank does; returns the JSON object on rank 0 (None elsewhere).  `make_workload` is the seam the CPU
    tests use to drive the rank logic (barrier, MAX over ranks, rank-0 line) with a stand-in workload under gloo."""
    world, rank, local_rank = rank_env()
    dist, local_rank, reduce_device = init_dist(world, rank, local_rank)
    batchgen = getattr(args, "workload", "clip") == "batchgen"
    if batchgen and make_workload is GpuWorkload:
        make_workload = BatchgenWorkload
    tracks = getattr(args, "workload", "clip") == "tracks"
    if tracks and make_workload is GpuWorkload:
        make_workload = TracksWorkload
    wl = make_workload(args, world, rank, local_rank, dist)
    elapsed = timed_steps(wl.step, wl.sync, args.steps, args.warmup, dist, reduce_device)
    line = None
    n = args.tracks * args.track_frames if tracks else args.frames
    total_frames = (args.total_frames if batchgen else n * world) * args.steps
    fps = total_frames / elapsed
    roof = wl.roofline(fps / world)                            # every rank runs it (keeps the ranks in step), rank 0 reports
    proof = None
    if world > 1 and hasattr(wl, "exchange_state"):            # every rank: one more step, then the slot-by-slot check of what the collective delivered
        wl.step()
        wl.sync()
        st = wl.exchange_state()
        proof = prove_exchange(st, world, rank, dist, st.get("comm"))
    if rank == 0:
        line = {"metric": f"frames/sec (224x224, {args.total_frames}-frame video directory, all-gather before the GRU)" if batchgen else f"frames/sec (224x224, {args.tracks} tracks x seq={args.track_frames}, crop overlapped with the forward)" if tracks else f"frames/sec (224x224, seq={n})",
                "value": round(fps, 2), "unit": "frames/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "strong" if batchgen else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                "config": wl.config(), "roofline": roof}
        if proof:
            line["config"].update(proof)
        elif world == 1:
            line["config"].update(exchange_ranks=1, frames_total=total_frames // args.steps)
        wl.extras(line)
    wl.close()
    if rank == 0:
        if world == 1 and make_workload is GpuWorkload and args.dtype == "f32" and args.frames == FRAMES_PER_GPU and not args.no_secondary and not batchgen:
            line["secondary"] = secondary_legs(args)
        print(json.dumps(line), file=out or sys.stdout, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return line


def secondary_legs(args):
    """The other single-GPU-runnable BASELINE configs under the same clock as the headline, in the same process, each with its own
    config.workload / dtype / roofline (round-4 review: 5 488 and 17 750 frames/s were builder-run numbers only):
      configs[2]  8 clips x 32 frames = 256 frames per call, bf16 storage / fp32 accumulation                    (20 steps)
      configs[3]  batch_generation over ONE 10 000-frame video at 1 GPU, fp32, with its phases                   (1 warm + 2 timed jobs, ~6 s)
      configs[4]  4 person tracks x 64 frames, bf16, crop on a side stream                                       (30 steps)
    No CPU-baseline leg here; configs[2]'s line carries `parity` (ok = the bf16 bound of tests/test_gpu_bf16.py on the first 8 frames of the 256-frame
    call: two oracle passes over 8 frames, ~2 s)."""
    import copy

    def leg(make, a, frames_per_step, metric, extra=None):
        try:
            wl = make(a, 1, 0, 0, None)
            elapsed = timed_steps(wl.step, wl.sync, a.steps, a.warmup, None, "cuda")
            fps = frames_per_step * a.steps / elapsed
            obj = {"config": wl.config(), "metric": metric, "dtype": a.dtype, "value": round(fps, 2), "unit": "frames/s", "steps": a.steps, "warmup": a.warmup,
                   "ms_per_step": round(elapsed / a.steps * 1e3, 4), "roofline": wl.roofline(fps)}
            if extra:
                obj.update(extra(wl))
            wl.close()
            return obj
        except Exception as e:                                 # the headline line must not be lost to a secondary leg
            return {"metric": metric, "error": f"{type(e).__name__}: {e}"}

    base = copy.copy(args)
    base.inflight, base.light, base.tune_cache, base.no_cpu_baseline = 1, True, None, True
    a2 = copy.copy(base)
    a2.dtype, a2.frames, a2.steps, a2.warmup = "bf16", 256, 20, 5
    a3 = copy.copy(base)
    a3.workload, a3.dtype, a3.steps, a3.warmup, a3.total_frames, a3.chunk = "batchgen", "f32", 2, 1, 10000, 400
    a4 = copy.copy(base)
    a4.workload, a4.dtype, a4.steps, a4.warmup, a4.tracks, a4.track_frames, a4.call_frames, a4.no_overlap = "tracks", "bf16", 30, 5, 4, 64, None, False
    return [leg(GpuWorkload, a2, 256, "frames/sec (224x224, 8 clips x seq=32)", lambda wl: {"parity": wl.parity_bf16()}),
            leg(BatchgenWorkload, a3, a3.total_frames, f"frames/sec (224x224, {a3.total_frames}-frame video directory, all-gather before the GRU)",
                lambda wl: {"scaling": "strong", "phases": wl.runner.phases()}),
            leg(TracksWorkload, a4, a4.tracks * a4.track_frames, f"frames/sec (224x224, {a4.tracks} tracks x seq={a4.track_frames}, crop overlapped with the forward)")]


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0 and args.gpus > 1:                           # no launcher around us: start the ranks as children, relay, exit
        return self_launch(argv, args.gpus)
    if world > 1 and args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
