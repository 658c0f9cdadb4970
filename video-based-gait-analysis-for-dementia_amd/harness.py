"""Host harness of the per-frame path: frame sharding across the GPUs of one node, the one
all-gather that reassembles the pose sequence, and the fixed-buffer clip runner the bench uses.

Reference counterparts: the model loops of ``demo.py:126-188`` and
``batch_generation.py:289-329`` (single process, single device there).  Frames are independent in
the reference's runnable configuration (``grnet.py:136-152``), so a clip shards into contiguous
frame ranges with no data-path collective; the only exchange is the all-gather of the per-frame
results before anything temporal (the GRU gait encoder) runs on the whole sequence.
"""
import ctypes as C

import torch

from . import _lib

# per-frame pose record gathered across ranks: name -> floats per frame
POSE_RECORD = (("theta", 85), ("kp_3d", 87), ("kp_2d", 58), ("point_local_feat", 3072))
POSE_RECORD_FLOATS = sum(s for _, s in POSE_RECORD)
# with the temporal branch (use_gait_feat, grnet.py:154-173) the second head pass also needs the first pass's cam_shape_feats
POSE_RECORD_GAIT = POSE_RECORD + (("cam_shape_feats", 1536),)


def shard_range(n_total, world, rank):
    """Contiguous frame range [lo, hi) of ``rank``: ceil(n/world) frames each, last ranks may be short/empty."""
    per = -(-n_total // world)
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


def plan_work_items(lengths, world, chunk=128):
    """Work items of a directory of videos for ``world`` ranks: every video is cut into runs of <= ``chunk`` consecutive frames
    (a short clip is ONE item: it is never split below the size at which the kernels run efficiently), and the items go, in
    order, to the rank with the fewest frames so far (ties: lowest rank).  Deterministic, so every rank computes the same plan
    and nobody communicates it.  Returns [(video index, lo, hi, rank)] in (video, lo) order."""
    load = [0] * world
    items = []
    for vi, n in enumerate(lengths):
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            r = min(range(world), key=lambda k: (load[k], k))
            load[r] += hi - lo
            items.append((vi, lo, hi, r))
    return items


def gather_work_items(items, local_rows, row_floats, world, rank, dist, device, comm=None):
    """ONE all-gather for a whole window of videos: every rank contributes the rows (frames) of its items, concatenated in item
    order and padded to the largest per-rank count; returns {video index: (n_frames_of_the_video, row_floats) tensor} in frame
    order (on every rank).  ``local_rows``: (count_of_this_rank, row_floats) tensor on ``device``."""
    counts = [0] * world
    for _, lo, hi, r in items:
        counts[r] += hi - lo
    assert local_rows.shape[0] == counts[rank], (local_rows.shape, counts, rank)
    cap = max(max(counts), 1)
    send = torch.zeros(cap, row_floats, dtype=torch.float32, device=device)
    send[:counts[rank]] = local_rows
    if world == 1:
        recv = send.unsqueeze(0)
    else:
        flat = torch.empty(world * cap, row_floats, dtype=torch.float32, device=device)
        _all_gather(flat, send, dist, comm)
        recv = flat.view(world, cap, row_floats)
    cursor = [0] * world
    per_video = {}
    for vi, lo, hi, r in items:
        per_video.setdefault(vi, []).append(recv[r, cursor[r]:cursor[r] + hi - lo])
        cursor[r] += hi - lo
    return {vi: torch.cat(parts, 0) for vi, parts in per_video.items()}


def _host_channel_tensor(values, dist, device):
    """uint8 tensor the launcher's process group can move: device memory on the nccl backend (RCCL moves nothing else), host memory otherwise."""
    t = torch.tensor(list(values), dtype=torch.uint8)
    return t.to(device) if dist.get_backend() == "nccl" else t


def agree_all_ok(ok, dist, device):
    """True when EVERY rank passed ok=True (a MIN over the launcher's process group): the ranks take the same branch afterwards."""
    t = _host_channel_tensor([1 if ok else 0], dist, device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.cpu()[0]))


def share_unique_id(raw, dist, device, status=0):
    """Rank 0's 128-byte communicator id to every rank over the launcher's process group, with a status byte in front: rank 0 ALWAYS enters this
    broadcast -- with status != 0 and a zeroed id when it could not draw one -- so the other ranks never wait for a sender that has left.
    ``raw``: the id on rank 0, any 128 bytes elsewhere.  Returns (status, id bytes) as rank 0 sent them."""
    t = _host_channel_tensor([status & 0xff] + list(bytearray(raw)), dist, device)
    dist.broadcast(t, src=0)
    got = bytes(t.cpu().tolist())
    return got[0], got[1:]


class RcclComm:
    """The C ABI's own RCCL communicator (grnet_comm_*, include/grnet_hip.h): the exchange then needs PyTorch only for the device buffers.

    Bootstrap, the same steps on every rank (no rank waits in a collective another one skips):
      1. grnet_comm_probe -- local: can RCCL be bound in this process? -- and a MIN over the ranks; if any rank cannot, EVERY rank raises here;
      2. rank 0 draws the 128-byte id (grnet_comm_unique_id) and broadcasts status + id over the launcher's process group (any backend: it carries
         129 bytes over the host); rank 0 enters the broadcast even when drawing failed (status 1, zeroed id) and then every rank raises;
      3. every rank enters grnet_comm_create (collective inside RCCL: it returns when all ranks have arrived; a rank whose ncclCommInitRank fails
         after that point is RCCL's to report on the others -- nothing above RCCL can unblock them).
    ``all_gather`` enqueues ONE ncclAllGather on the caller's stream.  RCCL wants one GPU per rank: two ranks sharing a device (the gloo
    rehearsals) cannot use this class.
    """

    def __init__(self, world, rank, device, dist=None, unique_id=None):
        self._lib = lib = _lib.load()
        self.world, self.rank, self.device = world, rank, torch.device(device)
        self._h = None
        if unique_id is None:
            if world > 1 and dist is None:
                raise ValueError("RcclComm: pass the launcher's torch.distributed (or the 128-byte unique_id of rank 0) when world > 1")
            probe = lib.grnet_comm_probe()
            why = "" if probe == 0 else lib.grnet_comm_last_error().decode()
            if world > 1 and not agree_all_ok(probe == 0, dist, self.device):
                raise RuntimeError("RcclComm: RCCL cannot be bound on every rank" + (f" (this rank: {why})" if why else " (this rank could)"))
            if world == 1 and probe != 0:
                _lib.check_comm(lib, probe, "grnet_comm_probe")
            buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
            status = 0
            if rank == 0:
                status = 0 if lib.grnet_comm_unique_id(buf, _lib.COMM_ID_BYTES) == 0 else 1
                why = lib.grnet_comm_last_error().decode() if status else ""
            if world > 1:
                status, unique_id = share_unique_id(buf.raw if status == 0 else bytes(_lib.COMM_ID_BYTES), dist, self.device, status)
            else:
                unique_id = buf.raw
            if status:
                raise RuntimeError("RcclComm: rank 0 could not draw the communicator id" + (f": {why}" if why else ""))
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError(f"RcclComm: the unique id has {len(unique_id)} bytes, not {_lib.COMM_ID_BYTES}")
        h = C.c_void_p()
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check_comm(lib, lib.grnet_comm_create(C.byref(h), unique_id, world, rank, index), "grnet_comm_create")
        self._h = h

    def all_gather(self, recv, send, stream=None):
        """recv (world * send.numel() elements) <- every rank's ``send``, rank-major; both contiguous device tensors of one dtype."""
        assert send.is_contiguous() and recv.is_contiguous() and recv.dtype == send.dtype
        assert recv.numel() == self.world * send.numel(), (recv.shape, send.shape, self.world)
        stream = stream if stream is not None else torch.cuda.current_stream(send.device)
        rc = self._lib.grnet_allgather(self._h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                                       send.numel() * send.element_size(), C.c_void_p(stream.cuda_stream))
        _lib.check_comm(self._lib, rc, "grnet_allgather")
        return recv

    def info(self):
        """(ranks, rank) as RCCL's communicator itself reports them (grnet_comm_info -> ncclCommCount / ncclCommUserRank)."""
        nr, rk = C.c_int(), C.c_int()
        _lib.check_comm(self._lib, self._lib.grnet_comm_info(self._h, C.byref(nr), C.byref(rk)), "grnet_comm_info")
        return nr.value, rk.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.grnet_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _all_gather(out, send, dist, comm):
    """The one collective of the path: the C ABI's communicator when the caller made one, else the launcher's process group."""
    if comm is not None:
        comm.all_gather(out, send)
    else:
        dist.all_gather_into_tensor(out, send)


def pack_layout(n_local, record=POSE_RECORD):
    """Offsets (in floats) of each field inside one rank's packed block: field-major, frame-minor."""
    off, layout = 0, {}
    for name, sz in record:
        layout[name] = (off, sz)
        off += sz * n_local
    return layout, off


def gather_pose_records(packed_local, n_local, world, dist, out=None, comm=None):
    """One all-gather of every rank's packed block (RCCL over xGMI on GPUs -- through the C ABI's grnet_allgather when ``comm`` is an
    RcclComm, else through the launcher's process group -- gloo in CPU tests).

    ``packed_local``: 1-D tensor of ``n_local * POSE_RECORD_FLOATS`` floats laid out by ``pack_layout``;
    every rank must pass the same ``n_local`` (pad the last shard).  Returns the (world, block) tensor.
    """
    block = packed_local.numel()
    if out is None:
        out = torch.empty(world * block, dtype=packed_local.dtype, device=packed_local.device)
    if world == 1:
        out.copy_(packed_local)
    else:
        _all_gather(out, packed_local, dist, comm)
    return out.view(world, block)


def unpack_sequence(gathered, n_local, n_total, record=POSE_RECORD):
    """(world, block) -> dict of whole-sequence tensors (n_total, ...) in frame order."""
    layout, _ = pack_layout(n_local, record)
    world = gathered.shape[0]
    seq = {}
    for name, (off, sz) in layout.items():
        seq[name] = gathered[:, off:off + sz * n_local].reshape(world * n_local, sz)[:n_total]
    seq["theta"] = seq["theta"].reshape(-1, 85)
    seq["kp_3d"] = seq["kp_3d"].reshape(-1, 29, 3)
    seq["kp_2d"] = seq["kp_2d"].reshape(-1, 29, 2)
    seq["point_local_feat"] = seq["point_local_feat"].reshape(-1, 128, 24)
    if "cam_shape_feats" in seq:
        seq["cam_shape_feats"] = seq["cam_shape_feats"].reshape(-1, 64, 24)
    return seq


def temporal_after_gather(model, seq, bbox, cimg, b, t):
    """BASELINE configs[3]: after the all-gather every rank holds the whole sequence's first-pass records; the temporal branch
    (cparams, GRU gait encoder, attention block, second head pass -- grnet.py:154-173) needs all frames of a clip at once, so it runs
    here, on the reassembled sequence (replicated: ~0.04 s for 10 000 frames against ~0.4 s of per-frame work per GPU).
    ``seq``: unpack_sequence(..., record=POSE_RECORD_GAIT); bbox (b,t,4), cimg (b,t,2)."""
    return model.gait_correct(seq["point_local_feat"], seq["cam_shape_feats"], seq["theta"], bbox, cimg, b, t)


class ClipRunner:
    """Runs the hot path over one resident shard of frames with fixed device buffers.

    Fixed pointers let the library replay ONE captured hipGraph per step (GRNET_OPT_USE_GRAPH); the
    small per-frame results are written straight into the packed block that the all-gather sends, so
    the exchange needs no packing kernel.  ``verts`` / ``rotmat`` stay sharded (rank-local).
    """

    def __init__(self, model, frames, use_graph=True, world=1, rank=0, dist=None, tune_level=1, tune_cache=None, record=POSE_RECORD, comm=None):
        self.model, self.frames, self.world, self.rank, self.dist, self.comm = model, frames.contiguous(), world, rank, dist, comm
        self.n = n = frames.shape[0]
        self.record = record
        dev = frames.device
        layout, block = pack_layout(n, record)
        self.packed = torch.zeros(block, dtype=torch.float32, device=dev)
        self.verts = torch.empty(n, 6890, 3, dtype=torch.float32, device=dev)
        self.rotmat = torch.empty(n, 24, 3, 3, dtype=torch.float32, device=dev)
        self.gathered = torch.empty(world * block, dtype=torch.float32, device=dev)
        self.out = _lib.Outputs()
        for name, (off, sz) in layout.items():
            setattr(self.out, name, self.packed[off:off + sz * n].data_ptr())
        self.out.verts = self.verts.data_ptr()
        self.out.rotmat = self.rotmat.data_ptr()
        model.finalize()
        if use_graph:
            model.set_option(_lib.OPT_USE_GRAPH, 1)
        if tune_level:                     # launch configurations AND schedule (graph replay / eager lane streams) measured on this GPU
            model.tune(n, level=tune_level, cache=tune_cache)
        self._lib, self._h = model._lib, model._h
        self._stream = torch.cuda.current_stream(dev)

    def step(self):
        rc = self._lib.grnet_forward(self._h, C.c_void_p(self.frames.data_ptr()), self.n, C.byref(self.out),
                                     C.c_void_p(self._stream.cuda_stream))
        _lib.check(self._lib, self._h, rc, "grnet_forward")
        if self.world > 1:
            gather_pose_records(self.packed, self.n, self.world, self.dist, out=self.gathered, comm=self.comm)

    def sequence(self, n_total=None):
        """Whole-clip results in frame order (after step())."""
        g = self.gathered.view(self.world, -1) if self.world > 1 else self.packed.view(1, -1)
        return unpack_sequence(g, self.n, n_total if n_total is not None else self.n * self.world, self.record)


class ShardedSequenceRunner:
    """BASELINE configs[3] as one job: a sequence of ``n_total`` frames (a 10 000-frame video directory) sharded over the ranks
    (shard_range: ceil(n/world) frames each), every rank running the per-frame path on its shard in calls of <= ``chunk`` frames
    (batch_generation.py:289-329; the reference feeds a video in calls of >= 400 frames -- MAX_seqlen, batch_generation.py:34,303 -- and bench.py / batch_generation.py
    here default to 400: 5 800 / 5 880 / 5 930 frames/s at 128 / 256 / 400 frames per call; --max_frames sizes the activation arena for it),
    ONE all-gather of the per-frame records (theta, kp_3d, kp_2d, point_local_feat, cam_shape_feats: 19.4 KB per frame, written by
    the kernels straight into the send block -- every call's output pointers aim at its frames' slots, no packing kernel), then
    the temporal branch (grnet.py:154-173: cparams, GRU gait encoder, corrector + attention block, second head pass) on the whole
    reassembled sequence, replicated on every rank (temporal_after_gather).  Strong scaling: the job is fixed, the shards shrink.

    ``frames``: this rank's shard, (count, 3, 224, 224) on the device, count = hi - lo of shard_range.  ``forward_chunk(lo, hi, out)``
    and ``temporal(seq)`` are the seams of the CPU tests (a stand-in model under gloo); by default they call grnet_forward /
    temporal_after_gather."""

    def __init__(self, model, frames, n_total, world=1, rank=0, dist=None, chunk=128, forward_chunk=None, temporal=None, bbox=None, cimg=None,
                 comm=None):
        self.model, self.frames, self.n_total, self.world, self.rank, self.dist, self.comm = model, frames, int(n_total), world, rank, dist, comm
        self.n_local = -(-self.n_total // world)
        lo, hi = shard_range(self.n_total, world, rank)
        self.count = hi - lo
        assert frames.shape[0] == self.count, (frames.shape, lo, hi)
        dev = frames.device
        self.record = POSE_RECORD_GAIT
        layout, block = pack_layout(self.n_local, self.record)
        self.packed = torch.zeros(block, dtype=torch.float32, device=dev)       # a short last shard leaves its padding rows zero
        self.gathered = torch.empty(world * block, dtype=torch.float32, device=dev)
        self.calls = []
        for c0 in range(0, self.count, chunk):
            c1 = min(self.count, c0 + chunk)
            out = _lib.Outputs()
            for name, (off, sz) in layout.items():
                setattr(out, name, self.packed[off + sz * c0:off + sz * c1].data_ptr())
            self.calls.append((c0, c1, out))
        self.bbox = bbox if bbox is not None else torch.tensor([112.0, 112.0, 224.0, 224.0], device=dev).repeat(1, self.n_total, 1)
        self.cimg = cimg if cimg is not None else torch.full((1, self.n_total, 2), 112.0, device=dev)
        self._forward_chunk = forward_chunk or self._grnet_forward
        self._temporal = temporal or (lambda seq: temporal_after_gather(self.model, seq, self.bbox, self.cimg, 1, self.n_total))
        self.result = None
        # phase marks of the latest step (per-frame path / exchange / temporal branch): four events on the caller's stream, read by phases()
        self._marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if dev.type == "cuda" else None

    def phases(self):
        """{per_frame_ms, gather_ms, temporal_ms} of the latest step (device time between events on the step's stream; synchronises).
        per_frame = this rank's forward calls, gather = the one all-gather + unpacking, temporal = cparams + GRU + corrector / attention
        block + second head pass on the whole sequence (replicated on every rank: the serial tail of the strong-scaling job)."""
        if self._marks is None or self.result is None:
            return None
        self._marks[3].synchronize()
        t = [self._marks[i].elapsed_time(self._marks[i + 1]) for i in range(3)]
        return {"per_frame_ms": round(t[0], 3), "gather_ms": round(t[1], 3), "temporal_ms": round(t[2], 3)}

    def _grnet_forward(self, c0, c1, out):
        m = self.model
        stream = C.c_void_p(torch.cuda.current_stream(self.frames.device).cuda_stream)
        rc = m._lib.grnet_forward(m._h, C.c_void_p(self.frames[c0:c1].data_ptr()), c1 - c0, C.byref(out), stream)
        _lib.check(m._lib, m._h, rc, "grnet_forward")

    def step(self):
        mark = (lambda i: self._marks[i].record()) if self._marks is not None else (lambda i: None)
        mark(0)
        for c0, c1, out in self.calls:
            self._forward_chunk(c0, c1, out)
        mark(1)
        g = gather_pose_records(self.packed, self.n_local, self.world, self.dist, out=self.gathered, comm=self.comm)
        self.seq = unpack_sequence(g, self.n_local, self.n_total, self.record)
        mark(2)
        self.result = self._temporal(self.seq)
        mark(3)
        return self.result


class OverlappedTrackRunner:
    """BASELINE configs[4] on one GPU: several person tracks of one video, each ``(raw uint8 frames (t,H,W,3) resident on the device, boxes (t,4))``.
    Frames are independent on this path, so tracks are packed -- in order, whole -- into forward calls of <= ``call_frames`` frames (default: as many as
    the handle's arena takes; 4 x 64 frames are ONE 256-frame call, where the kernels are 1.3x more efficient than at 64).  Per call: the boxes' crop maps
    go up (80 B per frame, from pinned memory) and the crop + normalise kernel (row f1, OpenCV's arithmetic, grnet_crop_normalise_cv_maps, one launch per
    track) fills one of TWO fixed crop buffers on a side HIP stream while the forward of the previous call (a replayed hipGraph when the tuned schedule
    says so: fixed crop buffers and per-call output blocks make the pointers repeat, grnet.cpp's cache is keyed by them) runs on the caller's stream;
    events order the hand-over in both directions, and the first call of the NEXT step is staged under the last forward of this one (the runner cycles
    over the same tracks, as a bench does; a video loop would hand in the next tracks).  Same loop as pipeline.run_tracks_overlapped (demo.py:126-188 per
    person), minus every per-step allocation.  Video decode is the host's (ffmpeg / PIL: out of scope)."""

    def __init__(self, model, raw, boxes, scale=1.1, use_graph=True, tune_level=1, overlap=True, call_frames=None):
        from .pipeline import cv_crop_maps
        self.model, self.raw, self.overlap = model, [r.contiguous() for r in raw], overlap
        dev = self.dev = raw[0].device
        self.t = [int(r.shape[0]) for r in raw]
        cap = int(call_frames or model.max_frames)
        if max(self.t) > cap or cap > model.max_frames:
            raise ValueError(f"a track has {max(self.t)} frames, a call takes {cap}, the handle {model.max_frames}")
        self.calls, cur = [], []                                                                       # [[(track, offset in the call)], ...]
        for k, t in enumerate(self.t):
            if cur and cur[-1][1] + self.t[cur[-1][0]] + t > cap:
                self.calls.append(cur)
                cur = []
            cur.append((k, cur[-1][1] + self.t[cur[-1][0]] if cur else 0))
        self.calls.append(cur)
        self.call_n = [c[-1][1] + self.t[c[-1][0]] for c in self.calls]
        n_max = max(self.call_n)
        self.maps_host = [torch.from_numpy(cv_crop_maps(b, scale)).pin_memory() for b in boxes]        # (t,10) float64 per track
        self.maps_dev = [torch.empty(n_max, 10, dtype=torch.float64, device=dev) for _ in range(2)]
        self.bufs = [torch.empty(n_max, 3, 224, 224, dtype=torch.float32, device=dev) for _ in range(2)]
        self.results, self.outs = [None] * len(self.t), []
        for c, n in zip(self.calls, self.call_n):                                                      # one output block per call; a track's results are slices of it
            res = {"theta": torch.empty(n, 85, device=dev), "kp_3d": torch.empty(n, 29, 3, device=dev), "kp_2d": torch.empty(n, 29, 2, device=dev),
                   "verts": torch.empty(n, 6890, 3, device=dev), "rotmat": torch.empty(n, 24, 3, 3, device=dev),
                   "point_local_feat": torch.empty(n, 128, 24, device=dev)}
            out = _lib.Outputs()
            for name, ten in res.items():
                setattr(out, name, ten.data_ptr())
            self.outs.append(out)
            for k, off in c:
                self.results[k] = {name: ten[off:off + self.t[k]] for name, ten in res.items()}
        model.finalize()
        if use_graph:
            model.set_option(_lib.OPT_USE_GRAPH, 1)
        if tune_level:
            for n in sorted(set(self.call_n)):
                model.tune(n, level=tune_level)
        self.side = torch.cuda.Stream(device=dev) if overlap else None
        self.ready = [torch.cuda.Event() for _ in range(2)]                                            # the crops of a call have landed in bufs[slot]
        self.free = [torch.cuda.Event() for _ in range(2)]                                             # the forward that read bufs[slot] is done
        self._lib, self._h = model._lib, model._h
        self._seq = 0                                                                                  # calls issued so far: call number -> slot = seq % 2
        self._staged = False                                                                           # the crops of call number _seq are already enqueued

    def _stage(self, c, slot, main):
        st = self.side if self.overlap else main
        with torch.cuda.stream(st):
            if self.overlap:
                st.wait_event(self.free[slot])
            for k, off in self.calls[c]:
                t, raw = self.t[k], self.raw[k]
                self.maps_dev[slot][off:off + t].copy_(self.maps_host[k], non_blocking=True)
                rc = self._lib.grnet_crop_normalise_cv_maps(self._h, raw.data_ptr(), t, raw.shape[1], raw.shape[2], 0, self.maps_dev[slot][off:].data_ptr(), 0,
                                                            self.bufs[slot][off:].data_ptr(), C.c_void_p(st.cuda_stream))
                _lib.check(self._lib, self._h, rc, "grnet_crop_normalise_cv_maps")
            if self.overlap:
                self.ready[slot].record(st)

    def step(self):
        main = torch.cuda.current_stream(self.dev)
        nc = len(self.calls)
        if self.overlap and self._seq == 0:
            for e in self.free:
                e.record(main)
        for c in range(nc):
            slot = self._seq % 2
            if not self._staged:
                self._stage(c, slot, main)
            if self.overlap:
                self._stage((c + 1) % nc, 1 - slot, main)                                              # overlaps the forward below; (nc-1) + 1 = the next step's first call
                self._staged = True
                main.wait_event(self.ready[slot])
            rc = self._lib.grnet_forward(self._h, C.c_void_p(self.bufs[slot].data_ptr()), self.call_n[c], C.byref(self.outs[c]), C.c_void_p(main.cuda_stream))
            _lib.check(self._lib, self._h, rc, "grnet_forward")
            if self.overlap:
                self.free[slot].record(main)
            self._seq += 1
        return self.results
