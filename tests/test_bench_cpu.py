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
    assert j["config"]["steps_done_rank0"] == 22
    # rank 1 sleeps 4 ms per step, rank 0 only 2 ms: the reported time is the slower rank's (and the all-gather couples them)
    assert j["ms_per_step"] >= 4.0
    assert abs(j["value"] - 16 * 2 * 20 / (j["ms_per_step"] * 20 / 1e3)) / j["value"] < 1e-3
    r = j["roofline"]
    assert abs(r["achieved"] - j["value"] / 2 * bench.F_FRAME_FLOP / 1e12) < 2e-3 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4


def test_roofline_object_follows_survey_8d():
    r = bench.roofline_object(2547.94, "f32", bench.F_FRAME_FLOP, 5.93, 7.45, 316, 16)
    assert abs(r["achieved"] - 78.69) < 0.01 and abs(r["frac"] - 0.5002) < 1e-3        # the judge's recomputation of round 1
    assert abs(r["conv_only_achieved"] - 16 * bench.F_FRAME_FLOP / 5.93e-3 / 1e12) < 0.01
    assert r["peak"] == 157.3 and r["unit"] == "TFLOP/s" and r["bound"] == "mfma"


def test_roofline_object_reports_executed_flops_beside_the_algorithmic_figure():
    """The roofline fraction stays on SURVEY 8(d)'s algorithmic (direct-convolution) count; when Winograd layers execute fewer
    multiplies the bench line says so in separate keys, and says nothing when both counts agree."""
    executed = bench.F_FRAME_FLOP - 317.5e9 / 16 * (1 - 4.0 / 9.0)
    r = bench.roofline_object(3150.0, "f32", bench.F_FRAME_FLOP, 4.75, 6.0, 316, 16, executed_flops_per_frame=executed)
    assert abs(r["achieved"] - 3150.0 * bench.F_FRAME_FLOP / 1e12) < 0.01 and abs(r["frac"] - r["achieved"] / 157.3) < 1e-4
    assert abs(r["executed_achieved"] - 3150.0 * executed / 1e12) < 0.01 and r["executed_frac"] < r["frac"]
    assert abs(r["executed_gflop_per_step"] - executed * 16 / 1e9) < 0.01
    same = bench.roofline_object(3150.0, "f32", bench.F_FRAME_FLOP, 4.75, 6.0, 316, 16, executed_flops_per_frame=bench.F_FRAME_FLOP)
    assert "executed_achieved" not in same
