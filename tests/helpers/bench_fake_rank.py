"""Stand-in rank for the CPU tests of bench.py's N > 1 path: bench.run_rank() with a workload that needs no GPU.
Launched by bench.self_launch() through `python -m torch.distributed.run` (gloo), exactly like the real ranks."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class FakeWorkload:
    """step() sleeps (rank 1 twice as long, so MAX-over-ranks is observable) and all-gathers a token like the real exchange."""

    def __init__(self, args, world, rank, local_rank, dist):
        import torch
        self.torch, self.world, self.rank, self.dist, self.n, self.steps = torch, world, rank, dist, args.frames, 0

    def step(self):
        time.sleep(0.002 * (1 + self.rank))
        if self.dist is not None:
            out = self.torch.empty(self.world, dtype=self.torch.float32)
            self.dist.all_gather_into_tensor(out, self.torch.tensor([float(self.rank)]))
            assert out.tolist() == [float(r) for r in range(self.world)]
        self.steps += 1

    def sync(self):
        pass

    def exchange_state(self):
        """As the real workloads: this rank's send block, the gathered (world, block) buffer, its frame range (16 frames per rank)."""
        torch = self.torch
        local = torch.arange(self.n, dtype=torch.float32) + 1000.0 * self.rank
        gathered = torch.empty(self.world * self.n, dtype=torch.float32)
        self.dist.all_gather_into_tensor(gathered, local)
        return {"local": local, "gathered": gathered.view(self.world, -1), "frames": (self.rank * self.n, (self.rank + 1) * self.n)}

    def config(self):
        return {"workload": "fake", "frames_per_gpu": self.n, "steps_done_rank0": self.steps}

    def roofline(self, fps_per_gpu):
        return bench.roofline_object(fps_per_gpu, "f32", bench.F_FRAME_FLOP, None, None, 316, self.n)

    def extras(self, line):
        pass

    def close(self):
        pass


if __name__ == "__main__":
    os.environ["GRNET_BENCH_BACKEND"] = "gloo"
    bench.run_rank(bench.parse_args(sys.argv[1:]), make_workload=FakeWorkload)
