"""Helper of tests/test_bench_cpu.py::test_n_gt_1_line_shape_under_gloo: runs bench.main() as one rank of a gloo job with
run_workload() replaced by a stub that performs the collectives a real rank would (so a rank that skipped or reordered a
workload would hang the test) and returns a plausible result -- the line ASSEMBLY for N > 1 is what is under test."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402

CALLS = []


def stub(workload, steps, warmup, seed, rank, world, local_rank, strong=False, want_roofline=True):
    CALLS.append((workload, strong))
    t = torch.tensor([float(len(CALLS))])
    dist.all_reduce(t)                                   # every rank must be here with the same call count
    assert t.item() == world * len(CALLS), (t.item(), CALLS)
    CFG, B_weak, label = bench.WORKLOADS[workload]
    Bg = bench.STRONG_GLOBAL_BATCH[workload] if strong else B_weak * world
    B = Bg // world
    return dict(workload=workload, label=label, B=B, Bg=Bg, elapsed=2.0, minibatches=steps * 100, ex_steps=steps * 100 * 8.0,
                sample_steps=steps * 100 * 8.0 * Bg * 0.5, cfg=CFG, roofline=None,
                collective="torch.distributed.all_reduce, backend gloo",
                collective_us={"stats_f64_allreduce_us": 11.0, "grads_f32_allreduce_us": 22.0},
                first_pass_ms_per_minibatch=0.07, first_pass_steps_per_minibatch=7.0, first_pass_ex_steps=steps * 7.0,
                first_pass_seconds=steps * 7e-5, dist_world=world)


bench.run_workload = stub
bench.main()
