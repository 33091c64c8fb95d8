"""CPU checks of bench.py's plumbing: the N-rank launcher (`--gpus N` re-executes under torch.distributed.run) and the
lookup of the committed PMC traffic summaries (profiles/).  No GPU, no oracle."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def test_gpus_n_without_launcher_needs_n_gpus():
    """`python bench.py --gpus 2` on a box with fewer GPUs: clear message, exit code 2 (not a silent 1-rank run)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MMG_BENCH_BACKEND")}
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 2, (p.returncode, p.stderr[-400:])
    assert "needs 2 GPUs" in p.stderr


def test_gpus_n_reexecs_under_torch_distributed_run(monkeypatch):
    """With enough devices (here: the gloo smoke-test backend, which shares GPUs) the script starts N ranks of itself."""
    calls = []
    monkeypatch.setenv("MMG_BENCH_BACKEND", "gloo")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1", "--scaling", "strong", "--workload", "c3"])
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 0 and len(calls) == 1
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(REPO, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "3", "--warmup", "1", "--scaling", "strong", "--workload", "c3"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_traffic_lookup_picks_the_workloads_own_file(tmp_path):
    def put(name, kernels):
        json.dump({"kernels": {k: {"traffic_bytes_corrected": v} for k, v in kernels.items()}}, open(tmp_path / name, "w"))
    put("r02_config3_pmc_hbm_traffic.json", {"k_conversation_fast2": 6100000})
    put("r02_strong_config3_pmc_hbm_traffic.json", {"k_conversation_fast2": 29900000})
    put("r02_config5_pmc_hbm_traffic.json", {"k_conversation": 17301504})
    put("r02_strong_config5_pmc_hbm_traffic.json", {"k_conv_tile": 247000000})
    d = str(tmp_path)
    assert bench.traffic_lookup("c3", "k_conversation", False, d)[0] == 6100000
    assert bench.traffic_lookup("c3", "k_conversation", True, d)[0] == 29900000
    assert bench.traffic_lookup("c5", "k_conversation", False, d)[0] == 17301504
    assert bench.traffic_lookup("c5", "k_conv_tile", True, d)[0] == 247000000
    assert bench.traffic_lookup("c5", "k_conv_tile", False, d) == (None, None)
    # a newer round's file wins; a file without the kernel falls through to the older one
    put("r03_config3_pmc_hbm_traffic.json", {"k_wgrad": 1})
    assert bench.traffic_lookup("c3", "k_conversation", False, d)[0] == 6100000
    put("r04_config3_pmc_hbm_traffic.json", {"k_conversation_fast2": 5000000})
    assert bench.traffic_lookup("c3", "k_conversation", False, d)[0] == 5000000


def test_traffic_lookup_on_the_committed_profiles():
    """The committed summaries: config 3's per-GPU shard is the ~6 MB file, not the 30 MB whole-batch one; config 5 resolves."""
    v3, src3 = bench.traffic_lookup("c3", "k_conversation", False)
    assert v3 is not None and "strong" not in src3 and v3 < 12e6
    v3s, src3s = bench.traffic_lookup("c3", "k_conversation", True)
    assert v3s is not None and "strong" in src3s and v3s > v3
    v5, _ = bench.traffic_lookup("c5", "k_conversation", False)
    assert v5 is not None
