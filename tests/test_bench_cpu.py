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


def test_rocprof_lookup_parses_the_newest_summary(tmp_path):
    def put(name, rows):
        with open(tmp_path / name, "w") as f:
            f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n')
            for n, c, tot in rows:
                f.write('"%s",%d,%d,%f,1.0,1,1,0\n' % (n, c, tot, tot / c))
    put("r04_config2_kernel_stats.csv", [("void mmg::k_conversation_fast3<256, 32, 64, 100, true>(mmg::Dims)", 100, 2600000),
                                         ("mmg::k_opt(mmg::JobTable const*)", 100, 500000)])
    put("r05_config2_kernel_stats.csv", [("void mmg::k_conversation_fast3<256, 32, 64, 100, true>(mmg::Dims)", 10, 200000),
                                         ("void mmg::k_bwd_conv_fast<256, 32, 64, 100, 30, true, true>(mmg::Dims)", 10, 150000),
                                         ("mmg::k_wgrad(mmg::JobTable const*)", 10, 100000), ("mmg::k_wreduce(mmg::JobTable const*)", 10, 30000),
                                         ("mmg::k_opt(mmg::JobTable const*)", 10, 50000), ("__amd_rocclr_copyBuffer", 7, 999999)])
    put("r05_strong_config3_kernel_stats.csv", [("mmg::k_wgrad(mmg::JobTable const*)", 10, 1000000), ("mmg::k_opt(mmg::JobTable const*)", 10, 50000)])
    r = bench.rocprof_lookup("c2", False, str(tmp_path))
    assert r["source"].endswith("r05_config2_kernel_stats.csv") and r["dominant"] == "k_conversation_fast3" and r["minibatches"] == 10
    assert abs(r["per_group"]["k_conversation"] - 20.0) < 1e-9 and abs(r["per_group"]["k_wgrad"] - 13.0) < 1e-9     # k_wgrad + k_wreduce: one launch group
    assert bench.rocprof_lookup("c3", True, str(tmp_path))["dominant"] == "k_wgrad"
    assert bench.rocprof_lookup("c3", False, str(tmp_path)) is None


def _committed_bench_lines():
    import glob
    import re
    out = []
    for f in glob.glob(os.path.join(REPO, "profiles", "r*_bench_line*.json")):
        m = re.match(r"r(\d+)_", os.path.basename(f))
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        if m and isinstance(d.get("roofline"), dict) and d["roofline"].get("rocprof_source"):
            out.append((int(m.group(1)), f, d))
    return sorted(out)


def test_roofline_of_the_committed_bench_line_follows_from_the_committed_rocprof_summary():
    """The newest committed bench line names its dominant kernel from a committed rocprofv3 summary, and `frac_rocprof` is that
    kernel's operand bytes (the line's own `algorithmic_amount`) / the summary's per-minibatch average / peak: recomputed here
    from the CSV the line cites.  sec8d_bytes_per_minibatch is SURVEY.md 8(d)'s 4 (B F + D V + B) + 24 P."""
    lines = _committed_bench_lines()
    if not lines:
        pytest.skip("no committed bench line carries roofline.rocprof_source yet")
    _, path, d = lines[-1]
    rf = d["roofline"]
    src = os.path.join(REPO, rf["rocprof_source"])
    assert os.path.exists(src), "the line cites %s, which is not committed" % rf["rocprof_source"]
    w = "c2"
    import re
    num = re.search(r"config(\w+?)_kernel_stats", os.path.basename(src)).group(1)
    assert num == "2", "the default line is configs[1]"
    rp = bench.rocprof_lookup(w, False, os.path.dirname(src))
    assert bench.GROUP_OF[rp["dominant"]] == rf["kernel"] and rp["dominant"] == rf["rocprof_kernel"]
    us = rp["per_group"][rf["kernel"]]
    assert abs(us - rf["rocprof_avg_us"]) < 1e-6 * max(1.0, us)
    scale, peak = (1e9, bench.HBM_PEAK_GBS) if rf["bound"] == "hbm" else (1e12, bench.MFMA_F32_PEAK_TFLOPS)
    want = rf["algorithmic_amount"] / (us * 1e-6) / scale / peak
    assert abs(want - rf["frac_rocprof"]) <= 1e-9 + 1e-6 * want
    assert rf["peak"] == peak and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["sec8d_bytes_per_minibatch"] == 4 * (64 * 512 + 30 * 100 + 64) + 24 * 384180
    assert abs(rf["step_frac"] - rf["sec8d_bytes_per_minibatch"] / (d["ms_per_step"] * 1e-3) / 1e9 / bench.HBM_PEAK_GBS) < 1e-9
    if rf.get("traffic") and rf["bound"] == "hbm":
        assert abs(rf["traffic_ratio"] - rf["traffic"] / rf["algorithmic_amount"]) < 1e-9
    assert "value_first_pass" in d and d["value_first_pass"] > 0


def test_sec8d_bytes_and_algorithmic_work_conventions():
    assert bench.sec8d_bytes_per_minibatch(bench.C2, 64, 384180) == 4 * (64 * 512 + 3000 + 64) + 24 * 384180      # 9.36 MB: SURVEY.md 8(d)
    # the baselines' hidden tiles (2 K floats per live row) are the implementation's tape, not algorithmic bytes
    b1 = bench.algorithmic_work("k_bwd_conv", bench.C2, 64, 3.0)[1]
    b2 = bench.algorithmic_work("k_bwd_conv", bench.C2, 64, 4.0)[1]
    per_row = (b2 - b1) / 64 / 4
    assert per_row == 2 * 256 + 6 * 32 + 13 * 64 + 30 + 100 + 16


def test_n_gt_1_line_shape_under_gloo(tmp_path):
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one JSON line on rank 0): the weak configs[1] figure AND
    the strong-scaling figures of configs[2] (512) / configs[4] (2048) with rccl_world and per-collective us in the SAME line."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, MMG_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(REPO, "tests", "bench_stub_driver.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["rccl_world"] == 2 and d["config"]["global_batch"] == 128
    # value = the exact --steps pass (2 ranks x 4 x 7.0 exchange steps in 4 x 70 us); the long window is reported beside it
    assert abs(d["value"] - 2 * (4 * 7.0) / (4 * 7e-5)) < 1e-6 * d["value"] and d["value_first_pass"] == d["value"]
    assert d["value_window"] == 2 * 400 * 8.0 / 2.0 and d["config"]["window"]["value"] == d["value_window"]
    assert abs(d["ms_per_step"] - 0.07) < 1e-9
    assert d["config"]["collective_us"]["grads_f32_allreduce_us"] == 22.0
    sc = d["strong_configs"]
    assert set(sc) == {"c3s", "c5s"}
    assert sc["c3s"]["global_batch"] == 512 and sc["c3s"]["per_gpu_batch"] == 256 and sc["c3s"]["scaling"] == "strong" and sc["c3s"]["rccl_world"] == 2
    assert sc["c5s"]["global_batch"] == 2048 and sc["c5s"]["per_gpu_batch"] == 1024
    assert sc["c5s"]["value_window"] == 30 * 100 * 8.0 / 2.0 and abs(sc["c5s"]["value"] - 1e5) < 1e-3 and "collective_us" in sc["c5s"]
    assert "other_configs" not in d and "cpu_baseline" not in d
