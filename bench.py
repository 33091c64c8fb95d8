#!/usr/bin/env python
"""Benchmark of the exchange path: exchange-steps/sec on BASELINE.json config 2
(Adaptive 30-class, batch 64 per GPU, max_exchange 10, rec_w_dim 32, img_h_dim 256, rec_hidden 64).

A bench "step" is one training minibatch (conversation + losses + backward + clip + RMSprop);
the reported value counts EXCHANGE steps = iterations of the loop at model.py:801 that the
reference semantics execute (up to the step at which every sample of the global minibatch has
stopped, model.py:866), summed over the timed minibatches, divided by the wall time.  The unit is the metric's
own: one exchange step of one 64-sample batch ("bs=64").  At N GPUs the global minibatch is 64*N samples, i.e.
every exchange step of the sharded game advances N such batches, so value = N * loop iterations / wall time
(the whole-job aggregate; identical to the 1-GPU definition at N = 1).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: one rank per GPU; the global minibatch is 64*N (weak scaling), sharded along the batch axis
  (multimodalgame_amd/dist.py).  Either the caller launches the ranks (torch.distributed.run sets WORLD_SIZE / RANK /
  LOCAL_RANK) or -- plain `python bench.py --gpus N` -- this script re-executes itself under
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (launch_ranks()).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

C2 = dict(n_classes=30, feat_dim=512, h_dim=256, w_dim=32, rec_hidden=64, wv_dim=100, bas_hidden=500,
          max_exchange=10, use_binary=True, fixed_exchange=False, s_prob_prod=True, entropy_s=0.08,
          entropy_sen=0.01, entropy_rec=0.01, first_rec=0.0, optim_type="RMSprop", learning_rate=1e-4, top_k=6)
PER_GPU_BATCH = 64
# --workload: the other BASELINE.json configs (parity-test cases; the default and the reported metric is configs[1])
STRONG_GLOBAL_BATCH = {"c3": 512, "c5": 2048}    # BASELINE.json configs[2] / configs[4]
WORKLOADS = {
    "c2": (C2, 64, "configs[1]: Adaptive 30-class, batch 64 per GPU, max_exchange 10, rec_w_dim 32, img_h_dim 256, rec_hidden 64, "
                   "RMSprop; one bench step = one training minibatch"),
    "c3": (dict(C2, fixed_exchange=True), 64, "configs[2]: Fixed-exchange 30-class, global batch 512 = 64 per GPU on 8 GPUs, max_exchange 10"),
    "c4": (dict(C2, w_dim=256, h_dim=1024), 64, "configs[3]: Adaptive 30-class, batch 64, rec_w_dim 256 / img_h_dim 1024 (sample-tile MFMA kernels, co-resident receiver / sender roles in one launch)"),
    "c4r256": (dict(C2, w_dim=256, h_dim=1024, rec_hidden=256), 64, "configs[3] with rec_hidden 256 (SURVEY.md 8d C4: 'use 64 and additionally report R = 256'): wide-receiver roles over 16-unit slices on the matrix cores, kernels_rc.h"),
    "c5": (dict(C2, use_binary=False, fixed_exchange=True, n_classes=1000), 256,
           "configs[4]: 1000 classes, continuous messages, global batch 2048 = 256 per GPU on 8 GPUs (256 samples per GPU: one workgroup per sample; the sample-tile MFMA kernels take over from 1024 samples per GPU, --scaling strong at N=1)"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak


def synthetic_dataset(n_samples, n_classes, feat_dim, wv_dim, seed=1234):
    """avgpool_512 = |N(0,1)|, Target uniform over classes, desc = 0.3 N(0,1)."""
    rs = np.random.RandomState(seed)
    feats = np.abs(rs.standard_normal((n_samples, feat_dim))).astype(np.float32)
    target = rs.randint(0, n_classes, size=(n_samples,)).astype(np.int64)
    desc = (0.3 * rs.standard_normal((n_classes, wv_dim))).astype(np.float32)
    return feats, target, desc


def conversation_flops(d, B, t_steps):
    """SURVEY.md 8(d) "minimal" forward flops of the conversation launch: per (step, sample) row the sender MLP, the GRU cell,
    A = W_y1h h, the y head over D classes (3 flops per (class, r): add, relu, fma), softmax . desc, the query head and the
    stop bit; per minibatch nothing (h_x / Cd belong to k_prep; the baselines are their own launch and do not run in
    continuous mode).  C1-C3 shard: 2 * 64 * 137 k = 17.5 MFLOP per exchange step; C5's 256-sample shard: 134 MFLOP."""
    H, W, R, V, D = (d[k] for k in ("h_dim", "w_dim", "rec_hidden", "wv_dim", "n_classes"))
    per_row = 2 * W * H + 2 * H * W + 6 * R * (W + R) + 2 * R * R + 3 * D * R + 2 * D * V + 2 * V * R + 2 * R * R + 2 * R * W + 2 * R
    return B * t_steps * per_row


def algorithmic_work(kernel, d, B, t_steps, lean=False, prep_inside=False):
    """Algorithmic (minimal) work of ALL launches of `kernel` in one minibatch: (bound, amount) with amount in bytes for
    HBM-bound kernels and flops for MFMA-bound ones.  B samples, t_steps = exchange steps a sample takes on average
    (B * t_steps live (step, sample) rows).  Formulas: DESIGN.md §3.
    lean: the fused continuous-mode step (include/mmg.h: run_all_steps == 2) stores only what its backward reads.
    prep_inside: the conversation launch carries k_prep's blocks as roles: their operands count as its bytes."""
    F, H, W, R, V, K, D, T = (d[k] for k in ("feat_dim", "h_dim", "w_dim", "rec_hidden", "wv_dim", "bas_hidden",
                                             "n_classes", "max_exchange"))
    rows = B * t_steps
    p_sender = H * W + W * H + 2 * H + 2 * W
    p_recv = 3 * R * (W + R) + 6 * R + R * R + R + R * V + W * R + W + R * R + 2 * R + 2 + D * R + D * V
    mac_recv = 3 * R * W + 3 * R * R + 2 * R * R + R + D * V + R * V + W * R      # products of one receiver step of one sample
    if kernel == "k_game":                    # conversation + backward recurrence in ONE launch: the operands of both, the tape written once
        a = algorithmic_work("k_conversation", d, B, t_steps, lean=lean, prep_inside=prep_inside)[1]
        b = algorithmic_work("k_bwd_conv", d, B, t_steps, lean=lean)[1]
        fwd_tape = rows * 4 * (H + 6 * W + 6 * R + D + min(V, 32) + 12)     # (not re-read: it stays in LDS between the two recurrences)
        return "hbm", a + b - fwd_tape - 4 * (p_sender + p_recv)
    if kernel in ("k_conversation", "k_conversation_mc"):
        if lean and not d["use_binary"]:
            # lean tape: per (step, sample) the message z [W], the GRU state h [R] and gates [4R], stop bit / prob / mask /
            # bookkeeping (~8 floats); per sample the output step's logits, outp, dist and softmax [4D]; no a / c / zr / dbar /
            # g / w and no per-step y
            tape = rows * 4 * (W + 5 * R + 8) + B * 4 * 4 * D
        else:
            # floats written per live (step, sample) row: a [H]; z, pz, w, pw, zr, c [6 W]; GRU gates [4 R], h [R], g [R]; the class
            # logits [D]; softmax(y) [32] (k_conversation_fast3; the generic kernels write dbar [V] instead) and ~12 scalars
            tape = rows * 4 * (H + 6 * W + 6 * R + D + min(V, 32) + 12)
        prep = 4 * (B * F + H * F + H + D * V + 2 * R * V + 2 * D * R + H * W) if prep_inside else 0   # x, W_i, desc, y1 / w_d rows in; Cd, Dd out
        return "hbm", 4 * (p_sender + p_recv) + tape + 4 * B * H + prep
    if kernel == "k_bwd_mc":                  # continuous many-class backward: softmax in, dy out, class tables once, GRU tape in, gate gradients out
        return "hbm", 4 * (2 * B * D + 3 * D * R + rows * 11 * R + 3 * R * R)
    if kernel == "k_bwd_conv":
        # read the forward tape + write the delta tape (+ dbar = softmax(y) . desc [V]); with the register-resident kernels the
        # launch also carries the baselines' forward pass over the live rows: their weights count once, their hidden tiles
        # [2 K per row] do NOT -- they are this implementation's tape (k_wgrad's operand), not SURVEY.md 8(d)'s algorithmic bytes
        tape = rows * 4 * (2 * H + 6 * W + 13 * R + D + V + 16)
        return "hbm", 4 * (p_sender + p_recv) + tape + 4 * (K * (W + R) + K * (H + W) + 4 * K)
    if kernel == "k_conv_tile":               # sample-tile recurrence on the matrix cores (kernels_tile.h)
        sender = 2 * H * W if H * W < 65536 else 0                      # large sender MLPs run in k_send_s1 / k_send_s2
        return "mfma", 2 * rows * (mac_recv + sender)
    if kernel in ("k_conv_persist", "k_conv_split", "k_conv_rc"):   # all roles of the conversation in one launch: receiver + whole sender
        return "mfma", 2 * rows * (mac_recv + 2 * H * W)
    if kernel == "k_send_s1":
        return "mfma", 2 * rows * H * W
    if kernel == "k_send_s2":
        return "mfma", 2 * rows * W * H
    if kernel == "k_bwd_tile":                # transposed receiver products of the reverse pass
        return "mfma", 2 * rows * (W * R + 2 * R * R + 3 * R * R) + 2 * B * R * R
    if kernel == "k_send_bwd":
        return "mfma", 2 * rows * W * H
    if kernel == "k_wgrad":                   # reduces over the live (step, sample) rows only
        TB = rows
        fl = 2 * TB * (3 * R * W + 3 * R * R + R * R + R * V + W * R + R + H * W + W * H + K * (W + R) + K + K * (H + W) + K)
        fl += 2 * B * (R * R + H * F) + 2 * D * R * V
        return "mfma", fl
    if kernel == "k_baselines":               # live rows only (k_baselines3)
        return "mfma", 2 * rows * K * (W + R + W + 2)
    if kernel.startswith("k_prep"):           # h_x GEMM + Cd
        return "mfma", 2 * B * H * F + 2 * D * R * V
    if kernel == "k_opt":                     # read w, g, state; write w, state
        p_total = ((H * F + H) + (H * W + H) + W + (W * H + W)
                   + 3 * R * (W + R) + 6 * R + (R * R + R) + R * V + (W * R + W) + (R * (R + V) + R) + 2 * (R + 1)
                   + (K * (W + R) + K + K + 1) + (K * (H + W) + K + K + 1))
        return "hbm", 4 * 5 * p_total
    if kernel == "k_stats":
        return "hbm", 5 * T * B * 16
    if kernel == "k_dC":
        return "hbm", 4 * (B * D + B * R + 2 * D * R)
    return "hbm", 0


MIN_TIMED_SECONDS = float(os.environ.get("MMG_BENCH_MIN_SECONDS", "2.0"))   # the timed region is repeated until it lasts at least this long


def traffic_lookup(workload, kernel, strong, profiles_dir=None):
    """HBM traffic (bytes per dispatch, FETCH_SIZE + WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes) of `kernel`
    from the newest committed PMC summary of THIS workload: profiles/rNN_config<N>_pmc_hbm_traffic.json for the per-GPU
    batch, profiles/rNN_strong_config<N>_pmc_hbm_traffic.json for --scaling strong (the whole global batch on one GPU).
    Counters need their own rocprofv3 passes and cannot be collected inside the bench run.  Returns (bytes | None, source)."""
    import glob
    import re
    pdir = profiles_dir or os.path.join(REPO, "profiles")
    num = {"c2": "2", "c3": "3", "c4": "4", "c5": "5", "c4r256": "4r256"}.get(workload, workload)
    pat = re.compile(r"^r(\d+)_%sconfig%s_pmc_hbm_traffic\.json$" % ("strong_" if strong else "", num))
    files = sorted((int(pat.match(os.path.basename(f)).group(1)), f) for f in glob.glob(os.path.join(pdir, "r*_pmc_hbm_traffic.json"))
                   if pat.match(os.path.basename(f)))
    alias = {"k_game": ("k_game_fast",), "k_conversation": ("k_conversation_fast3", "k_conversation_fast2", "k_conversation_mc3p", "k_conversation_mc3", "k_conversation_mc", "k_conversation"),
             "k_conversation_mc": ("k_conversation_mc3p", "k_conversation_mc3", "k_conversation_mc"), "k_bwd_conv": ("k_bwd_conv_fast", "k_bwd_conv"),
             "k_baselines": ("k_baselines3", "k_baselines4", "k_baselines2", "k_baselines")}
    for _, f in reversed(files):
        try:
            pmc = json.load(open(f))["kernels"]
        except Exception:
            continue
        for key in alias.get(kernel, (kernel,)):
            if key in pmc and pmc[key].get("traffic_bytes_corrected") is not None:
                return pmc[key]["traffic_bytes_corrected"], "%s (per dispatch of %s, committed file; not measured in this run)" % (os.path.relpath(f, REPO), key)
    return None, None


# rocprofv3 kernel function -> the launch group (library hook mmg_set_profiling brackets groups of launches under these names)
GROUP_OF = {
    "k_conversation_fast3": "k_conversation", "k_conversation_fast2": "k_conversation", "k_conversation": "k_conversation",
    "k_game_fast": "k_game",
    "k_conversation_mc3p": "k_conversation_mc", "k_conversation_mc3": "k_conversation_mc", "k_conversation_mc": "k_conversation_mc",
    "k_conv_persist": "k_conv_persist", "k_conv_tile": "k_conv_tile", "k_conv_split": "k_conv_split",
    "k_rc_persist": "k_conv_rc", "k_rc_gru": "k_conv_rc", "k_rc_heads": "k_conv_rc", "k_rc_query": "k_conv_rc", "k_rc_tail": "k_conv_rc",
    "k_send_s1": "k_send_s1", "k_send_s2": "k_send_s2",
    "k_bwd_conv_fast": "k_bwd_conv", "k_bwd_conv": "k_bwd_conv",
    "k_bwd_pre_send": "k_bwd_tile", "k_bwd_pre": "k_bwd_tile", "k_bwd_sample": "k_bwd_tile", "k_bwd_tile": "k_bwd_tile", "k_rc_bwd": "k_bwd_tile",
    "k_send_bwd": "k_send_bwd", "k_dhx": "k_send_bwd", "k_bwd_mc1": "k_bwd_mc", "k_bwd_mc2": "k_bwd_mc",
    "k_dC_tile": "k_dC", "k_dC": "k_dC", "k_wgrad": "k_wgrad", "k_wreduce": "k_wgrad",
    "k_baselines": "k_baselines", "k_baselines2": "k_baselines", "k_baselines3": "k_baselines", "k_baselines4": "k_baselines", "k_gemm_nt": "k_baselines",
    "k_prep": "k_prep+h_x", "k_stats": "k_stats", "k_bas_stats": "k_bas_stats", "k_gradnorm": "k_gradnorm", "k_opt": "k_opt",
}


def rocprof_lookup(workload, strong, profiles_dir=None):
    """Per-minibatch kernel times of THIS workload from the newest committed rocprofv3 --kernel-trace --stats summary
    (profiles/rNN_[strong_]config<N>_kernel_stats.csv, written by scripts/round_evidence.sh from the same bench command).
    Returns None or dict(source=..., per_kernel={function: us per minibatch}, per_group={launch group: us per minibatch},
    dominant=function with the largest share).  One minibatch = one k_opt dispatch."""
    import csv
    import glob
    import re
    pdir = profiles_dir or os.path.join(REPO, "profiles")
    num = {"c2": "2", "c3": "3", "c4": "4", "c5": "5", "c4r256": "4r256"}.get(workload, workload)
    pat = re.compile(r"^r(\d+)_%sconfig%s_kernel_stats\.csv$" % ("strong_" if strong else "", num))
    files = sorted((int(pat.match(os.path.basename(f)).group(1)), f) for f in glob.glob(os.path.join(pdir, "r*_kernel_stats.csv"))
                   if pat.match(os.path.basename(f)))
    for _, f in reversed(files):
        try:
            rows = list(csv.DictReader(open(f)))
        except Exception:
            continue
        per, calls = {}, {}
        for r in rows:
            m = re.search(r"mmg::(k_[A-Za-z0-9_]+)", r.get("Name", ""))
            if not m:
                continue
            per[m.group(1)] = per.get(m.group(1), 0.0) + float(r["TotalDurationNs"])
            calls[m.group(1)] = calls.get(m.group(1), 0) + int(r["Calls"])
        # one minibatch = one optimizer step: a k_opt dispatch, or (fused steps: k_wgrad<OPT> carries the optimizer) a k_wgrad one
        n_mb = calls.get("k_opt", 0) or calls.get("k_wgrad", 0)
        if not n_mb:
            continue
        per_kernel = {k: v / n_mb * 1e-3 for k, v in per.items()}
        per_group = {}
        for k, v in per_kernel.items():
            g = GROUP_OF.get(k, k)
            per_group[g] = per_group.get(g, 0.0) + v
        return dict(source=os.path.relpath(f, REPO), per_kernel=per_kernel, per_group=per_group,
                    dominant=max(per_kernel, key=per_kernel.get), minibatches=n_mb)
    return None


def sec8d_bytes_per_minibatch(d, B, n_params):
    """SURVEY.md 8(d) "bytes per optimizer step": data 4 (B F + D V + B) + parameter / optimizer traffic 4 P * 6
    (read w, write + read g, read + write the RMSprop state, write w)."""
    return 4 * (B * d["feat_dim"] + d["n_classes"] * d["wv_dim"] + B) + 24 * n_params


def event_floor_us(dev, n=200):
    """What a HIP-event pair measures with NOTHING in between, recorded in a stream of small kernels on the engine's stream
    (= torch's current stream): the bracket overhead every HIP-event kernel time of this file contains and rocprofv3's
    kernel-trace durations do not (profiles/README.md)."""
    x = torch.zeros(64, device=dev)
    evs = []
    for _ in range(n):
        x.add_(1.0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); b.record()
        evs.append((a, b))
    x.add_(1.0)
    torch.cuda.synchronize(dev)
    return float(np.median([a.elapsed_time(b) for a, b in evs]) * 1e3)


def build_batches(CFG, Bg, B, rank, n, dev):
    """n DISTINCT minibatches resident in HBM, formed as the epoch loop forms them (misc.py:257-302 order: seeded shuffle,
    consecutive slices, sorted indices inside a batch) from a synthetic file of at least n * Bg samples: no sample is seen twice
    in the first n minibatches (25 cycled minibatches let the pair overfit them within the timed window)."""
    import random
    feats, target, desc = synthetic_dataset(max(100 * CFG["n_classes"], 4 * Bg, n * Bg), CFG["n_classes"], CFG["feat_dim"], CFG["wv_dim"])
    order = list(range(feats.shape[0]))
    random.seed(11)
    random.shuffle(order)
    nb = feats.shape[0] // Bg
    xs, ts = [], []
    for i in range(n):
        idx = sorted(order[(i % nb) * Bg:(i % nb + 1) * Bg])[rank * B:(rank + 1) * B]
        xs.append(feats[idx]); ts.append(target[idx])
    return torch.from_numpy(np.stack(xs)).to(dev), torch.from_numpy(np.stack(ts)).to(dev), torch.from_numpy(desc).to(dev)


def collective_us(dp, eng, dev, n=100):
    """us per all-reduce of the two collectives of a data-parallel step (multimodalgame_amd/dist.py), enqueued back to back on
    the engine's stream exactly as the step enqueues them; every rank calls it."""
    out = {}
    for name, t in (("stats_f64_allreduce_us", eng.stats), ("grads_f32_allreduce_us", eng.flat_grads)):
        if name.startswith("stats") and not eng.use_binary:
            out[name] = None                      # continuous messages: no statistics collective
            continue
        buf = torch.zeros_like(t)
        for _ in range(5):
            dp._all_reduce(buf)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            dp._all_reduce(buf)
        torch.cuda.synchronize(dev)
        out[name] = (time.perf_counter() - t0) / n * 1e6
        out[name.replace("_us", "_bytes")] = buf.numel() * buf.element_size()
    return out


def run_workload(workload, steps, warmup, seed, rank, world, local_rank, strong=False, want_roofline=True):
    """Times `steps` training minibatches of one workload (repeated until MIN_TIMED_SECONDS); returns a dict."""
    from multimodalgame_amd.engine import Engine
    from multimodalgame_amd.dist import DataParallel
    from multimodalgame_amd.agents import init_state_dicts
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())   # (ranks may share a GPU in the gloo smoke test)
    torch.cuda.set_device(dev)
    CFG, B_weak, label = WORKLOADS[workload]
    if strong:
        Bg = STRONG_GLOBAL_BATCH[workload]
        assert Bg % world == 0, "global batch %d does not divide over %d ranks" % (Bg, world)
        B = Bg // world
    else:
        B, Bg = B_weak, B_weak * world
    eng = Engine(device=dev, batch=B, global_batch=Bg, batch_offset=rank * B, **CFG)
    # random-init agents (reference init: Xavier-normal weights, zero biases, N(0,1) code_bias; baselines torch-default
    # uniform) from a fixed seed -- identical on every rank
    eng.load_state_dicts(init_state_dicts(eng, seed=0))
    # warm-up + the timed --steps + the per-kernel pass all draw FRESH minibatches (no sample repeats before the window below)
    n_cycle = min(steps + warmup + 20, 1024)
    xs, ts, desc_d = build_batches(CFG, Bg, B, rank, n_cycle, dev)
    dp = DataParallel(eng)

    def one(i):
        k = i % n_cycle
        if world > 1:
            dp.train_step(xs[k], ts[k], desc_d, seed=seed)
        else:
            eng.train_step(xs[k], ts[k], desc_d, seed=seed)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(warmup):
        one(i)
    sync()
    totals_before = eng.tape["totals"].cpu().numpy().copy()   # device-side running sums (semantic exchange steps, ..., sample-steps)
    # ---- THE timed region of the contract: EXACTLY `steps` minibatches, none of them seen before, from the SAME start (warmup
    # minibatches after the seed-0 initialisation) in every round and on every box, bracketed by barrier + synchronize on both
    # sides.  `value` and `ms_per_step` are derived from it (round 6; rounds 2-5 reported the 2 s window below as `value`).
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + i)
    sync()
    first_elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([first_elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        first_elapsed = float(tt.item())
    totals_first = eng.tape["totals"].cpu().numpy().copy()
    first_steps = float(totals_first[0] - totals_before[0])
    done = steps
    # ---- per-kernel durations at THIS point of the trajectory (right after the first pass, before the window trains the agents
    # further): HIP events on the launch stream (library hook mmg_set_profiling; rank 0 records, every rank runs the steps:
    # collectives!); per-step launches of one kernel are summed per minibatch.  Untimed.
    kern_ms, tstar_live, floor_us = {}, None, 0.0
    if want_roofline:
        reps = min(20, steps)
        for i in range(reps):
            if rank == 0:
                eng.set_profiling(True)
            one(warmup + done + i)
            torch.cuda.synchronize(dev)
            if rank == 0:
                per = {}
                for name, ms in eng.kernel_times(max_kernels=512):
                    per[name] = per.get(name, 0.0) + ms
                for name, ms in per.items():
                    kern_ms.setdefault(name, []).append(ms)
        if rank == 0:
            eng.set_profiling(False)
            tstar_live = eng.tape["tstar"].float().mean().item() + 1.0      # live steps per sample (B * tstar live rows)
            floor_us = event_floor_us(dev)
        done += reps
        sync()
    # ---- the long window (`value_window`): passes of EXACTLY `steps` minibatches, enqueued back to back and synchronised ONCE, as
    # many as MIN_TIMED_SECONDS (2 s) needs (same count on every rank): a 20-step run of a 60 us minibatch is 1.2 ms, too short for
    # any outside sampler.  It cycles the resident minibatches, i.e. it is a training run of many epochs over them: the agents'
    # conversations lengthen (live_rows_per_sample / exchange_steps_per_minibatch are reported beside it)
    totals_w0 = eng.tape["totals"].cpu().numpy().copy()
    more = torch.tensor([max(1.0, MIN_TIMED_SECONDS / max(first_elapsed, 1e-6))], device=dev)
    if world > 1:
        dist.all_reduce(more, op=dist.ReduceOp.MAX)
    n_pass = min(int(np.ceil(more.item())), 1 << 20)
    sync()
    t0 = time.perf_counter()
    for _ in range(n_pass):
        for i in range(steps):
            one(warmup + done + i)
        done += steps
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    totals_after = eng.tape["totals"].cpu().numpy()
    timed_mb = n_pass * steps
    ex_steps = float(totals_after[0] - totals_w0[0])
    sample_steps = float(totals_after[3] - totals_w0[3])            # sum_t n_active,t over the GLOBAL minibatches
    tstar_window = eng.tape["tstar"].float().mean().item() + 1.0    # live (step, sample) rows per sample at the END of the window
    eng.check_sync()                                                # no in-launch dependency wait may have timed out
    collective, coll_us = "none (single rank)", None
    if world > 1:
        collective = ("direct RCCL communicator on the engine's stream (multimodalgame_amd/rccl.py), world %d" % dp.comm.world
                      if dp.comm is not None else "torch.distributed.all_reduce, backend %s" % dist.get_backend())
        coll_us = collective_us(dp, eng, dev)
    res = dict(workload=workload, label=label, B=B, Bg=Bg, elapsed=elapsed, minibatches=timed_mb, ex_steps=ex_steps,
               sample_steps=sample_steps, cfg=CFG, roofline=None, collective=collective, collective_us=coll_us,
               first_pass_ms_per_minibatch=1e3 * first_elapsed / steps, first_pass_steps_per_minibatch=first_steps / steps,
               first_pass_ex_steps=first_steps, first_pass_seconds=first_elapsed,
               first_pass_sample_steps=float(totals_first[3] - totals_before[3]), window_live_rows_per_sample=tstar_window,
               degraded=eng.degraded(), dist_world=(dist.get_world_size() if world > 1 else 1))
    if want_roofline and rank == 0:
        avg = {k: float(np.mean(v)) for k, v in kern_ms.items()}           # ms per minibatch, HIP events (raw brackets)
        n_params = sum(e["rows"] * max(e["cols"], 1) for e in eng.param_entries)
        lean = not CFG["use_binary"]                                # (mmg_train_step and the phased DP step both run the lean tape)
        prep_inside = not any(k.startswith("k_prep") for k in avg)         # (the register-resident path with a CU per role)
        # The dominant kernel is NAMED by the newest committed rocprofv3 summary of this workload (profiles/, stable across
        # runs and boxes); the live HIP-event brackets -- which contain event_floor_us of bracket overhead each -- only decide
        # when no summary is committed.  `achieved` = this kernel's operand bytes (or flops) / its live duration minus the
        # bracket floor; `frac_rocprof` = the same amount / the committed rocprofv3 average (tests/test_bench_cpu.py recomputes it).
        rp = rocprof_lookup(workload, strong)
        dom = None
        if rp is not None:
            g = GROUP_OF.get(rp["dominant"], rp["dominant"])
            dom = g if g in avg else None
        if dom is None:
            dom = max(avg, key=avg.get)
        bound, amount = algorithmic_work(dom, CFG, B, tstar_live, lean=lean, prep_inside=prep_inside)
        n_launch = {"k_wgrad": 1}.get(dom, 1)
        live_us = max(avg[dom] * 1e3 - floor_us * n_launch, 1e-3)
        secs = live_us * 1e-6
        if bound == "hbm":
            achieved, peak, unit, scale = amount / secs / 1e9, HBM_PEAK_GBS, "GB/s", 1e9
        else:
            achieved, peak, unit, scale = amount / secs / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s", 1e12
        # HBM traffic of that kernel from the committed PMC summary of THIS workload (counters need their own rocprofv3
        # passes and cannot be collected inside this run): per-dispatch FETCH_SIZE / WRITE_SIZE
        traffic, traffic_source = traffic_lookup(workload, dom, strong)
        per_kernel = {}
        for k, ms in avg.items():
            bk, amt = algorithmic_work(k, CFG, B, tstar_live, lean=lean, prep_inside=prep_inside)
            if amt:
                a_k = amt / (max(ms * 1e3 - floor_us, 1e-3) * 1e-6) / (1e9 if bk == "hbm" else 1e12)
                per_kernel[k] = dict(bound=bk, achieved=round(a_k, 3), unit="GB/s" if bk == "hbm" else "TFLOP/s",
                                     frac=round(a_k / (HBM_PEAK_GBS if bk == "hbm" else MFMA_F32_PEAK_TFLOPS), 5))
        s8 = sec8d_bytes_per_minibatch(CFG, B, n_params)
        per_mb_s = first_elapsed / steps                           # (= the line's ms_per_step)
        roof = dict(bound=bound, kernel=dom, achieved=achieved, peak=peak, unit=unit, frac=achieved / peak,
                    traffic=traffic, traffic_source=traffic_source,
                    traffic_ratio=(traffic / amount if (traffic and bound == "hbm" and amount) else None),
                    launch_us=live_us, launch_us_raw=avg[dom] * 1e3, event_floor_us=floor_us,
                    algorithmic_amount=amount, live_rows_per_sample=tstar_live,
                    rocprof_kernel=(rp["dominant"] if rp else None), rocprof_source=(rp["source"] if rp else None),
                    rocprof_avg_us=(rp["per_group"].get(dom) if rp else None),
                    frac_rocprof=((amount / (rp["per_group"][dom] * 1e-6) / scale / peak) if rp and rp["per_group"].get(dom) else None),
                    sec8d_bytes_per_minibatch=s8,
                    step_frac=s8 / per_mb_s / 1e9 / HBM_PEAK_GBS,
                    window_step_frac=s8 / (elapsed / timed_mb) / 1e9 / HBM_PEAK_GBS,
                    note="launch_us = HIP-event time of ALL launches of `kernel` in one minibatch, taken right after the first pass, minus "
                         "event_floor_us (an empty event bracket); kernels_us are the raw brackets; `kernel` is the launch group of "
                         "rocprof_kernel, the largest entry of rocprof_source; step_frac = SURVEY.md 8(d) bytes per optimizer step / "
                         "ms_per_step / HBM peak: the path is latency-bound (DESIGN.md section 0)",
                    kernels_us={k: round(v * 1e3, 2) for k, v in sorted(avg.items(), key=lambda kv: -kv[1])},
                    launches_per_minibatch=len(avg),
                    per_kernel=per_kernel)
        # the conversation launch against the OTHER roof as well: SURVEY.md 8(d)'s minimal forward flops of the shard over the
        # fp32 matrix peak (most of them -- the 3 B D R relu-dot of the y head -- are VALU work by nature)
        conv = next((k for k in avg if k.startswith(("k_conversation", "k_conv_", "k_game"))), None)
        if conv is not None:
            fl = conversation_flops(CFG, B, tstar_live)
            cs = max(avg[conv] * 1e3 - floor_us, 1e-3) * 1e-6
            roof["conversation_flop"] = dict(kernel=conv, flops=fl, tflops=fl / cs / 1e12, flop_frac=fl / cs / 1e12 / MFMA_F32_PEAK_TFLOPS,
                                             peak="fp32 MFMA %.1f TFLOP/s" % MFMA_F32_PEAK_TFLOPS)
        res["roofline"] = roof
    del eng
    return res


def same_initial_weights(models):
    """Loads into the oracle's modules the very weights run_workload() starts the GPU from (agents.init_state_dicts, seed 0,
    drawn in the flat buffer's tensor order), so that both legs train the same agents from the same point."""
    from multimodalgame_amd import _lib
    from multimodalgame_amd.agents import init_state_dicts

    class _Shapes(object):
        pass
    st = _Shapes()
    st.params = {a: {} for a in _lib.AGENTS}
    for e in _lib.param_table(_lib.make_config(batch=PER_GPU_BATCH, **C2)):
        st.params[e["agent"]][e["name"]] = torch.zeros((e["rows"], e["cols"]) if e["cols"] else (e["rows"],))
    sd = init_state_dicts(st, seed=0)
    for a, m in models.items():
        m.load_state_dict(sd[a])


def cpu_baseline(seconds_budget=24.0):
    """The CPU oracle (literal restatement of the reference, oracle/cpu_ref.py) timed on this host: same config, same
    synthetic data, same initial weights, RMSprop, data loading excluded.  Thread counts {1, 4, 8, 16, all} are tried and
    the best is reported."""
    from oracle import cpu_ref
    fl = cpu_ref.Flags(use_binary=True, fixed_exchange=False, max_exchange=10, batch_size=64, learning_rate=1e-4,
                       entropy_s=0.08, entropy_sen=0.01, entropy_rec=0.01, img_feat_dim=512, img_h_dim=256,
                       rec_w_dim=32, sender_out_dim=32, rec_hidden=64, wv_dim=100, baseline_hid_dim=500,
                       top_k_train=6)
    feats, target, desc = synthetic_dataset(3000, 30, 512, 100)
    desc_t = torch.from_numpy(desc)
    ncpu = os.cpu_count() or 1
    all_threads = torch.get_num_threads()
    counts = sorted(set([c for c in (1, 4, 8, 16) if c <= ncpu] + [all_threads]))
    out = {}
    for threads in counts:
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        np.random.seed(0)
        models = cpu_ref.build_agents(fl)
        same_initial_weights(models)
        opts = cpu_ref.build_optimizers(models, fl)
        steps, n_mb, t_used, s_steps = 0, 0, 0.0, 0.0
        for i in range(3 + 400):
            idx = np.arange((i % 46) * 64, (i % 46 + 1) * 64)
            x, t = torch.from_numpy(feats[idx]), torch.from_numpy(target[idx])
            t0 = time.perf_counter()
            res = cpu_ref.train_minibatch(models, opts, x, t, desc_t, fl)
            dt = time.perf_counter() - t0
            if i >= 3:
                steps += res["n_steps"]; n_mb += 1; t_used += dt
                s_steps += float(sum(float(m.sum()) for m in res["s_masks"][:res["n_steps"]]))       # sum_t n_active,t
                if t_used > seconds_budget / len(counts):
                    break
        out[threads] = dict(steps_per_s=steps / t_used, minibatches=n_mb, seconds=t_used, threads=threads, steps_per_mb=steps / max(n_mb, 1),
                            sample_steps_per_s=s_steps / t_used)
    torch.set_num_threads(all_threads)
    best = max(out.values(), key=lambda v: v["steps_per_s"])
    return dict(value=best["steps_per_s"], unit="exchange-steps/s", cores=best["threads"], kind="port",
                sample_steps_per_s=best["sample_steps_per_s"], exchange_steps_per_minibatch=best["steps_per_mb"], ms_per_minibatch=1e3 * best["seconds"] / best["minibatches"],
                sample="%d minibatches of config 1 (B=64) from the GPU leg's initial weights, %.2f exchange steps each, in %.1f s on %d thread(s); "
                       "sweep %s exchange-steps/s; host has %d logical CPUs" % (
                    best["minibatches"], best["steps_per_mb"], best["seconds"], best["threads"],
                    ", ".join("%d thr: %.1f" % (k, v["steps_per_s"]) for k, v in sorted(out.items())), ncpu))


def run_cli(seconds=2.0, epochs=None, dist_backend=None, log_dev=10 ** 9):
    """The SHIPPED training loop -- `python -m multimodalgame_amd.model`, i.e. model.run(): flag parsing, description
    pipeline, device-resident HDF5 epoch loop (misc.load_hdf5), Game.train_step per minibatch, a log line every 50 steps --
    on synthetic HDF5 / CSV / GloVe files of configs[1]'s shape (3000 train samples, 30 classes).  Returns
    exchange-steps/s of the epoch loop as model.run() itself measures it (device-synchronised, dev evaluation and
    checkpointing pushed beyond the run: the metric excludes them)."""
    import shutil
    import tempfile
    from multimodalgame_amd import flags as _flags, model as _model, misc as _misc
    tmp = tempfile.mkdtemp(prefix="mmg_cli_")
    try:
        paths = _misc.write_synthetic_dataset(os.path.join(tmp, "data"), n_classes=30, per_class=100, feat_dim=512, wv_dim=100)
        per_epoch = 3000 // 64
        if epochs is None:
            epochs = max(2, int(seconds / (per_epoch * 80e-6)))
        stats = {}
        argv = ["model.py", "-experiment_name", "bench_cli", "-log_path", os.path.join(tmp, "logs"), "-model_type", "Adaptive",
                "-batch_size", "64", "-max_exchange", "10", "-rec_w_dim", "32", "-sender_out_dim", "32", "-img_h_dim", "256",
                "-rec_hidden", "64", "-learning_rate", "1e-4", "-entropy_rec", "0.01", "-entropy_sen", "0.01", "-entropy_s", "0.08",
                "-use_binary", "-max_epoch", str(epochs), "-log_dev", str(log_dev), "-save_after", str(10 ** 9), "-exchange_samples", "0",
                "-top_k_train", "6"] + [a for k, v in paths.items() for a in ("-" + k, v)]
        if dist_backend:
            argv += ["-dist_backend", dist_backend]
        argv += os.environ.get("MMG_CLI_EXTRA", "").split()          # (experiments: scripts/cli_run.py)
        _flags.define_flags()
        _flags.FLAGS.Reset()
        _flags.FLAGS(argv)
        _flags.default_flags(argv)
        import contextlib
        with open(os.devnull, "w") as devnull, contextlib.redirect_stderr(devnull):      # (FileLogger echoes every line to stderr)
            _model.run(stats=stats)
        _flags.FLAGS.Reset()
        if log_dev < 10 ** 9:
            # the reference's DEFAULT cadence (-log_dev 1000, -batch_size_dev 50: eval_dev over the 3000 dev samples every 1000
            # steps, model.py:1545-1576): the loop INCLUDING its dev evaluations
            wall = stats["train_seconds"] + stats["eval_seconds"]
            return dict(cli_default_flags_steps_per_s=stats["exchange_steps"] / wall,
                        eval_dev_ms=1e3 * stats["eval_seconds"] / max(stats["evals"], 1), eval_dev_calls=stats["evals"],
                        eval_dev_what="model.eval_dev over 3000 dev samples, -batch_size_dev 50 (60 batches: one eval-mode launch + device-side "
                                      "reductions each, one copy to the host at the end), every 1000 training steps (-log_dev 1000, the reference's default)",
                        cli_default_flags_eval_share=stats["eval_seconds"] / wall)
        return dict(cli_steps_per_s=stats["exchange_steps"] / stats["train_seconds"],
                    cli_ms_per_minibatch=1e3 * stats["train_seconds"] / stats["minibatches"],
                    cli_minibatches=stats["minibatches"], cli_seconds=stats["train_seconds"],
                    cli_exchange_steps_per_minibatch=stats["exchange_steps"] / stats["minibatches"],
                    cli_what="model.run() of `python -m multimodalgame_amd.model -model_type Adaptive -batch_size 64 -max_exchange 10 ...` on "
                             "synthetic HDF5 (3000 samples, %d epochs), log line every 50 steps, dev evaluation / checkpoints excluded" % epochs)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script on this node (one per GPU, RCCL) under
    torch.distributed.run and return its exit code.  MMG_BENCH_BACKEND=gloo lets N ranks share the visible GPUs (smoke
    test of this path on a 1-GPU box); with the default nccl (= RCCL) backend every rank needs its own GPU."""
    import socket
    import subprocess
    backend = os.environ.get("MMG_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if backend == "nccl" and have < n:
        sys.stderr.write("bench.py: --gpus %d needs %d GPUs on this node, %d visible (one rank per GPU over RCCL; "
                         "MMG_BENCH_BACKEND=gloo shares GPUs between ranks for a smoke test)\n" % (n, n, have))
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MMG_BENCH_LAUNCHED="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short c3 / c4 / c5 runs of the default invocation")
    ap.add_argument("--cli", action="store_true", help="only time the shipped training loop (model.run() on synthetic HDF5) and print its numbers")
    ap.add_argument("--no-cli", action="store_true", help="skip the model.run() measurement of the default invocation (config.cli_steps_per_s)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c2",
                    help="c2 = BASELINE.json's metric config (default); c3/c4/c5 = the other listed configs, for reference")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="strong (c3 / c5 only): the GLOBAL minibatch is fixed (512 / 2048 samples) and sharded over the ranks; "
                         "value = global-batch exchange steps per second (SURVEY.md 8d)")
    args = ap.parse_args()
    if args.scaling == "strong" and args.workload not in STRONG_GLOBAL_BATCH:
        ap.error("--scaling strong needs --workload c3 or c5 (the configs BASELINE.json defines with a global batch)")
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))       # the ranks print the JSON line (rank 0)
    if args.cli:
        # --cli --gpus N: N ranks of the SHIPPED data-parallel epoch loop (model.run() reads WORLD_SIZE / RANK / LOCAL_RANK; every
        # 64-sample minibatch is sharded over the ranks, i.e. strong scaling of configs[1]'s batch)
        world = int(os.environ.get("WORLD_SIZE", "1"))
        out = run_cli(dist_backend=os.environ.get("MMG_BENCH_BACKEND", "nccl") if world > 1 else None)
        if int(os.environ.get("RANK", "0")) == 0:
            out.update(n_gpus=world, scaling="strong" if world > 1 else "weak")
            print(json.dumps(out))
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; reporting n_gpus=%d\n" % (args.gpus, world, world))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; MMG_BENCH_BACKEND=gloo only exists to smoke-test this code path on a 1-GPU box
        dist.init_process_group(os.environ.get("MMG_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    strong = args.scaling == "strong"
    r = run_workload(args.workload, args.steps, args.warmup, args.seed, rank, world, local_rank, strong=strong)
    if rank == 0:
        per_mb = r["elapsed"] / r["minibatches"]
        unit_batch = r["Bg"] if strong else WORKLOADS[args.workload][1]
        # weak: one unit = an exchange step of one per-GPU-sized batch, a global minibatch of B*N samples advances N of them;
        # strong: one unit = an exchange step of the fixed global batch
        value_window = (1.0 if strong else world) * r["ex_steps"] / r["elapsed"]
        value = value_first = (1.0 if strong else world) * r["first_pass_ex_steps"] / r["first_pass_seconds"]
        per_mb = r["first_pass_seconds"] / args.steps
        line = {
            "metric": ("exchange-steps/sec (whole node), 30-class Adaptive max_exchange=10 bs=64" if args.workload == "c2" else
                       "exchange-steps/sec (whole node) of reference workload %s -- NOT BASELINE.json's metric" % args.workload),
            "value": value, "value_window": value_window, "value_first_pass": value_first, "unit": "exchange-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * per_mb, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": r["label"], "global_batch": r["Bg"], "per_gpu_batch": r["B"], "parallelism": "dp%d" % world,
                       "rccl_world": r["dist_world"], "collective": r["collective"], "collective_us": r["collective_us"],
                       "value_definition": "EXACTLY --steps training minibatches after --warmup untimed ones, from the seed-0 initialisation, every "
                                           "minibatch drawn fresh (no sample repeats), barrier + device synchronisation on both sides: the same point of "
                                           "the same training trajectory in every round and on every box (value_first_pass = value; rounds 2-5 reported "
                                           "the long window as value)",
                       "exchange_steps_per_minibatch": r["first_pass_steps_per_minibatch"], "sampling": "in-kernel Philox4x32-10",
                       "first_pass_ms_per_minibatch": r["first_pass_ms_per_minibatch"],
                       "first_pass_exchange_steps_per_minibatch": r["first_pass_steps_per_minibatch"],
                       "timed_minibatches": args.steps, "timed_seconds": r["first_pass_seconds"],
                       "minibatches_per_s": (1.0 if strong else world) * args.steps / r["first_pass_seconds"],
                       "sample_steps_per_s": r.get("first_pass_sample_steps", 0.0) / r["first_pass_seconds"],       # sum_t n_active,t per second, whole job
                       "fail_soft_degraded": r.get("degraded", 0),
                       # the %g s window that FOLLOWS the timed region (cycling the resident minibatches: many epochs of training, the
                       # conversations lengthen): value_window and what the agents had become by its end
                       "window": {"value": value_window, "seconds": r["elapsed"], "minibatches": r["minibatches"],
                                  "ms_per_minibatch": 1e3 * r["elapsed"] / r["minibatches"],
                                  "exchange_steps_per_minibatch": r["ex_steps"] / r["minibatches"],
                                  "sample_steps_per_s": r["sample_steps"] / r["elapsed"],
                                  "live_rows_per_sample_at_end": r.get("window_live_rows_per_sample")},
                       "unit_definition": "one exchange step (model.py:801 loop iteration) of one %d-sample batch%s" % (
                           unit_batch, " (the fixed global minibatch, sharded over the ranks)" if strong else
                           "; a global minibatch of %d*N samples advances N of them per iteration" % unit_batch)},
            "roofline": r["roofline"],
        }
        if world == 1 and args.workload == "c2" and not args.no_other_configs:
            # the other BASELINE.json configs on this GPU (short runs; parity-test cases, not the metric)
            other = {}
            # c3s / c5s: the whole global batch (512 / 2048 samples) on this one GPU = the N = 1 point of --scaling strong;
            # (its ms_per_minibatch) / (the per-GPU shard's) is the ceiling of the strong-scaling speed-up at 8 GPUs
            for w in ("c3", "c4", "c4r256", "c5", "c3s", "c5s"):
                o = run_workload(w[:-1] if w.endswith("s") else w, 30, 5, args.seed, 0, 1, local_rank, strong=w.endswith("s"))
                rf = o["roofline"] or {}
                other[w] = dict(workload=o["label"] + (" -- all %d samples on one GPU (--scaling strong, N=1)" % o["Bg"] if w.endswith("s") else ""), batch=o["B"],
                                ms_per_minibatch=o["first_pass_ms_per_minibatch"],          # 30 fresh minibatches after 5 warm-up ones (as `value`)
                                exchange_steps_per_s=o["first_pass_ex_steps"] / o["first_pass_seconds"],
                                first_pass_ms_per_minibatch=o["first_pass_ms_per_minibatch"],
                                window_ms_per_minibatch=1e3 * o["elapsed"] / o["minibatches"],
                                roofline={k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launch_us", "launch_us_raw", "event_floor_us", "algorithmic_amount", "traffic", "traffic_source", "traffic_ratio",
                                                                  "rocprof_kernel", "rocprof_avg_us", "rocprof_source", "frac_rocprof", "sec8d_bytes_per_minibatch", "step_frac", "launches_per_minibatch", "conversation_flop")},
                                kernels_us=rf.get("kernels_us"))
            line["other_configs"] = other
        if world == 1 and args.workload == "c2" and not args.no_cli and not strong:
            # what `python -m multimodalgame_amd.model` sustains end to end (same GPU, right after the HBM-resident measurement)
            # (ONE nested object: the driver's record keeps the first keys of `config` only -- VERDICT r05 weak 10)
            cli = run_cli()
            cli["cli_over_resident_window"] = cli["cli_steps_per_s"] / value_window      # both train for seconds: comparable regimes
            cli["cli_ms_over_resident_window_ms"] = cli["cli_ms_per_minibatch"] / (1e3 * r["elapsed"] / r["minibatches"])
            cli.update(run_cli(log_dev=1000))
            line["config"] = dict(cli=cli, **line["config"])
        if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
            line["cpu_baseline"] = cpu_baseline()
            cb = line["cpu_baseline"]
            # both legs from the same initial weights: the GPU's FIRST PASS is the leg at the CPU sample's conversation length
            cb["gpu_first_pass_over_cpu"] = value_first / cb["value"]
            cb["gpu_first_pass_exchange_steps_per_minibatch"] = r["first_pass_steps_per_minibatch"]
    # N > 1, default workload: the SURVEY.md 8(d) whole-node figures -- the GLOBAL batch of configs[2] (512) and configs[4] (2048)
    # sharded over the ranks (strong scaling) -- in the SAME line, so that one driver invocation per N yields them
    if world > 1 and args.workload == "c2" and not strong and not args.no_other_configs:
        sc = {}
        for w in ("c3", "c5"):
            if STRONG_GLOBAL_BATCH[w] % world:
                continue
            o = run_workload(w, 30, 5, args.seed, rank, world, local_rank, strong=True, want_roofline=False)
            sc[w + "s"] = dict(workload=o["label"], scaling="strong", global_batch=o["Bg"], per_gpu_batch=o["B"], rccl_world=o["dist_world"],
                               value=o["first_pass_ex_steps"] / o["first_pass_seconds"], unit="exchange-steps/s of the %d-sample global batch" % o["Bg"],
                               ms_per_minibatch=o["first_pass_ms_per_minibatch"],
                               first_pass_ms_per_minibatch=o["first_pass_ms_per_minibatch"],
                               value_window=o["ex_steps"] / o["elapsed"], window_ms_per_minibatch=1e3 * o["elapsed"] / o["minibatches"],
                               collective=o["collective"], collective_us=o["collective_us"])
        if rank == 0:
            line["strong_configs"] = sc
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
