"""`python -m multimodalgame_amd.model <reference flags>` -- the reference's entry point
(model.py:1813-1820) on the MI355X path: same flags / presets / derived file names, same log-line
templates (model.py:1348-1377, 1561-1584), same checkpoint dict (misc.py:58-75), same epoch / batch
order (misc.py:257-302).  The per-minibatch block is ONE fused call (Game.train_step); scalars are read
back only when a log line needs them."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

from . import flags as _flags
from .agents import Baseline, Receiver, Sender
from .flags import FLAGS
from .game import Game, exchange, get_rec_outp
from .misc import (FileLogger, VisdomLogger, cbow, embed, load_epoch, load_hdf5, read_data, torch_load, torch_save,
                   write_synthetic_dataset)
from .sparks import sparks


def _desc_matrix(csv_path, glove_path, wv_dim):
    """model.py:1072-1104."""
    descr, word_dict, _, label_id_to_idx, _ = read_data(csv_path)
    word_dict = embed(word_dict, glove_path)
    descr = cbow(descr, word_dict)
    desc = torch.cat([descr[i]["cbow"].view(1, -1) for i in descr.keys()], 0)
    return desc, (lambda x: label_id_to_idx.get(x))


EVAL_FLUSH_BATCHES = 64     # eval_dev reduces the stacked tape slices every this many dev batches (device memory bound)


def _eval_reduce(groups, acc, conv_lens, ham_sen, ham_rec, T, W, n_cls, top_k):
    """Pass 2 of eval_dev: the reference's per-batch host arithmetic (model.py:648-691) on the device, batched over the dev
    batches stacked since the last call; accumulates into acc (hit count, confusion matrix, classes seen) and the three lists."""
    for _bs, g in groups.items():
        mask = torch.stack(g["mask"]).to(torch.float32)                      # [NB, T + 1, B]: m_0 .. m_T, running minimum of the stop bits
        NB = mask.size(0)
        dv = mask.device
        steps = torch.arange(T, device=dv)
        if FLAGS.fixed_exchange:
            n = torch.full((NB,), T, dtype=torch.int64, device=dv)
            tsel = torch.full((NB, _bs), T - 1, dtype=torch.int64, device=dv)
        else:
            dead = mask[:, 1:].sum(2) == 0                                   # [NB, T]: step after which nobody is alive (model.py:866)
            n = torch.where(dead.any(1), torch.argmax(dead.to(torch.int8), 1) + 1, torch.full((NB,), T, dtype=torch.int64, device=dv))
            # y_masks[t] = min(1 - m'_{t+1}, m'_t) with m'_n forced to 0 (model.py:870, 1261): the masks are a running minimum
            # that starts at 1, so the selected step of a sample is the number of t in 1..n-1 with m_t = 1
            tsel = (mask[:, 1:] * (steps.view(1, T) + 1 < n.view(NB, 1)).to(mask.dtype).view(NB, T, 1)).sum(1).to(torch.int64)
        live = (steps.view(1, T) < n.view(NB, 1)).to(torch.float32)          # [NB, T]: the steps the reference executed
        y = torch.stack(g["y"]).to(torch.float32)                            # [NB, T, B, D]
        outp = y.gather(1, tsel.view(NB, 1, _bs, 1).expand(NB, 1, _bs, y.size(3)))[:, 0]
        dist = F.log_softmax(outp, dim=2)                                    # [NB, B, D]
        tgt = torch.stack(g["target"]).to(dv)                                # [NB, B]
        top_k_ind = dist.topk(min(top_k, dist.size(2)), dim=2).indices       # (= argsort()[:, -top_k:] as a set, model.py:658)
        c = (top_k_ind == tgt.unsqueeze(2)).sum()
        acc["correct"] = c if acc["correct"] is None else acc["correct"] + c.to(acc["correct"].device)
        pred = dist.argmax(2)
        if acc["conf_flat"] is None:
            acc["conf_flat"] = torch.zeros(n_cls * n_cls, dtype=torch.int64, device=dv)
            acc["seen"] = torch.zeros(n_cls, dtype=torch.int64, device=dv)
        conf_flat, seen = acc["conf_flat"], acc["seen"]
        conf_flat.index_add_(0, (tgt * n_cls + pred).view(-1), torch.ones(NB * _bs, dtype=torch.int64, device=dv))
        seen.index_add_(0, torch.cat([tgt.view(-1), pred.view(-1)]), torch.ones(2 * NB * _bs, dtype=torch.int64, device=dv))
        conv_lens.append((torch.stack(g["s"]).to(torch.float32) * live.view(NB, T, 1)).sum(1).view(-1))
        for name, hl in (("z", ham_sen), ("w", ham_rec)):
            msg = torch.stack(g[name]).to(torch.float32)                     # [NB, T, B, W]
            prev = torch.cat([torch.zeros(NB, 1, _bs, W, device=dv), msg[:, :-1]], 1)
            per_step = (msg - prev).abs().sum(3).mean(2)                     # [NB, T]: mean over the batch of the Hamming distance
            hl.append((per_step * live).sum(1) / n.to(torch.float32))       # [NB]: mean over the executed steps (model.py:679, 684)


def eval_dev(dev_file, batch_size, epoch, shuffle, top_k, game, desc, map_labels, conf_mat_path, device, dump=None):
    """model.py:580-722 ON THE DEVICE: deterministic conversations on the dev set, top-k accuracy (nominal batch size in the
    denominator, line 667), confusion matrix, conversation length and Hamming statistics.

    One pass per dev batch with NO host synchronisation: the eval-mode conversation (Game.eval_forward: one launch, all T
    steps of every sample) leaves masks / stop bits / messages / class logits on the engine's tape, and everything the
    reference then does on the host per batch -- the early-break step count n (model.py:866), output selection
    (get_rec_outp, 879-904), log-softmax, top-k membership (657-668), argmax, conversation lengths (671-672), the per-step
    mean Hamming distance of both agents' messages averaged over the n executed steps (675-691) -- is a handful of torch ops
    enqueued on the device: the tape slices of every batch are stacked (five device copies per batch) and the arithmetic runs
    ONCE per batch size over all batches together (hit count, a [D, D] confusion matrix by index_add_, the conversation
    lengths, the two Hamming means); the results are copied to the host ONCE.
    (Round 4 transcribed the host loop literally: numpy argsort per batch and ~20 float() syncs per batch.)

    dump: optional dict that receives the last batch's engine (the dev sample dump of model.py:1463-1518 reads its tape)."""
    W = FLAGS.rec_w_dim
    n_cls = desc.size(0)
    T = game.max_exchange
    # ---- pass 1: one eval-mode launch per batch; what the statistics need of its tape is copied into per-batch-size stacks
    # (five device copies per batch, no host synchronisation, no per-batch reduction kernels)
    groups = {}                                        # batch size -> dict of lists
    total = 0.0
    eng = None
    acc = dict(correct=None, conf_flat=None, seen=None)
    conv_lens, ham_sen, ham_rec = [], [], []
    pending = 0
    for batch in load_hdf5(dev_file, batch_size, epoch, shuffle, truncate_final_batch=True, map_labels=map_labels,
                           feats=(FLAGS.img_feat,), device=device):
        target, data = batch["target"], batch[FLAGS.img_feat]
        _bs = target.size(0)
        eng = game.eval_forward(data, target, desc)
        tp = eng.tape
        g = groups.setdefault(_bs, dict(mask=[], s=[], z=[], w=[], y=[], target=[]))
        g["mask"].append(tp["mask"].view(T + 1, _bs).clone()); g["s"].append(tp["s"].view(T, _bs).clone())
        g["z"].append(tp["z"].view(T, _bs, W).clone()); g["w"].append(tp["w"].view(T, _bs, W).clone())
        g["y"].append(tp["y"].view(T, _bs, -1).clone()); g["target"].append(target.view(-1))
        total += float(batch_size)                                           # model.py:667: the NOMINAL batch size
        pending += 1
        if pending >= EVAL_FLUSH_BATCHES:              # bound the stacked tape copies (a 1000-class dev set of 50k samples would hold GBs)
            _eval_reduce(groups, acc, conv_lens, ham_sen, ham_rec, T, W, n_cls, top_k)
            groups, pending = {}, 0
    _eval_reduce(groups, acc, conv_lens, ham_sen, ham_rec, T, W, n_cls, top_k)
    correct, conf_flat, seen = acc["correct"], acc["conf_flat"], acc["seen"]
    # ---- ONE copy to the host
    correct_h = int(correct.item()) if correct is not None else 0
    conf_full = conf_flat.view(n_cls, n_cls).cpu().numpy() if conf_flat is not None else np.zeros((n_cls, n_cls), np.int64)
    occ = np.nonzero(seen.cpu().numpy() > 0)[0] if seen is not None else np.zeros(0, np.int64)
    # sklearn.metrics.confusion_matrix (model.py:709): rows / columns = the SORTED CLASSES THAT OCCUR in truth or prediction
    np.savetxt(conf_mat_path, conf_full[np.ix_(occ, occ)], delimiter=",", fmt="%d")
    cl = torch.cat(conv_lens).cpu().numpy().astype(np.float64) if conv_lens else np.zeros(0)
    hs = torch.cat(ham_sen).cpu().numpy().astype(np.float64) if ham_sen else np.zeros(0)
    hr = torch.cat(ham_rec).cpu().numpy().astype(np.float64) if ham_rec else np.zeros(0)
    extra = dict(conversation_lengths_mean=cl.mean(), conversation_lengths_std=cl.std(),
                 hamming_sen_mean=hs.mean(), hamming_rec_mean=hr.mean())
    if dump is not None:
        dump["engine"] = eng
    return correct_h / total, extra


def _device(local_rank):
    if not torch.cuda.is_available():
        raise RuntimeError("multimodalgame_amd runs on MI355X only: no GPU visible and there is no CPU fallback")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    return dev


def _sync(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def parallel_env():
    """(rank, world, local_rank) of the data-parallel job: -rank / -world_size, else what torch.distributed.run exports."""
    world = FLAGS.world_size if FLAGS.world_size > 0 else int(os.environ.get("WORLD_SIZE", "1"))
    rank = FLAGS.rank if FLAGS.rank >= 0 else int(os.environ.get("RANK", "0"))
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world_size %d" % (rank, world))
    return rank, world, int(os.environ.get("LOCAL_RANK", str(rank)))


class _Silent(object):
    """Ranks > 0 of a data-parallel job: rank 0 alone writes the log, the flag dump, the confusion matrix and checkpoints."""

    def Log(self, message, level=1):
        pass


def run(stats=None):
    """The reference's run() (model.py:1001-1592) on the MI355X path.  One process per GPU: started under
    `python -m torch.distributed.run --nproc-per-node N -m multimodalgame_amd.model ...` the OUTER EPOCH LOOP is data
    parallel -- every rank forms the reference's global batch order (misc.py:257-302), keeps its B/N rows of every minibatch
    (load_hdf5(shard=...)), and Game.train_step runs dist.DataParallel.train_step (statistics all-reduce + ONE gradient
    all-reduce over RCCL, clip on the reduced gradient: the update of model.py:1307-1330 on the whole minibatch, identical on
    all ranks).  Rank 0 logs, evaluates (eval_dev) and checkpoints; losses / training accuracy in its log lines are those of
    the GLOBAL minibatch (they derive from the all-reduced statistics).

    stats: optional dict; the training loop leaves there what bench.py --cli reports: wall seconds of the epoch loop
    (device-synchronised at both ends, eval_dev time excluded), minibatches and exchange steps (device-side count)."""
    _flags.check_supported(FLAGS)                     # unsupported reference switches fail before anything is written
    rank, world, local_rank = parallel_env()
    if world > 1 and FLAGS.batch_size % world:
        raise ValueError("-batch_size %d does not divide over %d ranks" % (FLAGS.batch_size, world))
    os.makedirs(FLAGS.log_path, exist_ok=True)
    flogger = FileLogger(FLAGS.log_file) if rank == 0 else _Silent()
    VisdomLogger(env=FLAGS.env, experiment_name=FLAGS.experiment_name, enabled=FLAGS.visdom and rank == 0)
    flogger.Log("Flag Values:\n" + json.dumps(FLAGS.FlagValuesDict(), indent=4, sort_keys=True))
    if rank == 0 and not os.path.exists(FLAGS.json_file):
        with open(FLAGS.json_file, "w") as f:
            f.write(json.dumps(FLAGS.FlagValuesDict(), indent=4, sort_keys=True))
    device = _device(local_rank)
    own_group = False
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(FLAGS.dist_backend, rank=rank, world_size=world)
            own_group = True
        flogger.Log("Data parallel: {} ranks, {} samples of every {}-sample minibatch per rank, backend {}".format(
            world, FLAGS.batch_size // world, FLAGS.batch_size, dist.get_backend()))
    try:
        _run(stats, flogger, device, rank, world)
    finally:
        if own_group:
            import torch.distributed as dist
            dist.destroy_process_group()


def _run(stats, flogger, device, rank, world):
    torch.manual_seed(FLAGS.seed)

    sender = Sender(feature_type=FLAGS.img_feat, feat_dim=FLAGS.img_feat_dim, h_dim=FLAGS.img_h_dim,
                    w_dim=FLAGS.rec_w_dim, bin_dim_out=FLAGS.sender_out_dim, use_binary=FLAGS.use_binary,
                    use_attn=FLAGS.visual_attn, attn_dim=FLAGS.attn_dim, attn_extra_context=FLAGS.attn_extra_context,
                    attn_context_dim=FLAGS.attn_context_dim)
    baseline_sen = Baseline(hid_dim=FLAGS.baseline_hid_dim, x_dim=FLAGS.img_h_dim, binary_dim=FLAGS.rec_w_dim, inp_dim=0)
    receiver = Receiver(hid_dim=FLAGS.rec_hidden, out_dim=FLAGS.rec_out_dim, z_dim=FLAGS.sender_out_dim,
                        desc_dim=FLAGS.wv_dim, w_dim=FLAGS.rec_w_dim, s_dim=FLAGS.rec_s_dim, use_binary=FLAGS.use_binary)
    baseline_rec = Baseline(hid_dim=FLAGS.baseline_hid_dim, x_dim=0, binary_dim=FLAGS.rec_w_dim, inp_dim=FLAGS.rec_hidden)
    for mod in (sender, baseline_sen, receiver, baseline_rec):
        flogger.Log("Architecture: {}".format(mod))
        flogger.Log("Total Parameters: {}".format(float(sum(p.numel() for p in mod.parameters()))))

    if FLAGS.wv_type != "glove.6B":
        raise NotImplementedError                                          # model.py:1108 (fake/none are broken upstream)
    desc_train, map_labels_train = _desc_matrix(FLAGS.descr_train, FLAGS.glove_path, FLAGS.wv_dim)
    desc_dev, map_labels_dev = _desc_matrix(FLAGS.descr_dev, FLAGS.glove_path, FLAGS.wv_dim)
    desc_train, desc_dev = desc_train.to(device), desc_dev.to(device)

    game = Game(sender, receiver, baseline_sen, baseline_rec, device=device, seed=FLAGS.seed)
    game.set_parallel(rank, world)
    game.train_engine_for(FLAGS.batch_size // world, desc_train.size(0))  # parameters move into the flat GPU buffer
    if world > 1:                                                        # every rank starts from rank 0's weights
        import torch.distributed as dist
        dist.broadcast(game.engine.flat_params, src=0)
    models_dict, optimizers_dict = game.models_dict(), game.optimizers_dict()

    epoch, step, best_dev_acc = 0, 0, 0
    if os.path.exists(FLAGS.checkpoint):                                   # model.py:1150-1156
        flogger.Log("Loading from: " + FLAGS.checkpoint)
        data = torch_load(FLAGS.checkpoint, models_dict, optimizers_dict)
        flogger.Log("Loaded at step: {} and best dev acc: {}".format(data["step"], data["best_dev_acc"]))
        step, best_dev_acc = data["step"], data["best_dev_acc"]
        if "mmg_minibatch_counter" in data:              # resume the sampling stream where the checkpoint left it
            game.set_counters(data["mmg_minibatch_counter"], game.counters()[1])

    def do_eval():                                                        # (rank 0 only in a data-parallel job)
        return eval_dev(FLAGS.dev_file, FLAGS.batch_size_dev, epoch, FLAGS.shuffle_dev, FLAGS.top_k_dev, game, desc_dev,
                        map_labels_dev, FLAGS.conf_mat, device)

    if (FLAGS.eval_only or FLAGS.binary_only) and rank != 0:
        return                                                             # single-process modes: rank 0 does them
    if FLAGS.eval_only:                                                    # model.py:1166-1180
        if not os.path.exists(FLAGS.checkpoint):
            raise Exception("Must provide valid checkpoint.")
        dev_acc, extra = do_eval()
        flogger.Log("Dev Accuracy: " + str(dev_acc))
        with open(FLAGS.eval_csv_file, "w") as f:
            f.write("checkpoint,eval_file,topk,step,best_dev_acc,eval_acc,convlen_mean,convlen_std\n")
            f.write("{},{},{},{},{},{},{},{}\n".format(FLAGS.checkpoint, FLAGS.dev_file, FLAGS.top_k_dev, step,
                                                       best_dev_acc, dev_acc, extra["conversation_lengths_mean"],
                                                       extra["conversation_lengths_std"]))
        return
    if FLAGS.binary_only:                                                  # model.py:1181-1187
        if not os.path.exists(FLAGS.checkpoint):
            raise Exception("Must provide valid checkpoint.")
        from .binary_vectors import extract_binary
        game.engine_for(FLAGS.batch_size_dev, desc_dev.size(0))
        extract_binary(FLAGS, FLAGS.dev_file, FLAGS.batch_size_dev, epoch, FLAGS.shuffle_dev, sender, receiver, desc_dev,
                       map_labels_dev, device)
        flogger.Log("Wrote " + FLAGS.binary_output)
        return

    # Training accuracy of a log line = top-k hits of the last min(log_interval, minibatches of this process) minibatches
    # (model.py:1333-1348).  The hits accumulate on the device (tape "totals"); a log line reads the running sum and
    # subtracts what the previous log line read -- nothing is copied or enqueued per minibatch.
    steps_run, steps_at_log, hits_at_log = 0, 0, 0.0
    totals0 = None
    eval_seconds, n_evals = 0.0, 0
    import time as _time

    # The log block of a minibatch (model.py:1342-1518) is ENQUEUED at its step -- device-side reductions + one non-blocking copy
    # to pinned memory -- and WRITTEN once the copy has landed, normally a few minibatches later: same lines, same order (every
    # other writer of the log flushes it first), and the launch queue never drains for it.
    pending_log = []
    log_state = dict(hits_at_log=0.0)

    def emit_log(blk):
        snap = _log_snapshot_end(blk["h"])
        L, hits_now = snap["losses"], snap["hits_total"]
        avg_batch_acc = (hits_now - log_state["hits_at_log"]) / float(FLAGS.batch_size) / blk["n_seen"]
        log_state["hits_at_log"] = hits_now
        pre = blk["pre"]
        flogger.Log(pre + "Training Accuracy: {}".format(avg_batch_acc))
        flogger.Log(pre + "Loss Sender: {}".format(L["loss_binary_sen"]))
        flogger.Log(pre + "Loss Receiver (Y): {}".format(L["nll_loss"]))
        if FLAGS.use_binary:
            flogger.Log(pre + "Loss Receiver (Z): {}".format(L["loss_binary_rec"]))
            if not FLAGS.fixed_exchange:
                flogger.Log(pre + "Loss Receiver (S): {}".format(L["loss_binary_s"]))
            flogger.Log(pre + "Loss Baseline (S): {}".format(L["loss_bas_sen"]))
            flogger.Log(pre + "Loss Baseline (R): {}".format(L["loss_bas_rec"]))
        for line in _entropy_lines(blk["eng"], blk["target"], L, snap):    # model.py:1379-1407
            flogger.Log(line)
        if "dump" in snap:                                                 # model.py:1411-1461 (train sample dump)
            flogger.Log(_sample_dump_snap(snap["dump"], "Train:", int(L["n_steps"])))
        if blk["h_eval"] is not None:                                      # model.py:1463-1518 (the same minibatch in evaluation mode)
            ev_snap = _log_snapshot_end(blk["h_eval"])
            flogger.Log(_sample_dump_snap(ev_snap["dump"], "Eval:", _executed_steps_snap(ev_snap["dump"], FLAGS.fixed_exchange, ev_snap["T"])))

    def flush_log(block):
        while pending_log:
            blk = pending_log[0]
            if not block and not (_log_snapshot_ready(blk["h"]) and (blk["h_eval"] is None or _log_snapshot_ready(blk["h_eval"]))):
                return
            pending_log.pop(0)
            emit_log(blk)

    def finish():
        flush_log(True)
        if stats is not None and totals0 is not None:
            _sync(device)
            tot = game._train_engine.tape["totals"].cpu().tolist()
            stats.update(train_seconds=_time.perf_counter() - t_loop - eval_seconds, minibatches=steps_run,
                         exchange_steps=tot[0] - totals0[0], sample_steps=tot[3] - totals0[3],
                         eval_seconds=eval_seconds, evals=n_evals)
    def special(s):
        """A minibatch the host handles itself: it writes a log block (run-all tape), is followed by a dev evaluation or by a
        checkpoint (model.py:1342, 1545, 1579) -- evaluated identically on every rank."""
        return (s % FLAGS.log_interval == 0 or s % FLAGS.log_dev == 0
                or (s >= FLAGS.save_after and s % FLAGS.save_interval == 0))

    def epoch_items():
        """(i_batch, batch, n) in the reference's order (misc.py:257-302).  Device-resident epoch (misc.load_epoch): a RUN of n
        consecutive plain minibatches comes as ONE item -- batch = (features [n * B, F], targets [n * B]) -- and is enqueued by
        one library call (include/mmg.h: mmg_train_steps); the special minibatches come one by one (n = 0) as batch dicts.
        A file that streams from the host yields every minibatch as a dict, as rounds 1-5 did."""
        ep = None if os.environ.get("MMG_LOOP_PER_STEP") else load_epoch(
            FLAGS.train_file, FLAGS.batch_size, epoch, FLAGS.shuffle_train, map_labels=map_labels_train,
            feats=(FLAGS.img_feat,), device=device, shard=(rank, world))
        if ep is None:
            for i, b in enumerate(load_hdf5(FLAGS.train_file, FLAGS.batch_size, epoch, FLAGS.shuffle_train,
                                            map_labels=map_labels_train, feats=(FLAGS.img_feat,), device=device,
                                            with_ids=False, shard=(rank, world))):
                yield i, b, 0
            return
        i, x_ep, B = 0, ep.feats[FLAGS.img_feat], ep.B
        while i < ep.n:
            if special(step):
                yield i, ep.batch(i), 0
                i += 1
                continue
            n, limit = 1, ep.n - i
            if FLAGS.max_steps:
                limit = min(limit, FLAGS.max_steps - step)
            while n < limit and not special(step + n):
                n += 1
            yield i, (x_ep[i * B:(i + n) * B], ep.target[i * B:(i + n) * B]), n
            i += n

    _sync(device)
    t_loop = _time.perf_counter()
    while epoch < FLAGS.max_epoch:
        flush_log(True)
        flogger.Log("Starting epoch: {}".format(epoch))
        if FLAGS.images != "mammal":
            raise NotImplementedError                                      # model.py:1211 (cifar branch is broken upstream)
        for i_batch, batch, n_run in epoch_items():
            if totals0 is None:                      # (the first minibatch creates the engine)
                b0 = batch[1].size(0) // n_run if n_run else batch["target"].size(0)
                totals0 = game.train_engine_for(b0, desc_train.size(0)).tape["totals"].cpu().tolist()
                log_state["hits_at_log"] = totals0[1]
            if n_run:                                # a run of plain minibatches: one library call, nothing else to do for them
                game.train_steps(batch[0], batch[1], desc_train, n_run)
                steps_run += n_run
                step += n_run
                flush_log(False)
                if FLAGS.max_steps and step >= FLAGS.max_steps:
                    finish()
                    flogger.Log("Finished training.")
                    return
                continue
            # (a minibatch that writes a log block keeps the whole tape: every sample runs all steps -- same update, see Game.train_step)
            eng = game.train_step(batch[FLAGS.img_feat], batch["target"], desc_train,
                                  full_tape=(step % FLAGS.log_interval == 0))                # model.py:1240-1339
            steps_run += 1
            flush_log(False)                                               # a log block whose snapshot has landed is written now
            if step % FLAGS.log_interval == 0 and rank == 0:               # model.py:1342-1377 (global-minibatch figures)
                flush_log(True)
                n_dump = FLAGS.exchange_samples if FLAGS.exchange_samples > 0 else 0
                blk = dict(pre="Epoch: {} Step: {} Batch: {} ".format(epoch, step, i_batch), eng=eng, target=batch["target"],
                           n_seen=steps_run - steps_at_log,                # = min(minibatches of this process, log_interval)
                           h=_log_snapshot_begin(eng, batch["target"], dump=n_dump), h_eval=None)
                steps_at_log = steps_run
                if blk["h"]["event"] is None:                              # (no device queue to keep busy: the block is written at once,
                    emit_log(blk)                                          #  before the evaluation pass reuses the engine's tape)
                if n_dump:
                    # model.py:1463-1518: the same minibatch once more in evaluation mode (rounded messages), same layout
                    ev = game.eval_forward(batch[FLAGS.img_feat], batch["target"], desc_train)
                    h_eval = _log_snapshot_begin(ev, None, dump=n_dump, losses=False)
                    if blk["h"]["event"] is None:
                        ev_snap = _log_snapshot_end(h_eval)
                        flogger.Log(_sample_dump_snap(ev_snap["dump"], "Eval:", _executed_steps_snap(ev_snap["dump"], FLAGS.fixed_exchange, ev_snap["T"])))
                    else:
                        blk["h_eval"] = h_eval
                if blk["h"]["event"] is not None:
                    pending_log.append(blk)
            if step % FLAGS.log_dev == 0 and rank == 0:                    # model.py:1545-1576
                flush_log(True)
                _sync(device)
                t_ev = _time.perf_counter()
                dev_acc, extra = do_eval()
                _sync(device)
                eval_seconds += _time.perf_counter() - t_ev
                n_evals += 1
                pre = "Epoch: {} Step: {} Batch: {} ".format(epoch, step, i_batch)
                flogger.Log(pre + "Development Accuracy: {}".format(dev_acc))
                flogger.Log(pre + "Conversation Length (avg/std): {}/{}".format(
                    extra["conversation_lengths_mean"], extra["conversation_lengths_std"]))
                flogger.Log(pre + "Mean Hamming Distance (R/S): {}/{}".format(extra["hamming_rec_mean"], extra["hamming_sen_mean"]))
                if step >= FLAGS.save_after and dev_acc > best_dev_acc:
                    best_dev_acc = dev_acc
                    flogger.Log("Checkpointing with best Development Accuracy: {}".format(best_dev_acc))
                    torch_save(FLAGS.checkpoint + "_best", dict(step=step, best_dev_acc=best_dev_acc, mmg_minibatch_counter=game.counters()[0]), models_dict, optimizers_dict)
            if step >= FLAGS.save_after and step % FLAGS.save_interval == 0 and rank == 0:   # model.py:1579-1584
                flush_log(True)
                flogger.Log("Checkpointing.")
                torch_save(FLAGS.checkpoint, dict(step=step, best_dev_acc=best_dev_acc, mmg_minibatch_counter=game.counters()[0]), models_dict, optimizers_dict)
            step += 1
            if FLAGS.max_steps and step >= FLAGS.max_steps:
                finish()
                flogger.Log("Finished training.")
                return
        epoch += 1
    finish()
    flogger.Log("Finished training.")


_LOSS_KEYS = ("nll_loss", "loss_binary_s", "loss_binary_rec", "loss_binary_sen", "loss_bas_rec", "loss_bas_sen", "n_steps", "hits")


_PINNED = {}        # (losses, k, numel) -> FREE pinned host buffers; a snapshot handle owns its buffer from begin to end


def _log_snapshot_begin(eng, target, dump=0, losses=True):
    """Everything the log block of a minibatch prints, gathered on the device into ONE flat f64 vector and copied to pinned host
    memory WITHOUT waiting (an event marks the copy): the eight loss scalars, the running hit count, the batch statistics, the
    per-step prediction entropies of the run-all tape, targets | argmax predictions and -- dump > 0 -- what the sample dump of
    the first `dump` samples prints (probabilities, bits, stop probabilities, masks of every step) plus the per-step count of
    live samples.  _log_snapshot_end() waits for the event and parses.  The training loop enqueues a log minibatch's snapshot and
    formats it a few minibatches later: the GPU never idles for a log block (round 5; the synchronous copy drained the launch
    queue every -log_interval steps and the host's formatting kept it empty: 79 us per minibatch against 63 resident)."""
    tp = eng.tape
    f64 = torch.float64
    dev = tp["dist"].device
    T = tp["y"].size(0)
    B = tp["dist"].size(0)
    if hasattr(eng, "log_snapshot") and not os.environ.get("MMG_LOG_TORCH"):
        # ONE launch of the library (csrc/kernels_bwd.h: k_log_snapshot) instead of ~16 small torch kernels: 0.2 ms of GPU time per
        # log block (round 6).  MMG_LOG_TORCH=1: the torch form below, for the cross-check test.
        k = min(int(dump), B)
        flat = eng.log_snapshot(target.to(dev) if (losses and target is not None) else None, dump=k, losses=losses)
        return _log_snapshot_copy(flat, eng.stats.numel() if losses else 0, T, B, k, tp["z"].size(2), losses)
    parts, n_stats = [], 0
    if losses:
        y = tp["y"].to(torch.float32)
        pr = F.softmax(y, dim=2)
        ent = (torch.log(pr + 1e-8) * pr).sum(2).mean(1)                   # [T] (model.py:880-886; the executed steps are picked at the end)
        n_stats = eng.stats.numel()
        parts += [tp["losses"].to(f64).view(-1), tp["totals"].to(f64).view(-1)[1:2], eng.stats.to(f64).view(-1), ent.to(f64).view(-1),
                  tp["dist"].argmax(1).to(f64).view(-1), target.to(dev).to(f64).view(-1)]
    k = min(int(dump), B)
    W = tp["z"].size(2)
    if k > 0:
        parts += [tp["mask"][1:, :, 0].to(f64).sum(1).view(-1)]             # [T] live samples after every step
        for name in ("pz", "pw", "z", "w"):
            parts.append(tp[name][:, :k].to(f64).reshape(-1))              # [T, k, W]
        parts += [tp["ps"][:, :k].to(f64).reshape(-1), tp["mask"][1:, :k, 0].to(f64).reshape(-1)]   # [T, k]
    return _log_snapshot_copy(torch.cat(parts), n_stats, T, B, k, W, losses)


def _log_snapshot_copy(flat, n_stats, T, B, k, W, losses):
    """The flat device vector -> pinned host memory without waiting (an event marks the copy)."""
    f64 = torch.float64
    ev = None
    if flat.is_cuda and os.environ.get("MMG_LOG_SYNC"):                    # cross-check switch: the synchronous log block of rounds 1-4
        flat = flat.cpu()
    if flat.is_cuda:
        key = (losses, k, flat.numel())
        free = _PINNED.setdefault(key, [])
        host = free.pop() if free else torch.empty(flat.numel(), dtype=f64, pin_memory=True)
        host.copy_(flat, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(flat.device))
    else:
        host = flat
    return dict(host=host, event=ev, n_stats=n_stats, T=T, B=B, k=k, W=W, losses=losses, key=(losses, k, flat.numel()) if ev is not None else None)


def _log_snapshot_ready(h):
    return h["event"] is None or h["event"].query()


def _log_snapshot_end(h):
    if h["event"] is not None:
        h["event"].synchronize()
    flat = h["host"].tolist()
    if h.get("key") is not None:                       # the pinned buffer goes back to the pool: two pending blocks never share one
        _PINNED.setdefault(h["key"], []).append(h["host"])
        h["key"] = None
    T, B, k, W, n_stats = h["T"], h["B"], h["k"], h["W"], h["n_stats"]
    out = dict(T=T, B=B)
    o = 0
    if h["losses"]:
        out["losses"] = dict(zip(_LOSS_KEYS, flat[o:o + 8])); o += 8
        out["hits_total"] = flat[o]; o += 1
        out["stats"] = flat[o:o + n_stats]; o += n_stats
        out["ent_y"] = flat[o:o + T]; o += T
        out["argmax"] = [int(v) for v in flat[o:o + B]]; o += B
        out["target"] = [int(v) for v in flat[o:o + B]]; o += B
    if k > 0:
        d = dict(k=k, W=W, alive=flat[o:o + T]); o += T
        for name in ("pz", "pw", "z", "w"):
            d[name] = [[flat[o + (t * k + i) * W:o + (t * k + i + 1) * W] for i in range(k)] for t in range(T)]; o += T * k * W
        d["ps"] = [[flat[o + t * k + i] for i in range(k)] for t in range(T)]; o += T * k
        d["mask"] = [[flat[o + t * k + i] for i in range(k)] for t in range(T)]; o += T * k
        out["dump"] = d
    return out


def _log_snapshot(eng, target):
    """The synchronous form: begin + end (one device -> host copy for the whole block)."""
    return _log_snapshot_end(_log_snapshot_begin(eng, target))


def _entropy_lines(eng, target, L, snap=None):
    """model.py:1379-1407: "Predictions" (targets over argmax predictions of the minibatch) and the per-step entropies the
    losses report -- "Entropy Sender Binary" / "Entropy Receiver Binary": minus the mean over the step's ACTIVE samples of
    sum_j p log(p + 1e-8) + (1 - p) log(1 - p + 1e-8) (calculate_loss_binary, 919-923), read from the batch statistics the
    loss kernels reduced anyway (sum of neg-entropies and count per (stream, step): csrc/layout.h, all-reduced in a
    data-parallel job, i.e. of the GLOBAL minibatch); "Entropy Receiver Predictions": minus the mean over the WHOLE batch
    (stopped samples too, model.py:880-886) of sum_d softmax(y_t) log(softmax(y_t) + 1e-8) at every executed step, from the
    run-all tape of this log minibatch (a data-parallel job prints rank 0's rows here and in "Predictions")."""
    if snap is None:
        snap = _log_snapshot(eng, target)
    n = int(L["n_steps"])
    T, B = snap["T"], snap["B"]
    out = ["Predictions: {}".format(torch.tensor([snap["target"], snap["argmax"]], dtype=torch.int64).view(-1, B))]
    if FLAGS.use_binary:
        st = snap["stats"]
        per = 5                                                             # layout.h: MMG_ST_PER (n, sum w, sum w^2, sum w logp, sum negent)
        for title, stream, count in (("Entropy Sender Binary", 2, n), ("Entropy Receiver Binary", 1, n - 1)):
            if count <= 0:
                continue
            msg = title
            for i in range(count):
                base = (stream * T + i) * per
                msg += "\n{}. {}".format(i, -(st[base + 4] / st[base]) if st[base] > 0 else 0.0)
            out.append(msg + "\n")
    if n > 0:
        msg = "Entropy Receiver Predictions"
        for i, e in enumerate(snap["ent_y"][:n]):
            msg += "\n{}. {}".format(i, -e)
        out.append(msg + "\n")
    return out


def _executed_steps_snap(d, fixed, T):
    """Steps the reference's exchange() executes (model.py:866: break once every sample has stopped), from a snapshot's live counts."""
    if fixed:
        return T
    for t, a in enumerate(d["alive"]):
        if a == 0:
            return t + 1
    return T


def _sample_dump_snap(d, title, n):
    """model.py:1415-1461 / 1463-1518: sparkline of the probabilities and the bits of the first samples at every one of the `n`
    executed steps, from the dump part of a log snapshot (the tape of a log minibatch / of an evaluation pass holds all of them)."""
    W = d["W"]
    out = title
    for i in range(d["k"]):
        prev_sen, prev_rec = [0.0] * W, [0.0] * W
        for t in range(n):
            sp, rp, stp = d["pz"][t][i], d["pw"][t][i], [d["ps"][t][i]]
            sb, rb = d["z"][t][i], d["w"][t][i]
            sh, rh = float(sum(abs(a - b) for a, b in zip(prev_sen, sb))), float(sum(abs(a - b) for a, b in zip(prev_rec, rb)))
            prev_sen, prev_rec = sb, rb
            out += ("\n{:>3}".format(i) if t == 0 else "\n   ")
            out += "        {}".format(sparks([1] + sp)[1:]) + "           {}    {}".format(sparks([1] + stp)[1:], sparks([1] + rp)[1:])
            out += "\n    {:>3} S: {} {:4}".format(t, "".join(str(int(v)) for v in sb), sh)
            # (the forced zero of the LAST mask, model.py:870)
            out += "    s={} R: {} {:4}".format(0 if t == n - 1 else int(d["mask"][t][i]), "".join(str(int(v)) for v in rb), rh)
    return out + "\n"


def main(argv=None):
    argv = sys.argv if argv is None else argv
    _flags.define_flags()
    FLAGS(argv)
    if FLAGS.synthetic_data:
        n_cls = 30
        paths = write_synthetic_dataset(FLAGS.synthetic_data, n_classes=n_cls, feat_dim=512, wv_dim=FLAGS.wv_dim)
        for k, v in paths.items():
            if not any(a.lstrip("-").split("=")[0] == k for a in argv[1:]):
                setattr(FLAGS, k, v)
    _flags.default_flags(argv)
    run()


if __name__ == "__main__":
    main()
