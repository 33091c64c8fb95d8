"""exchange() and the per-minibatch training block of the reference (model.py:725-876, 1240-1339)
on top of the HIP engine, with the reference's calling conventions and return structures."""
import torch

from . import flags as _flags
from .engine import Engine


class FlatOptimizer(object):
    """Checkpoint-facing stand-in for the reference's four torch.optim objects (model.py:1110-1142):
    the update itself runs inside libmmg (k_gradnorm/k_opt); this class only converts the agent's slice
    of the flat optimizer state to and from torch.optim's state_dict layout (misc.py:61-62, 89-90): state index i is
    the i-th entry of module.parameters(), so a state_dict saved by torch.optim.RMSprop / Adam built on the reference's
    modules loads here and vice versa (tests/test_cli_gpu.py::test_optimizer_state_roundtrip_with_torch_optim)."""

    def __init__(self, game, agent):
        self.game, self.agent = game, agent

    def _params(self):
        """(name, view into the flat buffer) in torch.optim's numbering: the order of module.parameters(), i.e. of
        named_parameters() -- direct nn.Parameters first (Sender: code_bias is index 0), then the sub-modules in
        registration order -- NOT the engine's flat-buffer order."""
        views = self.game.engine.params[self.agent]
        mod = self.game.modules.get(self.agent)
        if mod is None:
            return list(views.items())
        return [(name, views[name]) for name, _ in mod.named_parameters()]

    def state_dict(self):
        eng, kind = self.game.engine, self.game.cfg["optim_type"]
        step = self.game.counters()[1]
        n = eng.n_params
        state = {}
        for i, (name, view) in enumerate(self._params()):
            off = view.storage_offset()
            if step == 0 or kind == "SGD":
                continue
            if kind == "RMSprop":
                state[i] = {"step": torch.tensor(float(step)),
                            "square_avg": eng.opt_state[off:off + view.numel()].view(view.shape).cpu().clone()}
            else:
                state[i] = {"step": torch.tensor(float(step)),
                            "exp_avg": eng.opt_state[off:off + view.numel()].view(view.shape).cpu().clone(),
                            "exp_avg_sq": eng.opt_state[n + off:n + off + view.numel()].view(view.shape).cpu().clone()}
        group = {"lr": self.game.cfg["learning_rate"], "params": list(range(len(self._params())))}
        if kind == "RMSprop":
            group.update(alpha=0.99, eps=1e-8, weight_decay=0, momentum=0, centered=False)
        elif kind == "Adam":
            group.update(betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False)
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        eng, kind = self.game.engine, self.game.cfg["optim_type"]
        n = eng.n_params
        step = 0
        for i, (name, view) in enumerate(self._params()):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is None:
                continue
            off = view.storage_offset()
            step = max(step, int(float(st.get("step", 0))))
            if kind == "RMSprop":
                eng.opt_state[off:off + view.numel()].copy_(st["square_avg"].reshape(-1).to(eng.device))
            elif kind == "Adam":
                eng.opt_state[off:off + view.numel()].copy_(st["exp_avg"].reshape(-1).to(eng.device))
                eng.opt_state[n + off:n + off + view.numel()].copy_(st["exp_avg_sq"].reshape(-1).to(eng.device))
        if step:
            for e in self.game.engines.values():
                e.tape["counter"][1:3] = step


class Game(object):
    """Binds the four agent modules to one flat parameter buffer on the GPU and caches one libmmg
    handle per (batch size, number of classes)."""

    def __init__(self, sender, receiver, baseline_sen, baseline_rec, flags=None, device=None, seed=0):
        fl = flags if flags is not None else _flags.FLAGS
        _flags.check_supported(fl)                    # -desc_attn, -sender_mix prod|mou, -flipout_*, -ignore_*, ... raise
        self.modules = dict(sender=sender, receiver=receiver, baseline_sen=baseline_sen, baseline_rec=baseline_rec)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.max_exchange = fl.max_exchange
        self.cfg = dict(feat_dim=sender.feat_dim, h_dim=sender.h_dim, w_dim=sender.w_dim, rec_hidden=receiver.hid_dim,
                        wv_dim=receiver.desc_dim, bas_hidden=baseline_sen.hid_dim if baseline_sen is not None else 500,
                        max_exchange=fl.max_exchange, use_binary=bool(sender.use_binary),
                        fixed_exchange=bool(fl.fixed_exchange), s_prob_prod=bool(fl.s_prob_prod),
                        entropy_s=fl.entropy_s, entropy_sen=fl.entropy_sen, entropy_rec=fl.entropy_rec,
                        first_rec=fl.first_rec, optim_type=fl.optim_type, learning_rate=fl.learning_rate,
                        top_k=fl.top_k_train)
        assert sender.bin_dim_out == sender.w_dim == receiver.w_dim == receiver.z_dim, \
            "Both sender and receiver should communicate with same dim vectors for now."     # model.py:1756
        self.seed = seed
        self._call = 0
        self.rank, self.world, self.group, self._dp = 0, 1, None, {}
        self.engines = {}
        self.engine = None            # the first engine owns the flat buffers
        self.optimizers = None
        for m in self.modules.values():
            if m is not None:
                m._game = self

    # the reference's dict names (model.py:1139-1142)
    def optimizers_dict(self):
        return dict(optimizer_rec=FlatOptimizer(self, "receiver"), optimizer_sen=FlatOptimizer(self, "sender"),
                    optimizer_bas_rec=FlatOptimizer(self, "baseline_rec"), optimizer_bas_sen=FlatOptimizer(self, "baseline_sen"))

    def models_dict(self):
        return dict(receiver=self.modules["receiver"], sender=self.modules["sender"],
                    baseline_rec=self.modules["baseline_rec"], baseline_sen=self.modules["baseline_sen"])

    def next_seed(self):
        return self.seed

    def set_parallel(self, rank, world, group=None):
        """Data-parallel training (dist.DataParallel): train_step() then takes THIS rank's rows of the global minibatch --
        [rank * B / world, (rank + 1) * B / world) of the reference's batch (misc.py:257-302 order) -- and every rank ends
        each step with the parameters of the single-process step on the whole batch."""
        self.rank, self.world, self.group = int(rank), int(world), group

    def _adopt(self, eng):
        """Move the modules' parameters into the engine's flat buffer (values preserved)."""
        for agent, m in self.modules.items():
            if m is None:
                continue
            for name, p in m.named_parameters():
                view = eng.params[agent][name]
                view.copy_(p.data.to(eng.device))
                p.data = view

    def engine_for(self, batch, n_classes=None, global_batch=None, batch_offset=0):
        n_classes = n_classes or getattr(self, "_last_classes", None) or 1
        self._last_classes = n_classes
        key = (batch, n_classes, global_batch or batch, batch_offset)
        if key not in self.engines:
            eng = Engine(device=self.device, share=self.engine, batch=batch, n_classes=n_classes,
                         global_batch=global_batch, batch_offset=batch_offset, **self.cfg)
            if self.engine is None:
                self.engine = eng
                self._adopt(eng)
            self.engines[key] = eng
        return self.engines[key]

    def train_engine_for(self, local_batch, n_classes=None):
        """The engine train_step() uses for `local_batch` samples on this rank (the whole batch on one GPU)."""
        return self.engine_for(local_batch, n_classes, global_batch=local_batch * self.world, batch_offset=self.rank * local_batch)

    # ------------------------------------------------------------------ model.py:725-876
    def exchange(self, exchange_args):
        data, target, desc = exchange_args["data"], exchange_args.get("target"), exchange_args["desc"]
        train = exchange_args["train"]
        break_early = exchange_args.get("break_early", False)
        if exchange_args.get("corrupt", False):
            raise NotImplementedError("-bit_flip message corruption is outside the accelerated hot path")
        if exchange_args.get("data_context") is not None:
            raise NotImplementedError("attention context is outside the accelerated hot path")
        for k, m in self.modules.items():
            if m is not None and (train or k in ("sender", "receiver")):
                m.train(train)
        B = data.size(0)
        eng = self.engine_for(B, desc.size(0))
        dev = eng.device
        data = data.to(dev, torch.float32).contiguous().view(B, -1)
        desc = desc.to(dev, torch.float32).contiguous()
        target = None if target is None else target.to(dev, torch.int64).contiguous()
        self._call += 1
        eng.forward(data, target, desc, seed=self.seed, train=train, run_all=True)
        tp = eng.tape
        T = self.max_exchange
        n = T
        if break_early:                                                   # model.py:866 (one host sync)
            alive = tp["mask"][1:, :, 0].sum(1).tolist()
            for t, a in enumerate(alive):
                if a == 0:
                    n = t + 1
                    break
        binary = self.cfg["use_binary"]
        masks = [tp["mask"][t].clone() for t in range(n + 1)]
        masks[-1].zero_()                                                 # model.py:870
        s = (masks, [tp["s"][t].clone() for t in range(n)], [tp["ps"][t].clone() for t in range(n)])
        sen_w = ([tp["z"][t].clone() for t in range(n)], [tp["pz"][t].clone() if binary else None for t in range(n)])
        rec_w = ([tp["w"][t].clone() for t in range(n)], [tp["pw"][t].clone() if binary else None for t in range(n)])
        y = [tp["y"][t].clone() for t in range(n)]
        bs = [tp["bs"][t].clone() for t in range(n)] if train and binary else []
        br = [tp["br"][t].clone() for t in range(n)] if train and binary else []
        self.modules["sender"].h_x = tp["hx"]
        self.modules["receiver"].h_z = tp["h"][n]
        self.modules["receiver"].h_w = tp["g"][n - 1]
        return s, sen_w, rec_w, y, bs, br

    def eval_forward(self, data, target, desc):
        """The eval-mode conversation of exchange() (rounded messages, cumulative-product stop bit, every sample runs all
        max_exchange steps) WITHOUT slicing / cloning the tape into the reference's per-step lists and without any host
        synchronisation: returns the engine, whose tape views (mask, s, ps, z, pz, w, pw, y, ...) stay valid until its next
        forward pass.  eval_dev() reduces them on the device (model.py:640-691)."""
        for k in ("sender", "receiver"):
            if self.modules.get(k) is not None:
                self.modules[k].train(False)
        B = data.size(0)
        eng = self.engine_for(B, desc.size(0))
        dev = eng.device
        data = data.to(dev, torch.float32).contiguous().view(B, -1)
        desc = desc.to(dev, torch.float32).contiguous()
        target = None if target is None else target.to(dev, torch.int64).contiguous()
        self._call += 1
        eng.forward(data, target, desc, seed=self.seed, train=False, run_all=True)
        return eng

    # ------------------------------------------------------------------ model.py:1240-1339
    def train_step(self, data, target, desc, uniforms=None, full_tape=False):
        """exchange + masks + losses + four backward/clip/optimizer blocks, fused on the device.
        Nothing is copied to the host; read ``losses()`` when a log line needs them.
        The step keeps only what training reads (include/mmg.h: run_all_steps == 2): per-(step, sample) tape arrays are valid on
        the LIVE rows (t <= tstar[b]) only, in Fixed mode tape["y"] holds the output step only, and in continuous mode
        (-nouse_binary) the arrays a / c / zr / dbar / g / w are NOT written -- code that wants them after a training step
        calls exchange() (run-all) instead.  model.run's sample dump reads live rows of binary runs only
        (flags.default_flags sets -exchange_samples 0 without -use_binary, model.py:1758-1759).

        full_tape=True (the minibatches that write a log block, model.py:1342-1542): the SAME update through the phased calls
        with every sample running all steps of the conversation (run-all), so that the tape holds what the reference's log
        block prints -- the class logits of stopped samples too ("Entropy Receiver Predictions" is a mean over the whole
        batch at every executed step, model.py:880-886) and every row of the sample dump.  Early exit == run-all and
        fused == phased are parity-tested (tests/test_hip_parity.py); the extra launches cost ~30 us once per log_interval."""
        B = data.size(0)
        eng = self.train_engine_for(B, desc.size(0))
        u = uniforms or (None, None, None)
        # the Philox minibatch counter and the optimizer step (Adam bias correction) live in each engine's workspace: hand
        # them over when the batch size / class count -- hence the engine -- changes between steps
        last = getattr(self, "_train_engine", None)
        if last is not None and last is not eng:
            eng.tape["counter"][:3].copy_(last.tape["counter"][:3])      # ([3]: the engine's own launch epoch, never handed over)
        self._train_engine = eng
        if self.world > 1:
            dp = self._dp.get(id(eng))
            if dp is None:
                from .dist import DataParallel
                dp = self._dp[id(eng)] = DataParallel(eng, group=self.group)
            dp.train_step(data, target, desc, u[0], u[1], u[2], seed=self.seed, full_tape=full_tape)
        elif full_tape:
            eng.forward(data, target, desc, u[0], u[1], u[2], seed=self.seed, train=True, run_all=True, log_tape=True)
            eng.loss_stats()
            eng.backward(data, target, desc)
            eng.clip_step()
        else:
            eng.train_step(data, target, desc, u[0], u[1], u[2], seed=self.seed)
        return eng

    def train_steps(self, data, target, desc, n):
        """n consecutive plain minibatches (no log block) enqueued by ONE library call: data [n * B, F] / target [n * B] in batch
        order (misc.Epoch).  Same updates, same sampling streams as n train_step() calls (tests/test_cli_gpu.py)."""
        B = data.size(0) // n
        eng = self.train_engine_for(B, desc.size(0))
        last = getattr(self, "_train_engine", None)
        if last is not None and last is not eng:
            eng.tape["counter"][:3].copy_(last.tape["counter"][:3])
        self._train_engine = eng
        if self.world > 1:
            dp = self._dp.get(id(eng))
            if dp is None:
                from .dist import DataParallel
                dp = self._dp[id(eng)] = DataParallel(eng, group=self.group)
            dp.train_steps(data, target, desc, n, seed=self.seed)
        elif hasattr(eng, "train_steps"):
            eng.train_steps(data, target, desc, n, seed=self.seed)
        else:                                            # (an engine stand-in without the C loop: tests/oracle_engine.py)
            for i in range(n):
                eng.train_step(data[i * B:(i + 1) * B], target[i * B:(i + 1) * B], desc, seed=self.seed)
        return eng

    def counters(self):
        """[minibatch counter (Philox stream), optimizer step] of the engine that trained last (checkpointed by model.py)."""
        eng = getattr(self, "_train_engine", None) or self.engine
        c = eng.tape["counter"].cpu().tolist()
        return [int(c[0]), int(max(c[1], c[2]))]

    def set_counters(self, minibatch, step):
        for eng in self.engines.values():
            eng.tape["counter"][0] = int(minibatch)
            eng.tape["counter"][1:3] = int(step)

    def losses(self, batch, n_classes):
        return self.engine_for(batch, n_classes).losses()


def exchange(sender, receiver, baseline_sen, baseline_rec, exchange_args):
    """Drop-in for model.py:725: same arguments, same returned structure
    ``(s, sen_w, rec_w, y, bs, br)`` of per-step tensor lists."""
    game = getattr(sender, "_game", None)
    if game is None:
        game = Game(sender, receiver, baseline_sen, baseline_rec)
    elif baseline_sen is not None and game.modules.get("baseline_sen") is None:
        game.modules["baseline_sen"], game.modules["baseline_rec"] = baseline_sen, baseline_rec
    return game.exchange(exchange_args)


def get_rec_outp(y, masks):
    """model.py:879-904 (host-side tensor ops on the returned lists; used by callers of exchange())."""
    import torch.nn.functional as F

    def negent(yy):
        probs = F.softmax(yy, dim=1)
        return (torch.log(probs + 1e-8) * probs).sum(1).mean()
    negentropy = [negent(yy) for yy in y]
    if masks is not None:
        batch_size = y[0].size(0)
        inp = torch.cat([yy.view(batch_size, 1, -1) for yy in y], 1)
        mask = torch.cat(masks, 1).view(batch_size, len(masks), 1).expand_as(inp)
        return torch.masked_select(inp, mask.bool()).view(batch_size, -1), negentropy
    return y[-1], negentropy
