"""CPU shard engine for testing the data-parallel protocol (TEST INFRASTRUCTURE, see cpu_ref.py).

Implements the engine interface multimodalgame_amd.dist.DataParallel drives -- forward / loss_stats /
backward / clip_step, `.stats` (f64 vector, same layout as layout.h: stat_stream / stat_bas / stat_glob)
and `.flat_grads` -- with the literal oracle for the conversation and torch autograd for the local
gradient of the surrogate loss built from the GLOBAL statistics (SURVEY.md §8e option A).  If the protocol
is right, two ranks with half the batch each end up with exactly the parameters of one process with the
whole batch (model.py:1240-1330 on the global minibatch)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import cpu_ref

ST_PER = 5


def stat_stream(T, k, t, j):
    return (k * T + t) * ST_PER + j


def stat_bas(T, which, t):
    return 3 * T * ST_PER + which * T + t


def stat_glob(T, k):
    return 3 * T * ST_PER + 2 * T + k


def stat_count(T):
    return 3 * T * ST_PER + 2 * T + 4


def coefficients(stats, T, flags):
    """Python port of kernels_bwd.h: loss_coefficients (model.py:912-916, 919-926, 947-967, 972-987)."""
    st = stats.tolist()
    cw, ce = np.zeros((3, T)), np.zeros((3, T))
    cb = np.zeros(T)
    lam = (flags.entropy_s, flags.entropy_rec, flags.entropy_sen)
    for k in range(3):
        nsum = sum(st[stat_stream(T, k, t, 0)] for t in range(T))
        ln = T - 1 if k == 1 else T
        for t in range(T):
            n, s1, s2 = (st[stat_stream(T, k, t, j)] for j in range(3))
            if n > 0 and nsum > 0:
                c_over_n = 1.0 / (ln * n) if flags.fixed_exchange else 1.0 / nsum
                denom = 1.0
                if n > 1:
                    mean = s1 / n
                    var = max((s2 - n * mean * mean) / (n - 1.0), 0.0)
                    denom = max(1.0, var ** 0.5)
                cw[k, t] = c_over_n / denom
                ce[k, t] = c_over_n * lam[k] if lam[k] is not None else 0.0
    nsum = sum(st[stat_stream(T, 2, t, 0)] for t in range(T))
    for t in range(T):
        n = st[stat_stream(T, 2, t, 0)]
        if n > 0 and nsum > 0:
            cb[t] = 2.0 * (1.0 / (T * n) if flags.fixed_exchange else 1.0 / nsum)
    return cw, ce, cb


class ShardEngine(object):
    def __init__(self, flags, models, global_batch):
        self.fl, self.models, self.Bg = flags, models, global_batch
        self.T = flags.max_exchange
        self.stats = torch.zeros(stat_count(self.T), dtype=torch.float64)
        self.order = [(a, k) for a in ("receiver", "sender", "baseline_rec", "baseline_sen")
                      for k, _ in models[a].named_parameters()]
        self.flat_grads = torch.zeros(sum(dict(models[a].named_parameters())[k].numel() for a, k in self.order))
        self.optimizers = cpu_ref.build_optimizers(models, flags)
        self.use_binary = bool(flags.use_binary)

    def forward(self, x, target, desc, u_z, u_s, u_w, seed=0, train=True, run_all=False, minimal=False, log_tape=False):
        fl, m = self.fl, self.models
        tape = cpu_ref.UniformTape(u_z, u_s, u_w)
        for a in ("sender", "receiver"):
            m[a].rng = tape
        args = dict(data=x, target=target, desc=desc, train=True, break_early=False)
        s, sen_w, rec_w, y, bs, br = cpu_ref.exchange(m["sender"], m["receiver"], m["baseline_sen"], m["baseline_rec"], args, fl)
        B, T = x.size(0), self.T
        sfe = torch.cat(s[1], 1)                                        # [B,T] stop bits
        tstar = torch.full((B,), T - 1, dtype=torch.long)
        if not fl.fixed_exchange:
            for b in range(B):
                zeros = (sfe[b] == 0).nonzero()
                if len(zeros):
                    tstar[b] = int(zeros[0])
        outp = torch.stack([y[int(tstar[b])][b] for b in range(B)])
        dist = F.log_softmax(outp, dim=1)
        logs = dist.detach().gather(1, target.view(-1, 1)).view(-1)
        self.saved = dict(s=s, sen_w=sen_w, rec_w=rec_w, bs=bs, br=br, tstar=tstar, dist=dist, logs=logs, target=target, y=y)

    @staticmethod
    def _lp_ne(q, p):
        lp = (q * torch.log(p + 1e-8) + (1 - q) * torch.log(1 - p + 1e-8)).sum(1)
        ne = (p * torch.log(p + 1e-8) + (1 - p) * torch.log(1 - p + 1e-8)).sum(1)
        return lp, ne

    def _streams(self):
        sv = self.saved
        T = self.T
        ts = sv["tstar"]
        out = []
        for t in range(T):
            act = (ts >= t)
            act_next = (ts > t)
            out.append(dict(
                s=(act, sv["s"][1][t], sv["s"][2][t], sv["br"][t]),
                rec=(act_next, sv["rec_w"][0][t], sv["rec_w"][1][t], sv["br"][t]),
                sen=(act, sv["sen_w"][0][t], sv["sen_w"][1][t], sv["bs"][t])))
        return out

    def loss_stats(self):
        T, fl, sv = self.T, self.fl, self.saved
        st = torch.zeros_like(self.stats)
        L = sv["logs"].double()
        for t, d in enumerate(self._streams()):
            for k, key in enumerate(("s", "rec", "sen")):
                if key == "s" and fl.fixed_exchange:
                    continue
                act, q, p, beta = d[key]
                lp, ne = self._lp_ne(q.detach(), p.detach())
                w = (L - beta.detach().view(-1).double())[act]
                st[stat_stream(T, k, t, 0)] = float(act.sum())
                st[stat_stream(T, k, t, 1)] = w.sum()
                st[stat_stream(T, k, t, 2)] = (w * w).sum()
                st[stat_stream(T, k, t, 3)] = (w * lp.double()[act]).sum()
                st[stat_stream(T, k, t, 4)] = ne.double()[act].sum()
            act = d["sen"][0]
            st[stat_bas(T, 0, t)] = ((sv["br"][t].detach().view(-1).double() - L)[act] ** 2).sum()
            st[stat_bas(T, 1, t)] = ((sv["bs"][t].detach().view(-1).double() - L)[act] ** 2).sum()
        st[stat_glob(T, 0)] = L.sum()
        self.stats.copy_(st)

    def backward(self, x, target, desc):
        T, fl, sv = self.T, self.fl, self.saved
        cw, ce, cb = coefficients(self.stats, T, fl)
        L = sv["logs"]
        loss_rec = -(sv["dist"].gather(1, sv["target"].view(-1, 1)).sum()) / self.Bg
        loss_sen = torch.zeros(())
        loss_br = torch.zeros(())
        loss_bs = torch.zeros(())
        # continuous messages: loss_rec = NLL, the other three agents are not trained (model.py:1297-1305, 1313)
        for t, d in enumerate(self._streams() if fl.use_binary else ()):
            for k, key in enumerate(("s", "rec", "sen")):
                if key == "s" and fl.fixed_exchange:
                    continue
                act, q, p, beta = d[key]
                if not act.any():
                    continue
                lp, ne = self._lp_ne(q.detach(), p)
                w = (L - beta.detach().view(-1))
                term = ((-w * float(cw[k, t]) * lp + float(ce[k, t]) * ne)[act]).sum()
                if key == "sen":
                    loss_sen = loss_sen + term
                else:
                    loss_rec = loss_rec + term
            act = d["sen"][0]
            if act.any():
                loss_br = loss_br + (0.5 * float(cb[t]) * ((sv["br"][t].view(-1) - L) ** 2)[act]).sum()
                loss_bs = loss_bs + (0.5 * float(cb[t]) * ((sv["bs"][t].view(-1) - L) ** 2)[act]).sum()
        for m in self.models.values():
            m.zero_grad()
        loss_rec.backward()
        if loss_sen.requires_grad:
            loss_sen.backward()
        if loss_br.requires_grad:
            loss_br.backward()
        if loss_bs.requires_grad:
            loss_bs.backward()
        chunks = []
        for a, k in self.order:
            p = dict(self.models[a].named_parameters())[k]
            chunks.append((p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1))
        self.flat_grads.copy_(torch.cat(chunks))

    def clip_step(self):
        off = 0
        for a, k in self.order:
            p = dict(self.models[a].named_parameters())[k]
            p.grad = self.flat_grads[off:off + p.numel()].view_as(p).clone()
            off += p.numel()
        for agent, opt in ((("receiver", "optimizer_rec"), ("sender", "optimizer_sen"),
                            ("baseline_rec", "optimizer_bas_rec"), ("baseline_sen", "optimizer_bas_sen"))
                           if self.fl.use_binary else (("receiver", "optimizer_rec"),)):
            nn.utils.clip_grad_norm_(self.models[agent].parameters(), max_norm=1.)
            self.optimizers[opt].step()
