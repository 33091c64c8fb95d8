"""CPU oracle: a literal restatement of the reference's REINFORCE exchange path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product path (``multimodalgame_amd``) never does and fails loudly when
its HIP library is missing.

Every function restates, op for op, the reference's sequence in modern PyTorch
(torch >= 2) with the legacy semantics the reference was written against
(PyTorch 0.1.12 / Python 2.7): per-step ``image_layer``,
materialised ``build_inp``, host-side Bernoulli sampling with numpy uniforms,
list-of-tensors bookkeeping, four separate backward passes and the stock
``clip_grad_norm_`` + ``torch.optim`` updates.  Citations are
``/root/reference/<file>:<line>``.

Parity pin: ``tests/golden/*.npz`` were produced by executing the reference's
own ``model.py`` classes/functions (compat-patched in memory, see
``tests/golden/make_golden.py``) and ``tests/test_oracle_golden.py`` checks
this file against them.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim


# ----------------------------------------------------------------------------
# Config: the subset of the reference's gflags the hot path reads
# (model.py:1639-1741).  The reference reads a global FLAGS inside forward();
# the oracle freezes it into an object passed explicitly.
# ----------------------------------------------------------------------------
class Flags(object):
    def __init__(self, **kw):
        self.use_binary = True          # model.py:1701
        self.fixed_exchange = True      # model.py:1737
        self.max_exchange = 3           # model.py:1736
        self.first_rec = 0.0            # model.py:1709
        self.s_prob_prod = True         # model.py:1713
        self.entropy_s = None           # model.py:1730
        self.entropy_sen = None         # model.py:1731
        self.entropy_rec = None         # model.py:1732
        self.optim_type = "RMSprop"     # model.py:1725
        self.learning_rate = 1e-4       # model.py:1728
        self.batch_size = 32            # model.py:1726
        self.top_k_train = 6            # model.py:1722
        self.top_k_dev = 6              # model.py:1721
        self.img_feat_dim = 512
        self.img_h_dim = 100
        self.rec_w_dim = 50
        self.sender_out_dim = 50
        self.rec_hidden = 128
        self.rec_out_dim = 1
        self.rec_s_dim = 1
        self.wv_dim = 100
        self.baseline_hid_dim = 500
        self.ignore_receiver = False    # model.py:1703
        self.flipout_sen = None
        self.flipout_rec = None
        self.flipout_dev = False
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError("unknown flag " + k)
            setattr(self, k, v)


class UniformTape(object):
    """Stands in for the global numpy RNG the reference samples from
    (model.py:227, 420, 460).  ``rand`` hands back pre-drawn uniforms in the
    reference's call order -- per exchange step z (B x W), s (B x 1), w (B x W)
    -- so that the oracle, the reference and the HIP path consume identical
    numbers.  ``u_z``/``u_w`` are [T,B,W], ``u_s`` is [T,B,1]."""

    def __init__(self, u_z=None, u_s=None, u_w=None):
        self.u = {"z": u_z, "s": u_s, "w": u_w}
        self.t = {"z": 0, "s": 0, "w": 0}

    def rand(self, kind, *shape):
        arr = self.u[kind]
        out = np.asarray(arr[self.t[kind]], dtype=np.float64).reshape(shape)
        self.t[kind] += 1
        return out


class NumpyRng(object):
    """The reference's behaviour: the process-global numpy generator."""

    def rand(self, kind, *shape):
        return np.random.rand(*shape)


# ----------------------------------------------------------------------------
# Init (misc.py:349-385)
# ----------------------------------------------------------------------------
def xavier_normal_(tensor, gain=1.0):
    fan_out, fan_in = tensor.size(0), tensor.size(1)      # misc.py:354-356
    std = gain * math.sqrt(2.0 / (fan_in + fan_out))      # misc.py:384
    with torch.no_grad():
        return tensor.normal_(0, std)


# ----------------------------------------------------------------------------
# Agents (model.py:49-551), non-attention, sender_mix == "sum" branch
# ----------------------------------------------------------------------------
class Sender(nn.Module):
    """model.py:49-238."""

    def __init__(self, feature_type="avgpool_512", feat_dim=512, h_dim=100, w_dim=50,
                 bin_dim_out=50, use_binary=True, use_attn=False, attn_dim=256,
                 attn_extra_context=False, attn_context_dim=4096, rng=None, flags=None):
        super().__init__()
        assert not use_attn, "attention branch is out of the hot-path scope (SURVEY.md §2)"
        self.feat_dim, self.h_dim, self.w_dim = feat_dim, h_dim, w_dim
        self.bin_dim_out, self.use_binary = bin_dim_out, use_binary
        self.rng = rng or NumpyRng()
        self.flags = flags or Flags()
        self.image_layer = nn.Linear(feat_dim, h_dim)             # model.py:67
        self.code_layer = nn.Linear(w_dim, h_dim)                 # model.py:68
        self.code_bias = nn.Parameter(torch.zeros(bin_dim_out))   # model.py:69
        self.binary_layer = nn.Linear(h_dim, bin_dim_out)         # model.py:76
        self.reset_parameters()

    def reset_parameters(self):                                   # model.py:90-97
        for m in self.modules():
            if isinstance(m, nn.Linear):
                xavier_normal_(m.weight.data)
                m.bias.data.zero_()
        self.code_bias.data.normal_()

    def reset_state(self):                                        # model.py:99-112
        pass

    def forward(self, x, w, g, t):
        self.h_x = h_x = self.image_layer(x)                      # model.py:195
        if t == 0:                                                # model.py:196-200
            batch_size = x.size(0)
            first_code = torch.sigmoid(self.code_bias.view(1, -1))
            h_w = self.code_layer(first_code).expand(batch_size, self.h_dim)
        else:
            h_w = self.code_layer(w)                              # model.py:207
        features = self.binary_layer(torch.tanh(h_x + h_w))       # model.py:216
        if self.use_binary:
            probs = torch.sigmoid(features)                       # model.py:223
            if self.training:                                     # model.py:224-227
                probs_ = probs.detach().cpu().numpy()
                binary = torch.from_numpy(
                    (self.rng.rand("z", *probs_.shape) < probs_).astype(probs_.dtype))      # (float32; float64 in a double re-run)
            else:
                binary = torch.round(probs).detach()              # model.py:229
            return binary, probs
        return features, None                                     # model.py:238


def build_inp(binary_features, descs):
    """model.py:519-551: row b*D+d = [h[b] || desc[d]]."""
    batch_size = binary_features.size(0)
    num_desc = descs.size(0)
    binary_index = torch.from_numpy(np.arange(batch_size).repeat(num_desc).astype(np.int64))
    binary_copied = torch.index_select(binary_features, 0, binary_index)
    desc_index = torch.from_numpy(
        np.concatenate([np.arange(num_desc)] * batch_size).astype(np.int64))
    desc_copied = torch.index_select(descs, 0, desc_index)
    return torch.cat([binary_copied, desc_copied], 1)


class Receiver(nn.Module):
    """model.py:241-477 (non-desc_attn branch)."""

    def __init__(self, z_dim=50, desc_dim=100, hid_dim=128, out_dim=1, w_dim=50, s_dim=1,
                 use_binary=True, rng=None, flags=None):
        super().__init__()
        self.z_dim, self.desc_dim, self.hid_dim = z_dim, desc_dim, hid_dim
        self.out_dim, self.w_dim, self.s_dim, self.use_binary = out_dim, w_dim, s_dim, use_binary
        self.rng = rng or NumpyRng()
        self.flags = flags or Flags()
        self.rnn = nn.GRUCell(z_dim, hid_dim)                     # model.py:256
        self.w_h = nn.Linear(hid_dim, hid_dim, bias=True)         # model.py:258
        self.w_d = nn.Linear(desc_dim, hid_dim, bias=False)       # model.py:259
        self.w = nn.Linear(hid_dim, w_dim)                        # model.py:260
        self.y1 = nn.Linear(hid_dim + desc_dim, hid_dim)          # model.py:262
        self.y2 = nn.Linear(hid_dim, out_dim)                     # model.py:263
        self.s = nn.Linear(hid_dim, s_dim)                        # model.py:265
        self.reset_parameters()
        self.reset_state()

    def reset_parameters(self):                                   # model.py:275-288
        for m in self.modules():
            if isinstance(m, nn.Linear):
                xavier_normal_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.GRUCell):
                for mm in m.parameters():
                    if mm.data.ndimension() == 2:
                        xavier_normal_(mm.data)
                    else:
                        mm.data.zero_()

    def reset_state(self):                                        # model.py:290-298
        self.h_z = None
        self.s_prob_prod = None

    def initial_state(self, batch_size):                          # model.py:300-301
        return torch.zeros(batch_size, self.hid_dim)

    def forward(self, z, desc, desc_set=None, desc_set_lens=None):
        batch_size = z.size(0)
        if self.h_z is None:                                      # model.py:336-337
            self.h_z = self.initial_state(batch_size)
        self.h_z = self.rnn(z, self.h_z)                          # model.py:340
        inp_with_desc = build_inp(self.h_z, desc)                 # model.py:412

        s_prob = torch.sigmoid(self.s(self.h_z))                  # model.py:414-415
        if self.training:                                         # model.py:416-420
            prob_ = s_prob.detach().cpu().numpy()
            s_binary = torch.from_numpy(
                (self.rng.rand("s", *prob_.shape) < prob_).astype(prob_.dtype))
        else:                                                     # model.py:421-427
            if self.s_prob_prod is None or not self.flags.s_prob_prod:
                self.s_prob_prod = s_prob
            else:
                self.s_prob_prod = self.s_prob_prod * s_prob
            s_binary = torch.round(self.s_prob_prod).detach()

        y = self.y1(inp_with_desc).clamp(min=0)                   # model.py:432
        y = self.y2(y).view(batch_size, -1)                       # model.py:433

        n_desc = y.size(1)
        y_scores = F.softmax(y, dim=1).detach()                   # model.py:441
        y_broadcast = y_scores.unsqueeze(2).expand(batch_size, n_desc, self.desc_dim)
        wd_inp = desc.unsqueeze(0).expand(batch_size, n_desc, self.desc_dim)
        wd_inp = (y_broadcast * wd_inp).sum(1)                    # model.py:449

        self.h_w = torch.tanh(self.w_h(self.h_z) + self.w_d(wd_inp))   # model.py:452
        w_scores = self.w(self.h_w)                               # model.py:454
        if self.use_binary:
            w_probs = torch.sigmoid(w_scores)                     # model.py:456
            if self.training:                                     # model.py:457-460
                probs_ = w_probs.detach().cpu().numpy()
                w_feats = torch.from_numpy(
                    (self.rng.rand("w", *probs_.shape) < probs_).astype(probs_.dtype))
            else:
                w_feats = torch.round(w_probs).detach()           # model.py:462
            if self.flags.ignore_receiver:                        # model.py:470-472
                w_feats = torch.zeros(w_feats.size())
        else:
            w_feats, w_probs = w_scores, None                     # model.py:474-475
        return (s_binary, s_prob), (w_feats, w_probs), y


class Baseline(nn.Module):
    """model.py:480-516; default nn.Linear init (no reset_parameters)."""

    def __init__(self, hid_dim, x_dim, binary_dim, inp_dim):
        super().__init__()
        self.x_dim, self.binary_dim, self.inp_dim, self.hid_dim = x_dim, binary_dim, inp_dim, hid_dim
        self.linear1 = nn.Linear(x_dim + binary_dim + inp_dim, hid_dim)
        self.linear2 = nn.Linear(hid_dim, 1)

    def forward(self, x, binary, inp):
        features = [f for f in (x, binary, inp) if f is not None]
        features = torch.cat(features, 1)                         # model.py:513
        hidden = self.linear1(features).clamp(min=0)              # model.py:514
        return self.linear2(hidden)                               # model.py:515


# ----------------------------------------------------------------------------
# Conversation (model.py:725-876)
# ----------------------------------------------------------------------------
def exchange(sender, receiver, baseline_sen, baseline_rec, exchange_args, flags):
    data = exchange_args["data"]
    desc = exchange_args["desc"]
    train = exchange_args["train"]
    break_early = exchange_args.get("break_early", False)
    batch_size = data.size(0)

    stop_mask = [torch.ones(batch_size, 1, dtype=torch.uint8)]    # model.py:775
    stop_feat, stop_prob, sen_feats, sen_probs = [], [], [], []
    rec_feats, rec_probs, y, bs, br = [], [], [], [], []

    w_binary = torch.full((batch_size, sender.w_dim), float(flags.first_rec))   # model.py:786

    if train:
        sender.train(); receiver.train(); baseline_sen.train(); baseline_rec.train()
    else:
        sender.eval(); receiver.eval()
    sender.reset_state()
    receiver.reset_state()

    for i_exchange in range(flags.max_exchange):                  # model.py:801
        z_r = w_binary
        z_binary, z_probs = sender(data.detach(), z_r.detach(), None, i_exchange)   # model.py:810
        z_s = z_binary
        (s_binary, s_prob), (w_binary, w_probs), outp = receiver(
            z_s.detach(), desc.detach(), None, None)              # model.py:826
        if train:
            baseline_sen_scores = baseline_sen(sender.h_x.detach(), z_r.detach(), None)   # model.py:835
            baseline_rec_scores = baseline_rec(None, z_s.detach(), receiver.h_z.detach())  # model.py:842
        outp = outp.view(batch_size, -1)
        stop_mask.append(torch.min(stop_mask[-1], s_binary.to(torch.uint8)))   # model.py:852
        stop_feat.append(s_binary)
        stop_prob.append(s_prob)
        sen_feats.append(z_binary)
        sen_probs.append(z_probs)
        rec_feats.append(w_binary)
        rec_probs.append(w_probs)
        y.append(outp)
        if train:
            br.append(baseline_rec_scores)
            bs.append(baseline_sen_scores)
        if break_early and stop_mask[-1].float().sum().item() == 0:   # model.py:866
            break

    stop_mask[-1].fill_(0)                                        # model.py:870
    return (stop_mask, stop_feat, stop_prob), (sen_feats, sen_probs), (rec_feats, rec_probs), y, bs, br


def get_rec_outp(y, masks):
    """model.py:879-904."""
    def negent(yy):
        probs = F.softmax(yy, dim=1)
        return (torch.log(probs + 1e-8) * probs).sum(1).mean()
    negentropy = [negent(yy) for yy in y]
    if masks is not None:
        batch_size = y[0].size(0)
        exchange_steps = len(masks)
        inp = torch.cat([yy.view(batch_size, 1, -1) for yy in y], 1)
        mask = torch.cat(masks, 1).view(batch_size, exchange_steps, 1).expand_as(inp)
        outp = torch.masked_select(inp, mask.bool()).view(batch_size, -1)
        return outp, negentropy
    return y[-1], negentropy


def calculate_loss_binary(binary_features, binary_probs, logs, baseline_scores, entropy_penalty):
    """model.py:907-927 with the 0.1.12 keepdim semantics of ``sum(1)`` (line 911)."""
    log_p_z = binary_features.detach() * torch.log(binary_probs + 1e-8) + \
        (1 - binary_features.detach()) * torch.log(1 - binary_probs + 1e-8)
    log_p_z = log_p_z.sum(1, keepdim=True)
    weight = logs.detach() - baseline_scores.detach()
    if logs.size(0) > 1:
        weight = weight / max(1.0, torch.std(weight).item())      # model.py:915
    loss = torch.mean(-1 * weight * log_p_z)
    initial_negent = (torch.log(binary_probs + 1e-8) * binary_probs).sum(1).mean()
    inverse_negent = (torch.log((1. - binary_probs) + 1e-8) * (1. - binary_probs)).sum(1).mean()
    negentropy = initial_negent + inverse_negent
    if entropy_penalty is not None:
        loss = loss + entropy_penalty * negentropy
    return loss, negentropy


def multistep_loss_binary(binary_features, binary_probs, logs, baseline_scores, masks, entropy_penalty):
    """model.py:930-968."""
    if masks is not None:
        def mapped_fn(feat, prob, scores, mask, mask_sums):
            if mask_sums == 0:
                return torch.zeros(1), torch.zeros(1)
            m = mask.bool()
            feat = feat[m.expand_as(feat)].view(-1, feat.size(1))
            prob = prob[m.expand_as(prob)].view(-1, prob.size(1))
            _logs = logs[m.expand_as(logs)].view(-1, logs.size(1))
            scores = scores[m.expand_as(scores)].view(-1, scores.size(1))
            return calculate_loss_binary(feat, prob, _logs, scores, entropy_penalty)
        _mask_sums = [m.float().sum().item() for m in masks]
        outp = [mapped_fn(*a) for a in zip(binary_features, binary_probs, baseline_scores, masks, _mask_sums)]
        losses = [o[0] for o in outp]
        entropies = [o[1] for o in outp]
        loss = sum(l * ms for l, ms in zip(losses, _mask_sums)) / sum(_mask_sums)
    else:
        outp = [calculate_loss_binary(feat, prob, logs, scores, entropy_penalty)
                for feat, prob, scores in zip(binary_features, binary_probs, baseline_scores)]
        losses = [o[0] for o in outp]
        entropies = [o[1] for o in outp]
        loss = sum(losses) / len(binary_features)
    return loss, entropies


def multistep_loss_bas(baseline_scores, logs, masks):
    """model.py:971-988 (exact float counts; the uint8 wrap of 0.1.12 is not reproduced,
    SURVEY.md Appendix B3)."""
    if masks is not None:
        losses = [F.mse_loss(scores[mask.bool()].view(-1, 1), logs[mask.bool()].view(-1, 1).detach())
                  for scores, mask in zip(baseline_scores, masks)]
        _mask_sums = [m.sum().float() for m in masks]
        loss = sum(l * ms for l, ms in zip(losses, _mask_sums)) / sum(_mask_sums)
    else:
        losses = [F.mse_loss(scores, logs.detach()) for scores in baseline_scores]
        loss = sum(losses) / len(baseline_scores)
    return loss


# ----------------------------------------------------------------------------
# Model/optimizer construction (model.py:1014-1142) and one minibatch of the
# training loop (model.py:1219-1339)
# ----------------------------------------------------------------------------
def build_agents(flags, rng=None):
    sender = Sender(feat_dim=flags.img_feat_dim, h_dim=flags.img_h_dim, w_dim=flags.rec_w_dim,
                    bin_dim_out=flags.sender_out_dim, use_binary=flags.use_binary, rng=rng, flags=flags)
    baseline_sen = Baseline(hid_dim=flags.baseline_hid_dim, x_dim=flags.img_h_dim,
                            binary_dim=flags.rec_w_dim, inp_dim=0)
    receiver = Receiver(hid_dim=flags.rec_hidden, out_dim=flags.rec_out_dim, z_dim=flags.sender_out_dim,
                        desc_dim=flags.wv_dim, w_dim=flags.rec_w_dim, s_dim=flags.rec_s_dim,
                        use_binary=flags.use_binary, rng=rng, flags=flags)
    baseline_rec = Baseline(hid_dim=flags.baseline_hid_dim, x_dim=0,
                            binary_dim=flags.rec_w_dim, inp_dim=flags.rec_hidden)
    return dict(sender=sender, receiver=receiver, baseline_sen=baseline_sen, baseline_rec=baseline_rec)


def build_optimizers(models, flags):
    """model.py:1110-1142."""
    cls = {"SGD": optim.SGD, "Adam": optim.Adam, "RMSprop": optim.RMSprop}[flags.optim_type]
    return dict(optimizer_rec=cls(models["receiver"].parameters(), lr=flags.learning_rate),
                optimizer_sen=cls(models["sender"].parameters(), lr=flags.learning_rate),
                optimizer_bas_rec=cls(models["baseline_rec"].parameters(), lr=flags.learning_rate),
                optimizer_bas_sen=cls(models["baseline_sen"].parameters(), lr=flags.learning_rate))


def train_minibatch(models, optimizers, data, target, desc, flags, update=True):
    """One iteration of the loop at model.py:1218: exchange, masks, losses, four
    backward/clip/step blocks, top-k accuracy.  Returns everything a parity test
    wants to look at."""
    sender, receiver = models["sender"], models["receiver"]
    baseline_sen, baseline_rec = models["baseline_sen"], models["baseline_rec"]
    exchange_args = dict(data=data, target=target, desc=desc, train=True,
                         break_early=not flags.fixed_exchange)
    s, sen_w, rec_w, y, bs, br = exchange(sender, receiver, baseline_sen, baseline_rec, exchange_args, flags)
    s_masks, s_feats, s_probs = s
    sen_feats, sen_probs = sen_w
    rec_feats, rec_probs = rec_w

    if flags.fixed_exchange:                                      # model.py:1248-1262
        binary_s_masks = binary_rec_masks = binary_sen_masks = None
        bas_rec_masks = bas_sen_masks = y_masks = None
    else:
        binary_s_masks = s_masks[:-1]
        binary_rec_masks = s_masks[1:-1]
        binary_sen_masks = s_masks[:-1]
        bas_rec_masks = s_masks[:-1]
        bas_sen_masks = s_masks[:-1]
        y_masks = [torch.min(1 - m1, m2) for m1, m2 in zip(s_masks[1:], s_masks[:-1])]

    outp, ent_y_rec = get_rec_outp(y, y_masks)                    # model.py:1264
    dist = F.log_softmax(outp, dim=1)                             # model.py:1267
    nll_loss = F.nll_loss(dist, target)                           # model.py:1271
    logs = dist.detach().gather(1, target.view(-1, 1))            # model.py:1274

    zero = torch.zeros(1)
    loss_binary_s = loss_binary_rec = loss_binary_sen = zero
    if flags.use_binary:
        if not flags.fixed_exchange:                              # model.py:1278-1280
            loss_binary_s, _ = multistep_loss_binary(
                s_feats, s_probs, logs, br, binary_s_masks, flags.entropy_s)
        if len(rec_feats[:-1]) > 0:                               # model.py:1284-1289
            loss_binary_rec, _ = multistep_loss_binary(
                rec_feats[:-1], rec_probs[:-1], logs, br[:-1], binary_rec_masks, flags.entropy_rec)
        else:
            loss_binary_rec = torch.zeros(1)
        loss_binary_sen, _ = multistep_loss_binary(
            sen_feats, sen_probs, logs, bs, binary_sen_masks, flags.entropy_sen)
        loss_bas_rec = multistep_loss_bas(br, logs, bas_rec_masks)
        loss_bas_sen = multistep_loss_bas(bs, logs, bas_sen_masks)

    loss_rec = nll_loss                                           # model.py:1296-1305
    if flags.use_binary:
        loss_rec = loss_rec + loss_binary_rec
        if not flags.fixed_exchange:
            loss_rec = loss_rec + loss_binary_s
        loss_sen = loss_binary_sen
    else:
        loss_sen = loss_bas_rec = loss_bas_sen = zero

    grads, grad_norms = {}, {}

    def _update(opt_key, model_key, loss):
        opt, model = optimizers[opt_key], models[model_key]
        opt.zero_grad()
        loss.backward()
        grads[model_key] = {k: p.grad.detach().clone() for k, p in model.named_parameters()
                            if p.grad is not None}
        grad_norms[model_key] = float(nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.))
        if update:
            opt.step()

    _update("optimizer_rec", "receiver", loss_rec)                # model.py:1308-1311
    if flags.use_binary:                                          # model.py:1313-1330
        _update("optimizer_sen", "sender", loss_sen)
        _update("optimizer_bas_rec", "baseline_rec", loss_bas_rec)
        _update("optimizer_bas_sen", "baseline_sen", loss_bas_sen)

    top_k_ind = torch.from_numpy(dist.detach().numpy().argsort()[:, -flags.top_k_train:]).long()   # model.py:1333
    target_exp = target.view(-1, 1).expand(target.size(0), flags.top_k_train)
    hits = int((top_k_ind == target_exp).sum())
    accuracy = hits / float(flags.batch_size)                     # model.py:1337

    return dict(
        n_steps=len(y), s_masks=s_masks, s_feats=s_feats, s_probs=s_probs,
        sen_feats=sen_feats, sen_probs=sen_probs, rec_feats=rec_feats, rec_probs=rec_probs,
        y=y, bs=bs, br=br, outp=outp, dist=dist, logs=logs,
        nll_loss=nll_loss, loss_binary_s=loss_binary_s, loss_binary_rec=loss_binary_rec,
        loss_binary_sen=loss_binary_sen, loss_bas_rec=loss_bas_rec, loss_bas_sen=loss_bas_sen,
        grads=grads, grad_norms=grad_norms, accuracy=accuracy, hits=hits, top_k_ind=top_k_ind)


def eval_batch(models, data, target, desc, flags, top_k=None):
    """Per-batch body of eval_dev (model.py:612-668): deterministic conversation,
    output selection, top-k hits."""
    top_k = top_k or flags.top_k_dev
    exchange_args = dict(data=data, target=target, desc=desc, train=False,
                         break_early=not flags.fixed_exchange)
    with torch.no_grad():
        s, sen_w, rec_w, y, _, _ = exchange(models["sender"], models["receiver"], None, None,
                                            exchange_args, flags)
        s_masks, s_feats, s_probs = s
        if flags.fixed_exchange:
            y_masks = None
        else:
            y_masks = [torch.min(1 - m1, m2) for m1, m2 in zip(s_masks[1:], s_masks[:-1])]
        outp, _ = get_rec_outp(y, y_masks)
        dist = F.log_softmax(outp, dim=1)
    top_k_ind = torch.from_numpy(dist.numpy().argsort()[:, -top_k:]).long()     # model.py:658
    hits = int((top_k_ind == target.view(-1, 1).expand(target.size(0), top_k)).sum())
    conv_len = torch.cat(s_feats, 1).float().sum(1).view(-1).tolist()           # model.py:671
    return dict(n_steps=len(y), outp=outp, dist=dist, top_k_ind=top_k_ind, hits=hits,
                conversation_lengths=conv_len, s_masks=s_masks, s_feats=s_feats, s_probs=s_probs,
                sen_feats=sen_w[0], sen_probs=sen_w[1], rec_feats=rec_w[0], rec_probs=rec_w[1], y=y)


# ----------------------------------------------------------------------------
# Deterministic fillers owned by the repo (weights / inputs are regenerated from
# seeds rather than stored in fixtures).  numpy's legacy RandomState stream is
# frozen across numpy versions.
# ----------------------------------------------------------------------------
PARAM_ORDER = {
    "sender": ["image_layer.weight", "image_layer.bias", "code_layer.weight", "code_layer.bias",
               "code_bias", "binary_layer.weight", "binary_layer.bias"],
    "receiver": ["rnn.weight_ih", "rnn.weight_hh", "rnn.bias_ih", "rnn.bias_hh",
                 "w_h.weight", "w_h.bias", "w_d.weight", "w.weight", "w.bias",
                 "y1.weight", "y1.bias", "y2.weight", "y2.bias", "s.weight", "s.bias"],
    "baseline_rec": ["linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias"],
    "baseline_sen": ["linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias"],
}


def fill_state_dicts(shapes, seed=0, bias_scale=0.05):
    """shapes: {agent: {param_name: shape}} -> {agent: {param_name: float32 ndarray}}.
    2-D tensors get Xavier-normal scale, 1-D tensors N(0, bias_scale) (non-zero biases so
    that parity tests exercise them; code_bias N(0,1) as in model.py:97)."""
    rs = np.random.RandomState(seed)
    out = {}
    for agent in ("sender", "receiver", "baseline_rec", "baseline_sen"):
        out[agent] = {}
        for name in PARAM_ORDER[agent]:
            shape = tuple(shapes[agent][name])
            if len(shape) == 2:
                std = math.sqrt(2.0 / (shape[0] + shape[1]))
            elif name == "code_bias":
                std = 1.0
            else:
                std = bias_scale
            out[agent][name] = (rs.standard_normal(shape) * std).astype(np.float32)
    return out


def load_filled(models, seed=0, bias_scale=0.05):
    shapes = {a: {k: tuple(v.shape) for k, v in m.state_dict().items()} for a, m in models.items()}
    filled = fill_state_dicts(shapes, seed, bias_scale)
    for a, m in models.items():
        m.load_state_dict({k: torch.from_numpy(v) for k, v in filled[a].items()})
    return filled


def synthetic_batch(batch_size, n_classes, feat_dim=512, wv_dim=100, seed=1234):
    """features |N(0,1)|, targets uniform over classes, desc 0.3*N(0,1)."""
    rs = np.random.RandomState(seed)
    x = np.abs(rs.standard_normal((batch_size, feat_dim))).astype(np.float32)
    target = rs.randint(0, n_classes, size=(batch_size,)).astype(np.int64)
    desc = (0.3 * rs.standard_normal((n_classes, wv_dim))).astype(np.float32)
    return x, target, desc


def draw_uniforms(max_exchange, batch_size, w_dim, seed=0):
    """Uniforms in the reference's consumption order (z, s, w per step), rounded to float32
    so that ``u < p`` is the same comparison on every backend."""
    rs = np.random.RandomState(seed)
    u_z = np.empty((max_exchange, batch_size, w_dim), np.float32)
    u_s = np.empty((max_exchange, batch_size, 1), np.float32)
    u_w = np.empty((max_exchange, batch_size, w_dim), np.float32)
    for t in range(max_exchange):
        u_z[t] = rs.rand(batch_size, w_dim)
        u_s[t] = rs.rand(batch_size, 1)
        u_w[t] = rs.rand(batch_size, w_dim)
    return u_z, u_s, u_w


# ----------------------------------------------------------------------------
# Packing of one minibatch's results into flat npz entries (shared by the golden
# generator, which packs the REFERENCE's outputs, and the tests, which pack the
# oracle's / the HIP path's outputs the same way).
# ----------------------------------------------------------------------------
def _stack(lst):
    if len(lst) == 0 or lst[0] is None:
        return np.zeros((0,), np.float32)
    return np.stack([t.detach().numpy() for t in lst], 0)


def pack_train(res, models, prefix=""):
    """Flatten one minibatch's results into npz entries.  Big gradient/parameter tensors are
    stored as (norm, strided sample) pairs, small ones in full."""
    out = {}
    p = prefix
    out[p + "n_steps"] = np.int64(res["n_steps"])
    out[p + "s_masks"] = _stack(res["s_masks"]).astype(np.uint8)
    for k in ("s_feats", "s_probs", "sen_feats", "sen_probs", "rec_feats", "rec_probs", "y", "bs", "br"):
        out[p + k] = _stack(res[k]).astype(np.float32)
    for k in ("outp", "dist", "logs"):
        out[p + k] = res[k].detach().numpy().astype(np.float32)
    out[p + "losses"] = np.array([float(res[k].detach()) for k in (
        "nll_loss", "loss_binary_s", "loss_binary_rec", "loss_binary_sen", "loss_bas_rec", "loss_bas_sen")], np.float64)
    out[p + "hits"] = np.int64(res["hits"])
    for agent in ("receiver", "sender", "baseline_rec", "baseline_sen"):
        if agent not in res["grads"]:
            continue
        out[p + "gradnorm." + agent] = np.float64(res["grad_norms"][agent])
        for name, g in res["grads"][agent].items():
            g = g.numpy()
            out[p + "g.%s.%s.norm" % (agent, name)] = np.float64(np.linalg.norm(g.astype(np.float64)))
            flat = g.reshape(-1)
            stride = max(1, flat.size // 512)
            out[p + "g.%s.%s.sample" % (agent, name)] = flat[::stride].astype(np.float32)
        for name, w in models[agent].state_dict().items():
            w = w.numpy()
            out[p + "p.%s.%s.sum" % (agent, name)] = np.float64(w.astype(np.float64).sum())
            flat = w.reshape(-1)
            stride = max(1, flat.size // 512)
            out[p + "p.%s.%s.sample" % (agent, name)] = flat[::stride].astype(np.float32)
    return out
