"""Agent modules with the reference's constructor signatures, forward() contracts, side-effect
attributes and state_dict keys (model.py:49-516), backed by the HIP library.

The modules hold ordinary ``nn.Parameter``s so ``state_dict`` / ``load_state_dict`` / ``.cuda()`` /
checkpoints behave as in the reference.  When a :class:`multimodalgame_amd.game.Game` adopts them,
each parameter's storage becomes a view into the engine's flat parameter buffer, which is what the
kernels read and the fused optimizer updates in place.

forward() is forward-only (no autograd graph): gradients are produced by the hand-written backward
kernels through ``Game.train_step`` -- the counterpart of model.py:1243-1330.
"""
import math

import torch
import torch.nn as nn

from . import flags as _flags


def xavier_normal(tensor, gain=1.0, generator=None):
    """misc.py:367-385: N(0, gain * sqrt(2 / (fan_in + fan_out)))."""
    fan_out, fan_in = tensor.size(0), tensor.size(1)
    std = gain * math.sqrt(2.0 / (fan_in + fan_out))
    with torch.no_grad():
        return tensor.normal_(0, std, generator=generator)


def _linear_default_(weight, bias, generator=None):
    """torch.nn.Linear.reset_parameters (the Baselines keep the default init, model.py:480-494)."""
    bound = 1.0 / math.sqrt(weight.size(1))
    with torch.no_grad():
        weight.uniform_(-bound, bound, generator=generator)
        if bias is not None:
            bias.uniform_(-bound, bound, generator=generator)


def init_state_dicts(engine, seed=0):
    """Reference initialisation (model.py:90-97, 275-288; Baseline default) for all four agents,
    from one CPU generator so that every rank gets identical weights."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for agent, d in engine.params.items():
        out[agent] = {}
        for name, view in d.items():
            t = torch.zeros(view.shape)
            if agent.startswith("baseline"):
                fan_in = engine.params[agent][name.split(".")[0] + ".weight"].shape[1]
                bound = 1.0 / math.sqrt(fan_in)
                t.uniform_(-bound, bound, generator=g)
            elif t.dim() == 2:
                xavier_normal(t, generator=g)
            elif name == "code_bias":
                t.normal_(generator=g)
            out[agent][name] = t
    return out


class _Agent(nn.Module):
    agent_name = None

    def __init__(self):
        super().__init__()
        self._game = None

    def _bound_game(self, batch_size):
        if self._game is None:
            raise RuntimeError(
                "%s is not attached to a Game: construct multimodalgame_amd.game.Game(sender, receiver, "
                "baseline_sen, baseline_rec, ...) (exchange() does this on first use)" % type(self).__name__)
        return self._game


class Sender(_Agent):
    """model.py:49-238 (non-attention, sender_mix == 'sum')."""
    agent_name = "sender"

    def __init__(self, feature_type, feat_dim, h_dim, w_dim, bin_dim_out, use_binary,
                 use_attn=False, attn_dim=256, attn_extra_context=False, attn_context_dim=4096):
        super().__init__()
        if use_attn:
            raise NotImplementedError("visual attention (-visual_attn) is outside the accelerated hot path "
                                      "(SURVEY.md §2); use -model_type Fixed/Adaptive")
        self.feature_type, self.feat_dim, self.h_dim, self.w_dim = feature_type, feat_dim, h_dim, w_dim
        self.bin_dim_out, self.use_binary, self.use_attn = bin_dim_out, use_binary, use_attn
        self.attn_dim, self.attn_extra_context, self.attn_context_dim = attn_dim, attn_extra_context, attn_context_dim
        self.image_layer = nn.Linear(feat_dim, h_dim)
        self.code_layer = nn.Linear(w_dim, h_dim)
        self.code_bias = nn.Parameter(torch.zeros(bin_dim_out))
        self.binary_layer = nn.Linear(h_dim, bin_dim_out)
        self.h_x = None
        self.reset_parameters()

    def reset_parameters(self):                                           # model.py:90-97
        for m in self.modules():
            if isinstance(m, nn.Linear):
                xavier_normal(m.weight.data)
                m.bias.data.zero_()
        self.code_bias.data.normal_()

    def reset_state(self):                                                # model.py:99-112
        self.attn_scores = []

    def forward(self, x, w, g, t):
        game = self._bound_game(x.size(0))
        msg, probs, h_x = game.engine_for(x.size(0)).sender_forward(
            x.contiguous(), None if t == 0 else w.contiguous(), t, self.training, seed=game.next_seed())
        self.h_x = h_x                                                    # model.py:195 side effect
        return msg, probs


class Receiver(_Agent):
    """model.py:241-477 (non-desc_attn)."""
    agent_name = "receiver"

    def __init__(self, z_dim, desc_dim, hid_dim, out_dim, w_dim, s_dim, use_binary):
        super().__init__()
        if out_dim != 1 or s_dim != 1:
            raise NotImplementedError("rec_out_dim and rec_s_dim must be 1 (the reference's only working setting)")
        self.z_dim, self.desc_dim, self.hid_dim = z_dim, desc_dim, hid_dim
        self.out_dim, self.w_dim, self.s_dim, self.use_binary = out_dim, w_dim, s_dim, use_binary
        self.rnn = nn.GRUCell(z_dim, hid_dim)
        self.w_h = nn.Linear(hid_dim, hid_dim, bias=True)
        self.w_d = nn.Linear(desc_dim, hid_dim, bias=False)
        self.w = nn.Linear(hid_dim, w_dim)
        self.y1 = nn.Linear(hid_dim + desc_dim, hid_dim)
        self.y2 = nn.Linear(hid_dim, out_dim)
        self.s = nn.Linear(hid_dim, s_dim)
        self.reset_parameters()
        self.reset_state()

    def reset_parameters(self):                                           # model.py:275-288
        for m in self.modules():
            if isinstance(m, nn.Linear):
                xavier_normal(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.GRUCell):
                for mm in m.parameters():
                    if mm.data.ndimension() == 2:
                        xavier_normal(mm.data)
                    else:
                        mm.data.zero_()

    def reset_state(self):                                                # model.py:290-298
        self.h_z = None
        self.s_prob_prod = None
        self.h_w = None
        self._t = 0

    def initial_state(self, batch_size):                                  # model.py:300-301
        return torch.zeros(batch_size, self.hid_dim, device=self.w.weight.device)

    def forward(self, z, desc, desc_set=None, desc_set_lens=None):
        game = self._bound_game(z.size(0))
        B = z.size(0)
        first = self.h_z is None
        h_z = self.initial_state(B) if first else self.h_z.clone()
        sprod = torch.ones(B, device=z.device) if self.s_prob_prod is None else self.s_prob_prod.view(-1).clone()
        s, s_prob, w, w_probs, y, h_w = game.engine_for(B).receiver_forward(
            z.contiguous(), desc.contiguous(), h_z, sprod, first, min(self._t, game.max_exchange - 1),
            self.training, seed=game.next_seed())
        self._t += 1
        self.h_z, self.h_w = h_z, h_w                                     # model.py:340, 452 side effects
        if not self.training:
            self.s_prob_prod = sprod.view(B, 1)                           # model.py:423-426
        return (s, s_prob), (w, w_probs), y


class Baseline(_Agent):
    """model.py:480-516."""

    def __init__(self, hid_dim, x_dim, binary_dim, inp_dim):
        super().__init__()
        self.x_dim, self.binary_dim, self.inp_dim, self.hid_dim = x_dim, binary_dim, inp_dim, hid_dim
        self.linear1 = nn.Linear(x_dim + binary_dim + inp_dim, hid_dim)
        self.linear2 = nn.Linear(hid_dim, 1)
        self.agent_name = "baseline_sen" if inp_dim == 0 else "baseline_rec"

    def forward(self, x, binary, inp):
        game = self._bound_game(binary.size(0))
        return game.engine_for(binary.size(0)).baseline_forward(
            self.agent_name, None if x is None else x.contiguous(), binary.contiguous(),
            None if inp is None else inp.contiguous())
