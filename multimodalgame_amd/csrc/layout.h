// layout.h -- single source of truth for (a) the flat parameter buffer and (b) the workspace
// ("tape") of the MI355X exchange path.  Host code only; kernels receive the resolved pointers
// in `Params` / `Tape` structs passed by value.
#pragma once
#include <stdint.h>
#include <string.h>
#include "../../include/mmg.h"
#ifndef __HIPCC__
#define __host__
#define __device__
#endif

namespace mmg {

// ---------------------------------------------------------------------------------------------
// Parameters.  Order: receiver, sender, baseline_rec, baseline_sen -- the order of the four
// update blocks at model.py:1307-1330; each agent's tensors are contiguous so that the per-agent
// gradient norm is a reduction over one range.  Tensor names are the reference's state_dict keys.
// ---------------------------------------------------------------------------------------------
enum ParamId {
    R_WIH, R_WHH, R_BIH, R_BHH, R_WH_W, R_WH_B, R_WD_W, R_W_W, R_W_B, R_Y1_W, R_Y1_B, R_Y2_W, R_Y2_B, R_S_W, R_S_B,
    S_IMG_W, S_IMG_B, S_CODE_W, S_CODE_B, S_CODE_BIAS, S_BIN_W, S_BIN_B,
    BR_L1_W, BR_L1_B, BR_L2_W, BR_L2_B,
    BS_L1_W, BS_L1_B, BS_L2_W, BS_L2_B,
    P_COUNT
};

struct ParamLayout {
    int64_t off[P_COUNT];
    int32_t rows[P_COUNT], cols[P_COUNT], agent[P_COUNT];
    const char* name[P_COUNT];
    int64_t agent_begin[5];   // agent a occupies [agent_begin[a], agent_begin[a+1])
    int64_t total;
};

inline ParamLayout param_layout(const mmg_config& c) {
    ParamLayout L;
    const int H = c.h_dim, W = c.w_dim, R = c.rec_hidden, V = c.wv_dim, K = c.bas_hidden, F = c.feat_dim;
    struct E { ParamId id; const char* n; int agent, rows, cols; };
    const E tab[P_COUNT] = {
        {R_WIH, "rnn.weight_ih", 0, 3 * R, W}, {R_WHH, "rnn.weight_hh", 0, 3 * R, R},
        {R_BIH, "rnn.bias_ih", 0, 3 * R, 0}, {R_BHH, "rnn.bias_hh", 0, 3 * R, 0},
        {R_WH_W, "w_h.weight", 0, R, R}, {R_WH_B, "w_h.bias", 0, R, 0}, {R_WD_W, "w_d.weight", 0, R, V},
        {R_W_W, "w.weight", 0, W, R}, {R_W_B, "w.bias", 0, W, 0},
        {R_Y1_W, "y1.weight", 0, R, R + V}, {R_Y1_B, "y1.bias", 0, R, 0},
        {R_Y2_W, "y2.weight", 0, 1, R}, {R_Y2_B, "y2.bias", 0, 1, 0},
        {R_S_W, "s.weight", 0, 1, R}, {R_S_B, "s.bias", 0, 1, 0},
        {S_IMG_W, "image_layer.weight", 1, H, F}, {S_IMG_B, "image_layer.bias", 1, H, 0},
        {S_CODE_W, "code_layer.weight", 1, H, W}, {S_CODE_B, "code_layer.bias", 1, H, 0},
        {S_CODE_BIAS, "code_bias", 1, W, 0},
        {S_BIN_W, "binary_layer.weight", 1, W, H}, {S_BIN_B, "binary_layer.bias", 1, W, 0},
        {BR_L1_W, "linear1.weight", 2, K, W + R}, {BR_L1_B, "linear1.bias", 2, K, 0},
        {BR_L2_W, "linear2.weight", 2, 1, K}, {BR_L2_B, "linear2.bias", 2, 1, 0},
        {BS_L1_W, "linear1.weight", 3, K, H + W}, {BS_L1_B, "linear1.bias", 3, K, 0},
        {BS_L2_W, "linear2.weight", 3, 1, K}, {BS_L2_B, "linear2.bias", 3, 1, 0},
    };
    int64_t o = 0;
    int cur_agent = -1;
    for (int i = 0; i < P_COUNT; ++i) {
        const E& e = tab[i];
        if (e.agent != cur_agent) { cur_agent = e.agent; L.agent_begin[cur_agent] = o; }
        L.off[e.id] = o; L.rows[e.id] = e.rows; L.cols[e.id] = e.cols; L.agent[e.id] = e.agent; L.name[e.id] = e.n;
        int64_t n = (int64_t)e.rows * (e.cols ? e.cols : 1);
        o += (n + 3) & ~(int64_t)3;     // every tensor starts on a 16-byte boundary (float4 loads)
    }
    L.agent_begin[4] = o;
    L.total = o;
    return L;
}

// Resolved device pointers of all parameter tensors (into the flat buffer).
struct Params { float* p[P_COUNT]; };

inline Params resolve_params(const ParamLayout& L, float* base) {
    Params P;
    for (int i = 0; i < P_COUNT; ++i) P.p[i] = base + L.off[i];
    return P;
}

// ---------------------------------------------------------------------------------------------
// Tape.  X(name, ctype, dtype_code, ndim, d0, d1, d2)
// Symbols available in the dims: B D F H W R V K T T1(=T+1) NSTAT NPART
// ---------------------------------------------------------------------------------------------
#define MMG_GN_BLOCKS 128      // blocks (= partial sums) of the gradient-norm kernel
#define MMG_TAPE_LIST(X)                                                          \
    /* ---- forward ---- */                                                        \
    X(hx, float, 0, 2, B, H, 1)          /* image_layer(x)            model.py:195 */ \
    X(Cd, float, 0, 2, D, R, 1)          /* desc . W_y1[:,R:]^T + b_y1 (App. A.2)  */ \
    X(descc, float, 0, 2, D, V, 1)       /* copy of the description matrix (GEMM operand inside the workspace) */ \
    X(Dd, float, 0, 2, D, R, 1)          /* desc . w_d^T: W_d (softmax(y) . desc) = softmax(y) . Dd (kernels_fast3.h)    */ \
    X(CdT, float, 0, 2, R, D, 1)         /* -Cd transposed: class index contiguous (kernels_tile.h, many-class y head) */ \
    X(cy, float, 0, 1, D, 1, 1)          /* b_y2 + sum_r w_y2[r] Cd[d][r]                                              */ \
    X(mstate, float, 0, 1, B, 1, 1)      /* running stop mask m_t between the per-step launches of one conversation */ \
    X(Apub, float, 0, 2, NTILE, 16 * R + 32, 1)  /* k_conv_split: A tile + row flags published by a tile's owner role       */ \
    X(cpart, float, 0, 3, NTILE * NHLP, 16, V + 2) /* k_conv_split: per helper: unnormalised mixture partial | slice max | slice sum */ \
    X(gip, float, 0, 3, NS2P, B, 3 * R)  /* per-sender-role partials of the GRU input product (k_conv_persist)           */ \
    X(zpart, float, 0, 3, NZP, 16, W)    /* per-SA-role partial message logits of a tile (k_conv_persist, fused sender roles) */ \
    X(rcgw, float, 0, 2, NRCT16, R, 1)   /* wide receiver (kernels_rc.h): w_h h + b_h of the step, [tile][R/16][r][q][16]: the query phase's MFMA accumulator order */ \
    X(rcyp, float, 0, 3, NRCT16, 32, 16) /* ... per-slice partial class logits [tile][slice quad][16 samples][32 classes][4], added in slice order by k_rc_query */ \
    X(rclw, float, 0, 3, 2 * NRCW, NRCB, 2)  /* ... per-16-bit-slice partial (log-likelihood, neg-entropy) of the receiver's message            */ \
    X(rcx, float, 0, 3, NRCX, 16, 3 * R) /* ... backward: the tile's gate gradients dgh_t, double-buffered by step parity (all-gather between k_rc_bwd's roles) */ \
    X(rcflags, uint32_t, 2, 1, 64 * 64, 1, 1) /* ... backward: one hand-off counter per sample tile (256-byte blocks), zeroed by the forward launch (k_rc_persist / k_rc_tail) */ \
    X(rcxa, float, 0, 3, NRCA, 16, 16)   /* ... the sender's hidden tile a_t in FRAGMENT order [tile][H/16 k-groups][16 samples][16]: a wave of an S2 role reads a k-group's 1 KB contiguously */ \
    X(rcxz, float, 0, 3, NRCZ, 16, 16)   /* ... z_t in fragment order [tile][W/16][16][16] (S2 roles -> GRU slices)                              */ \
    X(rcxw, float, 0, 3, NRCZ, 16, 16)   /* ... w_t in fragment order [tile][W/16][16][16] (message slices -> S1 roles)                        */ \
    X(rcxh, float, 0, 3, 2 * NRCH, 16, 16) /* ... h_{t+1} in fragment order [parity][tile][R/16][16][16] (GRU slices -> heads, next GRU step)   */ \
    X(rcst, float, 0, 2, 4, B, 1)        /* ... [0..1] running stop mask m_t, double-buffered by step parity; [2] take-output flag of the step */ \
    X(pflags, uint32_t, 2, 1, 20 * 64 * 64, 1, 1) /* k_conv_persist: per (kind, sample tile) counters, one per 256-byte block */ \
    X(pll_w, float, 0, 3, NPLT, B, 2 * W)   /* k_conv_persist<LL>: (value, tag) pairs of the receiver message w_t, tag = (launch epoch << 4) | step: receiver sample roles -> SA roles */ \
    X(pll_m, float, 0, 2, NPLT, 2 * B, 1)   /* ... pairs of the running stop mask after step t (the sender roles' live rows; all zero: the tile's conversation is over) */ \
    X(pll_z, float, 0, 3, NPLT, B, 2 * W)   /* ... pairs of the sender message z_t, binary mode: probability with the bit in the SIGN (+p: 1, -p: 0): SB roles -> receiver sample roles */ \
    X(pll_zp, float, 0, 3, NPLT * NZP, 16, 2 * W) /* ... pairs of the SA roles' partial message logits [tile][role][16][W]: SA -> SB roles */ \
    X(pll_gi, float, 0, 3, NPLT * NPLS, B, 2 * 3 * R) /* ... pairs of the SB roles' partials of the GRU input product [W / 16][B][3R]: SB -> receiver sample roles */ \
    X(mcA, float, 0, 3, NMC, 16, R + 4)  /* k_conversation_mc: A rows (+ take flag) published by the 16 members of a tile       */ \
    X(mcpart, float, 0, 3, NMC * 16, 16, 104) /* k_conversation_mc: per (tile, class slice): [16 samples][V mixture terms | m | s | pad] */ \
    X(mc3A, float, 0, 3, NMC, 16, 2 * (R + 4)) /* k_conversation_mc3: (value, epoch) pairs of the A rows (+ take flag) of a tile's 16 members */ \
    X(mc3P, float, 0, 3, NMC * 16, 16, 2 * 68) /* k_conversation_mc3: (value, epoch) pairs per (tile, class slice): [16 samples][R mixture terms | m | s | pad] */ \
    X(mcflags, uint32_t, 2, 1, 2 * NMC * 64, 1, 1) /* k_conversation_mc: per (kind, tile) hand-off counters, one per 256-byte block (zeroed by k_prep) */ \
    X(mcdA, float, 0, 3, NMCB, B, R)     /* k_bwd_mc1: per class block: partial dA of every sample                        */ \
    X(mcdys, float, 0, 2, NMCB, B, 1)    /* k_bwd_mc1: per class block: partial sum_d dy                                  */ \
    X(mcdC, float, 0, 3, NMCB, 2 * D, R) /* k_bwd_mc1: per sample group: partial dC | Py2                                 */ \
    X(prepll, float, 0, 1, 2 * (B * H + H + 2 * D * R), 1, 1) /* (value, launch epoch) pairs of h_x | hw0 | Cd | Dd: hand-off from the prep roles to the sample roles of ONE launch (kernels_fast3.h) */ \
    X(wrep, float, 0, 1, 30 * 256 * 4, 1, 1) /* the register-resident backward's transposed weight fragments, repacked per lane by k_prep's blocks: [30 float4][256 threads] (kernels_fwd.h: prep_repack) */ \
    X(cd32, float, 0, 1, 8 * R * 4, 1, 1) /* Cd of the first 32 classes as [8 float4][R] (class 4 j + c of unit r at ((j R + r) 4 + c)): the register-resident backward's column of Cd in 8 lane-consecutive loads */ \
    X(basell, float, 0, 1, 2 * B * K, 1, 1) /* k_game_fast: (value, epoch) pairs of basehx [B][K]: hand-off from the basehx tiles to the baseline roles of ONE launch (kernels_game.h) */ \
    X(gamell, float, 0, 1, 2 * (2 * B + 16), 1, 1) /* k_game_fast: (value, epoch) pairs [0, B): pair A = t*(b), published by sample role b once the rows the baseline roles multiply (z, z_r, h) have been written through; [B, B + 8): "repack block k is through"; [B + 16, 2 B + 16): pair B = t*(b) again, once what the statistics roles read (reward, hit, log-likelihood sums) is through */ \
    X(alive, int32_t, 2, 1, T + 2, 1, 1) /* [t]: sample tiles with a live sample when step t starts (kernels_tile.h)  */ \
    X(hw0, float, 0, 1, H, 1, 1)         /* code_layer(sigmoid(code_bias)) :199-200*/ \
    X(dsig, float, 0, 1, W, 1, 1)        /* sigmoid'(code_bias)                    */ \
    X(c, float, 0, 3, T, B, W)           /* sender code input per step             */ \
    X(zr, float, 0, 3, T, B, W)          /* z_r: baseline_sen binary input   :836  */ \
    X(a, float, 0, 3, T, B, H)           /* tanh(h_x + h_w)                  :216  */ \
    X(z, float, 0, 3, T, B, W)           /* sender message (sen_feats)       :855  */ \
    X(pz, float, 0, 3, T, B, W)          /* sender probs   (sen_probs)       :856  */ \
    X(h, float, 0, 3, T1, B, R)          /* GRU state, h[0]=0, h[t+1]=h_z after t  */ \
    X(gru, float, 0, 3, T, B, 4 * R)     /* r, u, n, W_hn h + b_hn per step        */ \
    X(s, float, 0, 3, T, B, 1)           /* stop bits (stop_feat)            :853  */ \
    X(ps, float, 0, 3, T, B, 1)          /* stop probs (stop_prob)           :854  */ \
    X(sprod, float, 0, 1, B, 1, 1)       /* eval: running product of stop probs    */ \
    X(y, float, 0, 3, T, B, D)           /* class logits per step            :859  */ \
    X(dbar, float, 0, 3, T, B, V)        /* softmax(y) . desc                :449  */ \
    X(pi, float, 0, 3, T, B, 32)         /* softmax(y_t), <= 32 classes (kernels_fast3.h): dbar = pi . desc is formed by roles of the backward launch */ \
    X(g, float, 0, 3, T, B, R)           /* receiver.h_w                     :452  */ \
    X(w, float, 0, 3, T, B, W)           /* receiver message (rec_feats)     :857  */ \
    X(pw, float, 0, 3, T, B, W)          /* receiver probs (rec_probs)       :858  */ \
    X(mask, uint8_t, 1, 3, T1, B, 1)     /* stop_mask list                   :775  */ \
    X(tstar, int32_t, 2, 1, B, 1, 1)     /* step whose logits are the output :1261 */ \
    X(lp_z, float, 0, 2, T, B, 1)        /* sum_j log-lik of sampled bits    :908  */ \
    X(ne_z, float, 0, 2, T, B, 1)        /* sum_j neg-entropy terms          :919  */ \
    X(lp_s, float, 0, 2, T, B, 1)                                                  \
    X(ne_s, float, 0, 2, T, B, 1)                                                  \
    X(lp_w, float, 0, 2, T, B, 1)                                                  \
    X(ne_w, float, 0, 2, T, B, 1)                                                  \
    X(hid_s, float, 0, 3, T, B, K)       /* baseline_sen relu hidden         :514  */ \
    X(hid_r, float, 0, 3, T, B, K)                                                 \
    X(basehx, float, 0, 2, B, K, 1)      /* h_x . baseline_sen.linear1.weight[:, :H]^T (same for every step of a sample) */ \
    X(bs_part, float, 0, 3, T, B, NPB)   /* partial scores per 64 hidden units     */ \
    X(br_part, float, 0, 3, T, B, NPB)                                             \
    X(bs, float, 0, 3, T, B, 1)          /* baseline_sen scores              :835  */ \
    X(br, float, 0, 3, T, B, 1)          /* baseline_rec scores              :842  */ \
    X(outp, float, 0, 2, B, D, 1)        /* get_rec_outp                     :1264 */ \
    X(dist, float, 0, 2, B, D, 1)        /* log_softmax(outp)                :1267 */ \
    X(sm, float, 0, 2, B, D, 1)          /* softmax(outp)                          */ \
    X(logs, float, 0, 1, B, 1, 1)        /* loglikelihood(dist, target)      :1274 */ \
    X(hit, int32_t, 2, 1, B, 1, 1)       /* target within top-k              :1333 */ \
    X(partll, float, 0, 1, 2 * 2 * T * B * NPB, 1, 1) /* (value, epoch) pairs of the baselines' partial scores [bs | br][T B][NPB]: hand-off from the baseline roles to the statistics roles of the backward launch */ \
    X(statll, float, 0, 1, 2 * (27 * T + 2 * T * B), 1, 1) /* (value, epoch) pairs of the stream statistics (per (stream, step): n | four f64 sums as 2 halves each) and of bs | br: hand-off from the statistics roles to the sample roles of the backward launch (kernels_fast.h) */ \
    X(stats, double, 3, 1, NSTAT, 1, 1)  /* batch statistics (all-reduced in DP)   */ \
    X(losses, float, 0, 1, 8, 1, 1)      /* nll, bin_s, bin_rec, bin_sen, bas_rec, bas_sen, n_steps, hits */ \
    X(counter, uint32_t, 2, 1, 4, 1, 1)  /* [0] minibatch counter (Philox), [1] optimizer step, [3] launch epoch of the (value, epoch) pair hand-offs: bumped with [0], never copied between engines (game.py), never reset */ \
    X(rmap, int32_t, 2, 1, T * B, 1, 1)  /* compacted list of the (step, sample) rows with t <= t*(b), in (t, b) order   */ \
    X(rcount, int32_t, 2, 1, 4, 1, 1)    /* [0] its length (k_wgrad reduces over these rows only)                          */ \
    X(sync, uint32_t, 2, 1, 512, 1, 1)    /* in-launch dependency counters between workgroup roles (device_utils.h: role_signal) */ \
    X(dbg, long long, 3, 1, 256, 1, 1)   /* debug timestamps (MMG_TIMING builds) */ \
    X(dbg2, long long, 3, 1, 16384, 1, 1) /* per-block start/end stamps of k_wgrad (MMG_TIMING builds) */ \
    X(totals, double, 3, 1, 4, 1, 1)     /* running sums over train steps: exchange steps, top-k hits, minibatches, sample-steps */ \
    /* ---- backward ---- */                                                       \
    X(dlz, float, 0, 3, T, B, W)         /* dL/d sender logits                     */ \
    X(dpre, float, 0, 3, T, B, H)        /* dL/d (h_x + h_w)                       */ \
    X(dhx, float, 0, 2, B, H, 1)         /* sum_t dpre                             */ \
    X(dc0, float, 0, 2, B, W, 1)         /* W_c^T dpre_0 (code_bias path)          */ \
    X(u0, float, 0, 1, H, 1, 1)          /* sum_b dpre_0[b, :] (code_bias path of the tile kernels) */ \
    X(dls, float, 0, 2, T, B, 1)         /* dL/d stop logit                        */ \
    X(dlw, float, 0, 3, T, B, W)         /* dL/d receiver message logits           */ \
    X(dgpre, float, 0, 3, T, B, R)       /* dL/d pre-tanh of h_w                   */ \
    X(dhin, float, 0, 3, T * NRCP, B, R)        /* dgpre W_h + dls w_s: what a step adds to dh besides the recurrence (k_bwd_pre) */ \
    X(dgi, float, 0, 3, T, B, 3 * R)     /* dL/d GRU input-side gate pre-acts      */ \
    X(dgh, float, 0, 3, T, B, 3 * R)     /* dL/d GRU hidden-side gate pre-acts     */ \
    X(dA, float, 0, 2, B, R, 1)          /* dL/d (W_y1h h) at t*                   */ \
    X(Astar, float, 0, 2, B, R, 1)       /* W_y1h h at t*                          */ \
    X(hstar, float, 0, 2, B, R, 1)       /* h at t*                                */ \
    X(dy, float, 0, 2, B, D, 1)          /* dNLL/d outp                            */ \
    X(dyT, float, 0, 2, D, B, 1)         /* its transpose (k_dC_tile reads a class's column coalesced) */ \
    X(dCpart, float, 0, 3, NDCS, 2 * D, R) /* per-sample-slice partials of dC | Py2 (k_dC_tile) */ \
    X(dysum, float, 0, 1, B, 1, 1)                                                 \
    X(dC, float, 0, 2, D, R, 1)          /* dL/d Cd                                */ \
    X(Py2, float, 0, 2, D, R, 1)         /* per-class partials of dL/d w_y2        */ \
    X(dbs, float, 0, 2, T, B, 1)                                                   \
    X(dbr, float, 0, 2, T, B, 1)                                                   \
    X(gnpart, float, 0, 1, 16384 + NPART, 1, 1)  /* squared grad-norm partials (k_wgrad blocks | k_gradnorm) */ \
    X(gnll, float, 0, 1, 2 * 16384, 1, 1) /* k_wgrad<OPT>: (value, epoch) pairs of the blocks' sums of squares (hand-off to the norm role of the same launch) */ \
    X(coefll, float, 0, 1, 2 * (4 * 64 + 4), 1, 1) /* k_wgrad<OPT>: (value, epoch) pairs of the four clip coefficients, 64 replicas each, then the four agents' squared norms */ \
    X(ones, float, 0, 1, 256, 1, 1)      /* 1.0f: B operand of bias gradients run as K = 1 GEMM jobs (many (step, sample) rows) */ \
    X(wcnt, int32_t, 2, 1, 16384, 1, 1)  /* k_wgrad: slices of an output tile that have written their partial tile (the last one adds them up and zeroes the count) */ \
    X(wpart, float, 0, 1, NWP, 1, 1)     /* raw partial tiles of weight gradients whose rows are split over workgroups (k_wgrad) */ \
    X(tables, uint8_t, 1, 1, 98304, 1, 1) /* GEMM / column-sum job descriptors      */

// statistics vector (f64).  Per stream (0 = stop bits, 1 = receiver msgs, 2 = sender msgs) and
// step: n, sum w, sum w^2, sum w*logp, sum negent; per baseline (0 = rec, 1 = sen) and step:
// sum (beta-L)^2; then sum logs, hits.
#define MMG_ST_PER 5
__host__ __device__ inline int stat_stream(int T, int stream, int t, int k) { return (stream * T + t) * MMG_ST_PER + k; }
__host__ __device__ inline int stat_bas(int T, int which, int t) { return 3 * T * MMG_ST_PER + which * T + t; }
__host__ __device__ inline int stat_glob(int T, int k) { return 3 * T * MMG_ST_PER + 2 * T + k; }
__host__ __device__ inline int stat_count(int T) { return 3 * T * MMG_ST_PER + 2 * T + 4; }

struct Tape {
#define X(name, ctype, code, nd, d0, d1, d2) ctype* name;
    MMG_TAPE_LIST(X)
#undef X
};

// row slices per output tile of k_wgrad's (step, sample)-row jobs: with thousands of rows the few dozen output tiles of the
// small receiver matrices would otherwise be a handful of long-running workgroups
__host__ __device__ inline int wgrad_nsplit(int TB, long long ptotal) {
    int n = TB / 2048;
    n = TB >= 4096 ? (n > 16 ? 16 : n) : 1;
    const int cap = (int)(12000 / (ptotal / 512 + 64));          // (k_wgrad addresses at most 16384 workgroups: ~ptotal / 512 output tiles x slices)
    return n > cap ? (cap < 1 ? 1 : cap) : n;
}

// ... per job: with more than 2048 (step, sample) rows a job with few output tiles (the receiver's small matrices: a dozen
// tiles that would each walk all the rows, 0.7 us per 64) is split further, down to ~5 chunks of 64 rows per workgroup
__host__ __device__ inline int wgrad_job_nsplit(int TB, long long ptotal, int job_tiles) {
    int n = wgrad_nsplit(TB, ptotal);
    if (TB > 2048 && job_tiles <= 64) { int m = TB / 320; m = m > 16 ? 16 : m; if (m > n) n = m; }
    return n;
}
__host__ __device__ inline bool wgrad_any_split(int TB, long long ptotal) { return TB > 2048 || wgrad_nsplit(TB, ptotal) > 1; }

// class helpers per sample tile of the many-class forward (k_conv_split): every CU the sample tiles leave idle takes a
// slice of the classes; 0: no split
__host__ __device__ inline int split_helpers(int B) { const int tiles = (B + 15) / 16; int nh = 224 / tiles - 1; return nh > 15 ? 15 : (nh < 0 ? 0 : nh); }   // (all roles must be co-resident: margin below the 256 CUs)

// sample slices of the class-side reduction (k_dC_tile): enough workgroups for the chip when there are few class blocks
// many-class register-resident conversation (kernels_mc.h): the small-agent shape with more classes than the 32 a sample's
// own workgroup holds, up to 16 slices x 64 classes
__host__ __device__ inline bool mc_shape(int H, int W, int R, int V, int D, int T) { return H == 256 && W == 32 && R == 64 && V == 100 && D > 32 && D <= 1024 && T <= 16; }

// wide receiver (kernels_rc.h): rec_hidden beyond the one-workgroup-per-tile forward (its LDS plan stops at R = 128 with a
// 256-bit message), beside the large sender / few samples of the per-step sender launches
__host__ __device__ inline bool rc_shape(int B, int H, int W, int R, int V, int D) {
    return R > 128 && R <= 256 && !(R & 15) && !(W & 15) && W <= 256 && (B + 15) / 16 < 64 && (long long)H * W >= 65536 &&
           D <= 32 && V <= 128 && !(V & 3) && !(H & 3);
}

// k_conv_persist with (value, epoch) pair hand-offs (kernels_tile.h: rs_role / sa_role / sb_role <LL>): the fused sender roles' shape
__host__ __device__ inline bool persist_ll_shape(int B, int H, int W, int R, int V, int D, int T) {
    return W == 256 && R == 64 && V == 100 && D <= 32 && T <= 15 && H % 64 == 0 && H / 64 <= 16 && H >= 256 && B <= 512;
}

__host__ __device__ inline int dc_slices(int B) { return B >= 1024 ? 4 : 1; }

struct TapeLayout {
    int n;
    mmg_tape_entry e[128];
    int64_t total;
};

inline TapeLayout tape_layout(const mmg_config& c) {
    TapeLayout L;
    L.n = 0;
    const int64_t B = c.batch, D = c.n_classes, F = c.feat_dim, H = c.h_dim, W = c.w_dim, R = c.rec_hidden,
                  V = c.wv_dim, K = c.bas_hidden, T = c.max_exchange, T1 = T + 1,
                  NSTAT = stat_count((int)T), NPART = MMG_GN_BLOCKS, NPB = (K + 63) / 64, NDCS = dc_slices((int)B), NS2P = (W + 15) / 16, NTILE = (B + 15) / 16, NZP = ((B + 15) / 16) * ((H + 63) / 64), NHLP = split_helpers((int)B) > 0 ? split_helpers((int)B) : 1,
                  NPLT = persist_ll_shape((int)B, (int)H, (int)W, (int)R, (int)V, (int)D, (int)T) ? 1 : 0,   /* the pair buffers of k_conv_persist<LL> exist (zero-sized otherwise) */
                  NPLS = W / 16,
                  NMC = mc_shape((int)H, (int)W, (int)R, (int)V, (int)D, (int)T) ? (B + 15) / 16 : 1,
                  NRCB = rc_shape((int)B, (int)H, (int)W, (int)R, (int)V, (int)D) ? B : 1,
                  NRCJ = rc_shape((int)B, (int)H, (int)W, (int)R, (int)V, (int)D) ? R / 16 : 1,
                  NRCX = rc_shape((int)B, (int)H, (int)W, (int)R, (int)V, (int)D) ? 2 * ((B + 15) / 16) : 1,
                  NRCP = rc_shape((int)B, (int)H, (int)W, (int)R, (int)V, (int)D) && R == 256 ? 4 : 1,   /* column bands of k_bwd_pre's partial dhin */
                  NRCA = rc_shape((int)B, (int)H, (int)W, (int)R, (int)V, (int)D) ? ((B + 15) / 16) * ((H + 15) / 16) : 1,
                  NRCZ = rc_shape((int)B, (int)H, (int)W, (int)R, (int)V, (int)D) ? ((B + 15) / 16) * (W / 16) : 1,
                  NRCH = rc_shape((int)B, (int)H, (int)W, (int)R, (int)V, (int)D) ? ((B + 15) / 16) * (R / 16) : 1,
                  NRCT16 = rc_shape((int)B, (int)H, (int)W, (int)R, (int)V, (int)D) ? 16 * ((B + 15) / 16) : 1,
                  NRCW = rc_shape((int)B, (int)H, (int)W, (int)R, (int)V, (int)D) ? W / 16 : 1,
                  NMCB = mc_shape((int)H, (int)W, (int)R, (int)V, (int)D, (int)T) && !c.use_binary ? 16 : 0,
                  NWP = wgrad_any_split((int)(T * B), param_layout(c).total) ? (int64_t)16 * (param_layout(c).total + 512 * 64) : 4;   /* (every job splits <= 16 ways) */
    (void)NPLT; (void)NPLS; (void)F; (void)V; (void)NPB; (void)NDCS; (void)NWP; (void)NS2P; (void)NTILE; (void)NHLP; (void)NZP; (void)NMC; (void)NMCB; (void)NRCB; (void)NRCJ; (void)NRCW; (void)NRCX; (void)NRCP; (void)NRCA; (void)NRCZ; (void)NRCH; (void)NRCT16;
    int64_t o = 0;
    const int64_t esz[4] = {4, 1, 4, 8};
#define X(name_, ctype, code, nd, d0, d1, d2)                                        \
    {                                                                               \
        mmg_tape_entry& e = L.e[L.n++];                                             \
        memset(&e, 0, sizeof(e));                                                   \
        strncpy(e.name, #name_, sizeof(e.name) - 1);                                \
        e.dtype = code; e.ndim = nd;                                                \
        e.dims[0] = (d0); e.dims[1] = (d1); e.dims[2] = (d2); e.dims[3] = 1;        \
        e.offset = o;                                                               \
        int64_t bytes = (int64_t)(d0) * (d1) * (d2) * esz[code];                    \
        o += (bytes + 255) & ~(int64_t)255;                                         \
    }
    MMG_TAPE_LIST(X)
#undef X
#define X(name_, ctype, code, nd, d0, d1, d2) +1
    static_assert(0 MMG_TAPE_LIST(X) <= (int)(sizeof(L.e) / sizeof(L.e[0])), "TapeLayout::e is too small for MMG_TAPE_LIST");
#undef X
    L.total = o;
    return L;
}

inline Tape resolve_tape(const TapeLayout& L, void* base) {
    Tape t;
    int i = 0;
#define X(name, ctype, code, nd, d0, d1, d2) t.name = (ctype*)((char*)base + L.e[i++].offset);
    MMG_TAPE_LIST(X)
#undef X
    return t;
}

// Dimensions handed to every kernel.
struct Dims {
    int B, Bg, boff, D, F, H, W, R, V, K, T;
    int use_binary, fixed, s_prob_prod, top_k;
    int has_es, has_esen, has_erec;
    float es, esen, erec, first_rec;
};

inline Dims make_dims(const mmg_config& c) {
    Dims d;
    d.B = c.batch; d.Bg = c.global_batch > 0 ? c.global_batch : c.batch; d.boff = c.batch_offset;
    d.D = c.n_classes; d.F = c.feat_dim; d.H = c.h_dim; d.W = c.w_dim; d.R = c.rec_hidden; d.V = c.wv_dim;
    d.K = c.bas_hidden; d.T = c.max_exchange;
    d.use_binary = c.use_binary; d.fixed = c.fixed_exchange; d.s_prob_prod = c.s_prob_prod; d.top_k = c.top_k;
    d.has_es = c.has_entropy_s; d.has_esen = c.has_entropy_sen; d.has_erec = c.has_entropy_rec;
    d.es = c.entropy_s; d.esen = c.entropy_sen; d.erec = c.entropy_rec; d.first_rec = c.first_rec;
    return d;
}

}  // namespace mmg
