// kernels_rc.h -- the receiver of a 16-sample tile SPLIT OVER WORKGROUPS by hidden-unit slice ("wide receiver": rec_hidden
// 129..256 beside the large sender of BASELINE config 4, SURVEY.md 8d "additionally report R = 256").  At R = W = 256 the
// receiver's per-step weights are 2.4 MB (W_ih and W_hh 768 KB each) and the one-workgroup-per-tile forward of kernels_tile.h
// needs 286 KB of LDS (three 16 x 772 gate tiles), so a step of the tile's receiver runs as three chip-wide launches whose
// workgroups own 16 hidden units (or 16 message bits) each and read their weight rows straight into MFMA B fragments:
//   k_rc_gru    (tiles x R/16)  gi = z_t W_ih^T, gh = h_t W_hh^T for the role's 3 x 16 gate columns (six 16x16x4 fp32 MFMA
//                               products, K split over the four waves, every load of the six in flight together), GRUCell
//                               (model.py:340) -> tape.gru, tape.h[t+1]; role 0: log-likelihood / neg-entropy of z_t
//   k_rc_heads  (tiles x R/16)  A = W_y1[:, :R] h, w_h h + b (16 columns each, App. A.2), the role's PARTIAL class logits
//                               sum_{r in slice} w_y2[r] relu(A[r] + Cd[d][r]) (model.py:432-433); role 0: stop bit
//                               (model.py:414-427) and the stop-mask bookkeeping (model.py:852)
//   k_rc_query  (tiles x W/16)  every role adds the R/16 partial logits, softmax (detached, model.py:441), description mixture
//                               (model.py:442-449), h_w = tanh(w_h h + w_d dbar) (model.py:452: 16 x V x R, recomputed by every
//                               role -- 13 MFLOP per tile-step, cheaper than one more launch), then ITS 16 message bits
//                               (model.py:454-475) and their partial log-likelihood sums
//   k_rc_tail   (tiles)         last step's message log-likelihood, output selection / log-softmax / reward / top-k
//                               (model.py:1264-1275, 1333-1339)
// between the per-step sender launches k_send_s1 / k_send_s2.  The launch boundary is the hand-off: no spin waits, no
// co-residency requirement.  Tape rows follow the live-row contract of k_conv_tile, so the tile backward (k_bwd_pre /
// k_bwd_tile / k_send_bwd / k_dC_tile) and k_wgrad run unchanged.  State between launches: rcst[0..1] = m_t double-buffered by
// step parity (a launch never writes the slot its own step reads), rcst[2] = take-output flag of the step, tape.tstar / sprod.
#pragma once

namespace mmg {

#define RC_MAXG 4                                   // k-groups of 16 per wave: K <= 256 over four waves
struct RcFrag { float4 v[RC_MAXG]; };

// this wave's k-groups [g0, g0 + RC_MAXG) of one operand row (clamped, branch-free: all loads of a phase go out together)
__device__ __forceinline__ void rc_load(RcFrag& f, const float* __restrict__ row, int K, int g0, int q) {
    const int kgroups = K >> 4;
#pragma unroll
    for (int u = 0; u < RC_MAXG; ++u) f.v[u] = *reinterpret_cast<const float4*>(row + min(g0 + u, kgroups - 1) * 16 + q * 4);
}
__device__ __forceinline__ f32x4 rc_mma(const RcFrag& a, const RcFrag& b, int n, f32x4 acc) {
#pragma unroll
    for (int u = 0; u < RC_MAXG; ++u) {
        if (u < n) {
            acc = mfma16(a.v[u].x, b.v[u].x, acc); acc = mfma16(a.v[u].y, b.v[u].y, acc);
            acc = mfma16(a.v[u].z, b.v[u].z, acc); acc = mfma16(a.v[u].w, b.v[u].w, acc);
        }
    }
    return acc;
}
// K share of a wave: k-groups [g0, g0 + n)
__device__ __forceinline__ void rc_share(int K, int wave, int& g0, int& n) {
    const int kg = K >> 4, per = (kg + 3) >> 2;
    g0 = wave * per;
    n = max(0, min(kg, g0 + per) - g0);
}

// rows of step t that are stored: valid sample, still in its conversation (k_conv_tile's TL_LIVE)
__device__ __forceinline__ bool rc_live(const Tape& tp, int B, int t, int b, bool valid, bool may_stop) {
    return valid && (!may_stop || t == 0 || tp.rcst[(size_t)(t & 1) * B + b] != 0.f);
}

// lp_w / ne_w of step tp_ from the per-role partials k_rc_query left (fixed summation tree: deterministic)
__device__ __forceinline__ void rc_sum_lw(const Dims& dm, const Tape& tp, int tp_, int b0, int nb, bool may_stop) {
    const int B = dm.B, NJW = dm.W >> 4, tid = threadIdx.x, m = tid >> 4, l16 = tid & 15, b = min(b0 + m, B - 1);
    float lpv = 0.f, nev = 0.f;
    for (int jw = l16; jw < NJW; jw += 16) {
        const float* p = tp.rclw + ((size_t)jw * B + b) * 2;
        lpv += p[0]; nev += p[1];
    }
    lpv = dpp_group_sum<16>(lpv); nev = dpp_group_sum<16>(nev);
    const bool live = rc_live(tp, B, tp_, b, m < nb, may_stop);
    const bool live2 = live && (!may_stop || tp.rcst[(size_t)((tp_ + 1) & 1) * B + b] != 0.f);
    if (l16 == 0 && live2) { tp.lp_w[(size_t)tp_ * B + b] = lpv; tp.ne_w[(size_t)tp_ * B + b] = nev; }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rc_gru(Dims dm, Params P, Tape tp, ConvArgs ar, int t, int skip) {
    __shared__ float s_acc[6][4][16][17];
    __shared__ float s_live[16];
    if (skip && tp.alive[t] == 0) return;
    const int B = dm.B, W = dm.W, R = dm.R, NJ = R >> 4;
    const int tile = blockIdx.x / NJ, j = blockIdx.x - tile * NJ;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    const bool may_stop = !ar.run_all && !dm.fixed && ar.train;
    if (tid < MMG_TM) s_live[tid] = rc_live(tp, B, t, min(b0 + tid, B - 1), tid < nb, may_stop) ? 1.f : 0.f;
    __syncthreads();
    if (may_stop) {
        bool any = false;
        for (int m = 0; m < MMG_TM; ++m) any = any || (s_live[m] != 0.f);
        if (!any) return;                                               // the tile's conversations are over
    }
    const size_t rowb = (size_t)t * B;
    {
        const int bx = min(b0 + i, B - 1), unit = 16 * j + i;
        int gw0, nw_, gr0, nr_;
        rc_share(W, wave, gw0, nw_); rc_share(R, wave, gr0, nr_);
        RcFrag az, ah, wi0, wi1, wi2, wh0, wh1, wh2;
        rc_load(az, tp.z + (rowb + bx) * W, W, gw0, q);
        rc_load(wi0, P.p[R_WIH] + (size_t)(unit) * W, W, gw0, q);
        rc_load(wi1, P.p[R_WIH] + (size_t)(R + unit) * W, W, gw0, q);
        rc_load(wi2, P.p[R_WIH] + (size_t)(2 * R + unit) * W, W, gw0, q);
        // (t == 0: h_0 = 0 is being written by this very launch -- the hidden-side product is b_hh alone)
        const int nh = (t > 0) ? nr_ : 0;
        rc_load(ah, tp.h + (rowb + bx) * R, R, gr0, q);
        rc_load(wh0, P.p[R_WHH] + (size_t)(unit) * R, R, gr0, q);
        rc_load(wh1, P.p[R_WHH] + (size_t)(R + unit) * R, R, gr0, q);
        rc_load(wh2, P.p[R_WHH] + (size_t)(2 * R + unit) * R, R, gr0, q);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a0 = rc_mma(az, wi0, nw_, z4), a1 = rc_mma(az, wi1, nw_, z4), a2 = rc_mma(az, wi2, nw_, z4);
        const f32x4 a3 = rc_mma(ah, wh0, nh, z4), a4 = rc_mma(ah, wh1, nh, z4), a5 = rc_mma(ah, wh2, nh, z4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s_acc[0][wave][q * 4 + r][i] = a0[r]; s_acc[1][wave][q * 4 + r][i] = a1[r]; s_acc[2][wave][q * 4 + r][i] = a2[r];
            s_acc[3][wave][q * 4 + r][i] = a3[r]; s_acc[4][wave][q * 4 + r][i] = a4[r]; s_acc[5][wave][q * 4 + r][i] = a5[r];
        }
    }
    __syncthreads();
    {   // GRUCell (model.py:340); gate order r, u, n
        const int m = tid >> 4, c = tid & 15, unit = 16 * j + c, b = min(b0 + m, B - 1);
        auto S = [&](int p) { return (s_acc[p][0][m][c] + s_acc[p][1][m][c]) + (s_acc[p][2][m][c] + s_acc[p][3][m][c]); };
        const float* bih = P.p[R_BIH]; const float* bhh = P.p[R_BHH];
        const float hprev = (t > 0) ? tp.h[(rowb + b) * R + unit] : 0.f;
        const float gir = S(0) + bih[unit], giu = S(1) + bih[R + unit], gin = S(2) + bih[2 * R + unit];
        const float ghr = S(3) + bhh[unit], ghu = S(4) + bhh[R + unit], ghn = S(5) + bhh[2 * R + unit];
        const float rr = fsigmoid(gir + ghr), uu = fsigmoid(giu + ghu);
        const float nn = ftanh(gin + rr * ghn);
        const float hv = nn + uu * (hprev - nn);
        if (t == 0 && m < nb) tp.h[(size_t)b * R + unit] = 0.f;         // h_{-1} = 0
        if (s_live[m] != 0.f) {
            float* gr = tp.gru + (rowb + b) * 4 * R;
            gr[unit] = rr; gr[R + unit] = uu; gr[2 * R + unit] = nn; gr[3 * R + unit] = ghn;
            tp.h[((size_t)(t + 1) * B + b) * R + unit] = hv;
        }
    }
    if (j == 0) {
        if (t == 0 && tid < nb) tp.mask[b0 + tid] = 1;                  // stop_mask[0] = ones   model.py:775
        if (dm.use_binary) {                                            // log-likelihood / neg-entropy of the sender's bits, model.py:908-922
            const int m = tid >> 4, l16 = tid & 15, b = min(b0 + m, B - 1);
            float lpv = 0.f, nev = 0.f;
            for (int k = l16; k < W; k += 16) {
                const float p = tp.pz[(rowb + b) * W + k], zz = tp.z[(rowb + b) * W + k];
                const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
                lpv += zz * l1 + (1.f - zz) * l0; nev += p * l1 + (1.f - p) * l0;
            }
            lpv = dpp_group_sum<16>(lpv); nev = dpp_group_sum<16>(nev);
            if (l16 == 0 && s_live[m] != 0.f) { tp.lp_z[rowb + b] = lpv; tp.ne_z[rowb + b] = nev; }
            if (t > 0) rc_sum_lw(dm, tp, t - 1, b0, nb, may_stop);       // the receiver's message of the step before
        }
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rc_heads(Dims dm, Params P, Tape tp, ConvArgs ar, int t, int skip) {
    __shared__ float s_acc[2][4][16][17];
    __shared__ __attribute__((aligned(16))) float s_A[16][20];
    __shared__ float s_live[16], s_mn[16];
    if (skip && tp.alive[t] == 0) return;
    const int B = dm.B, R = dm.R, V = dm.V, D = dm.D, T = dm.T, NJ = R >> 4;
    const int tile = blockIdx.x / NJ, j = blockIdx.x - tile * NJ;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    const bool may_stop = !ar.run_all && !dm.fixed && ar.train, train = ar.train != 0;
    if (tid < MMG_TM) s_live[tid] = rc_live(tp, B, t, min(b0 + tid, B - 1), tid < nb, may_stop) ? 1.f : 0.f;
    __syncthreads();
    if (may_stop) {
        bool any = false;
        for (int m = 0; m < MMG_TM; ++m) any = any || (s_live[m] != 0.f);
        if (!any) return;
    }
    const size_t rowb = (size_t)t * B, rowh = (size_t)(t + 1) * B;
    {
        const int bx = min(b0 + i, B - 1), unit = 16 * j + i;
        int g0, n;
        rc_share(R, wave, g0, n);
        RcFrag ah, wa, wg;
        rc_load(ah, tp.h + (rowh + bx) * R, R, g0, q);
        rc_load(wa, P.p[R_Y1_W] + (size_t)unit * (R + V), R, g0, q);
        rc_load(wg, P.p[R_WH_W] + (size_t)unit * R, R, g0, q);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a0 = rc_mma(ah, wa, n, z4), a1 = rc_mma(ah, wg, n, z4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { s_acc[0][wave][q * 4 + r][i] = a0[r]; s_acc[1][wave][q * 4 + r][i] = a1[r]; }
    }
    __syncthreads();
    {
        const int m = tid >> 4, c = tid & 15, unit = 16 * j + c;
        auto S = [&](int p) { return (s_acc[p][0][m][c] + s_acc[p][1][m][c]) + (s_acc[p][2][m][c] + s_acc[p][3][m][c]); };
        s_A[m][c] = S(0);                                               // A (App. A.2)
        if (m < nb) tp.rcgw[(size_t)(b0 + m) * R + unit] = S(1) + P.p[R_WH_B][unit];     // w_h h + b_h
    }
    __syncthreads();
    {   // this slice's share of y[m][d] = b_y2 + sum_r w_y2[r] relu(A[m][r] + Cd[d][r])     (model.py:432-433)
        const int m = tid >> 4, c = tid & 15;
        const float4* w4 = reinterpret_cast<const float4*>(P.p[R_Y2_W] + 16 * j);
        const float4 wq0 = w4[0], wq1 = w4[1], wq2 = w4[2], wq3 = w4[3];
        const float4* a4 = reinterpret_cast<const float4*>(&s_A[m][0]);
        const float4 aq0 = a4[0], aq1 = a4[1], aq2 = a4[2], aq3 = a4[3];
        for (int d = c; d < D; d += 16) {
            const float4* c4 = reinterpret_cast<const float4*>(tp.Cd + (size_t)d * R + 16 * j);
            const float4 cq0 = c4[0], cq1 = c4[1], cq2 = c4[2], cq3 = c4[3];
            float s0 = 0.f, s1 = 0.f;
            s0 = fmaf(wq0.x, fmax_nn(aq0.x + cq0.x, 0.f), s0); s1 = fmaf(wq0.y, fmax_nn(aq0.y + cq0.y, 0.f), s1);
            s0 = fmaf(wq0.z, fmax_nn(aq0.z + cq0.z, 0.f), s0); s1 = fmaf(wq0.w, fmax_nn(aq0.w + cq0.w, 0.f), s1);
            s0 = fmaf(wq1.x, fmax_nn(aq1.x + cq1.x, 0.f), s0); s1 = fmaf(wq1.y, fmax_nn(aq1.y + cq1.y, 0.f), s1);
            s0 = fmaf(wq1.z, fmax_nn(aq1.z + cq1.z, 0.f), s0); s1 = fmaf(wq1.w, fmax_nn(aq1.w + cq1.w, 0.f), s1);
            s0 = fmaf(wq2.x, fmax_nn(aq2.x + cq2.x, 0.f), s0); s1 = fmaf(wq2.y, fmax_nn(aq2.y + cq2.y, 0.f), s1);
            s0 = fmaf(wq2.z, fmax_nn(aq2.z + cq2.z, 0.f), s0); s1 = fmaf(wq2.w, fmax_nn(aq2.w + cq2.w, 0.f), s1);
            s0 = fmaf(wq3.x, fmax_nn(aq3.x + cq3.x, 0.f), s0); s1 = fmaf(wq3.y, fmax_nn(aq3.y + cq3.y, 0.f), s1);
            s0 = fmaf(wq3.z, fmax_nn(aq3.z + cq3.z, 0.f), s0); s1 = fmaf(wq3.w, fmax_nn(aq3.w + cq3.w, 0.f), s1);
            if (m < nb) tp.rcyp[((size_t)j * B + b0 + m) * D + d] = s0 + s1;
        }
    }
    if (j != 0) return;
    {   // stop bit (model.py:414-427) and the stop-mask bookkeeping (model.py:852) -- per sample
        const int m = tid >> 4, l16 = tid & 15, b = min(b0 + m, B - 1);
        const float* hr = tp.h + (rowh + b) * R;
        const float* ws = P.p[R_S_W];
        float acc = 0.f;
        for (int r = l16 * 4; r < R; r += 64) {
            const float4 hq = *reinterpret_cast<const float4*>(hr + r), wq = *reinterpret_cast<const float4*>(ws + r);
            acc = fmaf(wq.x, hq.x, acc); acc = fmaf(wq.y, hq.y, acc); acc = fmaf(wq.z, hq.z, acc); acc = fmaf(wq.w, hq.w, acc);
        }
        acc = dpp_group_sum<16>(acc);
        if (l16 == 0) {
            const bool valid = m < nb, live = s_live[m] != 0.f;
            const float p = fsigmoid(acc + P.p[R_S_B][0]);
            float sv, prod = 1.f;
            if (train) {
                const float u = ar.u_s ? ar.u_s[rowb + b] : philox_uniform(ar.seed, (uint32_t)(t * dm.Bg + dm.boff + b), tp.counter[0], 1u);
                sv = (u < p) ? 1.f : 0.f;                                                   // model.py:420
            } else {
                const float before = (t == 0) ? 1.f : tp.sprod[b];
                prod = dm.s_prob_prod ? before * p : p;                                     // model.py:423-426
                sv = rintf(prod);                                                           // model.py:427
            }
            const float m_t = (t == 0) ? 1.f : tp.rcst[(size_t)(t & 1) * B + b];
            const float m_next = fminf(m_t, sv);
            const int ts_before = (t == 0) ? -1 : tp.tstar[b];
            const bool take = dm.fixed ? (t == T - 1) : (ts_before < 0 && (m_next == 0.f || t == T - 1));
            s_mn[m] = valid ? m_next : 0.f;
            if (valid) {
                tp.rcst[(size_t)((t + 1) & 1) * B + b] = m_next;
                tp.rcst[(size_t)2 * B + b] = take ? 1.f : 0.f;
                tp.tstar[b] = take ? t : ts_before;
                tp.mstate[b] = m_next;
                if (!train) tp.sprod[b] = prod;
            }
            if (live) {
                tp.s[rowb + b] = sv; tp.ps[rowb + b] = p;
                const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
                tp.lp_s[rowb + b] = sv * l1 + (1.f - sv) * l0;
                tp.ne_s[rowb + b] = p * l1 + (1.f - p) * l0;
                tp.mask[rowh + b] = (uint8_t)(m_next != 0.f);
            }
        }
        __syncthreads();
        if (tid == 0 && t + 1 < T) {
            bool alive = false;
            for (int mm = 0; mm < nb; ++mm) alive = alive || (s_mn[mm] != 0.f);
            if (alive) atomicAdd(&tp.alive[t + 1], 1);
        }
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rc_query(Dims dm, Params P, Tape tp, ConvArgs ar, int t, int skip) {
    __shared__ __attribute__((aligned(16))) float s_y[16][68];
    __shared__ __attribute__((aligned(16))) float s_dbar[16][132];
    __shared__ __attribute__((aligned(16))) float s_g[16][260];
    __shared__ float s_acc[4][16][17];
    __shared__ float s_live[16], s_live2[16], s_take[16];
    if (skip && tp.alive[t] == 0) return;
    const int B = dm.B, W = dm.W, R = dm.R, V = dm.V, D = dm.D, NJ = R >> 4, NJW = W >> 4;
    const int tile = blockIdx.x / NJW, jw = blockIdx.x - tile * NJW;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    const bool may_stop = !ar.run_all && !dm.fixed && ar.train, train = ar.train != 0, binary = dm.use_binary != 0;
    if (tid < MMG_TM) {
        const int b = min(b0 + tid, B - 1);
        const bool live = rc_live(tp, B, t, b, tid < nb, may_stop);
        s_live[tid] = live ? 1.f : 0.f;
        s_live2[tid] = (live && (!may_stop || tp.rcst[(size_t)((t + 1) & 1) * B + b] != 0.f)) ? 1.f : 0.f;
        s_take[tid] = (tid < nb && tp.rcst[(size_t)2 * B + b] != 0.f) ? 1.f : 0.f;
    }
    for (int idx = tid; idx < 16 * 132; idx += 256) (&s_dbar[0][0])[idx] = 0.f;        // K padding of the w_d product
    __syncthreads();
    if (may_stop) {
        bool any = false;
        for (int m = 0; m < MMG_TM; ++m) any = any || (s_live[m] != 0.f);
        if (!any) return;
    }
    const size_t rowb = (size_t)t * B;
    const float b2 = P.p[R_Y2_B][0];
    // ---- class logits: the R/16 partials in role order
    for (int idx = tid; idx < MMG_TM * D; idx += 256) {
        const int m = idx / D, d = idx - m * D, b = min(b0 + m, B - 1);
        float acc = 0.f;
        for (int jj = 0; jj < NJ; jj += 8) {
            float pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pv[u] = tp.rcyp[((size_t)min(jj + u, NJ - 1) * B + b) * D + d];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (jj + u < NJ) ? pv[u] : 0.f;
        }
        const float yv = acc + b2;
        s_y[m][d] = yv;
        if (jw == 0 && m < nb) {
            if ((ar.y_last_only ? s_take[m] : s_live[m]) != 0.f) tp.y[(rowb + b) * D + d] = yv;
            if (s_take[m] != 0.f) tp.outp[(size_t)b * D + d] = yv;       // the output step, model.py:1261-1264
        }
    }
    __syncthreads();
    // ---- softmax(y) (detached, model.py:441): wave per sample row, in place (D <= 64: one class per lane)
    for (int m = wave; m < MMG_TM; m += 4) {
        const float v = (lane < D) ? s_y[m][lane] : -3.0e38f;
        const float mx = dpp_wave_max(v);
        const float e = (lane < D) ? __expf(v - mx) : 0.f;
        const float se = dpp_wave_sum(e);
        if (lane < D) s_y[m][lane] = e * __builtin_amdgcn_rcpf(se);
    }
    __syncthreads();
    // ---- description mixture (model.py:442-449)
    for (int idx = tid; idx < MMG_TM * V; idx += 256) {
        const int m = idx / V, v = idx - m * V;
        float acc = 0.f;
        for (int d0 = 0; d0 < D; d0 += 8) {
            float dv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) dv[u] = ar.desc[(size_t)min(d0 + u, D - 1) * V + v];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf((d0 + u < D) ? s_y[m][min(d0 + u, D - 1)] : 0.f, dv[u], acc);
        }
        s_dbar[m][v] = acc;
        if (jw == 0 && s_live2[m] != 0.f) tp.dbar[(rowb + b0 + m) * V + v] = acc;
    }
    __syncthreads();
    // ---- h_w = tanh(w_h h + b_h + w_d dbar)   (model.py:452): all R columns, four 16-column tiles per wave
    {
        const int kg = (V + 15) >> 4;                                   // <= 8
        for (int tn = wave; tn < NJ; tn += 4) {
            const float* wrow = P.p[R_WD_W] + (size_t)(tn * 16 + i) * V;
            float4 bq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) bq[u] = ldrow4c<true>(wrow, min(u, kg - 1) * 16 + q * 4, V);
            float gwv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) gwv[r] = tp.rcgw[(size_t)min(b0 + q * 4 + r, B - 1) * R + tn * 16 + i];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u < kg) {
                    const float4 a = *reinterpret_cast<const float4*>(&s_dbar[i][u * 16 + q * 4]);
                    acc = mfma16(a.x, bq[u].x, acc); acc = mfma16(a.y, bq[u].y, acc);
                    acc = mfma16(a.z, bq[u].z, acc); acc = mfma16(a.w, bq[u].w, acc);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = q * 4 + r, n = tn * 16 + i;
                const float gv = ftanh(gwv[r] + acc[r]);
                s_g[m][n] = gv;
                if (jw == 0 && s_live2[m] != 0.f) tp.g[(rowb + b0 + m) * R + n] = gv;
            }
        }
    }
    __syncthreads();
    // ---- this role's 16 bits of the receiver's message (model.py:454-475)
    {
        int g0, n;
        rc_share(R, wave, g0, n);
        RcFrag ag, ww;
        rc_load(ww, P.p[R_W_W] + (size_t)(16 * jw + i) * R, R, g0, q);
        const int kgr = R >> 4;
#pragma unroll
        for (int u = 0; u < RC_MAXG; ++u) ag.v[u] = *reinterpret_cast<const float4*>(&s_g[i][min(g0 + u, kgr - 1) * 16 + q * 4]);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a0 = rc_mma(ag, ww, n, z4);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_acc[wave][q * 4 + r][i] = a0[r];
    }
    __syncthreads();
    {
        const int m = tid >> 4, c = tid & 15, n = 16 * jw + c, b = min(b0 + m, B - 1);
        const float lw = (s_acc[0][m][c] + s_acc[1][m][c]) + (s_acc[2][m][c] + s_acc[3][m][c]) + P.p[R_W_B][n];
        float wv = lw, lpv = 0.f, nev = 0.f;
        const bool st = s_live2[m] != 0.f;
        if (binary) {
            const float pp = fsigmoid(lw);
            if (train) {
                const float u = ar.u_w ? ar.u_w[(rowb + b) * W + n]
                                       : philox_uniform(ar.seed, (uint32_t)((t * dm.Bg + dm.boff + b) * W + n), tp.counter[0], 2u);
                wv = (u < pp) ? 1.f : 0.f;                                                  // model.py:460
            } else wv = rintf(pp);                                                          // model.py:462
            if (st) tp.pw[(rowb + b) * W + n] = pp;
            const float l1 = flog(pp + MMG_EPS), l0 = flog(1.f - pp + MMG_EPS);
            lpv = wv * l1 + (1.f - wv) * l0; nev = pp * l1 + (1.f - pp) * l0;
        }
        if (st) tp.w[(rowb + b) * W + n] = wv;
        if (binary) {
            lpv = dpp_group_sum<16>(lpv); nev = dpp_group_sum<16>(nev);
            if (c == 0 && m < nb) { float* pl = tp.rclw + ((size_t)jw * B + b) * 2; pl[0] = lpv; pl[1] = nev; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rc_tail(Dims dm, Params P, Tape tp, ConvArgs ar) {
    const int B = dm.B, D = dm.D, T = dm.T;
    const int b0 = blockIdx.x * MMG_TM, nb = min(MMG_TM, B - b0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool may_stop = !ar.run_all && !dm.fixed && ar.train;
    if (dm.use_binary) rc_sum_lw(dm, tp, T - 1, b0, nb, may_stop);
    // output selection, log-softmax, reward, top-k (model.py:1264-1275, 1333-1339): wave per sample, D <= 64
    for (int m = wave; m < nb; m += 4) {
        const int b = b0 + m;
        const float v = (lane < D) ? tp.outp[(size_t)b * D + lane] : -3.0e38f;
        const float mx = dpp_wave_max(v);
        const float se = dpp_wave_sum((lane < D) ? __expf(v - mx) : 0.f);
        const float lse = mx + flog(se);
        const int tgt = ar.target ? (int)ar.target[b] : -1;
        const float dt = (tgt >= 0) ? (tp.outp[(size_t)b * D + max(tgt, 0)] - lse) : 0.f;
        const float ld = v - lse;
        float above = 0.f;
        if (lane < D) {
            tp.dist[(size_t)b * D + lane] = ld;
            tp.sm[(size_t)b * D + lane] = __expf(ld);
            if (tgt >= 0 && ld > dt) above = 1.f;
        }
        above = dpp_wave_sum(above);
        if (lane == 0) {
            tp.logs[b] = dt;
            tp.hit[b] = (tgt >= 0 && above < (float)dm.top_k) ? 1 : 0;
        }
    }
}

}  // namespace mmg
