// kernels_rc.h -- the receiver of a 16-sample tile SPLIT OVER WORKGROUPS by hidden-unit slice ("wide receiver": rec_hidden
// 129..256 beside the large sender of BASELINE config 4, SURVEY.md 8d "additionally report R = 256").  At R = W = 256 the
// receiver's per-step weights are 2.4 MB (W_ih and W_hh 768 KB each) and the one-workgroup-per-tile forward of kernels_tile.h
// needs 286 KB of LDS (three 16 x 772 gate tiles), so a tile's receiver step is cut into three phases whose workgroups own 16
// hidden units (or 16 message bits) each and read their weight rows straight into MFMA B fragments (v_mfma_f32_16x16x4_f32, K
// split over the four waves):
//   GRU     (R/16 slices)  gi = z_t W_ih^T, gh = h_t W_hh^T for the slice's 3 x 16 gate columns, GRUCell (model.py:340)
//   heads   (R/16 slices)  A = W_y1[:, :R] h, w_h h + b (App. A.2), the slice's PARTIAL class logits (model.py:432-433)
//   query   (W/16 slices)  adds the partial logits, softmax (model.py:441), h_w = tanh(w_h h + b + softmax(y) . Dd) with Dd = desc . W_d^T
//                          (model.py:442-452; recomputed by every slice: cheaper than one more hand-off), ITS 16 message bits (model.py:454-475)
// Default: k_rc_persist -- ONE launch per conversation of co-resident roles (S1 / S2 sender slices, receiver slices running the three
// phases, one stop role per tile) that hand their results on through memory + counters, every payload in the consumer's fragment
// order; the backward's output-step prelude and reverse-time loop likewise as k_rc_bwd's roles.  Fallback (roles do not fit the
// device, MMG_NO_RC_PERSIST=1): the same phase bodies as the launches k_rc_gru / k_rc_heads / k_rc_query / k_rc_tail between the
// per-step sender launches k_send_s1 / k_send_s2 -- the launch boundary is the hand-off, no spin waits, no co-residency requirement.
// Tape rows follow the live-row contract of k_conv_tile, so k_bwd_pre / k_send_bwd / k_dC_tile / k_wgrad run unchanged.  State between
// phases: rcst[0..1] = m_t double-buffered by step parity (a phase never writes the slot its own step reads), rcst[2] = take-output
// flag of the step, rcst[3] = "a conversation of the tile goes on", tape.tstar / sprod.  DESIGN.md 3e.
#pragma once

namespace mmg {

#define RC_MAXG 4                                   // k-groups of 16 per wave: K <= 256 over four waves
struct RcFrag { float4 v[RC_MAXG]; };

// PS (persistent launch, k_rc_persist): what another workgroup of the SAME launch wrote is read with agent-scope loads and
// written with write-through stores (device_utils.h: ld_cc / st_wt); the per-step kernels use plain accesses
template <bool PS> __device__ __forceinline__ float rc_ld(const float* p) { return PS ? ld_cc(p) : *p; }
template <bool PS> __device__ __forceinline__ float4 rc_ld4(const float* p) { return PS ? ld_cc4(p) : *reinterpret_cast<const float4*>(p); }
template <bool PS> __device__ __forceinline__ void rc_st(float* p, float v) { if (PS) st_wt(p, v); else *p = v; }

// an opaque zero added to the addresses of step-invariant operands that are NOT meant to stay in registers across the persistent
// launch's step loop: hipcc otherwise hoists every such load out of the loop and spills (k_rc_persist: 256 + 256 registers, 1.6 KB scratch)
__device__ __forceinline__ int rc_opaque0() { int z = 0; asm volatile("" : "+v"(z)); return z; }

// this wave's k-groups [g0, g0 + RC_MAXG) of one operand row (clamped, branch-free: all loads of a phase go out together)
__device__ __forceinline__ void rc_load(RcFrag& f, const float* __restrict__ row, int K, int g0, int q) {
    const int kgroups = K >> 4;
#pragma unroll
    for (int u = 0; u < RC_MAXG; ++u) f.v[u] = *reinterpret_cast<const float4*>(row + min(g0 + u, kgroups - 1) * 16 + q * 4);
}
template <bool PS>
__device__ __forceinline__ void rc_load_act(RcFrag& f, const float* row, int K, int g0, int q) {      // activation rows (hand-off payload when PS)
    const int kgroups = K >> 4;
#pragma unroll
    for (int u = 0; u < RC_MAXG; ++u) f.v[u] = rc_ld4<PS>(row + min(g0 + u, kgroups - 1) * 16 + q * 4);
}
// the same k-groups from a FRAGMENT-ORDER copy [k-group][16 samples][16] of the tile (hand-off payload of the persistent launch): lane
// (i, q) reads the 16 bytes at [g][i][4 q], a wave 1 KB contiguous per k-group
__device__ __forceinline__ void rc_load_frag(RcFrag& f, const float* tilebase, int K, int g0, int i, int q) {
    const int kgroups = K >> 4;
#pragma unroll
    for (int u = 0; u < RC_MAXG; ++u) f.v[u] = ld_cc4(tilebase + (size_t)min(g0 + u, kgroups - 1) * 256 + i * 16 + q * 4);
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
template <bool PS>
__device__ __forceinline__ bool rc_live(const Tape& tp, int B, int t, int b, bool valid, bool may_stop) {
    return valid && (!may_stop || t == 0 || rc_ld<PS>(&tp.rcst[(size_t)(t & 1) * B + b]) != 0.f);
}

// lp_w / ne_w of step tp_ from the per-role partials k_rc_query left (fixed summation tree: deterministic)
template <bool PS>
__device__ __forceinline__ void rc_sum_lw(const Dims& dm, const Tape& tp, int tp_, int b0, int nb, bool may_stop) {
    const int B = dm.B, NJW = dm.W >> 4, tid = threadIdx.x, m = tid >> 4, l16 = tid & 15, b = min(b0 + m, B - 1);
    float lpv = 0.f, nev = 0.f;
    for (int jw = l16; jw < NJW; jw += 16) {
        const float* p = tp.rclw + (((size_t)(tp_ & 1) * NJW + jw) * B + b) * 2;
        lpv += rc_ld<PS>(p); nev += rc_ld<PS>(p + 1);
    }
    lpv = dpp_group_sum<16>(lpv); nev = dpp_group_sum<16>(nev);
    // stored rows of the message of step tp_: the conversation goes on after it (m_{tp_+1} != 0 implies m_{tp_} != 0; only
    // the slot of m_{tp_+1} is read -- the other one may already hold m_{tp_+2})
    const bool live2 = (m < nb) && (!may_stop || rc_ld<PS>(&tp.rcst[(size_t)((tp_ + 1) & 1) * B + b]) != 0.f);
    if (l16 == 0 && live2) { tp.lp_w[(size_t)tp_ * B + b] = lpv; tp.ne_w[(size_t)tp_ * B + b] = nev; }
}

// ---------------------------------------------------------------------------------------------
// step-invariant operands of a role's phases: the persistent launch loads them ONCE, ahead of its step loop (register-resident
// weight fragments: 24 + 8 + 4 float4 per lane), the per-step kernels at every launch
struct RcGruW { RcFrag wi0, wi1, wi2, wh0, wh1, wh2; float bir, biu, bin_, bhr, bhu, bhn; };
__device__ __forceinline__ void rc_gru_w(RcGruW& w, const Dims& dm, const Params& P, const int j) {
    const int W = dm.W, R = dm.R, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    const int unit = 16 * j + i, unit_e = 16 * j + (tid & 15);
    int gw0, nw_, gr0, nr_;
    rc_share(W, wave, gw0, nw_); rc_share(R, wave, gr0, nr_);
    gw0 += rc_opaque0(); gr0 += rc_opaque0();
    rc_load(w.wi0, P.p[R_WIH] + (size_t)(unit) * W, W, gw0, q);
    rc_load(w.wi1, P.p[R_WIH] + (size_t)(R + unit) * W, W, gw0, q);
    rc_load(w.wi2, P.p[R_WIH] + (size_t)(2 * R + unit) * W, W, gw0, q);
    rc_load(w.wh0, P.p[R_WHH] + (size_t)(unit) * R, R, gr0, q);
    rc_load(w.wh1, P.p[R_WHH] + (size_t)(R + unit) * R, R, gr0, q);
    rc_load(w.wh2, P.p[R_WHH] + (size_t)(2 * R + unit) * R, R, gr0, q);
    const float* bih = P.p[R_BIH]; const float* bhh = P.p[R_BHH];
    w.bir = bih[unit_e]; w.biu = bih[R + unit_e]; w.bin_ = bih[2 * R + unit_e];
    w.bhr = bhh[unit_e]; w.bhu = bhh[R + unit_e]; w.bhn = bhh[2 * R + unit_e];
}
// PS = false: returns early (false) when the tile's conversations are over; PS = true: always runs through (its signals must go out)
template <bool PS>
__device__ __forceinline__ bool rc_gru_body(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int t, const int tile, const int j,
                                            const RcGruW& w) {
    __shared__ float s_acc[6][4][16][17];
    __shared__ float s_live[16];
    const int B = dm.B, W = dm.W, R = dm.R;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0);
    // (per-thread indices re-derived from an OPAQUE copy of the thread id in every call: inside the persistent launch's step loop hipcc
    //  otherwise hoists each phase's row / column / tape-address arithmetic out of the loop and keeps hundreds of registers live across it)
    const int tid = rc_opaque0() + (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, q = lane >> 4;
    const bool may_stop = !ar.run_all && !dm.fixed && ar.train;
    const float live_in = rc_live<PS>(tp, B, t, min(b0 + (tid & 15), B - 1), (tid & 15) < nb, may_stop) ? 1.f : 0.f;
    const size_t rowb = (size_t)t * B;
    // operands of the epilogue (biases, h_{t-1} of this thread's unit): in flight with the product's operands
    const int m = tid >> 4, c = tid & 15, unit_e = 16 * j + c, b = min(b0 + m, B - 1);
    const float hprev = (t > 0) ? rc_ld<PS>(&tp.h[(rowb + b) * R + unit_e]) : 0.f;
    const float bir = w.bir, biu = w.biu, bin_ = w.bin_, bhr = w.bhr, bhu = w.bhu, bhn = w.bhn;
    {
        const int bx = min(b0 + i, B - 1);
        int gw0, nw_, gr0, nr_;
        rc_share(W, wave, gw0, nw_); rc_share(R, wave, gr0, nr_);
        RcFrag az, ah;
        // (t == 0: h_0 = 0 is being written by this very launch -- the hidden-side product is b_hh alone)
        const int nh = (t > 0) ? nr_ : 0;
        if (PS) {
            rc_load_frag(az, tp.rcxz + (size_t)tile * (W >> 4) * 256, W, gw0, i, q);
            rc_load_frag(ah, tp.rcxh + ((size_t)(t & 1) * ((B + 15) >> 4) * (R >> 4) + (size_t)tile * (R >> 4)) * 256, R, gr0, i, q);
        } else {
            rc_load_act<PS>(az, tp.z + (rowb + bx) * W, W, gw0, q);
            rc_load_act<PS>(ah, tp.h + (rowb + bx) * R, R, gr0, q);
        }
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a0 = rc_mma(az, w.wi0, nw_, z4), a1 = rc_mma(az, w.wi1, nw_, z4), a2 = rc_mma(az, w.wi2, nw_, z4);
        const f32x4 a3 = rc_mma(ah, w.wh0, nh, z4), a4 = rc_mma(ah, w.wh1, nh, z4), a5 = rc_mma(ah, w.wh2, nh, z4);
        MMG_RSTAMP(PS && tile == 0 && j == 0 && t == 3, 160);
        __syncthreads();                                                // (PS: the LDS of the phase before is free)
        if (tid < MMG_TM) s_live[tid] = live_in;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s_acc[0][wave][q * 4 + r][i] = a0[r]; s_acc[1][wave][q * 4 + r][i] = a1[r]; s_acc[2][wave][q * 4 + r][i] = a2[r];
            s_acc[3][wave][q * 4 + r][i] = a3[r]; s_acc[4][wave][q * 4 + r][i] = a4[r]; s_acc[5][wave][q * 4 + r][i] = a5[r];
        }
    }
    __syncthreads();
    if (!PS && may_stop) {
        bool any = false;
        for (int mm = 0; mm < MMG_TM; ++mm) any = any || (s_live[mm] != 0.f);
        if (!any) return false;                                         // the tile's conversations are over
    }
    {   // GRUCell (model.py:340); gate order r, u, n
        MMG_RSTAMP(PS && tile == 0 && j == 0 && t == 3, 161);
        auto S = [&](int p) { return (s_acc[p][0][m][c] + s_acc[p][1][m][c]) + (s_acc[p][2][m][c] + s_acc[p][3][m][c]); };
        const float gir = S(0) + bir, giu = S(1) + biu, gin = S(2) + bin_;
        const float ghr = S(3) + bhr, ghu = S(4) + bhu, ghn = S(5) + bhn;
        const float rr = fsigmoid(gir + ghr), uu = fsigmoid(giu + ghu);
        const float nn = ftanh(gin + rr * ghn);
        const float hv = nn + uu * (hprev - nn);
        if (t == 0 && m < nb) tp.h[(size_t)b * R + unit_e] = 0.f;       // h_{-1} = 0
        if (s_live[m] != 0.f) {
            float* gr = tp.gru + (rowb + b) * 4 * R;
            gr[unit_e] = rr; gr[R + unit_e] = uu; gr[2 * R + unit_e] = nn; gr[3 * R + unit_e] = ghn;
            rc_st<PS>(&tp.h[((size_t)(t + 1) * B + b) * R + unit_e], hv);
        }
        // (PS) h_{t+1} in fragment order for the heads phase and the next GRU step, by step parity (a role writes h_{t+2} while others still read h_{t+1})
        if (PS) st_wt(&tp.rcxh[((((size_t)((t + 1) & 1) * ((B + 15) >> 4) + tile) * (R >> 4) + j) * 16 + m) * 16 + c], (m < nb) ? hv : 0.f);
    }
    return true;
}
// role 0 of a tile, off the hand-off's critical path (PS: after h_{t+1} has been signalled): stop_mask[0], the log-likelihood /
// neg-entropy of the sender's bits (model.py:908-922) and the sums of the receiver's message of the step before
template <bool PS>
__device__ __forceinline__ void rc_gru_extras(const Dims& dm, const Tape& tp, const ConvArgs& ar, const int t, const int tile) {
    const int B = dm.B, W = dm.W;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0), tid = threadIdx.x;
    const bool may_stop = !ar.run_all && !dm.fixed && ar.train;
    const size_t rowb = (size_t)t * B;
    if (t == 0 && tid < nb) tp.mask[b0 + tid] = 1;                      // stop_mask[0] = ones   model.py:775
    if (dm.use_binary) {
        const int m = tid >> 4, l16 = tid & 15, b = min(b0 + m, B - 1);
        const bool live = rc_live<PS>(tp, B, t, b, m < nb, may_stop);
        float lpv = 0.f, nev = 0.f;
        for (int k0 = l16 * 4; k0 < W; k0 += 64) {
            const float4 pq = rc_ld4<PS>(&tp.pz[(rowb + b) * W + k0]), zq = rc_ld4<PS>(&tp.z[(rowb + b) * W + k0]);
            const float pv[4] = {pq.x, pq.y, pq.z, pq.w}, zv[4] = {zq.x, zq.y, zq.z, zq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float l1 = flog(pv[e] + MMG_EPS), l0 = flog(1.f - pv[e] + MMG_EPS);
                lpv += zv[e] * l1 + (1.f - zv[e]) * l0; nev += pv[e] * l1 + (1.f - pv[e]) * l0;
            }
        }
        lpv = dpp_group_sum<16>(lpv); nev = dpp_group_sum<16>(nev);
        if (l16 == 0 && live) { tp.lp_z[rowb + b] = lpv; tp.ne_z[rowb + b] = nev; }
        if (t > 0) rc_sum_lw<PS>(dm, tp, t - 1, b0, nb, may_stop);       // the receiver's message of the step before
    }
}
__global__ __launch_bounds__(256) void k_rc_gru(Dims dm, Params P, Tape tp, ConvArgs ar, int t, int skip) {
    if (skip && tp.alive[t] == 0) return;
    const int NJ = dm.R >> 4;
    RcGruW w;
    rc_gru_w(w, dm, P, blockIdx.x % NJ);
    if (rc_gru_body<false>(dm, P, tp, ar, t, blockIdx.x / NJ, blockIdx.x % NJ, w) && blockIdx.x % NJ == 0)
        rc_gru_extras<false>(dm, tp, ar, t, blockIdx.x / NJ);
}

// ---------------------------------------------------------------------------------------------
struct RcHeadsW { RcFrag wa, wg; float bh, bs; float4 wq[4], cq[2][4], sq[4]; };
// frags: the two weight fragments and the scalars (what the persistent launch keeps across steps); the class rows, w_y2 and s.weight
// slices are loaded in any case (keeping all of it resident next to the GRU's 24 fragments spills)
__device__ __forceinline__ void rc_heads_w(RcHeadsW& w, const Dims& dm, const Params& P, const Tape& tp, const int j, const bool frags) {
    const int R = dm.R, V = dm.V, D = dm.D, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4, c = tid & 15;
    int g0, n;
    rc_share(R, wave, g0, n);
    const int oz = rc_opaque0();
    if (frags) {
        rc_load(w.wa, P.p[R_Y1_W] + (size_t)(16 * j + i) * (R + V) + oz, R, g0, q);
        rc_load(w.wg, P.p[R_WH_W] + (size_t)(16 * j + i) * R + oz, R, g0, q);
        w.bh = P.p[R_WH_B][16 * j + c + oz]; w.bs = P.p[R_S_B][oz];
    }
    const float4* w4 = reinterpret_cast<const float4*>(P.p[R_Y2_W] + 16 * j + oz);
#pragma unroll
    for (int e = 0; e < 4; ++e) w.wq[e] = w4[e];
#pragma unroll
    for (int e = 0; e < 2; ++e) {                                       // Cd[d][slice] of this thread's first two classes (d = c, c + 16)
        const float4* c4 = reinterpret_cast<const float4*>(tp.Cd + (size_t)min(c + 16 * e, D - 1) * R + 16 * j + oz);
        w.cq[e][0] = c4[0]; w.cq[e][1] = c4[1]; w.cq[e][2] = c4[2]; w.cq[e][3] = c4[3];
    }
    if (j == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) w.sq[e] = *reinterpret_cast<const float4*>(P.p[R_S_W] + min(c * 4 + 64 * e, R - 4) + oz);      // (role 0: s.weight)
    }
}
// returns (role 0 only; true elsewhere): a sample of the tile goes on after this step
template <bool PS>
__device__ __forceinline__ bool rc_heads_body(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int t, const int tile, const int j,
                                              const RcHeadsW& w) {
    __shared__ float s_acc[2][4][16][17];
    __shared__ __attribute__((aligned(16))) float s_A[16][20];
    __shared__ float s_live[16], s_mn[16];
    const int B = dm.B, R = dm.R, V = dm.V, D = dm.D, T = dm.T;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0);
    // (per-thread indices re-derived from an OPAQUE copy of the thread id in every call: inside the persistent launch's step loop hipcc
    //  otherwise hoists each phase's row / column / tape-address arithmetic out of the loop and keeps hundreds of registers live across it)
    const int tid = rc_opaque0() + (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, q = lane >> 4;
    const int m = tid >> 4, c = tid & 15, b = min(b0 + m, B - 1);
    const bool may_stop = !ar.run_all && !dm.fixed && ar.train, train = ar.train != 0;
    const size_t rowb = (size_t)t * B, rowh = (size_t)(t + 1) * B;
    // ---- every operand of the phase goes out in ONE round trip: the products' fragments, the class rows / w_y2 slice of the
    //      partial logits, and (role 0) the stop head's h rows and the bookkeeping state
    const float live_in = rc_live<PS>(tp, B, t, min(b0 + (tid & 15), B - 1), (tid & 15) < nb, may_stop) ? 1.f : 0.f;
    const int bx = min(b0 + i, B - 1);
    int g0, n;
    rc_share(R, wave, g0, n);
    RcFrag ah;
    if (PS) rc_load_frag(ah, tp.rcxh + ((size_t)((t + 1) & 1) * ((B + 15) >> 4) + tile) * (R >> 4) * 256, R, g0, i, q);
    else rc_load_act<PS>(ah, tp.h + (rowh + bx) * R, R, g0, q);
    const float bh = w.bh;
    const float4 wq0 = w.wq[0], wq1 = w.wq[1], wq2 = w.wq[2], wq3 = w.wq[3];
    float4 hq[4];                                                       // role 0: h_{t+1} of sample m, 16 lanes x 4 floats x 4
    float m_t = 1.f, sp_before = 1.f, u_s = 0.f;
    int ts_before = -1;
    if (j == 0 && !PS) {                                                // (PS: the stop head is the tile's stop role, rc_stop_role)
        const float* hr = tp.h + (rowh + b) * R;
#pragma unroll
        for (int e = 0; e < 4; ++e) hq[e] = rc_ld4<PS>(hr + min(c * 4 + 64 * e, R - 4));
        if (t > 0) {
            m_t = rc_ld<PS>(&tp.rcst[(size_t)(t & 1) * B + b]);
            ts_before = PS ? __hip_atomic_load(&tp.tstar[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tp.tstar[b];
            if (!train) sp_before = rc_ld<PS>(&tp.sprod[b]);
        }
        if (train) u_s = ar.u_s ? ar.u_s[rowb + b] : philox_uniform(ar.seed, (uint32_t)(t * dm.Bg + dm.boff + b), tp.counter[0], 1u);
    }
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 a0 = rc_mma(ah, w.wa, n, z4), a1 = rc_mma(ah, w.wg, n, z4);
    MMG_RSTAMP(PS && tile == 0 && j == 0 && t == 3, 165);
    __syncthreads();                                                    // (PS: the LDS of the phase before is free)
    if (tid < MMG_TM) s_live[tid] = live_in;
#pragma unroll
    for (int r = 0; r < 4; ++r) { s_acc[0][wave][q * 4 + r][i] = a0[r]; s_acc[1][wave][q * 4 + r][i] = a1[r]; }
    __syncthreads();
    if (!PS && may_stop) {
        bool any = false;
        for (int mm = 0; mm < MMG_TM; ++mm) any = any || (s_live[mm] != 0.f);
        if (!any) return false;
    }
    {
        const int unit = 16 * j + c;
        auto S = [&](int p) { return (s_acc[p][0][m][c] + s_acc[p][1][m][c]) + (s_acc[p][2][m][c] + s_acc[p][3][m][c]); };
        s_A[m][c] = S(0);                                               // A (App. A.2)
        // w_h h + b_h in the MFMA accumulator order of the query phase: [tile][column tile j][r = m & 3][q = m >> 2][c]
        rc_st<PS>(&tp.rcgw[((((size_t)tile * (R >> 4) + j) * 4 + (m & 3)) * 4 + (m >> 2)) * 16 + c], S(1) + bh);
    }
    __syncthreads();
    {   // this slice's share of y[m][d] = b_y2 + sum_r w_y2[r] relu(A[m][r] + Cd[d][r])     (model.py:432-433)
        MMG_RSTAMP(PS && tile == 0 && j == 0 && t == 3, 166);
        const float4* a4 = reinterpret_cast<const float4*>(&s_A[m][0]);
        const float4 aq0 = a4[0], aq1 = a4[1], aq2 = a4[2], aq3 = a4[3];
        auto part = [&](const float4& c0, const float4& c1, const float4& c2, const float4& c3) {
            float s0 = 0.f, s1 = 0.f;
            s0 = fmaf(wq0.x, fmax_nn(aq0.x + c0.x, 0.f), s0); s1 = fmaf(wq0.y, fmax_nn(aq0.y + c0.y, 0.f), s1);
            s0 = fmaf(wq0.z, fmax_nn(aq0.z + c0.z, 0.f), s0); s1 = fmaf(wq0.w, fmax_nn(aq0.w + c0.w, 0.f), s1);
            s0 = fmaf(wq1.x, fmax_nn(aq1.x + c1.x, 0.f), s0); s1 = fmaf(wq1.y, fmax_nn(aq1.y + c1.y, 0.f), s1);
            s0 = fmaf(wq1.z, fmax_nn(aq1.z + c1.z, 0.f), s0); s1 = fmaf(wq1.w, fmax_nn(aq1.w + c1.w, 0.f), s1);
            s0 = fmaf(wq2.x, fmax_nn(aq2.x + c2.x, 0.f), s0); s1 = fmaf(wq2.y, fmax_nn(aq2.y + c2.y, 0.f), s1);
            s0 = fmaf(wq2.z, fmax_nn(aq2.z + c2.z, 0.f), s0); s1 = fmaf(wq2.w, fmax_nn(aq2.w + c2.w, 0.f), s1);
            s0 = fmaf(wq3.x, fmax_nn(aq3.x + c3.x, 0.f), s0); s1 = fmaf(wq3.y, fmax_nn(aq3.y + c3.y, 0.f), s1);
            s0 = fmaf(wq3.z, fmax_nn(aq3.z + c3.z, 0.f), s0); s1 = fmaf(wq3.w, fmax_nn(aq3.w + c3.w, 0.f), s1);
            return s0 + s1;
        };
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int d = c + 16 * e;
            if (d < D && m < nb) rc_st<PS>(&tp.rcyp[((((size_t)tile * 4 + (j >> 2)) * 16 + m) * 32 + d) * 4 + (j & 3)], part(w.cq[e][0], w.cq[e][1], w.cq[e][2], w.cq[e][3]));
        }
        // (D <= 32: layout.h rc_shape -- two classes per thread cover them)
    }
    if (PS || j != 0) return true;
    MMG_RSTAMP(PS && tile == 0 && j == 0 && t == 3, 167);
    {   // stop bit (model.py:414-427) and the stop-mask bookkeeping (model.py:852) -- per sample
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c * 4 + 64 * e < R) {
                acc = fmaf(w.sq[e].x, hq[e].x, acc); acc = fmaf(w.sq[e].y, hq[e].y, acc);
                acc = fmaf(w.sq[e].z, hq[e].z, acc); acc = fmaf(w.sq[e].w, hq[e].w, acc);
            }
        }
        acc = dpp_group_sum<16>(acc);
        if (c == 0) {
            const bool valid = m < nb, live = s_live[m] != 0.f;
            const float p = fsigmoid(acc + w.bs);
            float sv, prod = 1.f;
            if (train) sv = (u_s < p) ? 1.f : 0.f;                                          // model.py:420
            else {
                prod = dm.s_prob_prod ? sp_before * p : p;                                  // model.py:423-426
                sv = rintf(prod);                                                           // model.py:427
            }
            const float m_next = fminf(m_t, sv);
            const bool take = dm.fixed ? (t == T - 1) : (ts_before < 0 && (m_next == 0.f || t == T - 1));
            s_mn[m] = valid ? m_next : 0.f;
            if (valid) {
                rc_st<PS>(&tp.rcst[(size_t)((t + 1) & 1) * B + b], m_next);
                rc_st<PS>(&tp.rcst[(size_t)2 * B + b], take ? 1.f : 0.f);
                if (PS) __hip_atomic_store(&tp.tstar[b], take ? t : ts_before, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else tp.tstar[b] = take ? t : ts_before;
                tp.mstate[b] = m_next;
                if (!train) rc_st<PS>(&tp.sprod[b], prod);
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
        bool alive = false;
        MMG_RSTAMP(PS && tile == 0 && j == 0 && t == 3, 168);
        for (int mm = 0; mm < nb; ++mm) alive = alive || (s_mn[mm] != 0.f);
        if (tid == 0 && t + 1 < T && alive) atomicAdd(&tp.alive[t + 1], 1);
        return alive;
    }
}
__global__ __launch_bounds__(256) void k_rc_heads(Dims dm, Params P, Tape tp, ConvArgs ar, int t, int skip) {
    if (skip && tp.alive[t] == 0) return;
    const int NJ = dm.R >> 4;
    RcHeadsW w;
    rc_heads_w(w, dm, P, tp, blockIdx.x % NJ, true);
    rc_heads_body<false>(dm, P, tp, ar, t, blockIdx.x / NJ, blockIdx.x % NJ, w);
}

// ---------------------------------------------------------------------------------------------
// part 0: the step's critical path (logits, softmax, h_w, message).  part 1 (role 0 of the tile, after its message has gone out): the
// description mixture dbar = softmax(y) . desc (model.py:442-449) for the tape -- only k_wgrad's w_d job reads it; the query head
// itself uses W_d dbar = softmax(y) . Dd with Dd = desc . W_d^T folded onto the classes by k_prep (tape.Dd, 30 KB in LDS).
struct RcQueryW { RcFrag ww; float bw, b2; };
__device__ __forceinline__ void rc_query_w(RcQueryW& w, const Dims& dm, const Params& P, const int jw) {
    const int R = dm.R, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    int g0, n;
    rc_share(R, wave, g0, n);
    const int oz = rc_opaque0();
    rc_load(w.ww, P.p[R_W_W] + (size_t)(16 * jw + i) * R + oz, R, g0, q);
    w.bw = P.p[R_W_B][16 * jw + (tid & 15) + oz]; w.b2 = P.p[R_Y2_B][oz];
}
template <bool PS>
__device__ __forceinline__ void rc_query_body(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int t, const int tile, const int jw, const int part,
                                              const RcQueryW& w) {
    __shared__ __attribute__((aligned(16))) float s_y[16][68];
    __shared__ __attribute__((aligned(16))) float s_g[16][260];
    __shared__ __attribute__((aligned(16))) float s_Dd[32][260];         // read once per launch (PS) instead of once per step
    __shared__ float s_acc[4][16][17];
    __shared__ float s_live[16], s_live2[16], s_take[16];
    const int B = dm.B, W = dm.W, R = dm.R, V = dm.V, D = dm.D, NJ = R >> 4;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0);
    // (per-thread indices re-derived from an OPAQUE copy of the thread id in every call: inside the persistent launch's step loop hipcc
    //  otherwise hoists each phase's row / column / tape-address arithmetic out of the loop and keeps hundreds of registers live across it)
    const int tid = rc_opaque0() + (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), i = lane & 15, q = lane >> 4;
    const int m = tid >> 4, c = tid & 15, b = min(b0 + m, B - 1), ncol = 16 * jw + c;
    const bool may_stop = !ar.run_all && !dm.fixed && ar.train, train = ar.train != 0, binary = dm.use_binary != 0;
    const size_t rowb = (size_t)t * B;
    if (part == 1) {
        // softmax(y) is still in s_y; desc from L2 (16 x D x V on the matrix cores: K = D in steps of 4, lane (i, q): A[i][4 s + q], B[4 s + q][16 vt + i])
        const int ks = (D + 3) >> 2, nv = (V + 15) >> 4;
        for (int vt = wave; vt < nv; vt += 4) {
            const int col = vt * 16 + i;
            float bv[8];
#pragma unroll
            for (int sk = 0; sk < 8; ++sk) { const int kk = 4 * sk + q; bv[sk] = (kk < D && col < V) ? ar.desc[(size_t)kk * V + col] : 0.f; }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sk = 0; sk < 8; ++sk) if (sk < ks) acc = mfma16(s_y[i][4 * sk + q], bv[sk], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = q * 4 + r;
                if (col < V && s_live2[mm] != 0.f) tp.dbar[(rowb + b0 + mm) * V + col] = acc[r];
            }
        }
        return;
    }
    // ---- operands that do not wait for anything this phase computes: row flags, the partial logits, this role's rows of W_w,
    //      its bias and Bernoulli uniforms -- one round trip
    float f_live = 0.f, f_next = 0.f, f_take = 0.f;
    if (tid < MMG_TM) {
        const int bb = min(b0 + tid, B - 1);
        f_live = rc_live<PS>(tp, B, t, bb, tid < nb, may_stop) ? 1.f : 0.f;
        f_next = may_stop ? rc_ld<PS>(&tp.rcst[(size_t)((t + 1) & 1) * B + bb]) : 1.f;
        f_take = (tid < nb) ? rc_ld<PS>(&tp.rcst[(size_t)2 * B + bb]) : 0.f;
    }
    float4 pv[2][4];                                                    // thread (m, d = c) and (m, d = c + 16): the 16 slice partials of its first two classes
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        // [tile][slice quad u][sample][class][4]: the 16 lanes of a sample read 256 contiguous bytes per (u, e)
        const float* yp = tp.rcyp + (((size_t)tile * 4 * 16 + m) * 32 + min(c + 16 * e, D - 1)) * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) pv[e][u] = rc_ld4<PS>(yp + (size_t)u * 16 * 32 * 4);
    }
    float gwa[4][4];                                                    // w_h h + b_h of this wave's (at most four: R <= 256) 16-column tiles of h_w
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) gwa[u][r] = rc_ld<PS>(&tp.rcgw[((((size_t)tile * NJ + min(wave + 4 * u, NJ - 1)) * 4 + r) * 4 + q) * 16 + i]);
    int gm0, nm;
    rc_share(R, wave, gm0, nm);
    const float bw = w.bw, b2 = w.b2;
    float u_w = 0.f;
    if (binary && train)
        u_w = ar.u_w ? ar.u_w[(rowb + b) * W + ncol] : philox_uniform(ar.seed, (uint32_t)((t * dm.Bg + dm.boff + b) * W + ncol), tp.counter[0], 2u);
    __syncthreads();                                                    // (PS: the LDS of the phase before is free)
    if (tid < MMG_TM) { s_live[tid] = f_live; s_live2[tid] = (f_live != 0.f && f_next != 0.f) ? 1.f : 0.f; s_take[tid] = f_take; }
    if (!PS || t == 0)
        for (int idx = tid * 4; idx < 32 * R; idx += 1024) {            // (rows D..31 zero: K padding of softmax(y) . Dd)
            const int d = idx / R, r = idx - d * R;
            *reinterpret_cast<float4*>(&s_Dd[d][r]) = (d < D) ? *reinterpret_cast<const float4*>(tp.Dd + (size_t)d * R + r) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    __syncthreads();
    if (!PS && may_stop) {
        bool any = false;
        for (int mm = 0; mm < MMG_TM; ++mm) any = any || (s_live[mm] != 0.f);
        if (!any) return;
    }
    // ---- class logits: the R/16 partials in role order
    MMG_RSTAMP(PS && tile == 0 && jw == 0 && t == 3, 170);
    for (int d = c, e = 0; d < D; d += 16, ++e) {
        float acc = 0.f;
        float4 pq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) pq[u] = (e == 0) ? pv[0][u] : pv[1][u];         // (D <= 32: two classes per thread)
#pragma unroll
        for (int u = 0; u < 4; ++u) {                                   // slice order
            acc += (4 * u < NJ) ? pq[u].x : 0.f; acc += (4 * u + 1 < NJ) ? pq[u].y : 0.f;
            acc += (4 * u + 2 < NJ) ? pq[u].z : 0.f; acc += (4 * u + 3 < NJ) ? pq[u].w : 0.f;
        }
        const float yv = acc + b2;
        s_y[m][d] = yv;
        if (jw == 0 && m < nb) {
            if ((ar.y_last_only ? s_take[m] : s_live[m]) != 0.f) tp.y[(rowb + b) * D + d] = yv;
            if (s_take[m] != 0.f) rc_st<PS>(&tp.outp[(size_t)b * D + d], yv);       // the output step, model.py:1261-1264
        }
    }
    __syncthreads();
    // ---- softmax(y) (detached, model.py:441): wave per sample row, in place (D <= 64: one class per lane)
    MMG_RSTAMP(PS && tile == 0 && jw == 0 && t == 3, 171);
    for (int mm = wave; mm < MMG_TM; mm += 4) {
        const float v = (lane < D) ? s_y[mm][lane] : -3.0e38f;
        const float mx = dpp_wave_max(v);
        const float e = (lane < D) ? __expf(v - mx) : 0.f;
        const float se = dpp_wave_sum(e);
        s_y[mm][lane] = (lane < D) ? e * __builtin_amdgcn_rcpf(se) : 0.f;      // (columns D..63 zero: K padding of the mixture product)
    }
    __syncthreads();
    MMG_RSTAMP(PS && tile == 0 && jw == 0 && t == 3, 172);
    MMG_RSTAMP(PS && tile == 0 && jw == 0 && t == 3, 173);
    // ---- h_w = tanh(w_h h + b_h + softmax(y) . Dd)   (model.py:452; Dd = desc . W_d^T): all R columns on the matrix cores, both operands
    //      in LDS (K = D in steps of 4), the 16-column tiles round-robin over the waves
    {
        const int ks = (D + 3) >> 2;
        // (every LDS operand of the four products is read before the first MFMA: one LDS round trip, not one per MFMA)
        float av[8], bv[4][8];
#pragma unroll
        for (int sk = 0; sk < 8; ++sk) av[sk] = s_y[i][4 * sk + q];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int sk = 0; sk < 8; ++sk) bv[u][sk] = s_Dd[4 * sk + q][min(wave + 4 * u, NJ - 1) * 16 + i];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tn = wave + 4 * u;
            if (tn >= NJ) break;
            const int col = tn * 16 + i;
            float gwv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) gwv[r] = gwa[u][r];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sk = 0; sk < 8; ++sk) if (sk < ks) acc = mfma16(av[sk], bv[u][sk], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = q * 4 + r;
                const float gv = ftanh(gwv[r] + acc[r]);
                s_g[mm][col] = gv;
                if (jw == 0 && s_live2[mm] != 0.f) tp.g[(rowb + b0 + mm) * R + col] = gv;
            }
        }
    }
    __syncthreads();
    MMG_RSTAMP(PS && tile == 0 && jw == 0 && t == 3, 174);
    // ---- this role's 16 bits of the receiver's message (model.py:454-475)
    {
        RcFrag ag;
        const int kgr = R >> 4;
#pragma unroll
        for (int u = 0; u < RC_MAXG; ++u) ag.v[u] = *reinterpret_cast<const float4*>(&s_g[i][min(gm0 + u, kgr - 1) * 16 + q * 4]);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a0 = rc_mma(ag, w.ww, nm, z4);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_acc[wave][q * 4 + r][i] = a0[r];
    }
    __syncthreads();
    MMG_RSTAMP(PS && tile == 0 && jw == 0 && t == 3, 175);
    {
        const float lw = (s_acc[0][m][c] + s_acc[1][m][c]) + (s_acc[2][m][c] + s_acc[3][m][c]) + bw;
        float wv = lw, lpv = 0.f, nev = 0.f;
        const bool st = s_live2[m] != 0.f;
        if (binary) {
            const float pp = fsigmoid(lw);
            wv = train ? ((u_w < pp) ? 1.f : 0.f) : rintf(pp);                              // model.py:460 / 462
            if (st) tp.pw[(rowb + b) * W + ncol] = pp;
            const float l1 = flog(pp + MMG_EPS), l0 = flog(1.f - pp + MMG_EPS);
            lpv = wv * l1 + (1.f - wv) * l0; nev = pp * l1 + (1.f - pp) * l0;
        }
        if (st) rc_st<PS>(&tp.w[(rowb + b) * W + ncol], wv);
        if (PS) st_wt(&tp.rcxw[(((size_t)tile * (W >> 4) + jw) * 16 + m) * 16 + c], wv);         // the S1 roles' copy, fragment order
        if (binary) {
            lpv = dpp_group_sum<16>(lpv); nev = dpp_group_sum<16>(nev);
            if (c == 0 && m < nb) { float* pl = tp.rclw + (((size_t)(t & 1) * (W >> 4) + jw) * B + b) * 2; rc_st<PS>(pl, lpv); rc_st<PS>(pl + 1, nev); }
        }
    }
}
__global__ __launch_bounds__(256) void k_rc_query(Dims dm, Params P, Tape tp, ConvArgs ar, int t, int skip) {
    if (skip && tp.alive[t] == 0) return;
    const int NJW = dm.W >> 4;
    RcQueryW w;
    rc_query_w(w, dm, P, blockIdx.x % NJW);
    rc_query_body<false>(dm, P, tp, ar, t, blockIdx.x / NJW, blockIdx.x % NJW, 0, w);
    if (blockIdx.x % NJW == 0) { __syncthreads(); rc_query_body<false>(dm, P, tp, ar, t, blockIdx.x / NJW, 0, 1, w); }
}

// ---------------------------------------------------------------------------------------------
template <bool PS>
__device__ __forceinline__ void rc_tail_body(const Dims& dm, const Tape& tp, const ConvArgs& ar, const int tile, const bool last_lw) {
    const int B = dm.B, D = dm.D, T = dm.T;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool may_stop = !ar.run_all && !dm.fixed && ar.train;
    if (dm.use_binary && last_lw) rc_sum_lw<PS>(dm, tp, T - 1, b0, nb, may_stop);
    // output selection, log-softmax, reward, top-k (model.py:1264-1275, 1333-1339): wave per sample, D <= 64
    for (int m = wave; m < nb; m += 4) {
        const int b = b0 + m;
        const float v = (lane < D) ? rc_ld<PS>(&tp.outp[(size_t)b * D + lane]) : -3.0e38f;
        const float mx = dpp_wave_max(v);
        const float se = dpp_wave_sum((lane < D) ? __expf(v - mx) : 0.f);
        const float lse = mx + flog(se);
        const int tgt = ar.target ? (int)ar.target[b] : -1;
        const float dt = (tgt >= 0) ? (rc_ld<PS>(&tp.outp[(size_t)b * D + max(tgt, 0)]) - lse) : 0.f;
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
__global__ __launch_bounds__(256) void k_rc_tail(Dims dm, Params P, Tape tp, ConvArgs ar) {
    rc_tail_body<false>(dm, tp, ar, blockIdx.x, true);
    if (threadIdx.x == 0) tp.rcflags[(size_t)blockIdx.x * 64] = 0u;    // the backward roles' hand-off counter of the tile (k_rc_bwd)
}

// ---------------------------------------------------------------------------------------------
// k_rc_persist: the same conversation as ONE launch of co-resident workgroup roles that hand their results on through memory
// + counters (kernels_tile.h: pf_signal / pf_wait; write-through payload stores, agent-scope payload loads, bounded spins ->
// error word) instead of 5 T + 1 launches.  Per tile of 16 samples:
//   S1 roles (H / 64)              64 units of a_t = tanh(h_x + w_{t-1} W_c^T + b_c)   (model.py:195-216), weight fragments in registers
//   S2 roles (W / 16)              16 bits of z_t ~ Bernoulli(sigmoid(a_t W_b^T + b_b)) (model.py:218-236), weight fragments in registers
//   RC roles (max(R, W) / 16)      GRU slice -> heads slice -> message slice (the three bodies above), two in-cluster hand-offs
//   stop role (1)                  the stop bit and the stop-mask bookkeeping of the tile's samples (off the receiver roles' path)
// Counters (tape.pflags, zeroed by k_prep; (kind * 40 + tile) x 256 bytes): 0 w_t out (RC -> S1), 1 a_t out (S1 -> S2), 2 z_t out
// (S2 -> RC), 3 h_{t+1} out, 4 partial logits / w_h h out, 5 the tile's conversations are over, 6 stop masks / row flags out.
// ---------------------------------------------------------------------------------------------
#define RC_MAXTILES 40
#ifndef RC_RES_HEADS
#define RC_RES_HEADS 0          // 1: the heads' two weight fragments stay in registers across steps (with the GRU's 24: spills)
#endif
#ifndef RC_RES_GRU
#define RC_RES_GRU 0          // 1: the GRU's 24 weight fragments as explicit registers across steps -- hipcc then spills them to scratch (816 B) instead of AGPRs
#endif
#ifndef RC_RES_QUERY
#define RC_RES_QUERY 0
#endif
__device__ __forceinline__ uint32_t* rc_ctr(const Tape& tp, int kind, int tile) { return tp.pflags + ((size_t)kind * RC_MAXTILES + tile) * 64; }

__device__ __forceinline__ void rc_s1_role(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int tile, const int sidx, const int nrc) {
    __shared__ float s_acc[4][4][16][17];
    const int B = dm.B, H = dm.H, W = dm.W, T = dm.T;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0), n0 = sidx * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    uint32_t* cW = rc_ctr(tp, 0, tile); uint32_t* cA = rc_ctr(tp, 1, tile); uint32_t* done = rc_ctr(tp, 5, tile);
    int g0, n;
    rc_share(W, wave, g0, n);
    RcFrag wc[4];                                                       // this wave's K share of the role's 4 x 16 rows of W_c: loaded once
#pragma unroll
    for (int u = 0; u < 4; ++u) rc_load(wc[u], P.p[S_CODE_W] + (size_t)min(n0 + u * 16 + i, H - 1) * W, W, g0, q);
    const int m = tid >> 4, c = tid & 15, b = min(b0 + m, B - 1);
    float hxv[4], bcv[4], hw0v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int hn = min(n0 + u * 16 + c, H - 1);
        hxv[u] = tp.hx[(size_t)b * H + hn]; bcv[u] = P.p[S_CODE_B][hn]; hw0v[u] = tp.hw0[hn];
    }
    for (int t = 0; t < T; ++t) {
        const size_t rowb = (size_t)t * B;
        if (t > 0) {
            MMG_RSTAMP(tile == 0 && sidx == 0 && t == 3, 100);
            if (!pf_wait<false>(cW, (uint32_t)(nrc * t), done, tp.sync)) return;
            MMG_RSTAMP(tile == 0 && sidx == 0 && t == 3, 101);
            RcFrag aw;
            rc_load_frag(aw, tp.rcxw + (size_t)tile * (W >> 4) * 256, W, g0, i, q);
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 a = rc_mma(aw, wc[u], n, z4);
#pragma unroll
                for (int r = 0; r < 4; ++r) s_acc[u][wave][q * 4 + r][i] = a[r];
            }
            __syncthreads();
            MMG_RSTAMP(tile == 0 && sidx == 0 && t == 3, 102);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int hn = n0 + u * 16 + c;
            const float hw = (t == 0) ? hw0v[u] : (s_acc[u][0][m][c] + s_acc[u][1][m][c]) + (s_acc[u][2][m][c] + s_acc[u][3][m][c]) + bcv[u];
            const float av = ftanh(hxv[u] + hw);                                               // model.py:216
            if (m < nb && hn < H) tp.a[(rowb + b) * H + hn] = av;                               // the tape (backward, k_wgrad): plain store
            // the S2 roles' copy in fragment order (k-group (n0 >> 4) + u of the tile: 16 samples x 16 columns = 1 KB contiguous)
            if (hn < H) st_wt(&tp.rcxa[(((size_t)tile * (H >> 4) + (n0 >> 4) + u) * 16 + m) * 16 + c], (m < nb) ? av : 0.f);
        }
        MMG_RSTAMP(tile == 0 && sidx == 0 && t == 3, 103);
        pf_signal(cA);
        MMG_RSTAMP(tile == 0 && sidx == 0 && t == 3, 104);
        if (sidx == 0) {                                                // code input rows of the tile (tapes c, zr): off the hand-off's critical path
            for (int idx = tid * 4; idx < nb * W; idx += 1024) {
                const int mm = idx / W, jj = idx - mm * W;              // (W is a multiple of 16: a quad never straddles rows)
                const float4 cv = (t == 0) ? make_float4(dm.first_rec, dm.first_rec, dm.first_rec, dm.first_rec)
                                           : ld_cc4(&tp.w[((size_t)(t - 1) * B + b0 + mm) * W + jj]);
                *reinterpret_cast<float4*>(&tp.zr[(rowb + b0 + mm) * W + jj]) = cv;        // z_r of baseline_sen, model.py:836
                const float4 cb = *reinterpret_cast<const float4*>(P.p[S_CODE_BIAS] + jj);
                *reinterpret_cast<float4*>(&tp.c[(rowb + b0 + mm) * W + jj]) =
                    (t == 0) ? make_float4(fsigmoid(cb.x), fsigmoid(cb.y), fsigmoid(cb.z), fsigmoid(cb.w)) : cv;
            }
        }
    }
}

__device__ __forceinline__ void rc_s2_role(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int tile, const int k, const int ns1) {
    __shared__ float s_acc[4][16][17];
    const int B = dm.B, H = dm.H, W = dm.W, T = dm.T;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0), n0 = k * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    uint32_t* cA = rc_ctr(tp, 1, tile); uint32_t* cZ = rc_ctr(tp, 2, tile); uint32_t* done = rc_ctr(tp, 5, tile);
    // K = H over the four waves: up to 16 k-groups each (H <= 1024), the role's W_b rows as register fragments
    const int kg = H >> 4, per = (kg + 3) >> 2, g0 = wave * per, n = max(0, min(kg, g0 + per) - g0);
    float4 wb[16];
    const float* wrow = P.p[S_BIN_W] + (size_t)(n0 + i) * H;
#pragma unroll
    for (int u = 0; u < 16; ++u) wb[u] = *reinterpret_cast<const float4*>(wrow + min(g0 + u, kg - 1) * 16 + q * 4);
    const int m = tid >> 4, c = tid & 15, b = min(b0 + m, B - 1), col = n0 + c;
    const float bb = P.p[S_BIN_B][col];
    const uint32_t mb_counter = tp.counter[0];
    for (int t = 0; t < T; ++t) {
        const size_t rowb = (size_t)t * B;
        float uz = 0.f;                                                 // the uniform of this thread's bit does not depend on the step's data: drawn before the wait
        if (dm.use_binary && ar.train)
            uz = ar.u_z ? ar.u_z[(rowb + b) * W + col] : philox_uniform(ar.seed, (uint32_t)((t * dm.Bg + dm.boff + b) * W + col), mb_counter, 0u);
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 110);
        if (!pf_wait<false>(cA, (uint32_t)(ns1 * (t + 1)), done, tp.sync)) return;
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 111);
        const float* afrag = tp.rcxa + (size_t)tile * (H >> 4) * 256 + i * 16 + q * 4;     // [k-group][sample i][4 q ..]
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        {
            float4 av[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) av[u] = ld_cc4(afrag + (size_t)min(g0 + u, kg - 1) * 256);
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                if (u < n) {
                    acc0 = mfma16(av[u].x, wb[u].x, acc0); acc0 = mfma16(av[u].y, wb[u].y, acc0);
                    acc0 = mfma16(av[u].z, wb[u].z, acc0); acc0 = mfma16(av[u].w, wb[u].w, acc0);
                }
                if (u + 1 < n) {
                    acc1 = mfma16(av[u + 1].x, wb[u + 1].x, acc1); acc1 = mfma16(av[u + 1].y, wb[u + 1].y, acc1);
                    acc1 = mfma16(av[u + 1].z, wb[u + 1].z, acc1); acc1 = mfma16(av[u + 1].w, wb[u + 1].w, acc1);
                }
            }
        }
        const f32x4 acc = acc0 + acc1;
#pragma unroll
        for (int r = 0; r < 4; ++r) s_acc[wave][q * 4 + r][i] = acc[r];
        __syncthreads();
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 112);
        {
            const float lz = (s_acc[0][m][c] + s_acc[1][m][c]) + (s_acc[2][m][c] + s_acc[3][m][c]) + bb;
            float zz = lz;
            if (dm.use_binary) {
                const float pp = fsigmoid(lz);
                zz = ar.train ? ((uz < pp) ? 1.f : 0.f) : rintf(pp);    // model.py:227 / 229
                if (m < nb) st_wt(&tp.pz[(rowb + b) * W + col], pp);
            }
            if (m < nb) st_wt(&tp.z[(rowb + b) * W + col], zz);
            st_wt(&tp.rcxz[(((size_t)tile * (W >> 4) + k) * 16 + m) * 16 + c], (m < nb) ? zz : 0.f);     // the GRU slices' copy, fragment order
        }
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 113);
        pf_signal(cZ);                                                  // (its barrier also frees s_acc for the next step)
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 114);
    }
}

// one poll loop over TWO counters (both requests in flight together): the receiver roles need the partial logits (counter 4) and
// the stop role's flags (counter 6) before the query phase
__device__ __forceinline__ bool rc_wait2(uint32_t* c1, uint32_t t1, uint32_t* c2, uint32_t t2, uint32_t* done, uint32_t* sync_err) {
    __shared__ int s_ok2;
    if (threadIdx.x == 0) {
        int ok = -1, spins = 0;
        while (ok < 0) {
            const uint32_t d = __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t a = __hip_atomic_load(c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t b = __hip_atomic_load(c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d != 0u) ok = 0;
            else if (a >= t1 && b >= t2) ok = 1;
            else {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > MMG_SPIN_LIMIT) { __hip_atomic_store(sync_err + MMG_SYNC_ERR, 100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; }
            }
        }
        s_ok2 = ok;
    }
    __syncthreads();
    const bool r = s_ok2 != 0;
    __syncthreads();
    return r;
}

// The stop head of a tile as a role of its own (one workgroup per tile): with it inside heads role 0 every role of the tile waited
// 2.5 us per step for that one (scripts/rc_timeline.py).  Per step: waits h_{t+1}, stop bit (model.py:414-427) + stop-mask
// bookkeeping (model.py:852) + the step's row flags (rcst) and "a conversation of the tile goes on" (rcst[3][first sample]), signals 6.
__device__ __forceinline__ void rc_stop_role(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int tile, const int nrc) {
    __shared__ float s_mn[16];
    const int B = dm.B, R = dm.R, T = dm.T;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0);
    const int tid = threadIdx.x, m = tid >> 4, c = tid & 15, b = min(b0 + m, B - 1);
    const bool may_stop = !ar.run_all && !dm.fixed && ar.train, train = ar.train != 0;
    uint32_t* cH = rc_ctr(tp, 3, tile); uint32_t* cF = rc_ctr(tp, 6, tile); uint32_t* done = rc_ctr(tp, 5, tile);
    float4 sq[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) sq[e] = *reinterpret_cast<const float4*>(P.p[R_S_W] + min(c * 4 + 64 * e, R - 4));
    const float bs = P.p[R_S_B][0];
    const uint32_t mb_counter = tp.counter[0];
    float m_t = 1.f, sprod = 1.f;                                       // carried by this role: nobody else writes them
    int tstar = -1;
    for (int t = 0; t < T; ++t) {
        const size_t rowb = (size_t)t * B, rowh = (size_t)(t + 1) * B;
        float u_s = 0.f;
        if (train) u_s = ar.u_s ? ar.u_s[rowb + b] : philox_uniform(ar.seed, (uint32_t)(t * dm.Bg + dm.boff + b), mb_counter, 1u);
        if (!pf_wait<false>(cH, (uint32_t)(nrc * (t + 1)), done, tp.sync)) return;
        const float* hr = tp.h + (rowh + b) * R;
        float4 hq[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hq[e] = ld_cc4(hr + min(c * 4 + 64 * e, R - 4));
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (c * 4 + 64 * e < R) {
                acc = fmaf(sq[e].x, hq[e].x, acc); acc = fmaf(sq[e].y, hq[e].y, acc);
                acc = fmaf(sq[e].z, hq[e].z, acc); acc = fmaf(sq[e].w, hq[e].w, acc);
            }
        }
        acc = dpp_group_sum<16>(acc);
        {
            // (all 16 lanes of a sample carry its state: lane 0 stores)
            const bool valid = m < nb, live = valid && (!may_stop || m_t != 0.f);
            const float p = fsigmoid(acc + bs);
            float sv;
            if (train) sv = (u_s < p) ? 1.f : 0.f;                                          // model.py:420
            else {
                sprod = dm.s_prob_prod ? sprod * p : p;                                     // model.py:423-426
                sv = rintf(sprod);                                                          // model.py:427
            }
            const float m_next = fminf(m_t, sv);
            const bool take = dm.fixed ? (t == T - 1) : (tstar < 0 && (m_next == 0.f || t == T - 1));
            if (take) tstar = t;
            if (c == 0) {
                s_mn[m] = valid ? m_next : 0.f;
                if (valid) {
                    st_wt(&tp.rcst[(size_t)((t + 1) & 1) * B + b], m_next);
                    st_wt(&tp.rcst[(size_t)2 * B + b], take ? 1.f : 0.f);
                    tp.tstar[b] = tstar;
                    tp.mstate[b] = m_next;
                    if (!train) tp.sprod[b] = sprod;
                }
                if (live) {
                    tp.s[rowb + b] = sv; tp.ps[rowb + b] = p;
                    const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
                    tp.lp_s[rowb + b] = sv * l1 + (1.f - sv) * l0;
                    tp.ne_s[rowb + b] = p * l1 + (1.f - p) * l0;
                    tp.mask[rowh + b] = (uint8_t)(m_next != 0.f);
                }
            }
            m_t = m_next;
        }
        __syncthreads();
        bool alive = false;
        for (int mm = 0; mm < nb; ++mm) alive = alive || (s_mn[mm] != 0.f);
        if (tid == 0) {
            st_wt(&tp.rcst[(size_t)3 * B + b0], alive ? 1.f : 0.f);
            if (t + 1 < T && alive) atomicAdd(&tp.alive[t + 1], 1);
        }
        pf_signal(cF);
        if (may_stop && !alive) return;                                 // (receiver role 0 ends the tile's conversation)
    }
}

// tile0 / tiles: the launch covers the sample tiles [tile0, tile0 + tiles) -- batches whose roles do not all fit the device run as
// consecutive launches over tile ranges (the conversations of different samples are independent)
__global__ __launch_bounds__(256) void k_rc_persist(Dims dm, Params P, Tape tp, ConvArgs ar, int tiles, int tile0) {
    const int T = dm.T, NJ = dm.R >> 4, NJW = dm.W >> 4, nrc = NJ > NJW ? NJ : NJW, ns1 = (dm.H + 63) >> 6, ns2 = NJW;
    const int per_tile = nrc + ns1 + ns2 + 1;                           // receiver slices, sender slices, the stop role
    // roles of a tile sit side by side in the grid (consecutive workgroups go round the XCDs)
    const int ltile = blockIdx.x / per_tile, slot = blockIdx.x - ltile * per_tile, tile = tile0 + ltile;
    if (ltile >= tiles) {
        // trailing workgroups (dispatched after every role, onto CUs the roles leave idle): 16 x 16 tiles of
        // basehx = h_x . baseline_sen.linear1.weight[:, :H]^T, which k_baselines4 needs next and nothing here produces
        gemm_nt_tile(blockIdx.x - tiles * per_tile, tp.hx, dm.H, P.p[BS_L1_W], dm.H + dm.W, nullptr, tp.basehx, dm.K, dm.B, dm.K, dm.H);
        return;
    }
    if (slot == nrc + ns2 + ns1) { rc_stop_role(dm, P, tp, ar, tile, nrc); return; }
    if (slot >= nrc + ns2) { rc_s1_role(dm, P, tp, ar, tile, slot - nrc - ns2, nrc); return; }
    if (slot >= nrc) { rc_s2_role(dm, P, tp, ar, tile, slot - nrc, ns1); return; }
    const int k = slot;
    uint32_t* cW = rc_ctr(tp, 0, tile); uint32_t* cZ = rc_ctr(tp, 2, tile); uint32_t* cH = rc_ctr(tp, 3, tile);
    uint32_t* cY = rc_ctr(tp, 4, tile); uint32_t* done = rc_ctr(tp, 5, tile); uint32_t* cF = rc_ctr(tp, 6, tile);
    const bool may_stop = !ar.run_all && !dm.fixed && ar.train;
    bool whole = true;
    // The weight fragments of a phase are requested right BEFORE the phase's wait: they arrive while the role polls, and the phase
    // then pays the payload's round trip only.  (Keeping all of them in registers across the whole step loop spills: RC_RES_GRU.)
    for (int t = 0; t < T; ++t) {
        RcGruW wgru; RcHeadsW wheads; RcQueryW wquery;
        rc_gru_w(wgru, dm, P, min(k, NJ - 1));
        const bool stamp = tile == 0 && t == 3 && (k == 0 || k == 5);
        const int so = (k == 0) ? 120 : 140;
        MMG_RSTAMP(stamp, so + 0);
        if (!pf_wait<false>(cZ, (uint32_t)(ns2 * (t + 1)), done, tp.sync)) return;
        MMG_RSTAMP(stamp, so + 1);
        if (k < NJ) rc_gru_body<true>(dm, P, tp, ar, t, tile, k, wgru);
        MMG_RSTAMP(stamp, so + 2);
        pf_signal(cH);
        MMG_RSTAMP(stamp, so + 3);
        rc_heads_w(wheads, dm, P, tp, min(k, NJ - 1), true);
        MMG_RSTAMP(stamp, so + 4);
        if (!pf_wait<false>(cH, (uint32_t)(nrc * (t + 1)), done, tp.sync)) return;
        MMG_RSTAMP(stamp, so + 5);
        bool alive = true;
        if (k < NJ) alive = rc_heads_body<true>(dm, P, tp, ar, t, tile, k, wheads);
        MMG_RSTAMP(stamp, so + 6);
        pf_signal(cY);
        rc_query_w(wquery, dm, P, min(k, NJW - 1));
        MMG_RSTAMP(stamp, so + 7);
        if (!rc_wait2(cY, (uint32_t)(nrc * (t + 1)), cF, (uint32_t)(t + 1), done, tp.sync)) return;
        if (k == 0) alive = ld_cc(&tp.rcst[(size_t)3 * dm.B + tile * MMG_TM]) != 0.f;       // (the stop role's verdict)
        MMG_RSTAMP(stamp, so + 8);
        if (k < NJW) rc_query_body<true>(dm, P, tp, ar, t, tile, k, 0, wquery);
        MMG_RSTAMP(stamp, so + 9);
        pf_signal(cW);
        MMG_RSTAMP(stamp, so + 10);
        if (k == 0) {                                                   // (while the sender roles work: rclw is double-buffered by step parity)
            rc_query_body<true>(dm, P, tp, ar, t, tile, 0, 1, wquery);
            rc_gru_extras<true>(dm, tp, ar, t, tile);
        }
        if (k == 0 && may_stop && !alive) {                            // every conversation of the tile has ended: the other roles stop at their next wait
            if (threadIdx.x == 0) __hip_atomic_store(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            whole = false;
            break;
        }
    }
    if (k != 0) return;
    if (whole && !pf_wait<false>(cW, (uint32_t)(nrc * T), nullptr, tp.sync)) return;       // the last message's partial sums
    __syncthreads();
    rc_tail_body<true>(dm, tp, ar, tile, whole);
    if (threadIdx.x == 0) {
        __hip_atomic_store(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tp.rcflags[(size_t)tile * 64] = 0u;                             // the backward roles' hand-off counter of the tile (k_rc_bwd)
    }
}

// (scripts/isa_stats.py -D MMG_ROLE_DIAG: the roles of k_rc_persist as kernels of their own -- diag_kernels.h)

// ---------------------------------------------------------------------------------------------
// k_rc_bwd: the reverse-time loop of k_bwd_tile (same math and tape contract) with a tile's hidden units split over R/16
// co-resident roles.  A role keeps ITS 16 columns of W_hh (3R x 16 = 48 KB) as MFMA B fragments in registers and the carried
// dh of its units in a register per thread; per reverse step it gathers the tile's gate gradients dgh_{t+1} (16 x 3R, published
// by all roles: ONE counter hand-off per step), forms dgh_{t+1} W_hh for its columns, runs the GRU cell backward of its units
// (model.py:340) and publishes its slice of dgh_t.
// ---------------------------------------------------------------------------------------------
// The output-step prelude (dy, h*, A*, dA, dA W_y1h) runs HERE too, per 16-unit slice (k_bwd_tile is not launched): every role forms dy of the
// tile (role 0 stores it), ITS 16 columns of A* = W_y1[:, :R] h* and of dA[m][r] = w_y2[r] sum_d dy[m][d] 1[A*[m][r] + Cd[d][r] > 0]
// -- both local to the slice -- then the roles all-gather dA (hand-off 0) and each forms its columns of dA W_y1h (what enters dh at
// the sample's output step).  The tile's counter is zeroed by the forward launch (k_rc_persist / k_rc_tail).  prelude & 2: role 0 of
// tile 0 also lists the live (step, sample) rows for k_wgrad / k_send_bwd.
__global__ __launch_bounds__(256) void k_rc_bwd(Dims dm, Params P, Tape tp, const int64_t* __restrict__ target, int zero_dead, int pre_bands, int prelude) {
    __shared__ float s_acc[4][16][17];
    __shared__ __attribute__((aligned(16))) float s_dy[16][36];
    const int B = dm.B, R = dm.R, T = dm.T, NJ = R >> 4, R3 = 3 * R;
    const int tile = blockIdx.x / NJ, j = blockIdx.x - tile * NJ;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0), u0 = 16 * j;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    const int m = tid >> 4, c = tid & 15, b = min(b0 + m, B - 1), unit = u0 + c;
    const bool binary = dm.use_binary != 0;
    uint32_t* ctr = tp.rcflags + (size_t)tile * 64;
    int tmax = 0;
    for (int mm = 0; mm < nb; ++mm) tmax = max(tmax, tp.tstar[b0 + mm]);
    const int ts = (m < nb) ? tp.tstar[b] : -1;                         // padded rows: never live
    float dam;
    int step = 0;
    {
        const int D = dm.D, V = dm.V;
        if ((prelude & 2) && blockIdx.x == 0 && tid < 64) build_row_map(dm, tp);        // live (step, sample) rows for k_wgrad / k_send_bwd
        // ---- dNLL/d outp (model.py:1264-1275): thread (m, d = c), (m, c + 16)
        for (int d = c; d < 32; d += 16) {
            float dv = 0.f;
            if (d < D && m < nb) {
                dv = (tp.sm[(size_t)b * D + d] - (d == (int)target[b] ? 1.f : 0.f)) / (float)dm.Bg;
                if (j == 0) { tp.dy[(size_t)b * D + d] = dv; tp.dyT[(size_t)d * B + b] = dv; }
            }
            s_dy[m][d] = dv;
        }
        // ---- A* slice: h* rows (h after the sample's output step) x the role's 16 rows of W_y1[:, :R]
        {
            const int bi = min(b0 + i, B - 1), tsi = max(tp.tstar[bi], 0);
            int g0, n;
            rc_share(R, wave, g0, n);
            RcFrag ah, wa;
            rc_load(ah, tp.h + ((size_t)(tsi + 1) * B + bi) * R, R, g0, q);
            rc_load(wa, P.p[R_Y1_W] + (size_t)(u0 + i) * (R + V), R, g0, q);
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 a0 = rc_mma(ah, wa, n, z4);
#pragma unroll
            for (int r = 0; r < 4; ++r) s_acc[wave][q * 4 + r][i] = a0[r];
        }
        __syncthreads();
        const float astar = (s_acc[0][m][c] + s_acc[1][m][c]) + (s_acc[2][m][c] + s_acc[3][m][c]);
        if (j == 0 && c == 0 && m < nb) {
            float ds = 0.f;
            for (int d = 0; d < D; ++d) ds += s_dy[m][d];
            tp.dysum[b] = ds;
        }
        // ---- dA slice: class rows of Cd for this unit, all requested together (D <= 32)
        float cv[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) cv[d] = tp.Cd[(size_t)min(d, D - 1) * R + unit];
        float da = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) da += (d < D && astar + cv[d] > 0.f) ? s_dy[m][d] : 0.f;
        da *= P.p[R_Y2_W][unit];
        if (m < nb) {
            const int tsm = max(ts, 0);
            tp.hstar[(size_t)b * R + unit] = tp.h[((size_t)(tsm + 1) * B + b) * R + unit];
            tp.Astar[(size_t)b * R + unit] = astar;
            st_wt(&tp.dA[(size_t)b * R + unit], da);                    // (read back by the tile's other roles)
        }
        pf_signal(ctr); ++step;
        // ---- this role's columns of dA W_y1h ("NN": W_y1 [out, in] IS [K, N]); its column fragments go out before the wait
        int g0, n;
        rc_share(R, wave, g0, n);
        float4 wy[RC_MAXG];
#pragma unroll
        for (int u = 0; u < RC_MAXG; ++u) {
            const float* w = P.p[R_Y1_W] + (size_t)(min(g0 + u, (R >> 4) - 1) * 16 + q * 4) * (R + V) + u0 + i;
            wy[u] = make_float4(w[0], w[R + V], w[2 * (R + V)], w[3 * (R + V)]);
        }
        if (!pf_wait<false>(ctr, (uint32_t)(NJ * step), nullptr, tp.sync)) return;
        RcFrag ad;
        rc_load_act<true>(ad, tp.dA + (size_t)min(b0 + i, B - 1) * R, R, g0, q);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < RC_MAXG; ++u) {
            if (u < n) {
                acc = mfma16(ad.v[u].x, wy[u].x, acc); acc = mfma16(ad.v[u].y, wy[u].y, acc);
                acc = mfma16(ad.v[u].z, wy[u].z, acc); acc = mfma16(ad.v[u].w, wy[u].w, acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s_acc[wave][q * 4 + r][i] = acc[r];
        __syncthreads();
        dam = (s_acc[0][m][c] + s_acc[1][m][c]) + (s_acc[2][m][c] + s_acc[3][m][c]);
        __syncthreads();                                                // (s_acc is rewritten by the time loop)
    }
    // this wave's K share of the role's 16 columns of W_hh ("NN" form: the PyTorch [out, in] matrix IS [K, N])
    const int kg = R3 >> 4, per = (kg + 3) >> 2, g0 = wave * per, n = max(0, min(kg, g0 + per) - g0);
    float4 wf[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
        const float* w = P.p[R_WHH] + (size_t)(min(g0 + u, kg - 1) * 16 + q * 4) * R + u0 + i;
        wf[u] = make_float4(w[0], w[R], w[2 * R], w[3 * R]);
    }
    float carry = 0.f;
    bool have = false;
    for (int t = zero_dead ? T - 1 : tmax; t >= 0; --t) {
        const size_t rowb = (size_t)t * B;
        // this step's forward tape (GRU gates, h_{t-1}) and dhin: in flight while the hand-off is awaited
        const float* gr = tp.gru + (rowb + b) * 4 * R;
        const float rr = gr[unit], uu = gr[R + unit], nn = gr[2 * R + unit], ghn = gr[3 * R + unit];
        const float fh = tp.h[(rowb + b) * R + unit];
        float fin = 0.f;                                                // (k_bwd_pre's column bands: partial products, added in band order)
        if (binary) {
            float fp[4];
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) fp[pb] = tp.dhin[((size_t)min(pb, pre_bands - 1) * T * B + rowb + b) * R + unit];
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) fin += (pb < pre_bands) ? fp[pb] : 0.f;
        }
        float prod = 0.f;
        if (have) {
            if (!pf_wait<false>(ctr, (uint32_t)(NJ * step), nullptr, tp.sync)) return;
            // (fragment order [tile][parity][k-group][16 samples][16]: a wave reads a k-group's 1 KB contiguously)
            const float* xfrag = tp.rcx + (size_t)(tile * 2 + ((t + 1) & 1)) * 16 * R3 + i * 16 + q * 4;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            float4 av[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) av[u] = ld_cc4(xfrag + (size_t)min(g0 + u, kg - 1) * 256);
#pragma unroll
            for (int u = 0; u < 12; u += 2) {
                if (u < n) {
                    acc0 = mfma16(av[u].x, wf[u].x, acc0); acc0 = mfma16(av[u].y, wf[u].y, acc0);
                    acc0 = mfma16(av[u].z, wf[u].z, acc0); acc0 = mfma16(av[u].w, wf[u].w, acc0);
                }
                if (u + 1 < n) {
                    acc1 = mfma16(av[u + 1].x, wf[u + 1].x, acc1); acc1 = mfma16(av[u + 1].y, wf[u + 1].y, acc1);
                    acc1 = mfma16(av[u + 1].z, wf[u + 1].z, acc1); acc1 = mfma16(av[u + 1].w, wf[u + 1].w, acc1);
                }
            }
            const f32x4 acc = acc0 + acc1;
#pragma unroll
            for (int r = 0; r < 4; ++r) s_acc[wave][q * 4 + r][i] = acc[r];
            __syncthreads();
            prod = (s_acc[0][m][c] + s_acc[1][m][c]) + (s_acc[2][m][c] + s_acc[3][m][c]);
        }
        // dh_t = dh_{t+1} u_{t+1} + dgh_{t+1} W_hh + dhin_t (+ dA W_y1h at the output step); GRU cell backward (model.py:340)
        const bool live = t <= ts;
        float dh = carry + prod;
        dh += (t == ts) ? dam : 0.f;
        dh += (binary && live) ? fin : 0.f;                             // (rows of steps a sample never took hold no dhin)
        if (!live) dh = 0.f;
        const float dn = dh * (1.f - uu), du = dh * (fh - nn);
        const float dnp = dn * (1.f - nn * nn), dup = du * uu * (1.f - uu);
        const float drp = dnp * ghn * rr * (1.f - rr);
        const float g0v = live ? drp : 0.f, g1v = live ? dup : 0.f, g2v = live ? dnp * rr : 0.f;
        carry = live ? dh * uu : 0.f;
        if (t > 0) {                                                    // dh_{t-1} receives dgh_t W_hh: the slice goes out to the tile's roles
            // gate g of unit u0 + c sits in k-group g (R / 16) + j, column c
            float* xw = tp.rcx + (size_t)(tile * 2 + (t & 1)) * 16 * R3 + ((size_t)j * 16 + m) * 16 + c;
            st_wt(xw, g0v); st_wt(xw + (size_t)NJ * 256, g1v); st_wt(xw + (size_t)2 * NJ * 256, g2v);
        }
        if (m < nb && (live || zero_dead)) {
            float* gi = tp.dgi + (rowb + b) * R3; float* gh = tp.dgh + (rowb + b) * R3;
            gi[unit] = g0v; gi[R + unit] = g1v; gi[2 * R + unit] = live ? dnp : 0.f;
            gh[unit] = g0v; gh[R + unit] = g1v; gh[2 * R + unit] = g2v;
        }
        if (t > 0) { pf_signal(ctr); ++step; have = true; }             // (its barrier also frees s_acc)
    }
}

}  // namespace mmg
