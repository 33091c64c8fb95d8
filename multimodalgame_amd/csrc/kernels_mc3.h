// kernels_mc3.h -- k_conversation_mc3: the many-class conversation of kernels_mc.h for CONTINUOUS messages (-nouse_binary: BASELINE
// config 5), re-cut the way kernels_fast3.h re-cut the 30-class kernel:
//   * 256 threads, one wave per SIMD; the per-sample agent phases are k_conversation_fast3's (code_layer | binary_layer | GRU cell |
//     heads on h | message), phases written as [LDS loads] [branch-free arithmetic] [stores];
//   * the description mixture is folded onto the classes: W_d (softmax(y) . desc) = softmax(y) . Dd, Dd = desc W_d^T [D, R] (k_prep).
//     A member's slice of the mixture is then e_k [16, 64] x Dd_k [64, R = 64] -- ONE 16-column MFMA tile per wave instead of seven
//     over V = 100 -- the all-to-all payload shrinks from V + 2 to R + 2 floats per (slice, sample), and the sample's combine step
//     yields w_d dbar directly (no dbar phase, no w_d phase).  dbar itself is never formed: in continuous mode w_d, w_h and w get no
//     gradient (every message input is detached, model.py:810, 1297-1305);
//   * W_hh (and code_layer) parked in LDS as per-lane spill slots; the hidden-side GRU product of the NEXT step runs between the
//     publication of A and the poll for the tile's rows, i.e. inside the first hand-off's wait.
// Hand-offs: the payload itself, as (value, epoch) pairs -- one aligned 8-byte write-through store / agent-scope load each
// (device_utils.h: st_ll), epoch = (minibatch counter, step).  A consumer loads the pairs it needs and loads again while any carries
// another epoch: one trip through memory once the data has landed.  k_conversation_mc's protocol (write-through payload, wait for
// the stores to complete, one counter increment per member; poll the counter, then agent-scope loads of the payload) is three
// dependent trips, ~1.5 us more per hand-off, two hand-offs per step.  The buffers are reused every step: a member's step t + 1
// rows go out only after it has passed the second hand-off of step t, i.e. after every member has read the step-t rows.
// ONE counter hand-off remains, after the last step: a slice owner's write-through stores of the selected logits (tape.outp) must
// have completed before the sample's owner reads them.
// Same tape contract and sampling streams as k_conversation_mc; binary messages with many classes stay on k_conversation_mc (their
// w_d gradient needs dbar).
#pragma once
#include "device_utils.h"
#include "kernels_fast3.h"
#include "kernels_mc.h"
#include "layout.h"

namespace mmg {

#define MMG_MC3_LDP 68                              // floats per (slice, sample) row of the partial buffer: R mixture terms | m | s | pad

// The y head of one (sample row, class, r-quarter): sum_j w2[j] max(A[j], -Cd[j]) over the quarter's 16 hidden units, as two
// chains over the even / the odd units kept in ONE packed accumulator (v_pk_fma_f32: half the FMA issue slots of the scalar
// form; the same products in the same order, so the sum is bit for bit that of p0 + p1 of rounds 3-5).
__device__ __forceinline__ float yhead16(const float4& a0, const float4& a1, const float4& a2, const float4& a3,
                                         const float (&ncd)[16], const float (&w2e)[16]) {
    f32x2 acc = {0.f, 0.f};
#define MMG_YH(ax, ay, j) acc = __builtin_elementwise_fma(f32x2{w2e[j], w2e[j + 1]}, f32x2{fmax_nn(ax, ncd[j]), fmax_nn(ay, ncd[j + 1])}, acc)
    MMG_YH(a0.x, a0.y, 0); MMG_YH(a0.z, a0.w, 2); MMG_YH(a1.x, a1.y, 4); MMG_YH(a1.z, a1.w, 6);
    MMG_YH(a2.x, a2.y, 8); MMG_YH(a2.z, a2.w, 10); MMG_YH(a3.x, a3.y, 12); MMG_YH(a3.z, a3.w, 14);
#undef MMG_YH
    return acc.x + acc.y;
}
// ... for the 16 rows of a tile: rows are fetched one ahead of their use (LDS latency under the previous row's 30 VALU slots; at
// one wave per SIMD nothing else hides it), two rows in flight keep the register footprint at 32 floats
template <int TM, int LDA, int LDY>
__device__ __forceinline__ void yhead_tile(const float* s_At, float* s_y, int e4, int cls, float cyv, const float (&ncd)[16], const float (&w2e)[16]) {
    const float* ar_ = s_At + 16 * e4;
    float4 c0 = *reinterpret_cast<const float4*>(ar_), c1 = *reinterpret_cast<const float4*>(ar_ + 4);
    float4 c2 = *reinterpret_cast<const float4*>(ar_ + 8), c3 = *reinterpret_cast<const float4*>(ar_ + 12);
#pragma unroll 2
    for (int i = 0; i < TM; ++i) {
        const float* nx = s_At + min(i + 1, TM - 1) * LDA + 16 * e4;
        const float4 n0 = *reinterpret_cast<const float4*>(nx), n1 = *reinterpret_cast<const float4*>(nx + 4);
        const float4 n2 = *reinterpret_cast<const float4*>(nx + 8), n3 = *reinterpret_cast<const float4*>(nx + 12);
        const float tot = dpp_group_sum<4>(yhead16(c0, c1, c2, c3, ncd, w2e));
        if (e4 == 0) s_y[i * LDY + cls] = tot + cyv;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
}

struct Mc3Lds {
    static constexpr int R = 64, W = 32, H = 256, TM = 16, LDA = R + 4, LDY = 64 + 4;
    static constexpr int a = 0;                          // [H]
    static constexpr int z = a + H;                      // [W]
    static constexpr int w = z + W;                      // [W]    the message the sender reads next
    static constexpr int h = w + W;                      // [2][R] state before / after the step
    static constexpr int Aown = h + 2 * R;               // [LDA]
    static constexpr int gh = Aown + LDA;                // [R]    w_h h + b_h
    static constexpr int g = gh + R;                     // [R]
    static constexpr int At = g + R;                     // [TM][LDA]
    static constexpr int y = At + TM * LDA;              // [TM][LDY]
    static constexpr int e = y + TM * LDY;               // [TM][LDY]
    static constexpr int P = e + TM * LDY;               // [TM][LDP]
    static constexpr int in = P + TM * MMG_MC3_LDP;      // [TM][LDP]  the 16 slices' partials of this sample
    static constexpr int ms = in + TM * MMG_MC3_LDP;     // m[16] | s[16]
    static constexpr int small = ms + 32;                // us[16] | mask[17] | red[16] | pad
    static constexpr int park = small + 64;              // [20][256] float4: W_hh (12) | code_layer (8)
    // the sample's tape, staged per step and flushed after the conversation (a hand-off signal waits for the member's
    // outstanding stores: only the hand-off payload is outstanding then)
    static constexpr int TMAX = 16;
    static constexpr int t_gru = park + 20 * 256 * 4;    // [TMAX][4 R]
    static constexpr int t_h = t_gru + TMAX * 4 * R;     // [TMAX + 1][R]
    static constexpr int t_z = t_h + (TMAX + 1) * R;     // [TMAX][W]
    static constexpr int t_sp = t_z + TMAX * W;          // s[16] | ps[16]
    static constexpr int t_a = t_sp + 32;                // [TMAX][H]      (not lean)
    static constexpr int t_g = t_a + TMAX * H;           // [TMAX][R]      (not lean)
    static constexpr int t_w = t_g + TMAX * R;           // [TMAX + 1][W]  (not lean): slot 0 = first_rec, slot t + 1 = w_t
    static constexpr int total = t_w + (TMAX + 1) * W;
};
__host__ __device__ inline int mc3_lds_bytes() { return Mc3Lds::total * 4; }

template <int H, int W, int R, int V, int CAP>
__global__ __launch_bounds__(256, 1) void k_conversation_mc3(Dims dm, Params P, Tape tp, ConvArgs ar, int ntile, int xcd_map, int y_last_only) {
    constexpr int NT = 256, TM = 16, LDA = Mc3Lds::LDA, LDY = Mc3Lds::LDY, LDP = MMG_MC3_LDP;
    static_assert(H == 256 && W == 32 && R == 64 && CAP == 64, "shape of the register-resident small agents");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef Mc3Lds L;
    float* const s_a = lds + L::a; float* const s_z = lds + L::z; float* const s_w = lds + L::w; float* const s_h = lds + L::h;
    float* const s_Aown = lds + L::Aown; float* const s_gh = lds + L::gh; float* const s_g = lds + L::g; float* const s_At = lds + L::At;
    float* const s_y = lds + L::y; float* const s_e = lds + L::e; float* const s_P = lds + L::P; float* const s_in = lds + L::in;
    float* const s_m = lds + L::ms; float* const s_s = s_m + 16;
    float* const s_us = lds + L::small; float* const s_mask = s_us + 16; float* const s_red = s_us + 40;
    float4* const s_park = reinterpret_cast<float4*>(lds + L::park);
    float* const t_gru = lds + L::t_gru; float* const t_h = lds + L::t_h; float* const t_z = lds + L::t_z; float* const t_sp = lds + L::t_sp;
    float* const t_a = lds + L::t_a; float* const t_g = lds + L::t_g; float* const t_w = lds + L::t_w;

    int tile, member;
    if (xcd_map) {
        const int wg = blockIdx.x, blk = wg >> 7, x = wg & 7, slot = (wg & 127) >> 3;
        tile = blk * 8 + x; member = slot;
        if (tile >= ntile) return;
    } else { tile = blockIdx.x >> 4; member = blockIdx.x & 15; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = dm.B, T = dm.T, D = dm.D;
    const int b_raw = tile * TM + member;
    const bool have = b_raw < B;                        // a real sample (otherwise: class-slice owner only)
    const bool have_full = have && !ar.lean;            // (lean: the fused training step keeps what its backward reads: z, h, gates, output step)
    const int b = have ? b_raw : B - 1;
    const bool train = ar.train != 0, inject = ar.u_s != nullptr;
    const bool l2h = ar.l2_handoff != 0 && xcd_map != 0;     // the tile's 16 members share this XCD's L2: the pairs stay there (device_utils.h: st_ll_l2)
    const int per = ar.per, c0 = member * per;
    const uint32_t mb_counter = tp.counter[0];
    const uint32_t ll_base = tp.counter[3] * 32u;          // epochs of this launch's pair hand-offs: launch epoch, step
    const uint32_t gb = (uint32_t)(dm.boff + b);
    const int tgt = ar.target ? (int)ar.target[b] : -1;
    MMG_MSTAMP(0);
    if (tid < T) s_us[tid] = (train && inject) ? ar.u_s[(size_t)tid * B + b] : philox_uniform(ar.seed, (uint32_t)(tid * dm.Bg + gb), mb_counter, 1u);
    // ------------------------------------------------------------ agent weights -> registers / LDS park (kernels_fast3.h lane maps)
    float4 park_tmp[20];
#pragma unroll
    for (int j = 0; j < 8; ++j) park_tmp[12 + j] = *reinterpret_cast<const float4*>(P.p[S_CODE_W] + (size_t)tid * W + 4 * j);
    const float bc = P.p[S_CODE_B][tid], hw0 = tp.hw0[tid], hx = tp.hx[(size_t)b * H + tid];
    const int m2 = tid >> 3, k2 = tid & 7;
    float wb[32], ww[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(P.p[S_BIN_W] + (size_t)m2 * H + (j * 8 + k2) * 4);
        wb[4 * j] = v.x; wb[4 * j + 1] = v.y; wb[4 * j + 2] = v.z; wb[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(P.p[R_W_W] + (size_t)m2 * R + k2 * 8 + 4 * j);
        ww[4 * j] = v.x; ww[4 * j + 1] = v.y; ww[4 * j + 2] = v.z; ww[4 * j + 3] = v.w;
    }
    const float bb = P.p[S_BIN_B][m2], bw = P.p[R_W_B][m2];
    const int u3 = tid >> 2, q3 = tid & 3;
    float wih[24];
#pragma unroll
    for (int gt = 0; gt < 3; ++gt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(P.p[R_WIH] + (size_t)(gt * R + u3) * W + q3 * 8 + 4 * j);
            wih[8 * gt + 4 * j] = v.x; wih[8 * gt + 4 * j + 1] = v.y; wih[8 * gt + 4 * j + 2] = v.z; wih[8 * gt + 4 * j + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            park_tmp[gt * 4 + i] = *reinterpret_cast<const float4*>(P.p[R_WHH] + (size_t)(gt * R + u3) * R + q3 * 16 + 4 * i);
    }
    const float b_r = P.p[R_BIH][u3] + P.p[R_BHH][u3], b_u = P.p[R_BIH][R + u3] + P.p[R_BHH][R + u3];
    const float b_in = P.p[R_BIH][2 * R + u3], b_hn = P.p[R_BHH][2 * R + u3];
    const int row4 = tid >> 1, half4 = tid & 1;
    float w4[32];
    {
        const float* src = (row4 < R) ? P.p[R_Y1_W] + (size_t)row4 * (R + V) : P.p[R_WH_W] + (size_t)(row4 - R) * R;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(src + (j * 2 + half4) * 4);
            w4[4 * j] = v.x; w4[4 * j + 1] = v.y; w4[4 * j + 2] = v.z; w4[4 * j + 3] = v.w;
        }
    }
    const float b4 = (row4 < R) ? 0.f : P.p[R_WH_B][row4 - R];
    const float4 ws4 = *reinterpret_cast<const float4*>(P.p[R_S_W] + (tid & 15) * 4);
    const float bs = P.p[R_S_B][0];
    // ------------------------------------------------------------ this member's class slice -> registers
    // y head: class slot cls = tid / 4, r-quarter e4 = tid % 4:  y[b, d] = cy[d] + sum_r w2[r] max(A[b, r], -Cd[d, r])
    const int cls = tid >> 2, e4 = tid & 3;
    const bool cls_ok = cls < per && c0 + cls < D;
    float ncd[16], w2e[16];
    {
        const float* crow = tp.Cd + (size_t)min(c0 + cls, D - 1) * R + 16 * e4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 u = *reinterpret_cast<const float4*>(crow + 4 * j), q = *reinterpret_cast<const float4*>(P.p[R_Y2_W] + 16 * e4 + 4 * j);
            ncd[4 * j] = -u.x; ncd[4 * j + 1] = -u.y; ncd[4 * j + 2] = -u.z; ncd[4 * j + 3] = -u.w;
            w2e[4 * j] = q.x; w2e[4 * j + 1] = q.y; w2e[4 * j + 2] = q.z; w2e[4 * j + 3] = q.w;
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(ncd[j]));        // (keep -Cd itself in the registers: hipcc re-negates Cd inside the y-head loop otherwise)
    const float cyv = cls_ok ? tp.cy[min(c0 + cls, D - 1)] : -3.0e38f;
    // mixture: wave w owns the columns 16 w .. 16 w + 15 of [16 samples, CAP classes] x Dd_k [CAP, R]; B fragment of k-step ks:
    // lane (fi = lane & 15, fq = lane >> 4) holds Dd[c0 + 4 ks + fq][16 w + fi]
    const int fi = lane & 15, fq = lane >> 4;
    float bfrag[CAP / 4];
#pragma unroll
    for (int ks = 0; ks < CAP / 4; ++ks) {
        const int c = 4 * ks + fq;
        const float dv = tp.Dd[(size_t)min(c0 + c, D - 1) * R + 16 * wave + fi];
        bfrag[ks] = (c < per && c0 + c < D) ? dv : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 20; ++i) s_park[i * NT + tid] = park_tmp[i];
    // ------------------------------------------------------------ conversation state
    if (tid < R) { s_h[tid] = 0.f; t_h[tid] = 0.f; }
    if (tid < W) { s_w[tid] = dm.first_rec; t_w[tid] = dm.first_rec; }
    if (tid == 0) s_mask[0] = 1.f;
    float m_run = 1.f, sprod = 1.f; int t_out = -1;        // (meaningful on lane 240: the stop head's lane)
    float ghp_r = 0.f, ghp_u = 0.f, ghn = b_hn;
    const float fixedm = dm.fixed ? 1.f : 0.f;
    const bool sprodm = dm.s_prob_prod != 0;
    __syncthreads();
    MMG_MSTAMP(2);
    uint32_t* cA = mc_ctr(tp, 0, tile, ntile);
    float* llA = tp.mc3A + (size_t)tile * TM * LDA * 2;                              // pairs [16 members][LDA]
    float* llP = tp.mc3P + ((size_t)tile * TM * TM) * LDP * 2;                       // pairs [16 members][16 samples][LDP]
    const size_t p_mine = (size_t)member * TM * LDP;                                 // ... this member's [16 samples][LDP]
    auto spin_expired = [&](int& spins) {
        if (++spins <= (1 << 16)) return false;
        if (lane == 0) __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    };
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * B + b;
        float* const hcur = s_h + (t & 1) * R;
        float* const hn = s_h + ((t + 1) & 1) * R;
        // ===== P1 sender: a = tanh(h_x + code_layer(c))
        {
            float hw = hw0;
            if (t > 0) {
                f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 wv = s_park[(12 + j) * NT + tid];
                    const float4 cv = *reinterpret_cast<const float4*>(s_w + 4 * j);
                    a01 = __builtin_elementwise_fma(f32x2{wv.x, wv.y}, f32x2{cv.x, cv.y}, a01);
                    a23 = __builtin_elementwise_fma(f32x2{wv.z, wv.w}, f32x2{cv.z, cv.w}, a23);
                }
                const f32x2 sm = a01 + a23;
                hw = bc + sm.x + sm.y;
            }
            const float av = ftanh(hx + hw);
            s_a[tid] = av;
            if (have_full) t_a[t * H + tid] = av;
        }
        __syncthreads();
        // ===== P2 sender message (continuous: the logits themselves)
        {
            const float lz = dpp_group_sum<8>(dot4p<8>(wb, s_a + k2 * 4, 32)) + bb;
            if (k2 == 0) { s_z[m2] = lz; t_z[t * W + m2] = lz; }
        }
        __syncthreads();
        // ===== P3 GRU cell
        {
            const float* zq = s_z + q3 * 8;
            const float4 z0 = *reinterpret_cast<const float4*>(zq), z1 = *reinterpret_cast<const float4*>(zq + 4);
            const float h_old = hcur[u3];
            auto gate = [&](const float* wg) {
                const f32x2 a = __builtin_elementwise_fma(f32x2{wg[0], wg[1]}, f32x2{z0.x, z0.y}, f32x2{wg[4], wg[5]} * f32x2{z1.x, z1.y});
                const f32x2 c = __builtin_elementwise_fma(f32x2{wg[2], wg[3]}, f32x2{z0.z, z0.w}, f32x2{wg[6], wg[7]} * f32x2{z1.z, z1.w});
                const f32x2 sm = a + c;
                return sm.x + sm.y;
            };
            const float xr = dpp_group_sum<4>(gate(wih) + ghp_r) + b_r;
            const float xu = dpp_group_sum<4>(gate(wih + 8) + ghp_u) + b_u;
            const float gin = dpp_group_sum<4>(gate(wih + 16)) + b_in;
            const float rr = fsigmoid(xr), uu = fsigmoid(xu);
            const float nn = ftanh(gin + rr * ghn);
            const float hv = nn + uu * (h_old - nn);
            t_gru[t * 4 * R + q3 * R + u3] = (q3 == 0) ? rr : (q3 == 1) ? uu : (q3 == 2) ? nn : ghn;
            if (q3 == 0) { hn[u3] = hv; t_h[(t + 1) * R + u3] = hv; }
        }
        __syncthreads();
        // ===== P4 heads on h: A = y1[:, :R] h, w_h h + b_h; stop head (lane 240 keeps it), masks / output step -> the take flag
        {
            const float4 hv4 = *reinterpret_cast<const float4*>(hn + (tid & 15) * 4);
            const float us_t = s_us[t];
            const float acc = dpp_group_sum<2>(dot4p<8>(w4, hn + half4 * 4, 8)) + b4;
            const float sv = dpp_group_sum<16>(fmaf(ws4.x, hv4.x, fmaf(ws4.y, hv4.y, fmaf(ws4.z, hv4.z, ws4.w * hv4.w))));
            const float p = fsigmoid(sv + bs);
            const float prod = sprodm ? sprod * p : p;
            sprod = train ? sprod : prod;
            const float sbit = train ? ((us_t < p) ? 1.f : 0.f) : rintf(prod);
            const float m_next = fminf(m_run, sbit);
            const bool last = (t == T - 1);
            const bool take = (fixedm != 0.f) ? last : (t_out < 0 && (m_next == 0.f || last));
            t_out = take ? t : t_out;
            m_run = m_next;
            if (half4 == 0) { if (row4 < R) s_Aown[row4] = acc; else s_gh[row4 - R] = acc; }
            if (tid == 240) {
                s_mask[t + 1] = m_next;
                s_Aown[R] = (take && have) ? 1.f : 0.f; s_Aown[R + 1] = 0.f; s_Aown[R + 2] = 0.f; s_Aown[R + 3] = 0.f;
                t_sp[t] = sbit; t_sp[16 + t] = p;
            }
        }
        __syncthreads();
        MMG_MSTAMP(16 + 8 * t + 0);
        if (tid < R && s_Aown[R] != 0.f) {                                 // output step of this sample: what the backward pass starts from
            tp.Astar[(size_t)b * R + tid] = s_Aown[tid];
            tp.hstar[(size_t)b * R + tid] = hn[tid];
        }
        // ----- hand-off 1: publish A (17 x 16 bytes); the hidden-side GRU product of the next step fills the wait
        const uint32_t ep = ll_base + (uint32_t)t + 1u;
        if (tid < LDA) { if (l2h) st_ll_l2(llA, (size_t)member * LDA + tid, s_Aown[tid], ep); else st_ll(llA, (size_t)member * LDA + tid, s_Aown[tid], ep); }
        {
            float4 pk[4], hq[4];
            park_load(pk, hq, s_park, 0, tid, hn + q3 * 16); ghp_r = park_fma(pk, hq);
            park_load(pk, hq, s_park, 1, tid, hn + q3 * 16); ghp_u = park_fma(pk, hq);
            park_load(pk, hq, s_park, 2, tid, hn + q3 * 16); ghn = dpp_group_sum<4>(park_fma(pk, hq)) + b_hn;
        }
        {
            constexpr int NA = (TM * LDA + NT - 1) / NT;                     // pairs of the tile's 16 rows per thread
            unsigned long long ua[NA];
            for (int spins = 0;; ) {
                bool fresh = true;
#pragma unroll
                for (int r = 0; r < NA; ++r) { ua[r] = ld_ll(llA, min(tid + NT * r, TM * LDA - 1)); fresh = fresh && ll_fresh(ua[r], ep); }
                if (!__any(!fresh) || spin_expired(spins)) break;
            }
            MMG_MSTAMP(16 + 8 * t + 1);
#pragma unroll
            for (int r = 0; r < NA; ++r) if (tid + NT * r < TM * LDA) s_At[tid + NT * r] = ll_value(ua[r]);
        }
        __syncthreads();
        MMG_MSTAMP(16 + 8 * t + 2);
        // ===== C1 slice logits for the 16 samples of the tile
        yhead_tile<TM, LDA, LDY>(s_At, s_y, e4, cls, cyv, ncd, w2e);
        __syncthreads();
        MMG_MSTAMP(16 + 8 * t + 7);
        // logits -> tape (every step: exchange() returns them; y_last_only: the output step's) and the selected rows -> outp;
        // slice softmax numerators: 16 lanes per sample, 4 classes per lane
        {
            const bool keep_y = !y_last_only || t == T - 1;
#pragma unroll
            for (int u = 0; u < TM * CAP / NT; ++u) {
                const int idx = tid + NT * u, i = idx / CAP, c = idx % CAP;
                const int bi = tile * TM + i;
                const float yv = s_y[i * LDY + c];
                if (bi < B && c < per && c0 + c < D) {
                    if (keep_y) tp.y[((size_t)t * B + bi) * D + c0 + c] = yv;
                    if (s_At[i * LDA + R] != 0.f) st_wt(&tp.outp[(size_t)bi * D + c0 + c], yv);     // model.py:1261-1264
                }
            }
            const int i = tid >> 4, l = tid & 15;
            float yv[CAP / 16];
            float m = -3.0e38f;
#pragma unroll
            for (int j = 0; j < CAP / 16; ++j) { yv[j] = s_y[i * LDY + l + 16 * j]; m = fmax_nn(m, yv[j]); }
            m = fmax_nn(m, dpp_f<MMG_DPP_QUAD_1032>(m)); m = fmax_nn(m, dpp_f<MMG_DPP_QUAD_2301>(m));
            m = fmax_nn(m, dpp_f<MMG_DPP_ROW_HALF_MIRROR>(m)); m = fmax_nn(m, dpp_f<MMG_DPP_ROW_MIRROR>(m));
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < CAP / 16; ++j) { const float e = __expf(yv[j] - m); s += e; s_e[i * LDY + l + 16 * j] = e; }
            s = dpp_group_sum<16>(s);
            if (l == 0) { s_P[i * LDP + R] = m; s_P[i * LDP + R + 1] = s; s_P[i * LDP + R + 2] = 0.f; s_P[i * LDP + R + 3] = 0.f; }
        }
        __syncthreads();
        MMG_MSTAMP(200 + t);
        // ===== C3 unnormalised mixture of the slice on the matrix cores: [16, CAP] x Dd_k [CAP, 16 columns of this wave]
        {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < CAP / 4; ks += 2) {
                acc0 = mfma16(s_e[fi * LDY + 4 * ks + fq], bfrag[ks], acc0);
                acc1 = mfma16(s_e[fi * LDY + 4 * (ks + 1) + fq], bfrag[ks + 1], acc1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s_P[(4 * fq + r) * LDP + 16 * wave + fi] = acc0[r] + acc1[r];
        }
        __syncthreads();
        MMG_MSTAMP(16 + 8 * t + 3);
        // ----- hand-off 2: this slice's partials out (16 x 17 x 16 bytes), the 16 slices of this sample in
        {
            constexpr int NP = (TM * LDP + NT - 1) / NT;                     // pairs per thread, out and in
#pragma unroll
            for (int r = 0; r < NP; ++r) if (tid + NT * r < TM * LDP) { if (l2h) st_ll_l2(llP, p_mine + tid + NT * r, s_P[tid + NT * r], ep); else st_ll(llP, p_mine + tid + NT * r, s_P[tid + NT * r], ep); }
            unsigned long long up[NP];
            int kq[NP];
#pragma unroll
            for (int r = 0; r < NP; ++r) { const int i = min(tid + NT * r, TM * LDP - 1); kq[r] = ((i / LDP) * TM + member) * LDP + i % LDP; }
            for (int spins = 0;; ) {
                bool fresh = true;
#pragma unroll
                for (int r = 0; r < NP; ++r) { up[r] = ld_ll(llP, (size_t)kq[r]); fresh = fresh && ll_fresh(up[r], ep); }
                if (!__any(!fresh) || spin_expired(spins)) break;
            }
            MMG_MSTAMP(16 + 8 * t + 4);
#pragma unroll
            for (int r = 0; r < NP; ++r) {
                const int i = tid + NT * r;
                if (i < TM * LDP) {
                    const int k = i / LDP, q = i % LDP;
                    const float v = ll_value(up[r]);
                    s_in[i] = v;
                    if (q == R) s_m[k] = v;
                    if (q == R + 1) s_s[k] = v;
                }
            }
        }
        __syncthreads();
        MMG_MSTAMP(16 + 8 * t + 5);
        // ===== K1 combine the slices (streaming softmax) straight into g = tanh(w_h h + b_h + sum_k P_k e^(m_k - M) / sum_k s_k e^(m_k - M))
        {
            const float4 m0 = *reinterpret_cast<const float4*>(s_m), m1 = *reinterpret_cast<const float4*>(s_m + 4);
            const float4 m2v = *reinterpret_cast<const float4*>(s_m + 8), m3 = *reinterpret_cast<const float4*>(s_m + 12);
            const float4 mq = *reinterpret_cast<const float4*>(s_m + 4 * q3), sq = *reinterpret_cast<const float4*>(s_s + 4 * q3);
            const float p0 = s_in[(4 * q3 + 0) * LDP + u3], p1 = s_in[(4 * q3 + 1) * LDP + u3];
            const float p2 = s_in[(4 * q3 + 2) * LDP + u3], p3 = s_in[(4 * q3 + 3) * LDP + u3];
            const float ghu = s_gh[u3];
            const float Ma = fmax_nn(fmax_nn(m0.x, m0.y), fmax_nn(m0.z, m0.w)), Mb = fmax_nn(fmax_nn(m1.x, m1.y), fmax_nn(m1.z, m1.w));
            const float Mc = fmax_nn(fmax_nn(m2v.x, m2v.y), fmax_nn(m2v.z, m2v.w)), Md = fmax_nn(fmax_nn(m3.x, m3.y), fmax_nn(m3.z, m3.w));
            const float M = fmax_nn(fmax_nn(Ma, Mb), fmax_nn(Mc, Md));
            const float e0 = __expf(mq.x - M), e1 = __expf(mq.y - M), e2 = __expf(mq.z - M), e3 = __expf(mq.w - M);
            const float Sp = dpp_group_sum<4>(fmaf(sq.x, e0, fmaf(sq.y, e1, fmaf(sq.z, e2, sq.w * e3))));
            const float Ap = dpp_group_sum<4>(fmaf(p0, e0, fmaf(p1, e1, fmaf(p2, e2, p3 * e3))));
            const float gv = ftanh(fmaf(Ap, __builtin_amdgcn_rcpf(Sp), ghu));
            if (q3 == 0) { s_g[u3] = gv; if (have_full) t_g[t * R + u3] = gv; }
        }
        __syncthreads();
        // ===== P7 receiver message (continuous: the logits)
        {
            const float lw = dpp_group_sum<8>(dot4p<2>(ww, s_g + k2 * 8, 4)) + bw;
            if (k2 == 0) { s_w[m2] = lw; if (have_full) t_w[(t + 1) * W + m2] = lw; }
        }
        __syncthreads();
        MMG_MSTAMP(16 + 8 * t + 6);
    }
    MMG_MSTAMP(3);
    // the one counter hand-off: every slice owner's selected-logit stores (tape.outp, write-through) have completed
    pf_signal(cA);
    if (!have) return;
    // ------------------------------------------------------------ the sample's tape, coalesced
    for (int i4 = tid; i4 < T * (4 * R / 4); i4 += NT) {
        const int t = i4 >> 6, c = i4 & 63;
        *reinterpret_cast<float4*>(tp.gru + ((size_t)t * B + b) * 4 * R + 4 * c) = *reinterpret_cast<const float4*>(t_gru + 4 * i4);
        if (have_full) *reinterpret_cast<float4*>(tp.a + ((size_t)t * B + b) * H + 4 * c) = *reinterpret_cast<const float4*>(t_a + 4 * i4);
    }
    for (int i4 = tid; i4 < (T + 1) * (R / 4); i4 += NT) {
        const int t = i4 >> 4, c = i4 & 15;
        *reinterpret_cast<float4*>(tp.h + ((size_t)t * B + b) * R + 4 * c) = *reinterpret_cast<const float4*>(t_h + 4 * i4);
        if (have_full && t < T) *reinterpret_cast<float4*>(tp.g + ((size_t)t * B + b) * R + 4 * c) = *reinterpret_cast<const float4*>(t_g + 4 * i4);
    }
    for (int i4 = tid; i4 < T * (W / 4); i4 += NT) {
        const int t = i4 >> 3, c = i4 & 7;
        const size_t o = ((size_t)t * B + b) * W + 4 * c;
        *reinterpret_cast<float4*>(tp.z + o) = *reinterpret_cast<const float4*>(t_z + 4 * i4);
        if (have_full) {
            const float4 cv = *reinterpret_cast<const float4*>(t_w + 4 * i4);              // slot t: what the sender read at step t
            *reinterpret_cast<float4*>(tp.zr + o) = cv;
            float4 c0v = cv;
            if (t == 0) { const float* sg = P.p[S_CODE_BIAS] + 4 * c; c0v = make_float4(fsigmoid(sg[0]), fsigmoid(sg[1]), fsigmoid(sg[2]), fsigmoid(sg[3])); }
            *reinterpret_cast<float4*>(tp.c + o) = c0v;                                      // model.py:199
            *reinterpret_cast<float4*>(tp.w + o) = *reinterpret_cast<const float4*>(t_w + W + 4 * i4);
        }
    }
    if (tid < T) {
        const float sbit = t_sp[tid], p = t_sp[16 + tid];
        const size_t rw = (size_t)tid * B + b;
        const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
        tp.s[rw] = sbit; tp.ps[rw] = p;
        tp.lp_s[rw] = sbit * l1 + (1.f - sbit) * l0; tp.ne_s[rw] = p * l1 + (1.f - p) * l0;
    }
    if (tid <= T) tp.mask[(size_t)tid * B + b] = (uint8_t)(s_mask[tid] != 0.f);
    // ------------------------------------------------------------ output selection / reward / top-k (model.py:1264-1275, 1333-1339)
    // every slice owner stored this sample's selected logits (write-through) before its partial hand-off of that step
    if (tid == 240) { s_red[8] = (float)t_out; s_red[9] = sprod; }
    mc_wait(cA, (uint32_t)TM, tp.sync);                 // (the tape flush above went out in its shadow)
    const int tstar = dm.fixed ? (T - 1) : (int)s_red[8];
    constexpr int NY = 4;                               // classes per thread: D <= 16 * CAP = NT * NY
    float o[NY];
    float mx = -3.0e38f;
#pragma unroll
    for (int u = 0; u < NY; ++u) {
        const int d = tid + NT * u;
        o[u] = (d < D) ? __hip_atomic_load(&tp.outp[(size_t)b * D + min(d, D - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -3.0e38f;
        mx = fmaxf(mx, o[u]);
    }
    mx = block_max(mx, s_red);
    float se = 0.f;
#pragma unroll
    for (int u = 0; u < NY; ++u) se += (tid + NT * u < D) ? __expf(o[u] - mx) : 0.f;
    se = block_sum(se, s_red);
    const float lse = mx + flog(se);
    float dtv = 0.f;
#pragma unroll
    for (int u = 0; u < NY; ++u) if (tid + NT * u == tgt) dtv = o[u] - lse;
    const float dt = block_sum(dtv, s_red);            // (exactly one thread holds the target's log-probability)
    float above = 0.f;
#pragma unroll
    for (int u = 0; u < NY; ++u) {
        const int d = tid + NT * u;
        if (d < D) {
            const float ld = o[u] - lse;
            tp.dist[(size_t)b * D + d] = ld;
            tp.sm[(size_t)b * D + d] = __expf(ld);
            above += (tgt >= 0 && ld > dt) ? 1.f : 0.f;
        }
    }
    above = block_sum(above, s_red);
    if (tid == 0) {
        tp.tstar[b] = tstar;
        tp.sprod[b] = s_red[9];
        tp.logs[b] = (tgt >= 0) ? dt : 0.f;
        tp.hit[b] = (tgt >= 0 && above < (float)dm.top_k) ? 1 : 0;
    }
}

}  // namespace mmg
