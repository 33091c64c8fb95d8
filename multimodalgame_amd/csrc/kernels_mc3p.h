// kernels_mc3p.h -- k_conversation_mc3p: k_conversation_mc3 (kernels_mc3.h: many classes, continuous messages, BASELINE config 5)
// for batches of several rounds of workgroups (B >= 512): ONE workgroup runs member m of TWO sample tiles on one copy of the agent
// weights and of its class slice, the two conversations software-pipelined half a step apart.
//
// Why.  k_conversation_mc3 holds one workgroup per CU (369 registers per lane, 151 KB of LDS: the agents' 188 KB of weights live in
// registers + an LDS park), so 2 048 samples are 8 rounds of 256 workgroups, and a step of a workgroup is a SERIAL chain
//   sample phases 1.4 us | publish A, wait for the tile's 16 rows 1.8 | class slice: logits, softmax, mixture 3.6 | publish the
//   slice's partials, wait for the 16 slices of this sample 2.0 | combine 0.4            (profiles/r05_mc_timeline_c5.log: 9.2 us)
// of which 3.8 us are hand-off latency with nothing to do.  Two workgroups per CU do not fit; two samples per workgroup do: the
// weights, the class slice (-Cd, w2, the Dd fragments) and the LDS park are shared, only the per-sample state, the tile buffers and
// the tape staging exist twice (2 x 35 KB at T = 10 beside the 80 KB park).  The two tiles run the SAME phase code, skewed:
//   S(0,t) pubA | pollP K(1,t-1) | S(1,t) pubA | pollA C(0,t) pubP | pollA C(1,t) pubP | pollP K(0,t) | S(0,t+1) pubA | ...
// every poll comes one or two compute blocks (1.4 - 3.6 us) after its publication: the hand-offs cost their LDS gather only.
// A pair of steps takes ~12.5 us instead of 2 x 9.2 and the batch needs half the rounds (measured: profiles/r06_*config5*).
// The launch is PERSISTENT: 256 workgroups (16 pairs x 16 members) walk the batch's pairs, so the 188 KB of weights and the class
// slice are fetched once per workgroup, not once per round, and a pair's 16 members stay in step from pair to pair.
// Same lane maps, arithmetic, tape contract and sampling streams as k_conversation_mc3 (bit-identical results: tests/test_hip_configs.py);
// lean tape only (the fused training step); the host selects it for B >= 512, T <= 13.
#pragma once
#include "kernels_mc3.h"

namespace mmg {

// LDS plan (floats): the park, then two slots
struct Mc3pLds {
    static constexpr int R = 64, W = 32, H = 256, TM = 16, LDA = R + 4, LDY = 64 + 4, LDP = MMG_MC3_LDP;
    static constexpr int park = 0;                          // [20][256] float4: W_hh (12) | code_layer (8)
    static constexpr int slots = 20 * 256 * 4;
    // inside a slot
    static constexpr int a = 0;                             // [H]
    static constexpr int z = a + H;                         // [W]
    static constexpr int w = z + W;                         // [W]
    static constexpr int h = w + W;                         // [2][R]
    static constexpr int Aown = h + 2 * R;                  // [LDA]
    static constexpr int gh = Aown + LDA;                   // [R]
    static constexpr int g = gh + R;                        // [R]
    static constexpr int At = g + R;                        // [TM][LDA]
    static constexpr int y = At + TM * LDA;                 // [TM][LDY]   (dead after the softmax phase: `in` lives here afterwards)
    static constexpr int e = y + TM * LDY;                  // [TM][LDY]
    static constexpr int P = e + TM * LDY;                  // [TM][LDP]
    static constexpr int ms = P + TM * LDP;                 // m[16] | s[16]
    static constexpr int small = ms + 32;                   // us[16] | mask[17] | red[16] | pad
    static constexpr int tape = small + 64;                 // t_gru [T][4 R] | t_h [T + 1][R] | t_z [T][W] | t_sp [32]
    __host__ __device__ static constexpr int slot_floats(int T) { return tape + T * 4 * R + (T + 1) * R + T * W + 32; }
    __host__ __device__ static constexpr int total(int T) { return slots + 2 * slot_floats(T); }
};
__host__ __device__ inline int mc3p_lds_bytes(int T) { return Mc3pLds::total(T) * 4; }
__host__ __device__ inline bool mc3p_shape(int B, int T, int D) { return B >= 512 && (B & 15) == 0 && T <= 13 && mc3p_lds_bytes(T) <= 160 * 1024 && D > 32; }

template <int H, int W, int R, int V, int CAP>
__global__ __launch_bounds__(256, 1) void k_conversation_mc3p(Dims dm, Params P, Tape tp, ConvArgs ar, int ntile, int y_last_only) {
    constexpr int NT = 256, TM = 16, LDA = Mc3pLds::LDA, LDY = Mc3pLds::LDY, LDP = MMG_MC3_LDP;
    static_assert(H == 256 && W == 32 && R == 64 && CAP == 64, "shape of the register-resident small agents");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef Mc3pLds L;
    float4* const s_park = reinterpret_cast<float4*>(lds + L::park);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = dm.B, T = dm.T, D = dm.D;
    const int slotf = L::slot_floats(T);
    // workgroup -> (pair of tiles, member): the 16 members of a pair are the workgroups i = x (mod 8) of a block of 128 (one XCD)
    const int npair = (ntile + 1) >> 1;
    const int wg = blockIdx.x, blk = wg >> 7, x = wg & 7, member = (wg & 127) >> 3;
    const int pair0 = blk * 8 + x, pair_stride = (int)(gridDim.x >> 7) * 8;      // PERSISTENT: this workgroup walks pairs pair0, pair0 + stride, ...
    if (pair0 >= npair) return;
    const bool train = ar.train != 0, inject = ar.u_s != nullptr;
    const bool l2h = ar.l2_handoff != 0;                     // a pair's 16 members share this XCD's L2: the pairs stay there (device_utils.h: st_ll_l2)
    const int per = ar.per, c0 = member * per;
    const uint32_t mb_counter = tp.counter[0];
    const uint32_t ll_base = tp.counter[3] * 32u;
    // ------------------------------------------------------------ per-slot state
    struct Slot {
        float* base; float* t_gru; float* t_h; float* t_z; float* t_sp;
        int tile, b; bool on;
        float hx; int tgt;
        float m_run, sprod; int t_out;
        float ghp_r, ghp_u, ghn;
        uint32_t* cA; float* llA; float* llP;
    };
    Slot sl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        Slot& S = sl[s];
        S.base = lds + L::slots + s * slotf;
        S.t_gru = S.base + L::tape; S.t_h = S.t_gru + T * 4 * R; S.t_z = S.t_h + (T + 1) * R; S.t_sp = S.t_z + T * W;
    }
    // a pair's samples: what changes from pair to pair (the weights, the class slice and the park stay)
    auto slot_begin = [&](Slot& S, int pair, int s) __attribute__((always_inline)) {
        S.tile = 2 * pair + s;
        S.on = S.tile < ntile;
        const int tl = S.on ? S.tile : 2 * pair;
        S.b = min(tl * TM + member, B - 1);                  // (B % 16 == 0: every member of a live tile is a real sample)
        S.hx = tp.hx[(size_t)S.b * H + tid];
        S.tgt = ar.target ? (int)ar.target[S.b] : -1;
        S.m_run = 1.f; S.sprod = 1.f; S.t_out = -1;
        S.ghp_r = 0.f; S.ghp_u = 0.f;
        S.cA = mc_ctr(tp, 0, tl, ntile);
        S.llA = tp.mc3A + (size_t)tl * TM * LDA * 2;
        S.llP = tp.mc3P + ((size_t)tl * TM * TM) * LDP * 2;
        const uint32_t gb = (uint32_t)(dm.boff + S.b);
        float* s_us = S.base + L::small;
        if (tid < T) s_us[tid] = (train && inject) ? ar.u_s[(size_t)tid * B + S.b] : philox_uniform(ar.seed, (uint32_t)(tid * dm.Bg + gb), mb_counter, 1u);
    };
    slot_begin(sl[0], pair0, 0); slot_begin(sl[1], pair0, 1);
    // ------------------------------------------------------------ agent weights -> registers / LDS park (kernels_mc3.h lane maps)
    float4 park_tmp[20];
#pragma unroll
    for (int j = 0; j < 8; ++j) park_tmp[12 + j] = *reinterpret_cast<const float4*>(P.p[S_CODE_W] + (size_t)tid * W + 4 * j);
    const float bc = P.p[S_CODE_B][tid], hw0 = tp.hw0[tid];
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
    // ------------------------------------------------------------ this member's class slice -> registers (shared by both tiles)
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
    auto slot_state = [&](Slot& S) __attribute__((always_inline)) {
        S.ghn = b_hn;
        float* s_h = S.base + L::h; float* s_w = S.base + L::w; float* s_mask = S.base + L::small + 16;
        if (tid < R) { s_h[tid] = 0.f; S.t_h[tid] = 0.f; }
        if (tid < W) s_w[tid] = dm.first_rec;
        if (tid == 0) s_mask[0] = 1.f;
    };
    slot_state(sl[0]); slot_state(sl[1]);
    const float fixedm = dm.fixed ? 1.f : 0.f;
    const bool sprodm = dm.s_prob_prod != 0;
    __syncthreads();
    auto spin_expired = [&](int& spins) {
        if (++spins <= (1 << 16)) return false;
        if (lane == 0) __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    };
    // ============================================================ the three blocks of a step, for one slot
    // ---- S: the sample's own phases (sender | GRU | heads on h), publication of A, the next step's hidden-side GRU product
    auto blockS = [&](Slot& S, int t) __attribute__((always_inline)) {
        float* const s_a = S.base + L::a; float* const s_z = S.base + L::z; float* const s_w = S.base + L::w; float* const s_h = S.base + L::h;
        float* const s_Aown = S.base + L::Aown; float* const s_gh = S.base + L::gh;
        float* const s_us = S.base + L::small; float* const s_mask = s_us + 16;
        float* const hcur = s_h + (t & 1) * R;
        float* const hn = s_h + ((t + 1) & 1) * R;
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
            s_a[tid] = ftanh(S.hx + hw);
        }
        __syncthreads();
        {
            const float lz = dpp_group_sum<8>(dot4p<8>(wb, s_a + k2 * 4, 32)) + bb;
            if (k2 == 0) { s_z[m2] = lz; S.t_z[t * W + m2] = lz; }
        }
        __syncthreads();
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
            const float xr = dpp_group_sum<4>(gate(wih) + S.ghp_r) + b_r;
            const float xu = dpp_group_sum<4>(gate(wih + 8) + S.ghp_u) + b_u;
            const float gin = dpp_group_sum<4>(gate(wih + 16)) + b_in;
            const float rr = fsigmoid(xr), uu = fsigmoid(xu);
            const float nn = ftanh(gin + rr * S.ghn);
            const float hv = nn + uu * (h_old - nn);
            S.t_gru[t * 4 * R + q3 * R + u3] = (q3 == 0) ? rr : (q3 == 1) ? uu : (q3 == 2) ? nn : S.ghn;
            if (q3 == 0) { hn[u3] = hv; S.t_h[(t + 1) * R + u3] = hv; }
        }
        __syncthreads();
        {
            const float4 hv4 = *reinterpret_cast<const float4*>(hn + (tid & 15) * 4);
            const float us_t = s_us[t];
            const float acc = dpp_group_sum<2>(dot4p<8>(w4, hn + half4 * 4, 8)) + b4;
            const float sv = dpp_group_sum<16>(fmaf(ws4.x, hv4.x, fmaf(ws4.y, hv4.y, fmaf(ws4.z, hv4.z, ws4.w * hv4.w))));
            const float p = fsigmoid(sv + bs);
            const float prod = sprodm ? S.sprod * p : p;
            S.sprod = train ? S.sprod : prod;
            const float sbit = train ? ((us_t < p) ? 1.f : 0.f) : rintf(prod);
            const float m_next = fminf(S.m_run, sbit);
            const bool last = (t == T - 1);
            const bool take = (fixedm != 0.f) ? last : (S.t_out < 0 && (m_next == 0.f || last));
            S.t_out = take ? t : S.t_out;
            S.m_run = m_next;
            if (half4 == 0) { if (row4 < R) s_Aown[row4] = acc; else s_gh[row4 - R] = acc; }
            if (tid == 240) {
                s_mask[t + 1] = m_next;
                s_Aown[R] = take ? 1.f : 0.f; s_Aown[R + 1] = 0.f; s_Aown[R + 2] = 0.f; s_Aown[R + 3] = 0.f;
                S.t_sp[t] = sbit; S.t_sp[16 + t] = p;
            }
        }
        __syncthreads();
        if (tid < R && s_Aown[R] != 0.f) {                                 // output step of this sample: what the backward pass starts from
            tp.Astar[(size_t)S.b * R + tid] = s_Aown[tid];
            tp.hstar[(size_t)S.b * R + tid] = hn[tid];
        }
        const uint32_t ep = ll_base + (uint32_t)t + 1u;
        if (tid < LDA) { if (l2h) st_ll_l2(S.llA, (size_t)member * LDA + tid, s_Aown[tid], ep); else st_ll(S.llA, (size_t)member * LDA + tid, s_Aown[tid], ep); }
        {
            float4 pk[4], hq[4];
            park_load(pk, hq, s_park, 0, tid, hn + q3 * 16); S.ghp_r = park_fma(pk, hq);
            park_load(pk, hq, s_park, 1, tid, hn + q3 * 16); S.ghp_u = park_fma(pk, hq);
            park_load(pk, hq, s_park, 2, tid, hn + q3 * 16); S.ghn = dpp_group_sum<4>(park_fma(pk, hq)) + b_hn;
        }
    };
    // ---- C: the tile's 16 rows of A in, this member's class slice for them (logits, tape, softmax numerators, mixture), partials out
    auto blockC = [&](Slot& S, int t) __attribute__((always_inline)) {
        float* const s_At = S.base + L::At; float* const s_y = S.base + L::y; float* const s_e = S.base + L::e; float* const s_P = S.base + L::P;
        const uint32_t ep = ll_base + (uint32_t)t + 1u;
        {
            constexpr int NA = (TM * LDA + NT - 1) / NT;
            unsigned long long ua[NA];
            for (int spins = 0;; ) {
                bool fresh = true;
#pragma unroll
                for (int r = 0; r < NA; ++r) { ua[r] = ld_ll(S.llA, min(tid + NT * r, TM * LDA - 1)); fresh = fresh && ll_fresh(ua[r], ep); }
                if (!__any(!fresh) || spin_expired(spins)) break;
            }
            MMG_MSTAMP(16 + 16 * t + 8 + (&S == &sl[1] ? 1 : 0));
#pragma unroll
            for (int r = 0; r < NA; ++r) if (tid + NT * r < TM * LDA) s_At[tid + NT * r] = ll_value(ua[r]);
        }
        __syncthreads();
        yhead_tile<TM, LDA, LDY>(s_At, s_y, e4, cls, cyv, ncd, w2e);
        __syncthreads();
        {
            const bool keep_y = !y_last_only || t == T - 1;
#pragma unroll
            for (int u = 0; u < TM * CAP / NT; ++u) {
                const int idx = tid + NT * u, i = idx / CAP, c = idx % CAP;
                const int bi = S.tile * TM + i;
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
        {
            constexpr int NP = (TM * LDP + NT - 1) / NT;
            const size_t p_mine = (size_t)member * TM * LDP;
#pragma unroll
            for (int r = 0; r < NP; ++r) if (tid + NT * r < TM * LDP) { if (l2h) st_ll_l2(S.llP, p_mine + tid + NT * r, s_P[tid + NT * r], ep); else st_ll(S.llP, p_mine + tid + NT * r, s_P[tid + NT * r], ep); }
        }
    };
    // ---- K: the 16 slices' partials of this sample in, combined into g; the receiver's message
    auto blockK = [&](Slot& S, int t) __attribute__((always_inline)) {
        float* const s_in = S.base + L::y;                                  // (the logits of the step are dead)
        float* const s_m = S.base + L::ms; float* const s_s = s_m + 16;
        float* const s_gh = S.base + L::gh; float* const s_g = S.base + L::g; float* const s_w = S.base + L::w;
        const uint32_t ep = ll_base + (uint32_t)t + 1u;
        {
            constexpr int NP = (TM * LDP + NT - 1) / NT;
            unsigned long long up[NP];
            int kq[NP];
#pragma unroll
            for (int r = 0; r < NP; ++r) { const int i = min(tid + NT * r, TM * LDP - 1); kq[r] = ((i / LDP) * TM + member) * LDP + i % LDP; }
            for (int spins = 0;; ) {
                bool fresh = true;
#pragma unroll
                for (int r = 0; r < NP; ++r) { up[r] = ld_ll(S.llP, (size_t)kq[r]); fresh = fresh && ll_fresh(up[r], ep); }
                if (!__any(!fresh) || spin_expired(spins)) break;
            }
            MMG_MSTAMP(16 + 16 * (t + (&S == &sl[1] ? 1 : 0)) + 10 + (&S == &sl[1] ? 1 : 0));
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
            if (q3 == 0) s_g[u3] = gv;
        }
        __syncthreads();
        {
            const float lw = dpp_group_sum<8>(dot4p<2>(ww, s_g + k2 * 8, 4)) + bw;
            if (k2 == 0) s_w[m2] = lw;
        }
        __syncthreads();
    };
    // ============================================================ the skewed schedule (header)
    // ... written so that each block is instantiated once per slot (six inlined bodies): iteration t runs
    //     S(0,t) | K(1,t-1) | S(1,t) | C(0,t) | C(1,t) | K(0,t)        (t = T: only K(1,T-1) is left)
    const bool two = sl[1].on;
  for (int pair = pair0; pair < npair; pair += pair_stride) {
    if (pair != pair0) {                                     // the next pair of this workgroup: fresh samples on the same weights
        slot_begin(sl[0], pair, 0); slot_begin(sl[1], pair, 1);
        slot_state(sl[0]); slot_state(sl[1]);
        __syncthreads();
    }
    // (timing build: stamps 16 + 16 t + k of workgroup 0 -- block starts k = 0..5, the ends of the four polls k = 8..11)
    MMG_MSTAMP(2);
    for (int t = 0; t <= T; ++t) {
        MMG_MSTAMP(16 + 16 * t + 0);
        if (t < T) blockS(sl[0], t);
        MMG_MSTAMP(16 + 16 * t + 1);
        if (t > 0 && two) blockK(sl[1], t - 1);
        if (t == T) break;
        MMG_MSTAMP(16 + 16 * t + 2);
        if (two) blockS(sl[1], t);
        MMG_MSTAMP(16 + 16 * t + 3);
        blockC(sl[0], t);
        MMG_MSTAMP(16 + 16 * t + 4);
        if (two) blockC(sl[1], t);
        MMG_MSTAMP(16 + 16 * t + 5);
        blockK(sl[0], t);
    }
    MMG_MSTAMP(3);
    // ============================================================ per slot: tape flush, output selection / reward / top-k
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        Slot& S = sl[s];
        if (!S.on) continue;
        float* const s_mask = S.base + L::small + 16; float* const s_red = S.base + L::small + 40;
        const int b = S.b;
        pf_signal(S.cA);                                     // every slice owner's selected-logit stores (tape.outp, write-through) have completed
        for (int i4 = tid; i4 < T * (4 * R / 4); i4 += NT) {
            const int t = i4 >> 6, c = i4 & 63;
            *reinterpret_cast<float4*>(tp.gru + ((size_t)t * B + b) * 4 * R + 4 * c) = *reinterpret_cast<const float4*>(S.t_gru + 4 * i4);
        }
        for (int i4 = tid; i4 < (T + 1) * (R / 4); i4 += NT) {
            const int t = i4 >> 4, c = i4 & 15;
            *reinterpret_cast<float4*>(tp.h + ((size_t)t * B + b) * R + 4 * c) = *reinterpret_cast<const float4*>(S.t_h + 4 * i4);
        }
        for (int i4 = tid; i4 < T * (W / 4); i4 += NT) {
            const int t = i4 >> 3, c = i4 & 7;
            *reinterpret_cast<float4*>(tp.z + ((size_t)t * B + b) * W + 4 * c) = *reinterpret_cast<const float4*>(S.t_z + 4 * i4);
        }
        if (tid < T) {
            const float sbit = S.t_sp[tid], p = S.t_sp[16 + tid];
            const size_t rw = (size_t)tid * B + b;
            const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
            tp.s[rw] = sbit; tp.ps[rw] = p;
            tp.lp_s[rw] = sbit * l1 + (1.f - sbit) * l0; tp.ne_s[rw] = p * l1 + (1.f - p) * l0;
        }
        if (tid <= T) tp.mask[(size_t)tid * B + b] = (uint8_t)(s_mask[tid] != 0.f);
        if (tid == 240) { s_red[8] = (float)S.t_out; s_red[9] = S.sprod; }
        mc_wait(S.cA, (uint32_t)TM, tp.sync);                // (the tape flush above went out in its shadow)
        const int tstar = dm.fixed ? (T - 1) : (int)s_red[8];
        constexpr int NY = 4;                                // classes per thread: D <= 16 * CAP = NT * NY
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
        for (int u = 0; u < NY; ++u) if (tid + NT * u == S.tgt) dtv = o[u] - lse;
        const float dt = block_sum(dtv, s_red);
        float above = 0.f;
#pragma unroll
        for (int u = 0; u < NY; ++u) {
            const int d = tid + NT * u;
            if (d < D) {
                const float ld = o[u] - lse;
                tp.dist[(size_t)b * D + d] = ld;
                tp.sm[(size_t)b * D + d] = __expf(ld);
                above += (S.tgt >= 0 && ld > dt) ? 1.f : 0.f;
            }
        }
        above = block_sum(above, s_red);
        const float sprod_out = s_red[9];
        __syncthreads();                                     // (s_red[8..9] were read by everybody before the next slot's reductions)
        if (tid == 0) {
            tp.tstar[b] = tstar;
            tp.sprod[b] = sprod_out;
            tp.logs[b] = (S.tgt >= 0) ? dt : 0.f;
            tp.hit[b] = (S.tgt >= 0 && above < (float)dm.top_k) ? 1 : 0;
        }
    }
    __syncthreads();                                         // (the slots' LDS is the next pair's)
  }
}

}  // namespace mmg
