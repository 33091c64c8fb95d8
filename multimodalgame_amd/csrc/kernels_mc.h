// kernels_mc.h -- "many classes": the conversation of the small agents (H=256, W=32, R=64, V=100: BASELINE configs 1-3 and 5)
// when the description matrix has hundreds to thousands of rows (BASELINE config 5: D = 1000).
//
// The per-sample kernels re-stream all D x (R + V) class rows from L2 for every sample and step (656 KB per sample-step at
// D = 1000: 168 MB per step at 256 samples -- the chip's L2 bandwidth, 38 us per step), the 16-sample MFMA tiles need one CU
// per 16 samples (16 busy CUs at 256 samples, 73 us per tile-step).  Here a workgroup is BOTH:
//   * the register-resident agents of ONE sample (k_conversation_fast2's lane layouts, kernels_fast.h), and
//   * the owner of 1/16 of the classes for the 16 samples of its tile: its slice of -Cd (8 registers per lane), of w_y2 and
//     of desc (MFMA B fragments, 16 registers per lane) never leave the register file.
// Per exchange step the 16 workgroups of a tile exchange twice through memory (model.py:432-449):
//   A_t[b] = W_y1h h_t (64 floats per sample)                        all-gather   -> every member holds the tile's A [16, R]
//   slice logits -> slice max m_k, e = exp(y - m_k), slice sum s_k, unnormalised mixture e . desc_k  [16, V]
//                                                                    all-to-all   -> sample b combines its 16 slices like a
//                                                                                    streaming softmax: dbar = sum_k P_k e^(m_k-M) / S
// Hand-off = write-through (sc1) stores of the payload + one relaxed agent-scope counter increment per member; the consumer
// polls the counter and reads the payload with agent-scope (sc1) loads -- no L2 write-back, no L2 invalidate.
// Dependencies exist only INSIDE a tile and its 16 workgroups have consecutive ids, so with the hardware's in-order
// workgroup dispatch a launch of any size makes progress with >= 16 resident workgroups (no chip-wide co-residency
// requirement, unlike k_conv_persist); every spin is bounded all the same (error word -> k_opt skips the update).
// No early exit: a sample that has stopped keeps stepping (its tile needs its class slice); the tape rows of its dead steps
// are never read by a training minibatch (include/mmg.h: live rows only).
#pragma once
#include "device_utils.h"
#include "kernels_fwd.h"
#include "kernels_fast.h"
#include "kernels_tile.h"
#include "layout.h"

namespace mmg {

#define MMG_MC_LDP 104                              // floats per (slice, sample) row of the partial buffer: V mixture terms | m | s | pad

__device__ __forceinline__ uint32_t* mc_ctr(const Tape& tp, int kind, int tile, int ntile) { return tp.mcflags + ((size_t)kind * ntile + tile) * 64; }

// consumer side of a hand-off: lane 0 polls, everybody leaves through a barrier.  The payload is then read with ld_cc2.
__device__ __forceinline__ void mc_wait(uint32_t* ctr, uint32_t target, uint32_t* sync_err) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > MMG_SPIN_LIMIT) { __hip_atomic_store(sync_err + MMG_SYNC_ERR, 101u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}

#ifdef MMG_TIMING
#define MMG_MSTAMP(slot) do { if (blockIdx.x == 0 && tid == 0) tp.dbg[(slot)] = (long long)wall_clock64(); } while (0)
#else
#define MMG_MSTAMP(slot) do {} while (0)
#endif

// grid = ntile * 16 workgroups of 512 threads; workgroup (tile, member) runs sample tile * 16 + member (members beyond the
// batch run a clamped copy of the last sample and store nothing) and owns the classes [member * per, member * per + per).
// xcd_map: the 16 members of a tile are the workgroups i = x (mod 8) of a block of 128 (one XCD, one L2) instead of 16
// consecutive ones.  y_last_only: the class logits of the output step only go to the tape (mmg_train_step, Fixed mode).
template <int H, int W, int R, int V, int CAP>
__global__ __launch_bounds__(512, 2) void k_conversation_mc(Dims dm, Params P, Tape tp, ConvArgs ar, int ntile, int xcd_map, int y_last_only) {
    constexpr int NT = 512, TM = 16;
    static_assert(H == 256 && W == 32 && R == 64 && V == 100 && (CAP == 64), "shape of the register-resident small agents");
    constexpr int LDA = R + 4;                       // published A row: 64 floats | take | pad
    constexpr int LDY = CAP + 4;
    __shared__ __attribute__((aligned(16))) float s_a[H];
    __shared__ __attribute__((aligned(16))) float s_c[W];
    __shared__ __attribute__((aligned(16))) float s_z[W];
    __shared__ __attribute__((aligned(16))) float s_h[R];
    __shared__ __attribute__((aligned(16))) float s_gi[3 * R];
    __shared__ __attribute__((aligned(16))) float s_gh[3 * R];
    __shared__ __attribute__((aligned(16))) float s_Aown[LDA];
    __shared__ __attribute__((aligned(16))) float s_dbar[V + 4];
    __shared__ __attribute__((aligned(16))) float s_g[R];
    __shared__ float s_lp[W], s_lpw[W];
    __shared__ float s_misc[8];
    constexpr int TMAX = 16;
    __shared__ float s_uz[TMAX * W], s_uw[TMAX * W], s_us[TMAX];
    __shared__ __attribute__((aligned(16))) float s_At[TM * LDA];          // the tile's A rows (+ take flags)
    __shared__ __attribute__((aligned(16))) float s_y[TM * LDY];           // slice logits
    __shared__ __attribute__((aligned(16))) float s_e[TM * LDY];           // e = exp(y - m_k): A operand of the mixture product
    __shared__ __attribute__((aligned(16))) float s_P[TM * MMG_MC_LDP];    // this slice's partials, staged for 16-byte stores
    __shared__ __attribute__((aligned(16))) float s_in[TM * MMG_MC_LDP];   // the 16 slices' partials of this sample
    __shared__ float s_red[16];

    int tile, member;
    if (xcd_map) {
        const int w = blockIdx.x, blk = w >> 7, x = w & 7, slot = (w & 127) >> 3;
        tile = blk * 8 + x; member = slot;
        if (tile >= ntile) return;
    } else { tile = blockIdx.x >> 4; member = blockIdx.x & 15; }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = dm.B, T = dm.T, D = dm.D;
    const int b_raw = tile * TM + member;
    const bool have = b_raw < B;                        // a real sample (otherwise: class-slice owner only)
    // ar.lean (fused training step, continuous messages): only the receiver is trained and through the NLL alone, so the
    // backward pass reads z, h, the GRU gates and the output step -- the sender's hidden layer, the code inputs, dbar, g and
    // the receiver's message are not stored (1.9 KB of 3.4 KB per sample-step)
    const bool have_full = have && !ar.lean;
    const int b = have ? b_raw : B - 1;
    const bool binary = dm.use_binary != 0, train = ar.train != 0;
    const bool inject = ar.u_s != nullptr;
    const int per = ar.per;                             // classes per member (multiple of 4, <= CAP)
    const int c0 = member * per;
    MMG_MSTAMP(0);
    const uint32_t mb_counter = tp.counter[0];
    const uint32_t gb = (uint32_t)(dm.boff + b);
    if (train && inject) {
        for (int i = tid; i < T * W; i += NT) {
            const int t = i / W, j = i - t * W;
            if (ar.u_z) s_uz[i] = ar.u_z[((size_t)t * B + b) * W + j];
            if (ar.u_w) s_uw[i] = ar.u_w[((size_t)t * B + b) * W + j];
        }
        if (tid < T) s_us[tid] = ar.u_s[(size_t)tid * B + b];
    }
    // ------------------------------------------------------------ agent weights -> registers (kernels_fast.h lane layouts)
    const int n1 = tid >> 1, h1 = tid & 1;
    constexpr int J1 = W / 8;
    float wc[4 * J1];
#pragma unroll
    for (int j = 0; j < J1; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(P.p[S_CODE_W] + (size_t)n1 * W + (h1 + 2 * j) * 4);
        wc[4 * j] = v.x; wc[4 * j + 1] = v.y; wc[4 * j + 2] = v.z; wc[4 * j + 3] = v.w;
    }
    const float bc = P.p[S_CODE_B][n1];
    const float hw0 = tp.hw0[n1];
    const float hx = tp.hx[(size_t)b * H + n1];
    constexpr int LB = NT / W, JB = H / (4 * LB), JW = R / (4 * LB);
    const int nb = tid / LB, kpb = tid % LB;
    float wb[4 * JB], ww[4 * JW];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(P.p[S_BIN_W] + (size_t)nb * H + kpb * 4 + 4 * LB * j);
        wb[4 * j] = v.x; wb[4 * j + 1] = v.y; wb[4 * j + 2] = v.z; wb[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < JW; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(P.p[R_W_W] + (size_t)nb * R + kpb * 4 + 4 * LB * j);
        ww[4 * j] = v.x; ww[4 * j + 1] = v.y; ww[4 * j + 2] = v.z; ww[4 * j + 3] = v.w;
    }
    const float bb = P.p[S_BIN_B][nb];
    const float bw = P.p[R_W_B][nb];
    const int n3 = tid >> 1, h3 = tid & 1;
    const bool gru_lane = n3 < 3 * R;
    constexpr int J3I = W / 8, J3H = R / 8;
    float wih[4 * J3I], whh[4 * J3H];
    float bih = 0.f, bhh = 0.f;
    {
        const int nr = gru_lane ? n3 : 0;
#pragma unroll
        for (int j = 0; j < J3I; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(P.p[R_WIH] + (size_t)nr * W + (h3 + 2 * j) * 4);
            wih[4 * j] = v.x; wih[4 * j + 1] = v.y; wih[4 * j + 2] = v.z; wih[4 * j + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < J3H; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(P.p[R_WHH] + (size_t)nr * R + (h3 + 2 * j) * 4);
            whh[4 * j] = v.x; whh[4 * j + 1] = v.y; whh[4 * j + 2] = v.z; whh[4 * j + 3] = v.w;
        }
        bih = P.p[R_BIH][nr]; bhh = P.p[R_BHH][nr];
    }
    constexpr int L4 = NT / R, J4 = R / (4 * L4);
    const int n4 = tid / L4, kp4 = tid % L4;
    float wy1[4 * J4], wh[4 * J4];
#pragma unroll
    for (int j = 0; j < J4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(P.p[R_Y1_W] + (size_t)n4 * (R + V) + kp4 * 4 + 4 * L4 * j);
        wy1[4 * j] = v.x; wy1[4 * j + 1] = v.y; wy1[4 * j + 2] = v.z; wy1[4 * j + 3] = v.w;
        const float4 u = *reinterpret_cast<const float4*>(P.p[R_WH_W] + (size_t)n4 * R + kp4 * 4 + 4 * L4 * j);
        wh[4 * j] = u.x; wh[4 * j + 1] = u.y; wh[4 * j + 2] = u.z; wh[4 * j + 3] = u.w;
    }
    const float bh = P.p[R_WH_B][n4];
    constexpr int JD = (V + L4 - 1) / L4;
    float wd[JD];
#pragma unroll
    for (int j = 0; j < JD; ++j) { const int k = kp4 + L4 * j; wd[j] = P.p[R_WD_W][(size_t)n4 * V + min(k, V - 1)]; if (k >= V) wd[j] = 0.f; }
    const float ws = P.p[R_S_W][lane];
    const float bs = P.p[R_S_B][0];
    const float sig_cb = (tid < W) ? fsigmoid(P.p[S_CODE_BIAS][tid]) : 0.f;
    // ------------------------------------------------------------ this member's class slice -> registers
    // y head: class slot cls = tid / 8, r-eighth e8 = tid % 8:  y[b, d] = cy[d] + sum_r w2[r] max(A[b, r], -Cd[d, r])
    // (relu(A + c) = max(A, -c) + c; cy[d] = b_y2 + sum_r w2[r] Cd[d, r] comes from k_prep)
    const int cls = tid >> 3, e8 = tid & 7;
    const bool cls_ok = cls < per && c0 + cls < D;
    float ncd[8], w2e[8];
    {
        const float* crow = tp.Cd + (size_t)min(c0 + cls, D - 1) * R + 8 * e8;
        const float4 u0 = *reinterpret_cast<const float4*>(crow), u1 = *reinterpret_cast<const float4*>(crow + 4);
        ncd[0] = -u0.x; ncd[1] = -u0.y; ncd[2] = -u0.z; ncd[3] = -u0.w; ncd[4] = -u1.x; ncd[5] = -u1.y; ncd[6] = -u1.z; ncd[7] = -u1.w;
        const float4 q0 = *reinterpret_cast<const float4*>(P.p[R_Y2_W] + 8 * e8), q1 = *reinterpret_cast<const float4*>(P.p[R_Y2_W] + 8 * e8 + 4);
        w2e[0] = q0.x; w2e[1] = q0.y; w2e[2] = q0.z; w2e[3] = q0.w; w2e[4] = q1.x; w2e[5] = q1.y; w2e[6] = q1.z; w2e[7] = q1.w;
    }
    const float cyv = cls_ok ? tp.cy[min(c0 + cls, D - 1)] : -3.0e38f;
    // mixture: wave w < 7 owns the columns 16 w .. 16 w + 15 of [16 samples, CAP classes] x desc_k [CAP, V]; B fragment of
    // k-step ks: lane (fi = lane & 15, fq = lane >> 4) holds desc[c0 + 4 ks + fq][16 w + fi]
    const int fi = lane & 15, fq = lane >> 4;
    float bfrag[CAP / 4];
#pragma unroll
    for (int ks = 0; ks < CAP / 4; ++ks) {
        const int c = 4 * ks + fq, v = 16 * wave + fi;
        const float dv = ar.desc[(size_t)min(c0 + c, D - 1) * V + min(v, V - 1)];
        bfrag[ks] = (c < per && c0 + c < D && v < V && wave < 7) ? dv : 0.f;
    }
    if (train && !inject) {                        // Philox draws of the whole conversation, while the weight loads are in flight
        for (int i = tid; i < T * W; i += NT) {
            const int t = i / W, j = i - t * W;
            const uint32_t e = (uint32_t)((t * dm.Bg + gb) * W + j);
            s_uz[i] = philox_uniform(ar.seed, e, mb_counter, 0u);
            s_uw[i] = philox_uniform(ar.seed, e, mb_counter, 2u);
        }
        if (tid < T) s_us[tid] = philox_uniform(ar.seed, (uint32_t)(tid * dm.Bg + gb), mb_counter, 1u);
    }
    MMG_MSTAMP(1);
    // ------------------------------------------------------------ conversation state
    if (tid < R) { s_h[tid] = 0.f; if (have) tp.h[(size_t)b * R + tid] = 0.f; }
    if (tid < W) s_c[tid] = dm.first_rec;
    if (tid == 0) { s_misc[0] = 1.f; s_misc[1] = -1.f; s_misc[2] = 1.f; if (have) tp.mask[b] = 1; }
    __syncthreads();
    uint32_t* cA = mc_ctr(tp, 0, tile, ntile); uint32_t* cP = mc_ctr(tp, 1, tile, ntile);
    float* pubA = tp.mcA + (size_t)tile * TM * LDA;
    float* part_mine = tp.mcpart + ((size_t)(tile * TM + member) * TM) * MMG_MC_LDP;      // [16 samples][LDP] written by this member
    const float* part_tile = tp.mcpart + ((size_t)tile * TM * TM) * MMG_MC_LDP;            // [16 members][16 samples][LDP]
    float stop_p = 0.5f, stop_bit = 0.f;
    MMG_MSTAMP(2);
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * B + b;
        // ===== (1) sender: h_w = code_layer(c), a = tanh(h_x + h_w)
        {
            float hw = hw0;
            if (t > 0) hw = bc + dpp_group_sum<2>(dot4<J1>(wc, s_c + h1 * 4, 8));
            const float av = ftanh(hx + hw);
            if (h1 == 0) { s_a[n1] = av; if (have_full) tp.a[row * H + n1] = av; }
            if (tid < W && have_full) {
                const float cv = s_c[tid];
                tp.zr[row * W + tid] = cv;
                tp.c[row * W + tid] = (t == 0) ? sig_cb : cv;
            }
        }
        float ghv = bhh + dpp_group_sum<2>(dot4<J3H>(whh, s_h + h3 * 4, 8));      // GRU hidden-side product (independent of z)
        __syncthreads();
        // ===== (2) sender logits + sample
        {
            float acc = dpp_group_sum<LB>(dot4<JB>(wb, s_a + kpb * 4, 4 * LB));
            if (kpb == 0) {
                const float lz = acc + bb;
                float zz = lz, pp = 0.f;
                if (binary) {
                    pp = fsigmoid(lz);
                    zz = train ? ((s_uz[t * W + nb] < pp) ? 1.f : 0.f) : rintf(pp);
                    if (have) tp.pz[row * W + nb] = pp;
                }
                s_z[nb] = zz; s_lp[nb] = pp;
                if (have) tp.z[row * W + nb] = zz;
            }
        }
        __syncthreads();
        // ===== (3) GRU gate pre-activations
        {
            const float giv = bih + dpp_group_sum<2>(dot4<J3I>(wih, s_z + h3 * 4, 8));
            if (gru_lane && h3 == 0) { s_gi[n3] = giv; s_gh[n3] = ghv; }
        }
        if (binary && wave == 6) {                                         // waves 6, 7 hold no GRU rows: sender log-lik terms
            float lpv = 0.f, nev = 0.f;
            if (lane < W) {
                const float p = s_lp[lane], zz = s_z[lane];
                const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
                lpv = zz * l1 + (1.f - zz) * l0;
                nev = p * l1 + (1.f - p) * l0;
            }
            lpv = dpp_wave_sum(lpv); nev = dpp_wave_sum(nev);
            if (lane == 0 && have) { tp.lp_z[row] = lpv; tp.ne_z[row] = nev; }
        }
        __syncthreads();
        // ===== (4) GRU state update
        if (tid < R) {
            const float rr = fsigmoid(s_gi[tid] + s_gh[tid]);
            const float uu = fsigmoid(s_gi[R + tid] + s_gh[R + tid]);
            const float ghn = s_gh[2 * R + tid];
            const float nn = ftanh(s_gi[2 * R + tid] + rr * ghn);
            const float hv = nn + uu * (s_h[tid] - nn);
            if (have) {
                float* gr = tp.gru + row * 4 * R;
                gr[tid] = rr; gr[R + tid] = uu; gr[2 * R + tid] = nn; gr[3 * R + tid] = ghn;
                tp.h[((size_t)(t + 1) * B + b) * R + tid] = hv;
            }
            s_h[tid] = hv;
        }
        __syncthreads();
        // ===== (5) heads on h: A = W_y1h h (-> the tile), h-part of g, stop bit
        float gpre_h;
        {
            const float accA = dpp_group_sum<L4>(dot4<J4>(wy1, s_h + kp4 * 4, 4 * L4));
            const float accH = dpp_group_sum<L4>(dot4<J4>(wh, s_h + kp4 * 4, 4 * L4));
            if (kp4 == 0) s_Aown[n4] = accA;
            gpre_h = accH + bh;
        }
        if (wave == 7) {
            const float sv = dpp_wave_sum(ws * s_h[lane]);
            if (lane == 0) {
                const float p = fsigmoid(sv + bs);
                float sbit;
                if (train) sbit = (s_us[t] < p) ? 1.f : 0.f;
                else {
                    const float prod = dm.s_prob_prod ? s_misc[2] * p : p;
                    s_misc[2] = prod;
                    sbit = rintf(prod);
                }
                s_misc[3] = sbit;
                if (have) { tp.s[row] = sbit; tp.ps[row] = p; }
                stop_p = p; stop_bit = sbit;
                // masks / output step (model.py:775, 852, 870, 1261): known here, published with A
                const float m_t = s_misc[0];
                const float m_next = fminf(m_t, sbit);
                const bool first_stop = (m_next == 0.f) && (s_misc[1] < 0.f);
                const bool take_out = dm.fixed ? (t == T - 1) : (first_stop || ((t == T - 1) && (s_misc[1] < 0.f)));
                if (have) tp.mask[(size_t)(t + 1) * B + b] = (uint8_t)(m_next != 0.f);
                if (take_out) s_misc[1] = (float)t;
                s_misc[0] = m_next;
                s_Aown[R] = (take_out && have) ? 1.f : 0.f; s_Aown[R + 1] = 0.f; s_Aown[R + 2] = 0.f; s_Aown[R + 3] = 0.f;
                const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
                if (have) { tp.lp_s[row] = sbit * l1 + (1.f - sbit) * l0; tp.ne_s[row] = p * l1 + (1.f - p) * l0; }
            }
        }
        __syncthreads();
        MMG_MSTAMP(16 + 8 * t + 0);
        if (tid < R && s_Aown[R] != 0.f) {                                 // output step of this sample: what the backward pass starts from
            tp.Astar[(size_t)b * R + tid] = s_Aown[tid];
            tp.hstar[(size_t)b * R + tid] = s_h[tid];
        }
        // ----- hand-off 1: publish A (17 x 16 bytes), gather the tile's 16 rows
        if (tid < LDA / 4) st_wt4(pubA + member * LDA + 4 * tid, *reinterpret_cast<const float4*>(s_Aown + 4 * tid));
        pf_signal(cA);
        mc_wait(cA, (uint32_t)(TM * (t + 1)), tp.sync);
        MMG_MSTAMP(16 + 8 * t + 1);
        for (int i = tid; i < TM * LDA / 2; i += NT) {
            const float2 v = ld_cc2(pubA + 2 * i);
            *reinterpret_cast<float2*>(s_At + 2 * i) = v;
        }
        __syncthreads();
        MMG_MSTAMP(16 + 8 * t + 2);
        // ===== (6) slice logits for the 16 samples of the tile
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float4 a0 = *reinterpret_cast<const float4*>(s_At + i * LDA + 8 * e8);
            const float4 a1 = *reinterpret_cast<const float4*>(s_At + i * LDA + 8 * e8 + 4);
            float p0 = w2e[0] * fmax_nn(a0.x, ncd[0]), p1 = w2e[1] * fmax_nn(a0.y, ncd[1]);
            p0 = fmaf(w2e[2], fmax_nn(a0.z, ncd[2]), p0); p1 = fmaf(w2e[3], fmax_nn(a0.w, ncd[3]), p1);
            p0 = fmaf(w2e[4], fmax_nn(a1.x, ncd[4]), p0); p1 = fmaf(w2e[5], fmax_nn(a1.y, ncd[5]), p1);
            p0 = fmaf(w2e[6], fmax_nn(a1.z, ncd[6]), p0); p1 = fmaf(w2e[7], fmax_nn(a1.w, ncd[7]), p1);
            const float tot = dpp_group_sum<8>(p0 + p1);
            if (e8 == 0) s_y[i * LDY + cls] = tot + cyv;
        }
        __syncthreads();
        // logits -> tape (every step: exchange() returns them; y_last_only: the output step's) and the selected rows -> outp
        {
            const bool keep_y = !y_last_only || t == T - 1;
#pragma unroll
            for (int u = 0; u < TM * CAP / NT; ++u) {
                const int idx = tid + NT * u, i = idx / CAP, c = idx % CAP;
                const int bi = tile * TM + i;
                const float yv = s_y[i * LDY + c];
                if (bi < B && c < per && c0 + c < D) {
                    if (keep_y) tp.y[((size_t)t * B + bi) * D + c0 + c] = yv;
                    if (s_At[i * LDA + R] != 0.f) st_wt(&tp.outp[(size_t)bi * D + c0 + c], yv);     // model.py:1261-1264; read back by the sample's own workgroup
                }
            }
        }
        // slice softmax numerators: 16 lanes per sample, 4 classes per lane (waves 0-3)
        if (tid < TM * 16) {
            const int i = tid >> 4, l = tid & 15;
            float yv[CAP / 16];
            float m = -3.0e38f;
#pragma unroll
            for (int j = 0; j < CAP / 16; ++j) { yv[j] = s_y[i * LDY + l + 16 * j]; m = fmaxf(m, yv[j]); }
            m = fmaxf(m, dpp_f<MMG_DPP_QUAD_1032>(m)); m = fmaxf(m, dpp_f<MMG_DPP_QUAD_2301>(m));
            m = fmaxf(m, dpp_f<MMG_DPP_ROW_HALF_MIRROR>(m)); m = fmaxf(m, dpp_f<MMG_DPP_ROW_MIRROR>(m));
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < CAP / 16; ++j) { const float e = __expf(yv[j] - m); s += e; s_e[i * LDY + l + 16 * j] = e; }
            s = dpp_group_sum<16>(s);
            if (l == 0) { s_P[i * MMG_MC_LDP + V] = m; s_P[i * MMG_MC_LDP + V + 1] = s; s_P[i * MMG_MC_LDP + V + 2] = 0.f; s_P[i * MMG_MC_LDP + V + 3] = 0.f; }
        }
        __syncthreads();
        // ===== (7) unnormalised mixture of the slice: [16, CAP] x [CAP, V] on the matrix cores
        if (wave < 7) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < CAP / 4; ks += 2) {
                acc0 = mfma16(s_e[fi * LDY + 4 * ks + fq], bfrag[ks], acc0);
                acc1 = mfma16(s_e[fi * LDY + 4 * (ks + 1) + fq], bfrag[ks + 1], acc1);
            }
            const int v = 16 * wave + fi;
            if (v < V) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_P[(4 * fq + r) * MMG_MC_LDP + v] = acc0[r] + acc1[r];
            }
        }
        __syncthreads();
        MMG_MSTAMP(16 + 8 * t + 3);
        // ----- hand-off 2: this slice's partials out (16 x 26 x 16 bytes), the 16 slices of this sample in
        if (tid < TM * (MMG_MC_LDP / 4)) st_wt4(part_mine + 4 * tid, *reinterpret_cast<const float4*>(s_P + 4 * tid));
        pf_signal(cP);
        mc_wait(cP, (uint32_t)(TM * (t + 1)), tp.sync);
        MMG_MSTAMP(16 + 8 * t + 4);
        for (int i = tid; i < TM * MMG_MC_LDP / 2; i += NT) {
            const int k = i / (MMG_MC_LDP / 2), q = i % (MMG_MC_LDP / 2);
            const float2 v = ld_cc2(part_tile + ((size_t)(k * TM + member)) * MMG_MC_LDP + 2 * q);
            *reinterpret_cast<float2*>(s_in + k * MMG_MC_LDP + 2 * q) = v;
        }
        __syncthreads();
        MMG_MSTAMP(16 + 8 * t + 5);
        // combine the slices (streaming softmax): dbar = sum_k P_k exp(m_k - M) / sum_k s_k exp(m_k - M)
        if (tid < 128) {
            const int v = min(tid, V - 1);
            float M = -3.0e38f;
#pragma unroll
            for (int k = 0; k < TM; ++k) M = fmaxf(M, s_in[k * MMG_MC_LDP + V]);
            float S = 0.f, acc = 0.f;
#pragma unroll
            for (int k = 0; k < TM; ++k) {
                const float sc = __expf(s_in[k * MMG_MC_LDP + V] - M);
                S = fmaf(s_in[k * MMG_MC_LDP + V + 1], sc, S);
                acc = fmaf(s_in[k * MMG_MC_LDP + v], sc, acc);
            }
            const float dv = acc * __builtin_amdgcn_rcpf(S);
            if (tid < V) { s_dbar[tid] = dv; if (have_full) tp.dbar[row * V + tid] = dv; }
        }
        __syncthreads();
        // ===== (8) h_w = tanh(w_h h + b_h + w_d dbar)
        {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int j = 0; j + 3 < JD; j += 4) {
                a0 = fmaf(wd[j], s_dbar[min(kp4 + L4 * j, V - 1)], a0); a1 = fmaf(wd[j + 1], s_dbar[min(kp4 + L4 * (j + 1), V - 1)], a1);
                a2 = fmaf(wd[j + 2], s_dbar[min(kp4 + L4 * (j + 2), V - 1)], a2); a3 = fmaf(wd[j + 3], s_dbar[min(kp4 + L4 * (j + 3), V - 1)], a3);
            }
#pragma unroll
            for (int j = JD & ~3; j < JD; ++j) a0 = fmaf(wd[j], s_dbar[min(kp4 + L4 * j, V - 1)], a0);
            const float acc = dpp_group_sum<L4>((a0 + a1) + (a2 + a3));
            if (kp4 == 0) {
                const float gv = ftanh(gpre_h + acc);
                s_g[n4] = gv;
                if (have_full) tp.g[row * R + n4] = gv;
            }
        }
        __syncthreads();
        // ===== (9) receiver message
        {
            float acc = dpp_group_sum<LB>(dot4<JW>(ww, s_g + kpb * 4, 4 * LB));
            if (kpb == 0) {
                const float lw = acc + bw;
                float wv = lw, pp = 0.f;
                if (binary) {
                    pp = fsigmoid(lw);
                    wv = train ? ((s_uw[t * W + nb] < pp) ? 1.f : 0.f) : rintf(pp);
                    if (have) tp.pw[row * W + nb] = pp;
                }
                s_c[nb] = wv; s_lpw[nb] = pp;
                if (have_full) tp.w[row * W + nb] = wv;
            }
        }
        __syncthreads();
        MMG_MSTAMP(16 + 8 * t + 6);
        if (binary && wave == 7) {                                         // overlaps with phase (1) of the next step
            float lpv = 0.f, nev = 0.f;
            if (lane < W) {
                const float p = s_lpw[lane], wv = s_c[lane];
                const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
                lpv = wv * l1 + (1.f - wv) * l0;
                nev = p * l1 + (1.f - p) * l0;
            }
            lpv = dpp_wave_sum(lpv); nev = dpp_wave_sum(nev);
            if (lane == 0 && have) { tp.lp_w[row] = lpv; tp.ne_w[row] = nev; }
        }
    }
    (void)stop_p; (void)stop_bit;
    MMG_MSTAMP(3);
    if (!have) return;
    // ------------------------------------------------------------ output selection / reward / top-k (model.py:1264-1275, 1333-1339)
    // every slice owner stored this sample's selected logits (write-through) before its partial hand-off of that step
    const int tstar = dm.fixed ? (T - 1) : (int)s_misc[1];
    const int tgt = ar.target ? (int)ar.target[b] : -1;
    constexpr int NY = 2;                               // classes per thread: D <= 16 * CAP = NT * NY
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
        tp.sprod[b] = s_misc[2];
        tp.logs[b] = (tgt >= 0) ? dt : 0.f;
        tp.hit[b] = (tgt >= 0 && above < (float)dm.top_k) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Backward pass of the many-class conversation in CONTINUOUS mode (-nouse_binary: the loss is the NLL of the selected logits,
// only the receiver is trained, model.py:1297-1305, 1313; every message input is detached, so the gradient reaches the GRU
// only through h at the output step).  Two launches replace k_bwd_conv + k_dC + k_stats:
//   k_bwd_mc1   the (sample, class, r) indicator pass, once:  dy = (softmax(outp) - onehot) / B                    model.py:1267-1271
//               dC[d, r]  = w2[r] sum_b dy[b, d] 1[A*[b, r] + Cd[d, r] > 0]      Py2[d, r] = sum_b dy[b, d] relu(A*[b, r] + Cd[d, r])
//               dA[b, r]  = w2[r] sum_d dy[b, d] 1[...]                          
//               grid = 16 class blocks x G sample groups; partial sums over the group's samples / the block's classes
//   k_bwd_mc2   B sample workgroups: dA = sum of the 16 class-block partials, dh = W_y1h^T dA at t*, GRU BPTT with W_hh^T in
//               registers (k_bwd_sample's recurrence); + workgroups that add the G partials of dC / Py2; + (fused step) one
//               workgroup with the batch statistics (sum of rewards, top-k hits)
// ---------------------------------------------------------------------------------------------
template <int R, int CAP>
__global__ __launch_bounds__(512, 2) void k_bwd_mc1(Dims dm, Params P, Tape tp, const int64_t* __restrict__ target, int per, int ntile, int ngroup) {
    constexpr int NT = 512, TM = 16, LDA = R + 4, LDY = CAP + 4, LDC = R + 2;
    static_assert(R == 64 && CAP == 64, "slice shape of k_conversation_mc");
    __shared__ __attribute__((aligned(16))) float s_A[TM * LDA];
    __shared__ __attribute__((aligned(16))) float s_dy[TM * LDY];
    __shared__ __attribute__((aligned(16))) float s_Cd[CAP * LDC];
    const int tid = threadIdx.x;
    const int cb = blockIdx.x & 15, grp = blockIdx.x >> 4;
    const int B = dm.B, D = dm.D;
    const int c0 = cb * per;
    const int cls = tid >> 3, e8 = tid & 7;
    const bool cls_ok = cls < per && c0 + cls < D;
    // Both passes on PACKED fp32 (v_pk_fma_f32, two units per issue slot).  The indicator 1[A + Cd > 0] is one packed
    // fused multiply-add with the clamp modifier, clamp((A + Cd) 2^100) -- exactly 0 or 1 unless 0 < |A + Cd| < 2^-100, which a
    // sum of two O(1) floats never is (kernels_tile.h uses the scalar form) -- and with it  dC += ind dy,  Py2 += dy ((A + Cd) ind):
    // the same values added in the same order as the compare / select / max form (x 1 and x 0 are exact), 5 packed instructions per
    // unit pair instead of 12 scalar ones; the sample side 2 instead of 8.  (Round 6; config 5 at 2 048 samples: 52.9 -> see DESIGN 3g)
    constexpr float BIG = 1.2676506e30f;                                     // 2^100
    float cd[8], w2e[8];
    f32x2 cd2[4], cdk2[4], dC2[4], P2[4];
    {
        const float* crow = tp.Cd + (size_t)min(c0 + cls, D - 1) * R + 8 * e8;
        const float4 u0 = *reinterpret_cast<const float4*>(crow), u1 = *reinterpret_cast<const float4*>(crow + 4);
        cd[0] = u0.x; cd[1] = u0.y; cd[2] = u0.z; cd[3] = u0.w; cd[4] = u1.x; cd[5] = u1.y; cd[6] = u1.z; cd[7] = u1.w;
        const float4 q0 = *reinterpret_cast<const float4*>(P.p[R_Y2_W] + 8 * e8), q1 = *reinterpret_cast<const float4*>(P.p[R_Y2_W] + 8 * e8 + 4);
        w2e[0] = q0.x; w2e[1] = q0.y; w2e[2] = q0.z; w2e[3] = q0.w; w2e[4] = q1.x; w2e[5] = q1.y; w2e[6] = q1.z; w2e[7] = q1.w;
#pragma unroll
        for (int j = 0; j < 8; ++j) s_Cd[cls * LDC + 8 * e8 + j] = cd[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cd2[q] = f32x2{cd[2 * q], cd[2 * q + 1]}; cdk2[q] = cd2[q] * BIG;
            dC2[q] = f32x2{0.f, 0.f}; P2[q] = f32x2{0.f, 0.f};
        }
    }
    // dA pass: sample i2 = tid / 32, columns 2 rp, 2 rp + 1
    const int i2 = tid >> 5, rp = tid & 31;
    const float w2a = P.p[R_Y2_W][2 * rp], w2b = P.p[R_Y2_W][2 * rp + 1];
    const float invB = 1.0f / (float)dm.Bg;
    const int tpg = (ntile + ngroup - 1) / ngroup;
    for (int tl = grp * tpg; tl < min(ntile, (grp + 1) * tpg); ++tl) {
        const int b0 = tl * TM;
        __syncthreads();                                                  // (the previous tile's passes are done with s_A / s_dy)
        // stage A* rows and dy = (sm - onehot) / B of this (tile, class block); dy also goes to the tape
        for (int idx = tid; idx < TM * R / 4; idx += NT) {
            const int i = idx / (R / 4), q = idx % (R / 4);
            const float4 v = *reinterpret_cast<const float4*>(tp.Astar + (size_t)min(b0 + i, B - 1) * R + 4 * q);
            *reinterpret_cast<float4*>(s_A + i * LDA + 4 * q) = v;
        }
#pragma unroll
        for (int u = 0; u < TM * CAP / NT; ++u) {
            const int idx = tid + NT * u, i = idx / CAP, c = idx % CAP;
            const int bi = b0 + i, d = c0 + c;
            const bool ok = bi < B && c < per && d < D;
            float v = 0.f;
            if (ok) {
                v = (tp.sm[(size_t)bi * D + d] - ((int)target[bi] == d ? 1.f : 0.f)) * invB;
                tp.dy[(size_t)bi * D + d] = v;
            }
            s_dy[i * LDY + c] = v;
        }
        __syncthreads();
        // ---- class side: sums over the tile's samples, accumulated in registers across the group's tiles
#pragma unroll 4
        for (int i = 0; i < TM; ++i) {
            const float4 a0 = *reinterpret_cast<const float4*>(s_A + i * LDA + 8 * e8);
            const float4 a1 = *reinterpret_cast<const float4*>(s_A + i * LDA + 8 * e8 + 4);
            const f32x2 av2[4] = {f32x2{a0.x, a0.y}, f32x2{a0.z, a0.w}, f32x2{a1.x, a1.y}, f32x2{a1.z, a1.w}};
            const float dyv = s_dy[i * LDY + cls];
            const f32x2 dy2 = {dyv, dyv};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 ind = pk_fma_clamp(av2[q], BIG, cdk2[q]);
                dC2[q] = __builtin_elementwise_fma(ind, dy2, dC2[q]);
                P2[q] = __builtin_elementwise_fma(dy2, (av2[q] + cd2[q]) * ind, P2[q]);
            }
        }
        // ---- sample side: sums over the block's classes -> partial dA of the tile's samples
        {
            const f32x2 ak2 = f32x2{s_A[i2 * LDA + 2 * rp], s_A[i2 * LDA + 2 * rp + 1]} * BIG;
            f32x2 acc2 = {0.f, 0.f};
            float dsum = 0.f;
#pragma unroll 16
            for (int c = 0; c < CAP; ++c) {
                const float2 cv = *reinterpret_cast<const float2*>(s_Cd + c * LDC + 2 * rp);
                const float dyv = s_dy[i2 * LDY + c];
                acc2 = __builtin_elementwise_fma(pk_fma_clamp(f32x2{cv.x, cv.y}, BIG, ak2), f32x2{dyv, dyv}, acc2);
                dsum += dyv;
            }
            const float acc_a = acc2.x, acc_b = acc2.y;
            const int bi = b0 + i2;
            if (bi < B) {
                *reinterpret_cast<float2*>(tp.mcdA + ((size_t)cb * B + bi) * R + 2 * rp) = make_float2(acc_a * w2a, acc_b * w2b);
                if (rp == 0) tp.mcdys[(size_t)cb * B + bi] = dsum;
            }
        }
    }
    // this group's partial dC | Py2 of the block's classes
    if (cls_ok) {
        float* o = tp.mcdC + ((size_t)grp * 2 * D + (c0 + cls)) * R + 8 * e8;
        *reinterpret_cast<float4*>(o) = make_float4(dC2[0].x * w2e[0], dC2[0].y * w2e[1], dC2[1].x * w2e[2], dC2[1].y * w2e[3]);
        *reinterpret_cast<float4*>(o + 4) = make_float4(dC2[2].x * w2e[4], dC2[2].y * w2e[5], dC2[3].x * w2e[6], dC2[3].y * w2e[7]);
        float* q = o + (size_t)D * R;
        *reinterpret_cast<float4*>(q) = make_float4(P2[0].x, P2[0].y, P2[1].x, P2[1].y);
        *reinterpret_cast<float4*>(q + 4) = make_float4(P2[2].x, P2[2].y, P2[3].x, P2[3].y);
    }
}

template <int R, int V>
__global__ __launch_bounds__(256, 1) void k_bwd_mc2(Dims dm, Params P, Tape tp, int ngroup, int nred, int with_stats) {
    constexpr int NT = 256, K4 = NT / R, TMAX = 16;
    static_assert(R == 64 && K4 == 4, "receiver shape of the register-resident kernels");
    __shared__ __attribute__((aligned(16))) float s_dh[R], s_dgh[3 * R], s_dA[R], s_dAy[R], s_part[4 * R];
    __shared__ __attribute__((aligned(16))) float t_gru[TMAX * 4 * R], t_h[(TMAX + 1) * R];
    const int tid = threadIdx.x;
    const int B = dm.B, T = dm.T, D = dm.D;
    if ((int)blockIdx.x >= B) {
        const int j = (int)blockIdx.x - B;
        if (j >= nred) {                                                    // fused step: the batch statistics (k_stats) ride along
            if (with_stats) stats_pairs<false>(dm, P, tp, 0, tid >> 6, NT / 64);
            return;
        }
        // dC | Py2 = sum of the sample groups' partials, in group order (deterministic): one float4 per thread
        const size_t n4 = (size_t)2 * D * R / 4, i4 = (size_t)j * NT + tid;
        if (i4 < n4) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int g0 = 0; g0 < ngroup; g0 += 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(tp.mcdC + (size_t)min(g0 + u, ngroup - 1) * 2 * D * R)[i4];
#pragma unroll
                for (int u = 0; u < 8; ++u) if (g0 + u < ngroup) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
            }
            const size_t half = (size_t)D * R / 4;
            if (i4 < half) reinterpret_cast<float4*>(tp.dC)[i4] = acc;
            else reinterpret_cast<float4*>(tp.Py2)[i4 - half] = acc;
        }
        return;
    }
    const int b = blockIdx.x;
    const int tstar = tp.tstar[b];
    const int k4 = tid / K4, p4 = tid % K4;
    float whhT[3 * R / K4], y1T[R / K4];
#pragma unroll
    for (int i = 0; i < 3 * R / K4; ++i) whhT[i] = P.p[R_WHH][(size_t)(p4 * (3 * R / K4) + i) * R + k4];
#pragma unroll
    for (int i = 0; i < R / K4; ++i) y1T[i] = P.p[R_Y1_W][(size_t)(p4 * (R / K4) + i) * (R + V) + k4];
    // the 16 class-block partials of dA[b, :]: thread (r = tid % 64, quarter = tid / 64) adds four of them
    float dap[4];
    {
        const int r = tid & 63, qd = tid >> 6;
#pragma unroll
        for (int u = 0; u < 4; ++u) dap[u] = tp.mcdA[((size_t)(4 * qd + u) * B + b) * R + r];
    }
    const float dys = (tid < 16) ? tp.mcdys[(size_t)tid * B + b] : 0.f;
    constexpr int NU_ = TMAX * 4 * R / NT, NH_ = ((TMAX + 1) * R + NT - 1) / NT;
    float ru_[NU_], rh_[NH_];
    const int Tm1 = T - 1;
#pragma unroll
    for (int u = 0; u < NU_; ++u) { const int i = tid + NT * u, t = min(i / (4 * R), min(tstar, Tm1)), j = i % (4 * R); ru_[u] = tp.gru[((size_t)t * B + b) * 4 * R + j]; }
#pragma unroll
    for (int u = 0; u < NH_; ++u) { const int i = tid + NT * u, t = min(i / R, min(tstar + 1, T)), j = i % R; rh_[u] = tp.h[((size_t)t * B + b) * R + j]; }
#pragma unroll
    for (int u = 0; u < NU_; ++u) t_gru[tid + NT * u] = ru_[u];
#pragma unroll
    for (int u = 0; u < NH_; ++u) { const int i = tid + NT * u; if (i < (TMAX + 1) * R) t_h[i] = rh_[u]; }
    s_part[tid] = (dap[0] + dap[1]) + (dap[2] + dap[3]);
    if (tid < R) s_dh[tid] = 0.f;
    if (tid < 64) {
        const float tot = dpp_wave_sum(dys);
        if (tid == 0) tp.dysum[b] = tot;
    }
    __syncthreads();
    if (tid < R) {
        const float v = (s_part[tid] + s_part[R + tid]) + (s_part[2 * R + tid] + s_part[3 * R + tid]);
        s_dA[tid] = v; tp.dA[(size_t)b * R + tid] = v;
    }
    __syncthreads();
    {
        float accy = 0.f;
#pragma unroll
        for (int i = 0; i < R / K4; ++i) accy = fmaf(y1T[i], s_dA[p4 * (R / K4) + i], accy);
        accy = lane_group_sum<K4>(accy);
        if (p4 == 0) s_dAy[k4] = accy;
    }
    for (int t = tstar + 1; t < T; ++t) {                                   // steps this sample never took (Adaptive): zero rows for k_wgrad
        const size_t row = (size_t)t * B + b;
        if (tid < 3 * R) { tp.dgi[row * 3 * R + tid] = 0.f; tp.dgh[row * 3 * R + tid] = 0.f; }
    }
    __syncthreads();
    // ---- the recurrence: [cell backward] barrier [W_hh^T dgh] barrier (k_bwd_sample)
    for (int t = tstar; t >= 0; --t) {
        const size_t row = (size_t)t * B + b;
        const float din = (t == tstar) ? s_dAy[k4] : 0.f;
        {
            const float dh = s_dh[k4] + din;
            const float* gr = t_gru + t * 4 * R;
            const float rr = gr[k4], uu = gr[R + k4], nn = gr[2 * R + k4], ghn = gr[3 * R + k4];
            const float hp = t_h[t * R + k4];
            const float dn = dh * (1.f - uu), du = dh * (hp - nn);
            const float dnp = dn * (1.f - nn * nn), dup = du * uu * (1.f - uu);
            const float drp = dnp * ghn * rr * (1.f - rr);
            float* gi = tp.dgi + row * 3 * R; float* gh = tp.dgh + row * 3 * R;
            if (p4 == 0)      { gi[k4] = drp; gh[k4] = drp; s_dgh[k4] = drp; }
            else if (p4 == 1) { gi[R + k4] = dup; gh[R + k4] = dup; s_dgh[R + k4] = dup; }
            else if (p4 == 2) { gi[2 * R + k4] = dnp; gh[2 * R + k4] = dnp * rr; s_dgh[2 * R + k4] = dnp * rr; }
        }
        __syncthreads();
        {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int i = 0; i < 3 * R / K4; i += 4) {
                const float4 dv = *reinterpret_cast<const float4*>(s_dgh + p4 * (3 * R / K4) + i);
                a0 = fmaf(whhT[i], dv.x, a0); a1 = fmaf(whhT[i + 1], dv.y, a1);
                a2 = fmaf(whhT[i + 2], dv.z, a2); a3 = fmaf(whhT[i + 3], dv.w, a3);
            }
            const float acc = lane_group_sum<K4>((a0 + a1) + (a2 + a3));
            if (p4 == 0) s_dh[k4] = __fmul_rn(s_dh[k4] + din, t_gru[t * 4 * R + R + k4]) + acc;
        }
        __syncthreads();
    }
}

}  // namespace mmg
