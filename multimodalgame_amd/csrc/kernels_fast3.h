// kernels_fast3.h -- k_conversation_fast3: the register-resident conversation of kernels_fast.h, re-cut for ONE wave per SIMD.
//
// What bounds a step of k_conversation_fast2 (measured with s_memtime / s_memrealtime stamps at 2.37 GHz): 9 barrier phases of
// ~700 cycles; each phase issues ~95 wave-instructions on each of the TWO waves of a SIMD (770 VALU per SIMD and step) of which
// 15 are FMAs -- the rest is what every wave pays per phase whatever its share of the layer: LDS operand reads, the DPP
// reduction tree, the activation, exec-mask bookkeeping of predicated stores, 64-bit tape addresses.  The phases are a
// dependent chain (w -> a -> z -> h -> A -> y -> pi -> g -> w), so the step time is (phases) x (latency + instructions one SIMD
// issues per phase).  This kernel cuts both factors:
//   * 256 threads, one wave per SIMD: a layer's per-phase overhead is paid by 4 waves instead of 8; the FMAs per lane double
//     (32 per phase) but are issued as packed v_pk_fma_f32 (16 instructions);
//   * 7 phases instead of 9: the description mixture and w_d are ONE product -- W_d (softmax(y) . desc) = softmax(y) . (desc W_d^T)
//     = softmax(y) . Dd with Dd[D, R] formed once per minibatch by k_prep (like Cd): K = D = 30 instead of
//     V = 100 then R x V, and no barrier between; the GRU's gate products and state update were one phase already;
//   * the hidden-side GRU product W_hh h_t (12 288 of a step's 53 k MACs; depends on nothing but h_t) is spread over the three
//     light phases that follow the state update, its weights parked in LDS as a per-lane spill area (conflict-free 16-byte
//     slots, read back by the lane that wrote them: no synchronisation);
//   * NO global store, address computation or predicated tape write inside the step loop: every activation is written once to
//     a per-step LDS slot (which the next phase reads anyway) and the whole tape goes out after the conversation as coalesced
//     16-byte stores; softmax(y) . desc (`dbar`, read only by the weight-gradient job of w_d) is formed after the loop from
//     the stored softmax rows.
// Lane maps (tid = 0..255):
//   P1 code_layer      row n = tid, K = 32                                   32 regs, no reduction
//   P2 binary_layer    row m = tid / 8, 8 lanes x 32 k (float4-interleaved)  32 regs, 3 DPP steps
//   P3 GRU             unit u = tid / 4, quarter q = tid % 4, gates r, u, n  24 regs (W_ih) + LDS-parked W_hh, 2 DPP steps
//   P4 y1[:, :R] | w_h rows 0..63 | 64..127, row = tid / 2, 2 lanes x 32 k   32 regs, 1 DPP step
//   P5 class logits    class d = tid / 8, 8 lanes x 8 r: w2 max(A, -Cd) + cy 16 regs, 3 DPP steps; stop head on lanes 240..255
//   P6 softmax (per wave) then g: unit r = tid / 4, 4 lanes x 8 classes of Dd  8 regs, 2 DPP steps
//   P7 w               row m = tid / 8, 8 lanes x 8 k                          8 regs, 3 DPP steps
// Same tape contract, sampling streams (injected uniforms or Philox keyed by the global sample index) and early exit as
// k_conversation_fast2; parity against the oracle: tests/test_hip_parity.py, test_hip_configs.py (every fast-shape case).
#pragma once
#include "device_utils.h"
#include "kernels_fast.h"
#include "layout.h"

namespace mmg {


// dot product of 4 * NV register weights with float4 operands read from LDS at base + stride * j: two packed accumulators
// (v_pk_fma_f32: two FMAs per issue slot), i.e. four independent chains
template <int NV>
__device__ __forceinline__ float dot4p(const float* __restrict__ w, const float* lds, int stride) {
    // NV >= 4: four packed accumulators (chains of NV / 2 dependent v_pk_fma_f32 instead of NV)
    f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f}, b01 = {0.f, 0.f}, b23 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(lds + stride * j);
        const f32x2 w01 = {w[4 * j], w[4 * j + 1]}, w23 = {w[4 * j + 2], w[4 * j + 3]};
        const f32x2 v01 = {v.x, v.y}, v23 = {v.z, v.w};
        if (NV >= 4 && (j & 1)) { b01 = __builtin_elementwise_fma(w01, v01, b01); b23 = __builtin_elementwise_fma(w23, v23, b23); }
        else { a01 = __builtin_elementwise_fma(w01, v01, a01); a23 = __builtin_elementwise_fma(w23, v23, a23); }
    }
    const f32x2 s = (a01 + a23) + (b01 + b23);
    return s.x + s.y;
}

// 16 parked weights of this lane (four float4 slots of the LDS park, written by this lane itself) times 16 floats of h:
// the loads (issued at the top of a phase) and the products (wherever the scheduler finds room)
__device__ __forceinline__ void park_load(float4 (&w)[4], float4 (&v)[4], const float4* park, int gate, int tid, const float* hq) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { w[i] = park[(gate * 4 + i) * 256 + tid]; v[i] = *reinterpret_cast<const float4*>(hq + 4 * i); }
}
__device__ __forceinline__ float park_fma(const float4 (&w)[4], const float4 (&v)[4]) {
    f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a01 = __builtin_elementwise_fma(f32x2{w[i].x, w[i].y}, f32x2{v[i].x, v[i].y}, a01);
        a23 = __builtin_elementwise_fma(f32x2{w[i].z, w[i].w}, f32x2{v[i].z, v[i].w}, a23);
    }
    const f32x2 s = a01 + a23;
    return s.x + s.y;
}
// LDS plan of k_conversation_fast3 (floats); the host asks for fast3_lds_bytes() of dynamic shared memory
struct Fast3Lds {
    static constexpr int TMAX = 16, H = 256, W = 32, R = 64;
    static constexpr int a = 0;                               // [TMAX][H]
    static constexpr int gru = a + TMAX * H;                  // [TMAX][4 R]   r | u | n | W_hn h + b_hn
    static constexpr int h = gru + TMAX * 4 * R;              // [TMAX + 1][R]
    static constexpr int g = h + (TMAX + 1) * R;              // [TMAX][R]
    static constexpr int z = g + TMAX * R;                    // [TMAX][W]
    static constexpr int pz = z + TMAX * W;
    static constexpr int w = pz + TMAX * W;                   // [TMAX + 1][W]: slot 0 = first_rec, slot t + 1 = w_t
    static constexpr int pw = w + (TMAX + 1) * W;
    static constexpr int y = pw + TMAX * W;                   // [TMAX][32]
    static constexpr int pi = y + TMAX * 32;                  // [TMAX][32]    softmax(y_t)
    static constexpr int uz = pi + TMAX * 32;                 // [TMAX][W]
    static constexpr int uw = uz + TMAX * W;
    static constexpr int e = uw + TMAX * W;                   // [4 waves][32]  wave-private softmax numerators
    static constexpr int A = e + 4 * 32;                      // [R]
    static constexpr int gh = A + R;                          // [R]            w_h h + b_h
    static constexpr int small = gh + R;                      // us[16] | ps[16] | sb[16] | mask[17] | misc[15] | sigmoid(code_bias)[32]
    static constexpr int park = small + 112;                   // [12][256] float4: W_hh fragments (after the loop: desc [D][V])
    static constexpr int total = park + 12 * 256 * 4;
};
__host__ __device__ inline int fast3_lds_bytes() { return Fast3Lds::total * 4; }

// MERGED: the launch carries k_prep's blocks as leading roles (ar.nprep of them); a compile-time switch -- a runtime branch
// around the prologue's loads makes hipcc drain them at the join
template <int H, int W, int R, int V, bool MERGED>
__global__ __launch_bounds__(256, 1) void k_conversation_fast3(Dims dm, Params P, Tape tp, ConvArgs ar) {
    constexpr int NT = 256, TMAX = Fast3Lds::TMAX;
    static_assert(H == 256 && W == 32 && R == 64 && V % 4 == 0 && V <= 128, "lane maps of k_conversation_fast3");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // Roles of the launch, in workgroup order:
    //   [0, B)                one conversation per sample
    //   [B, B + nprep)        (MERGED) k_prep's blocks (parameter constants Cd / cy / Dd / hw0, tiles of h_x): nothing they need
    //                         comes from this launch, and the sample roles spend their first ~4.5 us loading weights -- a launch
    //                         of its own (5.5 us + the boundary) behind that prologue; hand-off by (value, epoch) pairs
    //                         (prep_body<true>).  The consumers sit AHEAD of their producers here, so that they start first: the
    //                         host merges only when every sample and prep role has a CU of its own, and the spin is bounded.
    //   then                  tiles of basehx for the baselines' launch (they spin on the h_x pairs they multiply)
    //   last                  (MERGED) the closing role: bumps the minibatch counter once every consumer holds its pairs
    constexpr bool merged = MERGED;
    const uint32_t n_consumers = (uint32_t)(dm.B + ar.nbase);
    if (merged && (int)blockIdx.x >= dm.B && (int)blockIdx.x < dm.B + ar.nprep) {
        prep_body<true>(dm, P, tp, ar.desc, ar.x, ar.prep_cpb, (int)blockIdx.x - dm.B, lds, ar.train);
#ifdef MMG_TIMING
        {   // when the first class block, the first hw0 block and the first / last h_x tile are through (scripts/timeline.py)
            const int nC = (dm.D + ar.prep_cpb - 1) / ar.prep_cpb, HB = (dm.H + 63) / 64, blk = (int)blockIdx.x - dm.B;
            const int slot = blk == 0 ? 123 : blk == nC ? 127 : blk == nC + HB ? 124 : blk == ar.nprep - 1 ? 125 : -1;
            if (slot >= 0 && threadIdx.x == 0) tp.dbg[slot] = (long long)wall_clock64();
        }
#endif
        return;
    }
    if (merged && (int)blockIdx.x == ar.nprep + dm.B + ar.nbase) { prep_closing_role(tp, n_consumers, ar.train); return; }
    if ((int)blockIdx.x >= ar.nprep + dm.B) {
        const int tile = (int)blockIdx.x - ar.nprep - dm.B;
        if (!merged) { gemm_nt_tile(tile, tp.hx, H, P.p[BS_L1_W], H + W, nullptr, tp.basehx, dm.K, dm.B, dm.K, H); return; }
        // (A operand = the h_x pairs of this launch's prep roles: the loads spin until they carry this launch's epoch)
        gemm_nt_tile<false, true>(tile, tp.prepll, H, P.p[BS_L1_W], H + W, nullptr, tp.basehx, dm.K, dm.B, dm.K, H, nullptr, tp.counter[3] + 1u, tp.sync);
        if (threadIdx.x == 0) prep_consumer_arrive(tp);
        return;
    }
    typedef Fast3Lds L;
    float* const s_a = lds + L::a; float* const s_gru = lds + L::gru; float* const s_h = lds + L::h; float* const s_g = lds + L::g;
    float* const s_z = lds + L::z; float* const s_pz = lds + L::pz; float* const s_w = lds + L::w; float* const s_pw = lds + L::pw;
    float* const s_y = lds + L::y; float* const s_pi = lds + L::pi; float* const s_uz = lds + L::uz; float* const s_uw = lds + L::uw;
    float* const s_e = lds + L::e; float* const s_A = lds + L::A; float* const s_gh = lds + L::gh;
    float* const s_us = lds + L::small; float* const s_ps = s_us + 16; float* const s_sb = s_us + 32; float* const s_mask = s_us + 48;
    float* const s_misc = s_us + 65; float* const s_sig = s_us + 80;
    float4* const s_park = reinterpret_cast<float4*>(lds + L::park);

    const int b = (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = dm.B, T = dm.T, Dr = dm.D;
    const bool binary = dm.use_binary != 0, train = ar.train != 0, inject = ar.u_s != nullptr;
    MMG_STAMP(0);
#ifdef MMG_TIMING
    if (b == 0 && tid == 0) tp.dbg[4] = (long long)__builtin_readcyclecounter();
#endif
    const uint32_t mb_counter = tp.counter[0] + ((merged && train) ? 1u : 0u);      // (merged: bumped by the launch's closing role -- training launches only)
    const uint32_t ll_epoch = tp.counter[3] + 1u;                         // epoch of this launch's (value, epoch) pairs
    const uint32_t gb = (uint32_t)(dm.boff + b);
    const int tgt = ar.target ? (int)ar.target[b] : -1;     // (read here: no dependent memory round trip after the conversation)
    if (train && inject) {
        for (int i = tid; i < T * W; i += NT) {
            const int t = i / W, j = i - t * W;
            if (ar.u_z) s_uz[i] = ar.u_z[((size_t)t * B + b) * W + j];
            if (ar.u_w) s_uw[i] = ar.u_w[((size_t)t * B + b) * W + j];
        }
        if (tid < T) s_us[tid] = ar.u_s[(size_t)tid * B + b];
    }
    // ------------------------------------------------------------ weights -> registers / LDS park (once)
    // P1 code_layer: row tid
    float wc[W];
#pragma unroll
    for (int j = 0; j < W / 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(P.p[S_CODE_W] + (size_t)tid * W + 4 * j);
        wc[4 * j] = v.x; wc[4 * j + 1] = v.y; wc[4 * j + 2] = v.z; wc[4 * j + 3] = v.w;
    }
    const float bc = P.p[S_CODE_B][tid];
    float hw0 = 0.f, hx = 0.f;
    if (!merged) { hw0 = tp.hw0[tid]; hx = tp.hx[(size_t)b * H + tid]; }
    // P2 binary_layer (m2, k2): float4 index j * 8 + k2 of row m2;  P7 w: floats k2 * 8 .. + 7 of row m2
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
    // P3 GRU (u3, q3): W_ih rows gate * R + u3, floats q3 * 8 .. + 7; W_hh floats q3 * 16 .. + 15 -> LDS park
    const int u3 = tid >> 2, q3 = tid & 3;
    float wih[24];
    float4 whh_tmp[12];                            // (parked in LDS below, once every load of the prologue is in flight)
#pragma unroll
    for (int gt = 0; gt < 3; ++gt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(P.p[R_WIH] + (size_t)(gt * R + u3) * W + q3 * 8 + 4 * j);
            wih[8 * gt + 4 * j] = v.x; wih[8 * gt + 4 * j + 1] = v.y; wih[8 * gt + 4 * j + 2] = v.z; wih[8 * gt + 4 * j + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            whh_tmp[gt * 4 + i] = *reinterpret_cast<const float4*>(P.p[R_WHH] + (size_t)(gt * R + u3) * R + q3 * 16 + 4 * i);
    }
    const float b_r = P.p[R_BIH][u3] + P.p[R_BHH][u3], b_u = P.p[R_BIH][R + u3] + P.p[R_BHH][R + u3];
    const float b_in = P.p[R_BIH][2 * R + u3], b_hn = P.p[R_BHH][2 * R + u3];
    // P4 (row4, half4): y1[:, :R] rows 0..63, w_h rows 64..127; float4 index j * 2 + half4
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
    // P5 (d5, r5): -Cd[d5][r5 * 8 .. + 7], w2[r5 * 8 .. + 7]; stop head on lanes 240..255: w_s[(tid - 240) * 4 .. + 3]
    const int d5 = tid >> 3, r5 = tid & 7;
    float ncd[8], w2[8];
    {
        const float* crow = tp.Cd + (size_t)min(d5, Dr - 1) * R + r5 * 8;
        float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
        if (!merged) { c0 = *reinterpret_cast<const float4*>(crow); c1 = *reinterpret_cast<const float4*>(crow + 4); }
        ncd[0] = -c0.x; ncd[1] = -c0.y; ncd[2] = -c0.z; ncd[3] = -c0.w; ncd[4] = -c1.x; ncd[5] = -c1.y; ncd[6] = -c1.z; ncd[7] = -c1.w;
        const float4 q0 = *reinterpret_cast<const float4*>(P.p[R_Y2_W] + r5 * 8), q1 = *reinterpret_cast<const float4*>(P.p[R_Y2_W] + r5 * 8 + 4);
        w2[0] = q0.x; w2[1] = q0.y; w2[2] = q0.z; w2[3] = q0.w; w2[4] = q1.x; w2[5] = q1.y; w2[6] = q1.z; w2[7] = q1.w;
    }
    float cy5 = merged ? 0.f : tp.cy[min(d5, Dr - 1)];  // b_y2 + sum_r w2[r] Cd[d][r]  (k_prep)
    const float4 ws4 = *reinterpret_cast<const float4*>(P.p[R_S_W] + (tid & 15) * 4);
    const float bs = P.p[R_S_B][0];
    // P6 (u3, q3): Dd[q3 * 8 + k][u3]
    float dd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int d = q3 * 8 + k; dd[k] = (d < Dr && !merged) ? tp.Dd[(size_t)d * R + u3] : 0.f; }
    if (tid < W) s_sig[tid] = fsigmoid(P.p[S_CODE_BIAS][tid]);
    // What the prep roles of this launch produce, as (value, epoch) pairs (device_utils.h: st_ll): loaded once this role's
    // weights are in (below) -- the producers need ~3.5 us, the ~400 KB of weight loads ~4.5 -- one more memory round trip; a pair
    // of another epoch -> load again.  (Issued behind the weight loads instead, 18 agent-scope loads per lane in the middle of
    // that stream: prologue 8.8 us against 6.2.)
    const size_t i_hw = prepll_hw0(dm) + tid, i_hx = prepll_hx(dm) + (size_t)b * H + tid;
    const size_t i_cd = prepll_cd(dm) + (size_t)min(d5, Dr - 1) * R + r5 * 8, i_dd = prepll_dd(dm) + u3;
    unsigned long long u[18];
    auto load_pairs = [&]() {
        u[0] = ld_ll(tp.prepll, i_hw); u[1] = ld_ll(tp.prepll, i_hx);
#pragma unroll
        for (int k = 0; k < 8; ++k) u[2 + k] = ld_ll(tp.prepll, i_cd + k);
#pragma unroll
        for (int k = 0; k < 8; ++k) u[10 + k] = ld_ll(tp.prepll, i_dd + (size_t)min(q3 * 8 + k, Dr - 1) * R);
    };
    if (train && !inject) {                        // Philox draws of the whole conversation, while the weight loads are in flight
        for (int i = tid; i < T * W; i += NT) {
            const int t = i / W, j = i - t * W;
            const uint32_t e = (uint32_t)((t * dm.Bg + gb) * W + j);
            s_uz[i] = philox_uniform(ar.seed, e, mb_counter, 0u);
            s_uw[i] = philox_uniform(ar.seed, e, mb_counter, 2u);
        }
        if (tid < T) s_us[tid] = philox_uniform(ar.seed, (uint32_t)(tid * dm.Bg + gb), mb_counter, 1u);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) s_park[i * NT + tid] = whh_tmp[i];
    if (merged) {
        int spins = 0;
        for (;;) {
            load_pairs();
            bool fresh = true;
#pragma unroll
            for (int k = 0; k < 18; ++k) fresh = fresh && ll_fresh(u[k], ll_epoch);
            if (!__any(!fresh)) break;
            if (++spins > (1 << 16)) { if (lane == 0) __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        MMG_STAMP(126);
#ifdef MMG_TIMING
        if (lane == 0) atomicAdd((unsigned long long*)&tp.dbg[118], (unsigned long long)spins);      // reloads of the pairs, summed over waves (scripts/timeline.py)
#endif
        hw0 = ll_value(u[0]); hx = ll_value(u[1]);
#pragma unroll
        for (int k = 0; k < 8; ++k) { ncd[k] = -ll_value(u[2 + k]); dd[k] = (q3 * 8 + k < Dr) ? ll_value(u[10 + k]) : 0.f; }
        // cy[d] = b_y2 + sum_r w2[r] Cd[d][r] from this lane group's own eighths (k_prep's value reaches later launches only)
        float part = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) part = fmaf(w2[k], -ncd[k], part);
        cy5 = dpp_group_sum<8>(part) + P.p[R_Y2_B][0];
        if (tid == 0) prep_consumer_arrive(tp);
    }
    MMG_STAMP(1);
    // ------------------------------------------------------------ conversation state
    if (tid < R) s_h[tid] = 0.f;
    if (tid < W) s_w[tid] = dm.first_rec;
    if (tid == 0) s_mask[0] = 1.f;
    float m_run = 1.f, sprod = 1.f; int t_out = -1;        // (meaningful on lane 240 only: the stop head's lane)
    float ghp_r = 0.f, ghp_u = 0.f, ghn = b_hn;            // W_hh h_0 = 0: partial gate sums of this lane's quarter, reduced n gate
    __syncthreads();

    int t_done = T, w_done = T;                         // steps executed / steps whose receiver message was formed
    MMG_STAMP(2);
    // Every phase below is written as  [all LDS loads] [branch-free arithmetic: uniform mode flags become selects] [stores]:
    // a predicated store or a mode branch in the middle of a phase splits the basic block, and the scheduler then issues the
    // loads of the next independent chain only after the previous chain has drained (measured in P5: the stop head's h read
    // waited for the class chain's store).
    const float fixedm = dm.fixed ? 1.f : 0.f;
    const bool sprodm = dm.s_prob_prod != 0;
    for (int t = 0; t < T; ++t) {
        MMG_STAMP(8 + 10 * t + 9);
        // ===== P1 sender: a = tanh(h_x + code_layer(c))
        {
            const float hw = (t > 0) ? bc + dot4p<W / 4>(wc, s_w + t * W, 4) : hw0;
            s_a[t * H + tid] = ftanh(hx + hw);
        }
        __syncthreads(); MMG_STAMP(8 + 10 * t + 0);
        // ===== P2 sender logits + sample
        {
            const float uz = s_uz[t * W + m2];
            const float lz = dpp_group_sum<8>(dot4p<8>(wb, s_a + t * H + k2 * 4, 32)) + bb;
            const float ps = fsigmoid(lz);
            const float pp = binary ? ps : 0.f;
            const float zz = binary ? (train ? ((uz < ps) ? 1.f : 0.f) : rintf(ps)) : lz;
            if (k2 == 0) { s_z[t * W + m2] = zz; s_pz[t * W + m2] = pp; }
        }
        __syncthreads(); MMG_STAMP(8 + 10 * t + 1);
        // ===== P3 GRU cell: gate pre-activations of this lane's quarter, 4-lane sums, state update
        {
            const float* zq = s_z + t * W + q3 * 8;
            const float4 z0 = *reinterpret_cast<const float4*>(zq), z1 = *reinterpret_cast<const float4*>(zq + 4);
            const float h_old = s_h[t * R + u3];
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
            s_gru[t * 4 * R + q3 * R + u3] = (q3 == 0) ? rr : (q3 == 1) ? uu : (q3 == 2) ? nn : ghn;
            if (q3 == 0) s_h[(t + 1) * R + u3] = hv;
        }
        __syncthreads(); MMG_STAMP(8 + 10 * t + 2);
        const float* hn = s_h + (t + 1) * R;
        // ===== P4 heads on h: A = y1[:, :R] h (rows 0..63), w_h h + b_h (rows 64..127)
        {
            const float acc = dpp_group_sum<2>(dot4p<8>(w4, hn + half4 * 4, 8)) + b4;
            if (half4 == 0) { if (row4 < R) s_A[row4] = acc; else s_gh[row4 - R] = acc; }
        }
        __syncthreads(); MMG_STAMP(8 + 10 * t + 3);
        // ===== P5 class logits | stop head (every wave computes it; lane 240 -- row 3 of wave 3, whose 16 lanes cover h -- keeps it) |
        //       hidden-side product of the r gate for the next step
        {
            const float4 a0 = *reinterpret_cast<const float4*>(s_A + r5 * 8), a1 = *reinterpret_cast<const float4*>(s_A + r5 * 8 + 4);
            const float4 hv = *reinterpret_cast<const float4*>(hn + (tid & 15) * 4);
            const float us_t = s_us[t];
            float4 pk[4], hq[4];
            park_load(pk, hq, s_park, 0, tid, hn + q3 * 16);
            float acc0 = w2[0] * fmax_nn(a0.x, ncd[0]), acc1 = w2[1] * fmax_nn(a0.y, ncd[1]);
            acc0 = fmaf(w2[2], fmax_nn(a0.z, ncd[2]), acc0); acc1 = fmaf(w2[3], fmax_nn(a0.w, ncd[3]), acc1);
            acc0 = fmaf(w2[4], fmax_nn(a1.x, ncd[4]), acc0); acc1 = fmaf(w2[5], fmax_nn(a1.y, ncd[5]), acc1);
            acc0 = fmaf(w2[6], fmax_nn(a1.z, ncd[6]), acc0); acc1 = fmaf(w2[7], fmax_nn(a1.w, ncd[7]), acc1);
            const float yv = dpp_group_sum<8>(acc0 + acc1) + cy5;
            const float sv = dpp_group_sum<16>(fmaf(ws4.x, hv.x, fmaf(ws4.y, hv.y, fmaf(ws4.z, hv.z, ws4.w * hv.w))));
            const float p = fsigmoid(sv + bs);
            const float prod = sprodm ? sprod * p : p;
            sprod = train ? sprod : prod;
            const float sbit = train ? ((us_t < p) ? 1.f : 0.f) : rintf(prod);
            const float m_next = fminf(m_run, sbit);
            const bool last = (t == T - 1);
            const bool take = (fixedm != 0.f) ? last : (t_out < 0 && (m_next == 0.f || last));
            t_out = take ? t : t_out;
            m_run = m_next;
            ghp_r = park_fma(pk, hq);
            if (r5 == 0) s_y[t * 32 + d5] = (d5 < Dr) ? yv : -3.0e38f;
            if (tid == 240) { s_ps[t] = p; s_sb[t] = sbit; s_mask[t + 1] = m_next; }
        }
        __syncthreads(); MMG_STAMP(8 + 10 * t + 4);
        if (!ar.run_all && !dm.fixed && train && s_mask[t + 1] == 0.f) { t_done = t + 1; w_done = t; break; }
        // ===== P6 softmax of the wave's own copy of y (lanes 0..31), then g = tanh(w_h h + b_h + softmax(y) . Dd)
        {
            const float yv = s_y[t * 32 + (lane & 31)];
            const float ghu = s_gh[u3];
            float4 pk[4], hq[4];
            park_load(pk, hq, s_park, 1, tid, hn + q3 * 16);
            float mx = fmax_nn(yv, dpp_f<MMG_DPP_QUAD_1032>(yv)); mx = fmax_nn(mx, dpp_f<MMG_DPP_QUAD_2301>(mx));
            mx = fmax_nn(mx, dpp_f<MMG_DPP_ROW_HALF_MIRROR>(mx)); mx = fmax_nn(mx, dpp_f<MMG_DPP_ROW_MIRROR>(mx));   // per 16-lane row
            const float m0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 0));
            const float m1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 16));
            const float e = __expf(yv - fmaxf(m0, m1));                      // (classes beyond Dr hold -3e38: e = 0; lanes 32..63 mirror 0..31)
            const float rs = dpp_group_sum<16>(e);
            const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs), 0));
            const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs), 16));
            const float inv = __builtin_amdgcn_rcpf(s0 + s1);
            if (lane < 32) s_e[wave * 32 + lane] = e;
            __builtin_amdgcn_wave_barrier();
            const float mix = dpp_group_sum<4>(dot4p<2>(dd, s_e + wave * 32 + q3 * 8, 4));
            const float gv = ftanh(fmaf(mix, inv, ghu));
            ghp_u = park_fma(pk, hq);
            if (q3 == 0) s_g[t * R + u3] = gv;
            if (tid < 32) s_pi[t * 32 + tid] = e * inv;
        }
        __syncthreads(); MMG_STAMP(8 + 10 * t + 5);
        // ===== P7 receiver message
        {
            const float uw = s_uw[t * W + m2];
            float4 pk[4], hq[4];
            park_load(pk, hq, s_park, 2, tid, hn + q3 * 16);
            const float lw = dpp_group_sum<8>(dot4p<2>(ww, s_g + t * R + k2 * 8, 4)) + bw;
            const float ps = fsigmoid(lw);
            const float pp = binary ? ps : 0.f;
            const float wv = binary ? (train ? ((uw < ps) ? 1.f : 0.f) : rintf(ps)) : lw;
            ghn = dpp_group_sum<4>(park_fma(pk, hq)) + b_hn;
            if (k2 == 0) { s_w[(t + 1) * W + m2] = wv; s_pw[t * W + m2] = pp; }
        }
        __syncthreads(); MMG_STAMP(8 + 10 * t + 6);
    }
    __syncthreads();
    MMG_STAMP(3);
#ifdef MMG_TIMING
    if (b == 0 && tid == 0) tp.dbg[5] = (long long)__builtin_readcyclecounter();
#endif
    // ------------------------------------------------------------ output selection / reward / top-k: wave 3, whose lane 48 (tid 240)
    // holds the output step in a register -- no LDS round trip
    if (wave == 3) {
        const int tstar = dm.fixed ? (T - 1) : __builtin_amdgcn_readlane(t_out, 48);
        const float o = (lane < 32) ? s_y[tstar * 32 + lane] : -3.0e38f;
        const float mx = dpp_wave_max(o);
        const float e = (lane < Dr) ? __expf(o - mx) : 0.f;
        const float lse = mx + flog(dpp_wave_sum(e));
        const float dt = (tgt >= 0) ? (__shfl(o, tgt, 64) - lse) : 0.f;
        const float ld = o - lse;
        if (lane < Dr) {
            tp.outp[(size_t)b * Dr + lane] = o;
            tp.dist[(size_t)b * Dr + lane] = ld;
            tp.sm[(size_t)b * Dr + lane] = __expf(ld);
        }
        const float above = dpp_wave_sum((lane < Dr && tgt >= 0 && ld > dt) ? 1.f : 0.f);
        if (lane == 0) {
            tp.tstar[b] = tstar;
            tp.logs[b] = dt;
            tp.hit[b] = (tgt >= 0 && above < (float)dm.top_k) ? 1 : 0;
        }
    }
    MMG_STAMP(120);
    // ------------------------------------------------------------ log-likelihood / neg-entropy sums of all steps (model.py:908-922):
    // step tt = tid / 16, bits 2 jj, 2 jj + 1 (jj = tid % 16): one pass, sums inside a 16-lane DPP row
    if (binary) {
        const int tt = tid >> 4, jj = tid & 15;
        const float2 pz2 = *reinterpret_cast<const float2*>(s_pz + tt * W + 2 * jj), qz2 = *reinterpret_cast<const float2*>(s_z + tt * W + 2 * jj);
        const float2 pw2 = *reinterpret_cast<const float2*>(s_pw + tt * W + 2 * jj), qw2 = *reinterpret_cast<const float2*>(s_w + (tt + 1) * W + 2 * jj);
        auto terms = [](float p, float q, float& lp, float& ne) {
            const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
            lp += q * l1 + (1.f - q) * l0; ne += p * l1 + (1.f - p) * l0;
        };
        float lz = 0.f, nz = 0.f, lw = 0.f, nw = 0.f;
        terms(pz2.x, qz2.x, lz, nz); terms(pz2.y, qz2.y, lz, nz);
        terms(pw2.x, qw2.x, lw, nw); terms(pw2.y, qw2.y, lw, nw);
        lz = dpp_group_sum<16>(lz); nz = dpp_group_sum<16>(nz); lw = dpp_group_sum<16>(lw); nw = dpp_group_sum<16>(nw);
        if (jj == 0 && tt < t_done) { tp.lp_z[(size_t)tt * B + b] = lz; tp.ne_z[(size_t)tt * B + b] = nz; }
        if (jj == 0 && tt < w_done) { tp.lp_w[(size_t)tt * B + b] = lw; tp.ne_w[(size_t)tt * B + b] = nw; }
    }
    if (tid < t_done) {
        const float p = s_ps[tid], sb = s_sb[tid];
        const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
        tp.lp_s[(size_t)tid * B + b] = sb * l1 + (1.f - sb) * l0;
        tp.ne_s[(size_t)tid * B + b] = p * l1 + (1.f - p) * l0;
        tp.s[(size_t)tid * B + b] = sb; tp.ps[(size_t)tid * B + b] = p;
    }
    if (tid <= t_done) tp.mask[(size_t)tid * B + b] = (uint8_t)(s_mask[tid] != 0.f);
    MMG_STAMP(121);
    // ------------------------------------------------------------ the tape, coalesced (rows t < t_done; g / w / pw / dbar: t < w_done).
    // Fixed trip counts, LDS reads unconditional (inside the plan), only the stores predicated: the reads of a block issue together.
    {
        float4 va[4], vg[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int i4 = tid + NT * r; va[r] = *reinterpret_cast<const float4*>(s_a + 4 * i4); vg[r] = *reinterpret_cast<const float4*>(s_gru + 4 * i4); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i4 = tid + NT * r, t = i4 >> 6, c = i4 & 63;
            if (t < t_done) {
                *reinterpret_cast<float4*>(tp.a + ((size_t)t * B + b) * H + 4 * c) = va[r];
                *reinterpret_cast<float4*>(tp.gru + ((size_t)t * B + b) * 4 * R + 4 * c) = vg[r];
            }
        }
        float4 vh[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) vh[r] = *reinterpret_cast<const float4*>(s_h + 4 * min(tid + NT * r, (TMAX + 1) * (R / 4) - 1));
        const float4 vgg = *reinterpret_cast<const float4*>(s_g + 4 * tid);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i4 = tid + NT * r, t = i4 >> 4, c = i4 & 15;
            if (t <= t_done && i4 < (TMAX + 1) * (R / 4)) *reinterpret_cast<float4*>(tp.h + ((size_t)t * B + b) * R + 4 * c) = vh[r];
        }
        { const int t = tid >> 4, c = tid & 15; if (t < w_done) *reinterpret_cast<float4*>(tp.g + ((size_t)t * B + b) * R + 4 * c) = vgg; }
        if (tid < TMAX * (W / 4)) {
            const int t = tid >> 3, c = tid & 7;
            const size_t o = ((size_t)t * B + b) * W + 4 * c;
            const float4 vz = *reinterpret_cast<const float4*>(s_z + 4 * tid), vpz = *reinterpret_cast<const float4*>(s_pz + 4 * tid);
            const float4 cv = *reinterpret_cast<const float4*>(s_w + 4 * tid);         // slot t: what the sender read at step t
            const float4 vw = *reinterpret_cast<const float4*>(s_w + W + 4 * tid), vpw = *reinterpret_cast<const float4*>(s_pw + 4 * tid);
            const float4 sg = *reinterpret_cast<const float4*>(s_sig + 4 * c);
            if (t < t_done) {
                *reinterpret_cast<float4*>(tp.z + o) = vz;
                if (binary) *reinterpret_cast<float4*>(tp.pz + o) = vpz;
                *reinterpret_cast<float4*>(tp.zr + o) = cv;
                *reinterpret_cast<float4*>(tp.c + o) = make_float4(t == 0 ? sg.x : cv.x, t == 0 ? sg.y : cv.y, t == 0 ? sg.z : cv.z, t == 0 ? sg.w : cv.w);   // model.py:199 (component-wise: a ?: over float4 values goes through scratch)
            }
            if (t < w_done) {
                *reinterpret_cast<float4*>(tp.w + o) = vw;
                if (binary) *reinterpret_cast<float4*>(tp.pw + o) = vpw;
                // softmax(y_t): dbar = softmax(y) . desc is formed by trailing workgroups of the backward launch (kernels_fast.h: dbar_role)
                *reinterpret_cast<float4*>(tp.pi + ((size_t)t * B + b) * 32 + 4 * c) = *reinterpret_cast<const float4*>(s_pi + 4 * tid);
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = tid + NT * r, t = i >> 5, d = i & 31;
            const float yv = s_y[i];
            if (t < t_done && d < Dr) tp.y[((size_t)t * B + b) * Dr + d] = yv;
        }
    }
    MMG_STAMP(122);
    MMG_STAMP(6);
}

}  // namespace mmg
