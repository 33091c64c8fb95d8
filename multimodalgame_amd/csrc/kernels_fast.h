// kernels_fast.h -- register-resident specialisations of the two per-sample kernels for the
// "small agent" shapes of BASELINE configs 1-3 (H=256, W=32, R=64, V=100, D<=32).
//
// Why: at these shapes one exchange step of one sample is ~107 k FMAs over 53.5 k weights.  The
// generic kernel (kernels_fwd.h) re-reads every weight from L2 at every step and pays one L2 round
// trip per layer on a dependent chain (~34 us per step measured).  Here the workgroup of a sample loads
// the per-step weights ONCE into registers, every layer with a fixed lane->weight mapping chosen at compile
// time, and then runs the T steps with activations in LDS: no global load sits on the critical path of the
// recurrence; tape stores are fire-and-forget.  The lane mappings are listed at each kernel.
// This file holds the BACKWARD kernel of that shape (k_bwd_conv_fast) and k_bas_stats; the forward conversation is
// kernels_fast3.h (one wave per SIMD; rounds 1-3's 512-thread k_conversation_fast2 was deleted in round 6).
#pragma once
#include "device_utils.h"
#include "kernels_fwd.h"
#include "layout.h"

namespace mmg {

template <int H, int W, int R, int V, int D>
struct FastDims {
    static constexpr int NT = 256;
    static constexpr bool ok = (H == NT) && (W == 32) && (R == 64) && (V % 4 == 0) && (V <= NT) && (D <= 32) && (3 * R <= NT);
};

// Hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp): absolute error ~1e-7 on
// sigmoid / tanh / log -- three orders of magnitude inside the 1e-4 parity bar.
__device__ __forceinline__ float fsigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }
__device__ __forceinline__ float flog(float x) { return __logf(x); }

// d loss / d logit of one Bernoulli unit (kernels_bwd.h: bit_seed) with hardware rcp / log
__device__ __forceinline__ float bit_seed_fast(float q, float p, float wh, float ce) {
    const float pe = p + MMG_EPS, qe = 1.f - p + MMG_EPS;
    const float rp = __builtin_amdgcn_rcpf(pe), rq = __builtin_amdgcn_rcpf(qe);
    float dLdp = -wh * (q * rp - (1.f - q) * rq);
    if (ce != 0.f) dLdp += ce * (flog(pe) + p * rp - flog(qe) - (1.f - p) * rq);
    return dLdp * p * (1.f - p);
}

// dot product of NV float4 register quads with float4 operands read from LDS at base + stride*j,
// as four independent FMA chains (x, y, z, w components) to keep the single wave per SIMD issuing.
template <int NV>
__device__ __forceinline__ float dot4(const float* __restrict__ w, const float* lds, int stride4) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(lds + stride4 * j);
        a0 = fmaf(w[4 * j], v.x, a0); a1 = fmaf(w[4 * j + 1], v.y, a1);
        a2 = fmaf(w[4 * j + 2], v.z, a2); a3 = fmaf(w[4 * j + 3], v.w, a3);
    }
    return (a0 + a1) + (a2 + a3);
}

template <int N>
__device__ __forceinline__ float lane_group_sum(float v) { return dpp_group_sum<N>(v); }

#ifdef MMG_TIMING
#define MMG_STAMP(slot) do { if (b == 0 && tid == 0) tp.dbg[(slot)] = (long long)wall_clock64(); } while (0)
#else
#define MMG_STAMP(slot) do {} while (0)
#endif

}  // namespace mmg

namespace mmg {

// ---------------------------------------------------------------------------------------------
// k_bwd_conv_fast: reverse-time pass of one sample with the TRANSPOSED weight fragments the
// recurrence needs resident in registers (w^T: 8, w_h^T: 16, W_hh^T: 48, binary_layer^T: 32 per lane).
// Same math, tape contract and zero-filling as k_bwd_conv (kernels_bwd.h).  The dh-independent work of all steps runs in
// three sweeps before the recurrence; a reverse step is two phases (cell backward | W_hh^T dgh).
// ---------------------------------------------------------------------------------------------
// MERGED: the grid starts with `n_stats` one-wave statistics roles (k_stats' pairs); the sample roles load their
// weights and forward tape while those run and wait for them right before the loss coefficients (device_utils.h).
// MERGE_DC: the grid ends with D class roles (k_dC).  Their inputs (dy, A*) exist after the FIRST reverse step of every
// sample, so they run concurrently with the rest of the recurrence instead of after it.
// (Rejected, measured: every sample role deriving the coefficients itself from the score partials -- in waves 0-3:
//  +64 live registers, AGPR spills, +4 us; in a dedicated fifth wave: +8 us, it outlasts the weight prologue.)
// dbar = softmax(y) . desc (model.py:442-449) for 16 (step, sample) rows, on the matrix cores: [16, 32] x [32, V].  The forward
// kernel of kernels_fast3.h folds the description product onto the classes (Dd) and never forms dbar; only the weight-gradient job
// of w_d reads it, so it is made HERE, by workgroups the backward launch carries along on idle CUs (nothing in this launch depends
// on them; k_wgrad is the next launch).  Dead rows hold stale softmax rows (finite): their dgpre is zero / they are not in the row list.
template <int V>
__device__ __forceinline__ void dbar_role(const Dims& dm, const Tape& tp, int tile) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    const int rows = dm.T * dm.B, r0 = tile * 16, Dr = dm.D;
    const float* prow = tp.pi + (size_t)min(r0 + i, rows - 1) * 32;
    float a[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) a[ks] = prow[4 * ks + q];
    constexpr int NTILE = (V + 15) / 16;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int nt = wave + 4 * u;
        if (nt >= NTILE) break;                                              // (wave-uniform)
        const int n = nt * 16 + i;
        float bv[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { const int d = 4 * ks + q; bv[ks] = (d < Dr && n < V) ? tp.descc[(size_t)min(d, Dr - 1) * V + min(n, V - 1)] : 0.f; }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) acc = mfma16(a[ks], bv[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int row = r0 + 4 * q + r; if (row < rows && n < V) tp.dbar[(size_t)row * V + n] = acc[r]; }
    }
}

// k_bas_stats: the phased (data-parallel) step's baselines AND batch statistics in one launch -- mmg_loss_stats of the
// register-resident path.  [n_stats statistics roles (4 (stream, step) pairs each, k_stats' pairs)] [baseline roles: k_baselines3's
// body per (16 live rows, 64 hidden units, baseline)]; the statistics roles spin on the baseline roles' partial-score pairs as in the
// fused step's backward launch.  13 spinning workgroups ahead of their producers: the host selects it with CUs to spare only.
__global__ __launch_bounds__(MMG_BLOCK) void k_bas_stats(Dims dm, Params P, Tape tp, int n_stats) {
    if ((int)blockIdx.x < n_stats) {            // (the partial scores arrive as (value, epoch) pairs: combine_score_ll spins on them)
        stats_pairs<false, true, true>(dm, P, tp, 1, (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), n_stats * 4, tp.counter[3]);
        return;
    }
    const int idx = (int)blockIdx.x - n_stats, npb = (dm.K + 63) / 64;
    const int window = idx / (2 * npb), rem = idx - window * 2 * npb, which = rem / npb, byi = rem - which * npb;
    baselines3_body<true>(dm, P, tp, window, byi, which, npb);
}

template <int H, int W, int R, int V, int D, bool MERGED, bool MERGE_DC>
__global__ __launch_bounds__(256, 1) void k_bwd_conv_fast(Dims dm, Params P, Tape tp, const int64_t* __restrict__ target, int n_stats,
                                                          int zero_dead, int n_dbar, int n_bas) {
    constexpr int NT = 256, K4 = NT / R;          // 4 lanes per output unit of the R-wide transposed products
    constexpr int TMAX = 16;
    static_assert(FastDims<H, W, R, V, D>::ok, "unsupported fast shape");
    __shared__ float s_coef[7 * 64];
    __shared__ __attribute__((aligned(16))) float s_dgh[2][3 * R];   // gate gradients of a step, double-buffered (ONE barrier per reverse step)
    __shared__ __attribute__((aligned(16))) float s_dy[32], s_A[R], s_dA[R];
    __shared__ float s_dls[TMAX], s_cf[4 * TMAX], s_dAy[R];
    __shared__ __attribute__((aligned(16))) float s_dhin[TMAX * R];
    // forward tape of this sample, loaded once: the time loop then issues stores only (a load inside it
    // would make its s_waitcnt vmcnt drain all outstanding delta-tape stores, ~1-2 us per wait on gfx950)
    __shared__ __attribute__((aligned(16))) float t_msg[4 * TMAX * W], t_g[TMAX * R];     // t_w | t_pw | t_z | t_pz
    float* t_w = t_msg; float* t_pw = t_msg + TMAX * W; float* t_z = t_msg + 2 * TMAX * W; float* t_pz = t_msg + 3 * TMAX * W;
    static_assert(2 * TMAX * W == TMAX * R, "a pair of message arrays holds one [16][R] tile");
    __shared__ float t_gru[TMAX * 4 * R], t_h[(TMAX + 1) * R], t_a[TMAX * H];
    __shared__ float s_statv[27 * TMAX];                 // the statistics roles' pair table, values only (MERGED)
    // MFMA A operands of the sweeps, rows PADDED (strides 34 / 66 floats): a lane (fi, fq) reads [fi * stride + 4 ks + fq], so with
    // the natural strides 32 / 64 all sixteen rows fi hit ONE LDS bank (16-way conflicts: 2.4 + 1.9 us of sweeps in rounds 2-4; round 5,
    // found in kernels_game.h: 2.2 + 0.64)
    constexpr int SW = 34, SR = 66;
    __shared__ float s_sbas[4 * TMAX * SW];              // seed bases: w1 | w2 | z1 | z2
    __shared__ float s_G1[TMAX * SR], s_G2p[TMAX * SR];  // dgpre bases
    // Workgroup roles.  n_bas == 0: [statistics n_stats][samples B][classes D][dbar n_dbar].
    // n_bas > 0 (the baselines' forward pass rides in this launch, kernels_fwd.h: baselines3_body): [samples B][statistics n_stats]
    // [baselines n_bas][classes D][dbar n_dbar] -- the sample roles start at once (their first ~10 us need no statistics), the
    // baseline roles fill the other CUs, the statistics roles wait for them.  (Consumers ahead of producers: the host only selects
    // this layout when B + n_stats workgroups leave CUs free for the producers.)
    int vbx = (int)blockIdx.x;                          // index in the n_bas == 0 layout
    if (MERGED && n_bas > 0) {
        const int bx = (int)blockIdx.x, B0 = dm.B;
        if (bx < B0) vbx = n_stats + bx;
        else if (bx < B0 + n_stats) vbx = bx - B0;
        else if (bx < B0 + n_stats + n_bas) {
            const int idx = bx - B0 - n_stats, npb = (dm.K + 63) / 64;
            const int window = idx / (2 * npb), rem = idx - window * 2 * npb, which = rem / npb, byi = rem - which * npb;
            baselines3_body<true>(dm, P, tp, window, byi, which, npb);
            return;
        } else vbx = bx - n_bas;
    }
    if (MERGED && vbx < n_stats) {                      // four pairs per workgroup (one per wave): few releasing workgroups
#ifdef MMG_TIMING
        if (vbx == 0 && threadIdx.x == 0) tp.dbg[128 + 48] = (long long)wall_clock64();
#endif
        if (n_bas > 0) {
            stats_pairs<true, true, true>(dm, P, tp, 1, vbx * 4 + (int)(threadIdx.x >> 6), n_stats * 4, tp.counter[3]);
        } else
            stats_pairs<true>(dm, P, tp, 1, vbx * 4 + (int)(threadIdx.x >> 6), n_stats * 4, tp.counter[3]);
#ifdef MMG_TIMING
        if (vbx == 0 && threadIdx.x == 0) tp.dbg[128 + 49] = (long long)wall_clock64();
#endif
        // (no counter: the sample roles spin on the (value, epoch) pairs stats_pairs wrote beside the statistics)
#ifdef MMG_TIMING
        if (vbx == 0 && threadIdx.x == 0) tp.dbg[128 + 50] = (long long)wall_clock64();
#endif
        return;
    }
    if ((int)blockIdx.x >= (int)gridDim.x - n_dbar) {      // trailing workgroups: dbar tiles (k_conversation_fast3 left softmax rows only)
        dbar_role<V>(dm, tp, (int)blockIdx.x - ((int)gridDim.x - n_dbar));
        return;
    }
    if (MERGE_DC && vbx >= n_stats + dm.B) {
        float* s_c = t_a; float* s_p = t_a + 256;
        if (vbx == n_stats + dm.B && threadIdx.x < 64) build_row_map(dm, tp);   // while waiting: rows k_wgrad will reduce over
        role_wait<8, false>(tp.sync, 1, (uint32_t)dm.B, (uint32_t)dm.D);
        dC_class<true>(dm, P, tp, vbx - n_stats - dm.B, s_c, s_p);
        return;
    }
    const int b = vbx - n_stats, tid = threadIdx.x, lane = tid & 63;
    const int B = dm.B, T = dm.T;
    const int Dr = dm.D;                                // classes of this run (<= D, the compile-time capacity)
    const bool binary = dm.use_binary != 0;
#ifdef MMG_TIMING
#define MMG_BSTAMP(slot) do { if (b == 0 && tid == 0) tp.dbg[128 + (slot)] = (long long)wall_clock64(); } while (0)
#else
#define MMG_BSTAMP(slot) do {} while (0)
#endif
    MMG_BSTAMP(0);
    // ---- prologue: issue EVERY independent global load before the first dependent use (one memory round trip for the
    // weight fragments, overlapping the tstar -> tape-preload chain), then stage the tape into LDS.
    CoefRegs creg;
    if (!MERGED) creg = coef_load<false>(dm, tp.stats);        // statistics first: they gate the first arithmetic of the kernel
    const int tstar = tp.tstar[b];
    const int tgt = (int)target[b];
    const float L = tp.logs[b];
    // transposed fragments: output unit k4 = tid/4, reduction slice p4 = tid%4 (n = p4 + 4*i)
    const int k4 = tid / K4, p4 = tid % K4;
    // (the five transposed weight fragments below: 30 lane-consecutive float4 loads from k_prep's repacked copy, tape.wrep --
    //  kernels_fwd.h: prep_repack -- instead of 120 strided dword loads from the parameter buffer)
    float whhT[3 * R / K4];
    const float wsk = P.p[R_S_W][k4];
    // the three dh-independent transposed products run for all steps at once on the matrix cores ([16 steps, K] x [K, N],
    // v_mfma_f32_16x16x4_f32): B fragments, lane (i = lane & 15, q = lane >> 4) holds Wm[4 ks + q][n0 + i] per k-step ks.
    // W_w^T: wave w owns units 16 w ..; W_h^T likewise; binary_layer^T: wave w owns columns 64 w .. 64 w + 63 (4 n-tiles).
    static_assert(R == 64 && H == 256 && NT == 256, "one n-tile of the R-wide products and four of the H-wide product per wave");
    const int wv = tid >> 6, fi = lane & 15, fq = lane >> 4;
    float wwF[W / 4], whF[R / 4], wbF[4][W / 4];
    float y1T[R / K4];                             // y1[:, :R]^T fragment (output step only)
    {
        static_assert(3 * R / K4 == 48 && R / K4 == 16 && W / 4 == 8 && R / 4 == 16 && MMG_REPACK_F4 == 30, "layout of tape.wrep");
        const float4* wr = reinterpret_cast<const float4*>(tp.wrep) + tid;
        float4 q[MMG_REPACK_F4];
#pragma unroll
        for (int j = 0; j < MMG_REPACK_F4; ++j) q[j] = wr[j * NT];
        auto put = [](float* dst, const float4& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; };
#pragma unroll
        for (int j = 0; j < 12; ++j) put(whhT + 4 * j, q[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) put(y1T + 4 * j, q[12 + j]);
#pragma unroll
        for (int j = 0; j < 2; ++j) put(wwF + 4 * j, q[16 + j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) put(whF + 4 * j, q[18 + j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) put(&wbF[j >> 1][4 * (j & 1)], q[22 + j]);
    }
    float y1r[R / K4];                             // y1[:, :R] row k4 fragment (forward product A = y1h . h*)
#pragma unroll
    for (int i = 0; i < R / K4; ++i) y1r[i] = P.p[R_Y1_W][(size_t)k4 * (R + V) + p4 * (R / K4) + i];
    const float sm_mine = (tid < Dr) ? tp.sm[(size_t)b * Dr + tid] : 0.f;
    const float w2_mine = (tid < R) ? P.p[R_Y2_W][tid] : 0.f;
    float cdcol[D];                                // Cd[:, tid] for tid < R
#pragma unroll
    for (int d = 0; d < D; ++d) cdcol[d] = 0.f;
    {   // k_prep's [8 float4][R] copy, tape.cd32 (classes beyond Dr: dy is zero there); every thread loads (clamped): no branch
        // around the loads of the prologue
        static_assert(D <= 32, "tape.cd32 holds the first 32 classes");
#pragma unroll
        for (int j = 0; j < (D + 3) / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(tp.cd32 + ((size_t)j * R + min(tid, R - 1)) * 4);
            cdcol[4 * j] = v.x; if (4 * j + 1 < D) cdcol[4 * j + 1] = v.y; if (4 * j + 2 < D) cdcol[4 * j + 2] = v.z; if (4 * j + 3 < D) cdcol[4 * j + 3] = v.w;
        }
    }
    // forward tape of this sample -> registers for ALL T steps (no dependence on tstar: every load of the prologue is
    // in flight at once; indices are clamped instead of guarded so the compiler keeps counted waits), LDS stores below
    constexpr int NW_ = TMAX * W / NT, NG_ = TMAX * R / NT, NU_ = TMAX * 4 * R / NT, NH_ = ((TMAX + 1) * R + NT - 1) / NT, NA_ = TMAX * H / NT;
    float rw_[NW_], rz_[NW_], rpw_[NW_], rpz_[NW_], rg_[NG_], ru_[NU_], rh_[NH_], ra_[NA_];
    const int Tm1 = T - 1;
#pragma unroll
    for (int u = 0; u < NW_; ++u) {
        const int i = tid + NT * u, t = min(i / W, Tm1), j = i % W;
        const size_t o = ((size_t)t * B + b) * W + j;
        rw_[u] = tp.w[o]; rz_[u] = tp.z[o]; rpw_[u] = tp.pw[o]; rpz_[u] = tp.pz[o];
    }
#pragma unroll
    for (int u = 0; u < NG_; ++u) { const int i = tid + NT * u, t = min(i / R, Tm1), j = i % R; rg_[u] = tp.g[((size_t)t * B + b) * R + j]; }
#pragma unroll
    for (int u = 0; u < NU_; ++u) { const int i = tid + NT * u, t = min(i / (4 * R), Tm1), j = i % (4 * R); ru_[u] = tp.gru[((size_t)t * B + b) * 4 * R + j]; }
#pragma unroll
    for (int u = 0; u < NH_; ++u) { const int i = tid + NT * u, t = min(i / R, T), j = i % R; rh_[u] = tp.h[((size_t)t * B + b) * R + j]; }
#pragma unroll
    for (int u = 0; u < NA_; ++u) { const int i = tid + NT * u, t = min(i / H, Tm1), j = i % H; ra_[u] = tp.a[((size_t)t * B + b) * H + j]; }
    const size_t so = (size_t)min(tid, Tm1) * B + b;
    float rbs_, rbr_;
    if (!MERGED) { rbs_ = tp.bs[so]; rbr_ = tp.br[so]; }
    const float rs_ = tp.s[so], rps_ = tp.ps[so];
    MMG_BSTAMP(1);
    LossCoef lc; lc.cw = s_coef; lc.ce = s_coef + 3 * T; lc.cb = s_coef + 6 * T;
    if (!MERGED) coef_compute(dm, creg, lc);                      // (the logged losses: a spare block of k_wgrad)
#pragma unroll
    for (int u = 0; u < NW_; ++u) { const int i = tid + NT * u; t_w[i] = rw_[u]; t_z[i] = rz_[u]; t_pw[i] = rpw_[u]; t_pz[i] = rpz_[u]; }
#pragma unroll
    for (int u = 0; u < NG_; ++u) t_g[tid + NT * u] = rg_[u];
#pragma unroll
    for (int u = 0; u < NU_; ++u) t_gru[tid + NT * u] = ru_[u];
#pragma unroll
    for (int u = 0; u < NH_; ++u) { const int i = tid + NT * u; if (i < (TMAX + 1) * R) t_h[i] = rh_[u]; }
#pragma unroll
    for (int u = 0; u < NA_; ++u) t_a[tid + NT * u] = ra_[u];
    // ---- output step t*, the part that needs no loss coefficient: NLL seed dy, A* = y1[:, :R] h*, dA (and the release of
    // dy / A* to the class roles).  Done here, while the statistics roles are still working, instead of inside the
    // first reverse step.
    const float dy_mine = (tid < Dr) ? (sm_mine - (tid == tgt ? 1.f : 0.f)) / (float)dm.Bg : 0.f;
    __syncthreads();                                    // the forward tape is staged (t_h)
    {
        const int t = tstar;
        if (tid < 64) {
            if (lane < Dr) {                            // (MERGE_DC: write-through store, see the signal below)
                if (MERGE_DC) __hip_atomic_store(&tp.dy[(size_t)b * Dr + lane], dy_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else tp.dy[(size_t)b * Dr + lane] = dy_mine;
            }
            if (lane < 32) s_dy[lane] = dy_mine;
            const float dsum = dpp_wave_sum(dy_mine);
            if (lane == 0) tp.dysum[b] = dsum;
        } else if (tid < 64 + R) {
            tp.hstar[(size_t)b * R + tid - 64] = t_h[(t + 1) * R + tid - 64];
        }
        {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < R / K4; ++i) acc = fmaf(y1r[i], t_h[(t + 1) * R + p4 * (R / K4) + i], acc);
            acc = lane_group_sum<K4>(acc);
            if (p4 == 0) s_A[k4] = acc;
        }
        __syncthreads();
        if (tid < R) {
            const float a = s_A[tid];
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) acc += (a + cdcol[d] > 0.f) ? s_dy[d] : 0.f;
            const float v = acc * w2_mine;
            s_dA[tid] = v; tp.dA[(size_t)b * R + tid] = v;
            if (MERGE_DC) __hip_atomic_store(&tp.Astar[(size_t)b * R + tid], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else tp.Astar[(size_t)b * R + tid] = a;
        }
        if (MERGE_DC && tid == 0) {
            // wave 0 wrote dy and A* with device-scope (write-through) stores: they need no L2 write-back, only to
            // have completed before the counter moves -- 64 sample roles each doing a full device-scope release
            // (buffer_wbl2) here would serialise on the L2s
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (a workgroup-scope fence does not wait for global stores)
            __hip_atomic_fetch_add(tp.sync + MMG_SYNC_ARR(1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- everything that does not depend on the carried dh, for ALL steps of the sample at once, and BEFORE the batch
    // statistics are needed.  A Bernoulli seed is linear in its two loss coefficients (App. A.4):
    //   seed = wh S1 + ce S2,   S1 = -(q / pe - (1 - q) / qe) p (1 - p),   S2 = (log pe + p / pe - log qe - (1 - p) / qe) p (1 - p)
    // and so is everything downstream of it.  The sweeps below run on the two BASIS vectors of every step while the
    // statistics roles of this launch are still reducing: dgpre = (dlw W_w)(1 - g^2), the sender's dpre = (dlz W_b)(1 - a^2)
    // and dgpre W_h as [16 steps, K] x [K, N] products on the matrix cores.  After the wait a step's gradients are two
    // multiply-adds of its bases with (wh_t, ce_t).
    MMG_BSTAMP(8);
    auto seed_basis = [](float q, float pr, float& S1, float& S2) {
        const float pe = pr + MMG_EPS, qe = 1.f - pr + MMG_EPS;
        const float rp = __builtin_amdgcn_rcpf(pe), rq = __builtin_amdgcn_rcpf(qe), pq = pr * (1.f - pr);
        S1 = -(q * rp - (1.f - q) * rq) * pq;
        S2 = (flog(pe) + pr * rp - flog(qe) - (1.f - pr) * rq) * pq;
    };
    constexpr int NS_ = TMAX * W / NT;
    float s1w[NS_], s2w[NS_], s1z[NS_], s2z[NS_];
#pragma unroll
    for (int u = 0; u < NS_; ++u) {
        const int i = tid + NT * u, t = i / W;
        float a1, a2, b1, b2;
        seed_basis(t_w[i], t_pw[i], a1, a2);
        seed_basis(t_z[i], t_pz[i], b1, b2);
        if (!(binary && t < tstar)) { a1 = 0.f; a2 = 0.f; }                  // receiver message: active while m_{t+1} == 1
        if (!(binary && t <= tstar)) { b1 = 0.f; b2 = 0.f; }
        s1w[u] = a1; s2w[u] = a2; s1z[u] = b1; s2z[u] = b2;
        const int j = i - t * W;
        s_sbas[t * SW + j] = a1; s_sbas[(TMAX + t) * SW + j] = a2; s_sbas[(2 * TMAX + t) * SW + j] = b1; s_sbas[(3 * TMAX + t) * SW + j] = b2;
    }
    __syncthreads();
    MMG_BSTAMP(9);
    // D fragments: lane holds rows t = 4 fq + r of column fi of its n-tile
    f32x4 g1 = {0.f, 0.f, 0.f, 0.f}, g2 = {0.f, 0.f, 0.f, 0.f}, p1[4], p2[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { p1[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; p2[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < W / 4; ++ks) {
        const int o = fi * SW + 4 * ks + fq;
        const float aw1 = s_sbas[o], aw2 = s_sbas[TMAX * SW + o], az1 = s_sbas[2 * TMAX * SW + o], az2 = s_sbas[3 * TMAX * SW + o];
        g1 = mfma16(aw1, wwF[ks], g1); g2 = mfma16(aw2, wwF[ks], g2);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { p1[nt] = mfma16(az1, wbF[nt][ks], p1[nt]); p2[nt] = mfma16(az2, wbF[nt][ks], p2[nt]); }
    }
    const int unit = 16 * wv + fi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float g = t_g[(4 * fq + r) * R + unit];
        g1[r] *= (1.f - g * g); g2[r] *= (1.f - g * g);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = t_a[(4 * fq + r) * H + 64 * wv + 16 * nt + fi];
            p1[nt][r] *= (1.f - a * a); p2[nt][r] *= (1.f - a * a);
        }
    // W_y1h^T dA enters dh at the output step only (unit k4, reduction slice p4)
    {
        float accy = 0.f;
#pragma unroll
        for (int i = 0; i < R / K4; ++i) accy = fmaf(y1T[i], s_dA[p4 * (R / K4) + i], accy);
        accy = lane_group_sum<K4>(accy);
        if (p4 == 0) s_dAy[k4] = accy;
    }
    float* s_H2 = t_msg + 2 * TMAX * W;                 // [16][R] over t_z | t_pz (read for the last time before the barrier above)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s_G1[(4 * fq + r) * SR + unit] = g1[r]; s_G2p[(4 * fq + r) * SR + unit] = g2[r]; }
    __syncthreads();
    MMG_BSTAMP(10);
    // The statistics roles of this launch are normally through by now (8 us after the start): their (value, epoch) pairs are
    // asked for HERE, a sweep ahead of their use -- the answer arrives while the matrix cores work.  Stale pairs: again, below.
    // The workgroup fetches the table of 27 T pairs ONCE (two per thread; every thread loading its own 25: 64 x 256 x 25 requests
    // for the same twenty cache lines) and the coefficient threads assemble their share from LDS.
    const uint32_t epoch = MERGED ? tp.counter[3] : 0u;
    const int i_st0 = min(tid, 27 * T - 1), i_st1 = min(tid + NT, 27 * T - 1);
    unsigned long long u_st0 = 0, u_st1 = 0, u_bs = 0, u_br = 0;
    auto load_stat_pairs = [&]() {
        u_st0 = ld_ll(tp.statll, i_st0); u_st1 = ld_ll(tp.statll, i_st1);
        u_bs = ld_ll(tp.statll, statll_bs(dm) + so); u_br = ld_ll(tp.statll, statll_br(dm) + so);
    };
    if (MERGED) { load_stat_pairs(); asm volatile("" ::: "memory"); }      // (the loads stay ahead of the sweep's LDS reads)
    {
        f32x4 h1 = {0.f, 0.f, 0.f, 0.f}, h2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < R / 4; ++ks) {
            h1 = mfma16(s_G1[fi * SR + 4 * ks + fq], whF[ks], h1);
            h2 = mfma16(s_G2p[fi * SR + 4 * ks + fq], whF[ks], h2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { s_dhin[(4 * fq + r) * R + unit] = h1[r]; s_H2[(4 * fq + r) * R + unit] = h2[r]; }
    }
    MMG_BSTAMP(11);
    if (MERGED) {                                       // the statistics roles of this launch publish stats, bs, br
        const bool need_b = binary && min(tid, Tm1) <= tstar;       // (the statistics roles write the baseline scores of live rows only)
        for (int spins = 0;; ) {
            const bool fresh = ll_fresh(u_st0, epoch) && ll_fresh(u_st1, epoch) && (!need_b || (ll_fresh(u_bs, epoch) && ll_fresh(u_br, epoch)));
            if (!__any(!fresh)) break;
            if (++spins > (1 << 16)) { if (lane == 0) __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            load_stat_pairs();
        }
        s_statv[i_st0] = ll_value(u_st0); s_statv[i_st1] = ll_value(u_st1);
        __syncthreads();
        MMG_BSTAMP(12);
        creg = coef_lds_regs(dm, s_statv);
        rbs_ = ll_value(u_bs); rbr_ = ll_value(u_br);
        coef_compute(dm, creg, lc);
    }
    MMG_BSTAMP(2);
    if (tid < TMAX) {
        // per-step scalars of the three streams (model.py:908-922): wh = (L - baseline) cw, ce; stop-bit and MSE seeds
        const int t = min(tid, Tm1);
        const bool on = binary && tid <= tstar;
        s_cf[tid] = on ? (L - rbr_) * lc.cw[T + t] : 0.f;          s_cf[TMAX + tid] = on ? lc.ce[T + t] : 0.f;          // receiver message
        s_cf[2 * TMAX + tid] = on ? (L - rbs_) * lc.cw[2 * T + t] : 0.f; s_cf[3 * TMAX + tid] = on ? lc.ce[2 * T + t] : 0.f;   // sender message
        float dls = 0.f;
        if (on && !dm.fixed) dls = bit_seed_fast(rs_, rps_, (L - rbr_) * lc.cw[t], lc.ce[t]);
        s_dls[tid] = dls;
        if (tid <= tstar) {
            const size_t row = (size_t)tid * B + b;
            tp.dls[row] = dls;
            tp.dbs[row] = binary ? lc.cb[t] * (rbs_ - L) : 0.f;                  // MSE seeds (model.py:971-988)
            tp.dbr[row] = binary ? lc.cb[t] * (rbr_ - L) : 0.f;
        }
    }

    // ---- zero the gradient tapes of steps this sample never took -- unless k_wgrad reduces over the live rows only
    // (build_row_map) and never looks at them: ~1.7 MB of stores per minibatch at config 2
    for (int t = zero_dead ? tstar + 1 : T; t < T; ++t) {
        const size_t row = (size_t)t * B + b;
        if (tid < W) { tp.dlz[row * W + tid] = 0.f; tp.dlw[row * W + tid] = 0.f; }
        tp.dpre[row * H + tid] = 0.f;
        if (tid < R) tp.dgpre[row * R + tid] = 0.f;
        if (tid < 3 * R) { tp.dgi[row * 3 * R + tid] = 0.f; tp.dgh[row * 3 * R + tid] = 0.f; }
        if (tid == 0) { tp.dls[row] = 0.f; tp.dbs[row] = 0.f; tp.dbr[row] = 0.f; }
    }
    __syncthreads();
    MMG_BSTAMP(3);
    // ---- the gradient tapes of the live steps: bases x coefficients (stores only; the recurrence below does not wait for them)
    {
        const float* c1r = s_cf, *c2r = s_cf + TMAX, *c1z = s_cf + 2 * TMAX, *c2z = s_cf + 3 * TMAX;
#pragma unroll
        for (int u = 0; u < NS_; ++u) {
            const int i = tid + NT * u, t = i / W, j = i - t * W;
            if (t <= tstar) {
                const size_t row = (size_t)t * B + b;
                tp.dlw[row * W + j] = fmaf(c1r[t], s1w[u], c2r[t] * s2w[u]);
                tp.dlz[row * W + j] = fmaf(c1z[t], s1z[u], c2z[t] * s2z[u]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 4 * fq + r;
            if (t <= tstar) tp.dgpre[((size_t)t * B + b) * R + unit] = fmaf(c1r[t], g1[r], c2r[t] * g2[r]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int hcol = 64 * wv + 16 * nt + fi;
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 4 * fq + r;
                if (t <= tstar) {
                    const float v = fmaf(c1z[t], p1[nt][r], c2z[t] * p2[nt][r]);
                    tp.dpre[((size_t)t * B + b) * H + hcol] = v;
                    part += v;
                }
            }
            part += __shfl_xor(part, 16); part += __shfl_xor(part, 32);       // the four row groups of the column
            if (fq == 0) tp.dhx[(size_t)b * H + hcol] = part;
        }
    }

    // ---- the recurrence: two phases and ONE barrier per step.  The carried dh of unit k4 lives in a register of each of the unit's
    // four lanes (the 4-lane sum leaves W_hh^T dgh in all of them: no LDS hop, no barrier before the next cell backward), and the gate
    // gradients are double-buffered: a wave that runs ahead writes the OTHER buffer and stops at the next step's barrier, which the
    // slowest wave reaches only after it has read this one.
    auto step_in = [&](int t) {                                        // what step t adds to dh besides the recurrence
        float v = fmaf(s_cf[t], s_dhin[t * R + k4], s_cf[TMAX + t] * s_H2[t * R + k4]) + wsk * s_dls[t];
        if (t == tstar) v += s_dAy[k4];
        return v;
    };
    float dh_c = 0.f;
    for (int t = tstar; t >= 0; --t) {
        const size_t row = (size_t)t * B + b;
        float* const dgb = s_dgh[t & 1];
        MMG_BSTAMP(16 + 2 * t);
        // ===== (4) GRU cell backward: all K4 lanes of unit k4 form the gate gradients, lane p4 stores the p4-th of them
        const float dh = dh_c + step_in(t);
        const float* gr = t_gru + t * 4 * R;
        const float rr = gr[k4], uu = gr[R + k4], nn = gr[2 * R + k4], ghn = gr[3 * R + k4];
        {
            const float hp = t_h[t * R + k4];
            const float dn = dh * (1.f - uu), du = dh * (hp - nn);
            const float dnp = dn * (1.f - nn * nn), dup = du * uu * (1.f - uu);
            const float drp = dnp * ghn * rr * (1.f - rr);
            float* gi = tp.dgi + row * 3 * R; float* gh = tp.dgh + row * 3 * R;
            static_assert(K4 == 4, "four lanes per unit share the stores");
            // (selects, then ONE predicated region: three divergent branches each with their own stores split the phase)
            const float vi = (p4 == 0) ? drp : (p4 == 1) ? dup : dnp;
            const float vh = (p4 == 2) ? dnp * rr : vi;
            const int idx = p4 * R + k4;
            if (p4 < 3) { gi[idx] = vi; gh[idx] = vh; dgb[idx] = vh; }
        }
        __syncthreads(); MMG_BSTAMP(16 + 2 * t + 1);
        // ===== (5) dh_{t-1} = dh * u + W_hh^T dgh
        {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int i = 0; i < 3 * R / K4; i += 4) {
                const float4 dv = *reinterpret_cast<const float4*>(dgb + p4 * (3 * R / K4) + i);
                a0 = fmaf(whhT[i], dv.x, a0); a1 = fmaf(whhT[i + 1], dv.y, a1);
                a2 = fmaf(whhT[i + 2], dv.z, a2); a3 = fmaf(whhT[i + 3], dv.w, a3);
            }
            const float acc = lane_group_sum<K4>((a0 + a1) + (a2 + a3));
            dh_c = __fmul_rn(dh, uu) + acc;
        }
    }
    MMG_BSTAMP(4);
    // (code_bias: k_wgrad's special column job forms dsig * W_c^T (sum_b dpre_0) once, instead of W_c^T dpre_0 per sample here)
}

}  // namespace mmg
