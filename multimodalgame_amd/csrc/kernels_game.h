// kernels_game.h -- k_game_fast: the WHOLE game of one minibatch in one launch for the small agents of BASELINE configs 1-2
// (Adaptive, binary messages, H = 256, W = 32, R = 64, V = 100, D <= 32, B <= 64): exchange() model.py:725-876, output selection /
// NLL / reward 879-904 + 1264-1275, the batch statistics of the REINFORCE / entropy / baseline losses 907-988 and the four
// backward() calls 1309-1328 -- what rounds 1-4 ran as k_conversation_fast3 followed by k_bwd_conv_fast.
//
// Why one launch: a sample's backward role re-read (2.4 + 1.8 us) the tape its forward role had staged in LDS one launch earlier,
// behind a ~2 us launch boundary that also made every sample wait for the LONGEST conversation of the minibatch before its
// backward prologue could start.  Here the workgroup of a sample runs its conversation (the body of kernels_fast3.h), publishes
// what the other roles need, and carries straight on into its reverse pass (the body of kernels_fast.h: k_bwd_conv_fast) on the
// SAME LDS slots: no tape preload, no staging, and the ~8 us of statistics-independent backward work of the short conversations
// overlap the long ones.
//
// Workgroup roles, in dispatch order (every one holds a CU: the launch's LDS size is the sample role's):
//   [0, B)             sample roles: forward -> epilogue -> backward
//   [B, +nprep)        k_prep's blocks (Cd / Dd / hw0 / h_x as (value, epoch) pairs, kernels_fwd.h: prep_body<true, true>; the
//                      backward's repacked weight fragments `wrep` by write-through stores + a "block through" pair)
//   [.., +nbase)       tiles of basehx = h_x . baseline_sen.linear1.weight[:, :H]^T, in and out as pairs
//   [.., +n_stats)     statistics roles (kernels_bwd.h: stats_pairs<.., LLIN>): wait for every sample's forward pass and for the
//                      baseline roles' partial scores, publish the (stream, step) sums + the baseline scores as pairs
//   [.., +n_bas)       baseline roles: both baselines' forward pass over the LIVE (step, sample) rows (k_baselines3's body); a role
//                      owns one (baseline, 64-hidden-unit block) and every (n_bas / 16)-th window of 16 live rows
//   [.., +D)           class roles (k_dC: dCd, Py2; the first one lists the live rows for k_wgrad)
// Hand-offs: the payload itself as (value, epoch) pairs wherever a consumer spins (device_utils.h: st_ll / ld_ll); epoch =
// counter[3] + 1 for every role of the launch -- the minibatch counter and the launch epoch are bumped by k_opt, the last launch
// of the fused step, so no role of this launch ever sees them move.  "Forward pass of sample b is through" is ONE pair per sample
// (tape.gamell[b] = t*(b)), stored after the sample's write-through stores of everything the statistics / baseline roles read
// (t*, reward, hit, log-likelihood sums, z / z_r / h rows) have completed.  Consumers sit AHEAD of some of their producers (sample
// roles wait for the statistics roles): the host selects the kernel only when B + n_stats + n_bas + D workgroups are co-resident
// with a margin (mmg.hip), and every spin is bounded (error word sync[511] -> k_opt skips the update, every later call fails).
// Parity: tests/test_hip_parity.py / test_hip_configs.py run every fast-shape Adaptive case through mmg_train_step, i.e. through this
// kernel; MMG_NO_GAME=1 (read at mmg_create) keeps the two-launch path, and both are compared (tests/test_hip_safety.py).
#pragma once
#include "device_utils.h"
#include "kernels_fast.h"
#include "kernels_fast3.h"
#include "layout.h"

namespace mmg {

// MMG_TIMING builds (scripts/game_timeline.py): wall-clock stamps of EVERY role in tape.dbg2 -- sample b: [2048 + 16 b + slot];
// statistics wave w: [3072 + 4 w + k]; class role d: [3328 + 2 d + k]; baseline role r: [3584 + 4 r + k] (k_wgrad's own stamps: below 2048)
#ifdef MMG_TIMING
#define MMG_GT(idx) do { if ((threadIdx.x & 63) == 0) tp.dbg2[(idx)] = (long long)wall_clock64(); } while (0)
#else
#define MMG_GT(idx) do {} while (0)
#endif

// LDS plan: the forward's (Fast3Lds) + what the backward adds.  s_cd / s_Astep are written during the forward pass; everything
// from `bwd` on overlays the W_hh park, which is dead once the conversation is over.
struct GameLds {
    typedef Fast3Lds F;
    static constexpr int TMAX = F::TMAX, R = F::R, W = F::W;
    static constexpr int cd = F::total;                       // [32][R]   Cd of this minibatch (from the prep roles' pairs)
    static constexpr int Astep = cd + 32 * R;                 // [TMAX][R] A_t = y1[:, :R] h_t of every step (A* without a product)
    static constexpr int desc = Astep + TMAX * R;             // [32][V = 100] the description matrix (B operand of dbar, after the loop)
    static constexpr int w2s = desc + 32 * 100;               // y2.weight [R] | s.weight [R]: the reverse pass reads them without a global load
    static constexpr int total = w2s + 2 * R;
    // overlay of the park [12][256] float4 = 12288 floats (after the loop)
    static constexpr int bwd = F::park;
    static constexpr int coef = bwd;                          // cw[3T] | ce[3T] | cb[T]  (7 * 64)
    static constexpr int dgh = coef + 7 * 64;                 // [2][3R]
    static constexpr int dy = dgh + 2 * 3 * R;                // [32]
    static constexpr int dA = dy + 32;                        // [R]
    static constexpr int dAy = dA + R;                        // [R]
    static constexpr int dls = dAy + R;                       // [TMAX]
    static constexpr int cf = dls + TMAX;                     // [4 TMAX]
    static constexpr int dhin = cf + 4 * TMAX;                // [TMAX][R]
    static constexpr int statv = dhin + TMAX * R;             // [27 TMAX]
    static constexpr int sm = statv + 27 * TMAX;              // [32] softmax(outp) | misc [16]
    static constexpr int misc = sm + 32;
    // MFMA A operands of the two sweeps, rows PADDED (strides 34 / 66 floats): a lane (fi, fq) reads [fi * stride + 4 ks + fq], so
    // with the natural strides 32 / 64 all sixteen rows fi hit ONE bank (16-way conflicts: most of the sweeps' time in rounds 2-4)
    static constexpr int SW = 34, SR = 66;
    static constexpr int sb = misc + 16;                      // [4][TMAX][SW]  seed bases: w1 | w2 | z1 | z2
    static constexpr int sg = sb + 4 * TMAX * SW;             // [2][TMAX][SR]  dgpre bases: g1 | g2
    static constexpr int bwd_end = sg + 2 * TMAX * SR;
    static_assert(bwd_end <= F::park + 12 * 256 * 4, "the backward's scratch overlays the W_hh park");
};
__host__ __device__ inline int game_lds_bytes() { return GameLds::total * 4; }

// 16 bytes of a row another role of this launch wrote through (agent-scope loads, device_utils.h)
template <int MAXQ>
__device__ __forceinline__ void frag_load_cc(float4 (&f)[MAXQ], const float* __restrict__ row, bool valid, int K, int q) {
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) {
        const int k = q * 4 + 16 * j;
        f[j] = (valid && k < K) ? ld_cc4(row + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// Baseline role `role` of `n_roles` (a multiple of 2 * npb / UB): (which, first unit block) = role % (2 npb / UB), window slot =
// role / (2 npb / UB).  UB = 2 (round 5, npb even): a role owns TWO 64-unit blocks of its baseline -- the rows of a window are loaded
// once for both, and the same number of roles covers twice as many window slots, so that the longer conversations of a trained
// pair (180+ live rows: 12+ windows) still cost every role ONE window between the last forward pass and the statistics.
// Partial scores stay per 64-unit block: the statistics roles add the same npb terms in the same order.
template <int UB>
__device__ __forceinline__ void game_baseline_role(const Dims& dm, const Params& P, const Tape& tp, int role, int n_roles, uint32_t epoch) {
    __shared__ int s_rid[64 * 16];                       // live-row ids of this role's windows, 16 per window (-1: none)
    __shared__ int s_nwin;
    __shared__ float s_part[UB][4][16];
    const int B = dm.B, H = dm.H, W = dm.W, R = dm.R, K = dm.K, T = dm.T;
    const int npb = (K + 63) / 64, cpw = npb / UB;       // unit-block groups per baseline
    const int combo = role % (2 * cpw), slot = role / (2 * cpw), nslots = n_roles / (2 * cpw);
    const int which = combo / cpw, byi0 = (combo - which * cpw) * UB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    // ---- weights first (parameters: stable since the previous minibatch's k_opt), while the conversations run
    const float* W1 = which ? P.p[BS_L1_W] : P.p[BR_L1_W];
    const int ldw = which ? H + W : W + R;
    int n[UB]; bool nv[UB]; float bias[UB], w2[UB];
    float4 w_msg[UB][4], w_st[UB][4];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        n[u] = ((byi0 + u) * 4 + wave) * 16 + i;
        nv[u] = n[u] < K;
        const float* wrow = W1 + (size_t)(nv[u] ? n[u] : 0) * ldw;
        bias[u] = nv[u] ? (which ? P.p[BS_L1_B][n[u]] : P.p[BR_L1_B][n[u]]) : 0.f;
        w2[u] = nv[u] ? (which ? P.p[BS_L2_W][n[u]] : P.p[BR_L2_W][n[u]]) : 0.f;
        frag_load(w_msg[u], wrow + (which ? H : 0), nv[u], W, q);
        if (!which) frag_load(w_st[u], wrow + W, nv[u], R, q);
    }
    for (int k = threadIdx.x; k < 64 * 16; k += blockDim.x) s_rid[k] = -1;
    __syncthreads();
    if (wave == 0) {
        MMG_GT(3584 + 4 * role);
        const int ts = game_wait_done(tp, B, epoch);     // every sample's forward pass is written through
        MMG_GT(3584 + 4 * role + 1);
        int base = 0;
        for (int t = 0; t < T; ++t) {
            const bool act = (lane < B) && (t <= ts);
            const unsigned long long m = __ballot(act);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            const int w = pos >> 4;
            if (act && (w % nslots) == slot && (w / nslots) < 64) s_rid[(w / nslots) * 16 + (pos & 15)] = t * B + lane;
            base += __popcll(m);
        }
        const int nw = (base + 15) >> 4;                  // windows of the minibatch; mine: slot, slot + nslots, ...
        if (lane == 0) s_nwin = nw > slot ? (nw - slot + nslots - 1) / nslots : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) { MMG_GT(5120 + 8 * role + 0); }
    const int nwin = s_nwin;
    float* hid = which ? tp.hid_s : tp.hid_r;
    float* part = which ? tp.bs_part : tp.br_part;
    // Two-stage pipeline over the role's windows: the rows (and the sender baseline's basehx pairs) of window kw + 1 are requested
    // BEFORE window kw's products, so a second window costs its MFMAs and stores, not another trip to memory (a trained pair
    // talks for ~7 steps: 440 live rows, 28 windows on 16 slots).
    float4 xmA[4], xtA[4], xmB[4], xtB[4];                 // (two named buffers: a runtime index would put them in scratch)
    unsigned long long ubA[UB][4], ubB[UB][4];
    auto request = [&](int kw, float4 (&xm_)[4], float4 (&xt_)[4], unsigned long long (&ub_)[UB][4]) {
        const int* rw = s_rid + kw * 16;
        const int rid = rw[i];
        const bool xv = rid >= 0;
        const size_t rr = (size_t)(xv ? rid : 0);
        frag_load_cc(xm_, (which ? tp.zr : tp.z) + rr * W, xv, W, q);
        if (!which) frag_load_cc(xt_, tp.h + (rr + B) * R, xv, R, q);
        if (which) {
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int o = rw[q * 4 + r]; ub_[u][r] = ld_ll(tp.basell, (size_t)((o >= 0 ? o : 0) % B) * K + min(n[u], K - 1)); }
        }
    };
    auto window = [&](int kw, float4 (&xm_)[4], float4 (&xt_)[4], unsigned long long (&ub_)[UB][4],
                      float4 (&xmN)[4], float4 (&xtN)[4], unsigned long long (&ubN)[UB][4]) {
        const int* rid_w = s_rid + kw * 16;
        if (kw + 1 < nwin) request(kw + 1, xmN, xtN, ubN);
        int orow[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) orow[r] = rid_w[q * 4 + r];
        f32x4 acc[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (which) {                                      // basehx of the sample: (value, epoch) pairs of this launch's tiles, written ~25 us ago
            bool fresh = true;
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) fresh = fresh && ll_fresh(ub_[u][r], epoch);
            if (__any(!fresh)) {                          // (not yet: spin on them)
                for (int spins = 0;; ) {
                    fresh = true;
#pragma unroll
                    for (int u = 0; u < UB; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ub_[u][r] = ld_ll(tp.basell, (size_t)((orow[r] >= 0 ? orow[r] : 0) % B) * K + min(n[u], K - 1));
                            fresh = fresh && ll_fresh(ub_[u][r], epoch);
                        }
                    if (!__any(!fresh)) break;
                    if (++spins > (1 << 16)) { __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[u][r] = (orow[r] >= 0) ? ll_value(ub_[u][r]) : 0.f;
        }
        if (threadIdx.x == 0 && kw == 0) { MMG_GT(5120 + 8 * role + 1); }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            frag_mfma(acc[u], xm_, w_msg[u], W, q);
            if (!which) { f32x4 acc2 = {0.f, 0.f, 0.f, 0.f}; frag_mfma(acc2, xt_, w_st[u], R, q); acc[u] += acc2; }
        }
#ifdef MMG_TIMING
        if (threadIdx.x == 0 && kw == 0) { asm volatile("" :: "v"(acc[0][0])); tp.dbg2[5120 + 8 * role + 2] = (long long)wall_clock64(); }
#endif
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaxf(acc[u][r] + bias[u], 0.f);                      // model.py:514
                if (orow[r] >= 0 && nv[u]) hid[(size_t)orow[r] * K + n[u]] = v; else v = 0.f;
                v = dpp_group_sum<16>(v * w2[u]);
                if (i == 0) s_part[u][wave][q * 4 + r] = v;
            }
        __syncthreads();
        if (threadIdx.x == 0 && kw == 0) { MMG_GT(5120 + 8 * role + 3); }
        if (threadIdx.x < 16 * UB) {
            const int u = threadIdx.x >> 4, rw = threadIdx.x & 15;
            if (rid_w[rw] >= 0) {
                const float v = (s_part[u][0][rw] + s_part[u][1][rw]) + (s_part[u][2][rw] + s_part[u][3][rw]);
                part[(size_t)rid_w[rw] * npb + byi0 + u] = v;
                st_ll(tp.partll, ((size_t)(which ? 0 : 1) * T * B + (size_t)rid_w[rw]) * npb + byi0 + u, v, epoch);
            }
        }
        __syncthreads();
    };
    if (nwin > 0) request(0, xmA, xtA, ubA);
    for (int kw = 0; kw < nwin; kw += 2) {
        window(kw, xmA, xtA, ubA, xmB, xtB, ubB);
        if (kw + 1 < nwin) window(kw + 1, xmB, xtB, ubB, xmA, xtA, ubA);
    }
    if (threadIdx.x == 0) { MMG_GT(3584 + 4 * role + 2); }
}

struct GameArgs { int n_stats, n_bas, bas_ub; };   // bas_ub: 64-unit blocks per baseline role (1 | 2)

template <int D>
__global__ __launch_bounds__(256, 1) void k_game_fast(Dims dm, Params P, Tape tp, ConvArgs ar, GameArgs ga) {
    constexpr int H = 256, W = 32, R = 64, V = 100;
    constexpr int NT = 256, TMAX = Fast3Lds::TMAX;
    static_assert(FastDims<H, W, R, V, D>::ok, "unsupported fast shape");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint32_t epoch = tp.counter[3] + 1u;           // of every pair of this launch (bumped by k_opt: mmg.hip, fused step)
    const int bx = (int)blockIdx.x, B = dm.B;
    // ---------------------------------------------------------------- the other roles
    if (bx >= B) {
        int r = bx - B;
        if (r < ar.nprep) {
            if (threadIdx.x == 0 && r < 256) { MMG_GT(4608 + 2 * r); }
            prep_body<true, true>(dm, P, tp, ar.desc, ar.x, ar.prep_cpb, r, lds, 0);
#ifdef MMG_TIMING
            __syncthreads();
            if (threadIdx.x == 0 && r < 256) { MMG_GT(4608 + 2 * r + 1); }
#endif
            return;
        }
        r -= ar.nprep;
        if (r < ar.nbase) {                              // basehx tile: A operand = the h_x pairs, result out as pairs too
            gemm_nt_tile<true, true>(r, tp.prepll, H, P.p[BS_L1_W], H + W, nullptr, tp.basehx, dm.K, B, dm.K, H, tp.basell, epoch, tp.sync);
            return;
        }
        r -= ar.nbase;
        if (r < ga.n_stats) {
            MMG_GT(3072 + 4 * (r * 4 + (int)(threadIdx.x >> 6)));
            stats_pairs<true, true, true, true>(dm, P, tp, 1, r * 4 + (int)(threadIdx.x >> 6), ga.n_stats * 4, epoch);
            MMG_GT(3072 + 4 * (r * 4 + (int)(threadIdx.x >> 6)) + 2);
            return;
        }
        r -= ga.n_stats;
        if (r < ga.n_bas) {
            if (ga.bas_ub == 2) game_baseline_role<2>(dm, P, tp, r, ga.n_bas, epoch);
            else game_baseline_role<1>(dm, P, tp, r, ga.n_bas, epoch);
            return;
        }
        r -= ga.n_bas;
        {
            float* s_c = lds; float* s_p = lds + 256;
            role_wait<8, false>(tp.sync, 1, (uint32_t)B, (uint32_t)dm.D);      // every sample has released dy / A*
            if (r == 0 && threadIdx.x < 64) build_row_map<true>(dm, tp);       // rows k_wgrad will reduce over
            if (threadIdx.x == 0) { MMG_GT(3328 + 2 * r); }
            dC_class<true>(dm, P, tp, r, s_c, s_p);
            if (threadIdx.x == 0) { MMG_GT(3328 + 2 * r + 1); }
        }
        return;
    }
    // ---------------------------------------------------------------- sample role
    typedef GameLds G;
    typedef Fast3Lds L;
    float* const s_a = lds + L::a; float* const s_gru = lds + L::gru; float* const s_h = lds + L::h; float* const s_g = lds + L::g;
    float* const s_z = lds + L::z; float* const s_pz = lds + L::pz; float* const s_w = lds + L::w; float* const s_pw = lds + L::pw;
    float* const s_y = lds + L::y; float* const s_pi = lds + L::pi; float* const s_uz = lds + L::uz; float* const s_uw = lds + L::uw;
    float* const s_e = lds + L::e; float* const s_A = lds + L::A; float* const s_gh = lds + L::gh;
    float* const s_us = lds + L::small; float* const s_ps = s_us + 16; float* const s_sb = s_us + 32; float* const s_mask = s_us + 48;
    float* const s_sig = s_us + 80;
    float4* const s_park = reinterpret_cast<float4*>(lds + L::park);
    float* const s_cd = lds + G::cd; float* const s_Astep = lds + G::Astep; float* const s_desc = lds + G::desc; float* const s_w2s = lds + G::w2s;

    const int b = bx, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = dm.T, Dr = dm.D;
    const bool inject = ar.u_s != nullptr;
    MMG_STAMP(0); if (tid == 0) { MMG_GT(2048 + 16 * b + 0); }
    const uint32_t mb_counter = tp.counter[0] + 1u;      // (the minibatch this step trains on; k_opt commits the bump)
    const uint32_t gb = (uint32_t)(dm.boff + b);
    const int tgt = (int)ar.target[b];
    int t_done = T, w_done = T;                           // steps executed / steps whose receiver message was formed
    {   // ============================================================ forward pass (kernels_fast3.h, MERGED, training, binary)
        for (int i = tid; i < L::uz; i += NT) lds[i] = 0.f;     // the tape slots: rows this conversation never writes read as zeros
        if (inject) {
            for (int i = tid; i < T * W; i += NT) {
                const int t = i / W, j = i - t * W;
                s_uz[i] = ar.u_z[((size_t)t * B + b) * W + j];
                s_uw[i] = ar.u_w[((size_t)t * B + b) * W + j];
            }
            if (tid < T) s_us[tid] = ar.u_s[(size_t)tid * B + b];
        }
        // ---- weights -> registers / LDS park (lane maps: kernels_fast3.h)
        float wc[W];
#pragma unroll
        for (int j = 0; j < W / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(P.p[S_CODE_W] + (size_t)tid * W + 4 * j);
            wc[4 * j] = v.x; wc[4 * j + 1] = v.y; wc[4 * j + 2] = v.z; wc[4 * j + 3] = v.w;
        }
        const float bc = P.p[S_CODE_B][tid];
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
        float4 whh_tmp[12];
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
        const int d5 = tid >> 3, r5 = tid & 7;
        float ncd[8], w2[8];
        {
            const float4 q0 = *reinterpret_cast<const float4*>(P.p[R_Y2_W] + r5 * 8), q1 = *reinterpret_cast<const float4*>(P.p[R_Y2_W] + r5 * 8 + 4);
            w2[0] = q0.x; w2[1] = q0.y; w2[2] = q0.z; w2[3] = q0.w; w2[4] = q1.x; w2[5] = q1.y; w2[6] = q1.z; w2[7] = q1.w;
        }
        const float4 ws4 = *reinterpret_cast<const float4*>(P.p[R_S_W] + (tid & 15) * 4);
        float dsc[(32 * V + NT - 1) / NT];               // the caller's description matrix -> LDS (below, with the park stores)
#pragma unroll
        for (int k = 0; k < (32 * V + NT - 1) / NT; ++k) { const int i = tid + NT * k; dsc[k] = (i < Dr * V) ? ar.desc[min(i, Dr * V - 1)] : 0.f; }
        const float bs = P.p[R_S_B][0];
        float dd[8];
        if (tid < W) s_sig[tid] = fsigmoid(P.p[S_CODE_BIAS][tid]);
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
        if (!inject) {                                   // Philox draws of the whole conversation, while the weight loads are in flight
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
#pragma unroll
        for (int k = 0; k < (32 * V + NT - 1) / NT; ++k) { const int i = tid + NT * k; if (i < 32 * V) s_desc[i] = dsc[k]; }
        if (tid == 0) { MMG_GT(2048 + 16 * b + 14); }
        {
            int spins = 0;
            for (;;) {
                load_pairs();
                bool fresh = true;
#pragma unroll
                for (int k = 0; k < 18; ++k) fresh = fresh && ll_fresh(u[k], epoch);
#ifdef MMG_TIMING
                if (tid == 0) tp.dbg2[2048 + 16 * b + 15] = (long long)(spins + 1);
#endif
                if (!__any(!fresh)) break;
                if (++spins > (1 << 16)) { if (lane == 0) __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        const float hw0 = ll_value(u[0]), hx = ll_value(u[1]);
#pragma unroll
        for (int k = 0; k < 8; ++k) { ncd[k] = -ll_value(u[2 + k]); dd[k] = (q3 * 8 + k < Dr) ? ll_value(u[10 + k]) : 0.f; }
        float cy5;
        {
            float part = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) part = fmaf(w2[k], -ncd[k], part);
            cy5 = dpp_group_sum<8>(part) + P.p[R_Y2_B][0];
        }
        // Cd of this minibatch for the reverse pass's output step (the column of Cd a unit needs is read back from here)
        if (d5 < Dr) {
            *reinterpret_cast<float4*>(s_cd + d5 * R + r5 * 8) = make_float4(-ncd[0], -ncd[1], -ncd[2], -ncd[3]);
            *reinterpret_cast<float4*>(s_cd + d5 * R + r5 * 8 + 4) = make_float4(-ncd[4], -ncd[5], -ncd[6], -ncd[7]);
        }
        // what the reverse pass would otherwise have to LOAD behind its 120 KB of weight fragments (a load's first use drains every
        // load and store in flight on this chip): y2 / s weights and the description matrix, parked in LDS while the weights stream in
        if (d5 == 0) { *reinterpret_cast<float4*>(s_w2s + r5 * 8) = make_float4(w2[0], w2[1], w2[2], w2[3]); *reinterpret_cast<float4*>(s_w2s + r5 * 8 + 4) = make_float4(w2[4], w2[5], w2[6], w2[7]); }
        if (tid < 16) *reinterpret_cast<float4*>(s_w2s + R + tid * 4) = ws4;
        MMG_STAMP(1); if (tid == 0) { MMG_GT(2048 + 16 * b + 1); }
        if (tid < R) s_h[tid] = 0.f;
        if (tid < W) s_w[tid] = dm.first_rec;
        if (tid == 0) s_mask[0] = 1.f;
        float m_run = 1.f; int t_out = -1;               // (meaningful on lane 240 only: the stop head's lane)
        float ghp_r = 0.f, ghp_u = 0.f, ghn = b_hn;
        __syncthreads();
        MMG_STAMP(2);
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
                const float zz = (uz < ps) ? 1.f : 0.f;
                if (k2 == 0) { s_z[t * W + m2] = zz; s_pz[t * W + m2] = ps; }
            }
            __syncthreads(); MMG_STAMP(8 + 10 * t + 1);
            // ===== P3 GRU cell
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
            // ===== P4 heads on h: A = y1[:, :R] h (kept per step: A* of the reverse pass), w_h h + b_h
            {
                const float acc = dpp_group_sum<2>(dot4p<8>(w4, hn + half4 * 4, 8)) + b4;
                if (half4 == 0) { if (row4 < R) { s_A[row4] = acc; s_Astep[t * R + row4] = acc; } else s_gh[row4 - R] = acc; }
            }
            __syncthreads(); MMG_STAMP(8 + 10 * t + 3);
            // ===== P5 class logits | stop head (lane 240 keeps it) | hidden-side product of the r gate for the next step
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
                const float sbit = (us_t < p) ? 1.f : 0.f;
                const float m_next = fminf(m_run, sbit);
                const bool last = (t == T - 1);
                const bool take = (t_out < 0 && (m_next == 0.f || last));
                t_out = take ? t : t_out;
                m_run = m_next;
                ghp_r = park_fma(pk, hq);
                if (r5 == 0) s_y[t * 32 + d5] = (d5 < Dr) ? yv : -3.0e38f;
                if (tid == 240) { s_ps[t] = p; s_sb[t] = sbit; s_mask[t + 1] = m_next; }
            }
            __syncthreads(); MMG_STAMP(8 + 10 * t + 4);
            if (s_mask[t + 1] == 0.f) { t_done = t + 1; w_done = t; break; }
            // ===== P6 softmax of the wave's own copy of y, then g = tanh(w_h h + b_h + softmax(y) . Dd)
            {
                const float yv = s_y[t * 32 + (lane & 31)];
                const float ghu = s_gh[u3];
                float4 pk[4], hq[4];
                park_load(pk, hq, s_park, 1, tid, hn + q3 * 16);
                float mx = fmax_nn(yv, dpp_f<MMG_DPP_QUAD_1032>(yv)); mx = fmax_nn(mx, dpp_f<MMG_DPP_QUAD_2301>(mx));
                mx = fmax_nn(mx, dpp_f<MMG_DPP_ROW_HALF_MIRROR>(mx)); mx = fmax_nn(mx, dpp_f<MMG_DPP_ROW_MIRROR>(mx));
                const float m0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 0));
                const float m1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 16));
                const float e = __expf(yv - fmaxf(m0, m1));
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
                const float wv = (uw < ps) ? 1.f : 0.f;
                ghn = dpp_group_sum<4>(park_fma(pk, hq)) + b_hn;
                if (k2 == 0) { s_w[(t + 1) * W + m2] = wv; s_pw[t * W + m2] = ps; }
            }
            __syncthreads(); MMG_STAMP(8 + 10 * t + 6);
        }
        __syncthreads();
        MMG_STAMP(3); if (tid == 0) { MMG_GT(2048 + 16 * b + 2); }
        // the output step travels from the stop head's lane (tid 240) to everybody through the mask slots' neighbour
        if (tid == 240) s_us[15] = (float)t_out;          // (s_us[T..15] are free: T <= 15 checked by the host)
    }
    __syncthreads();
    const int tstar = (int)s_us[15];
    float* const s_sm = lds + G::sm; float* const s_gm = lds + G::misc;
    const int wv = tid >> 6, fi = lane & 15, fq = lane >> 4;
    // ================================================================ epilogue 0: the rows the BASELINE roles multiply -- z_t, z_r
    // (= w_{t-1}), h_{t+1} of the live steps -- written through first thing; once they have completed: pair A, gamell[b] = t*
    {
        float4 vh[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) vh[r] = *reinterpret_cast<const float4*>(s_h + 4 * min(tid + NT * r, (TMAX + 1) * (R / 4) - 1));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i4 = tid + NT * r, t = i4 >> 4, c = i4 & 15;
            if (t <= t_done && i4 < (TMAX + 1) * (R / 4)) st_wt4(tp.h + ((size_t)t * B + b) * R + 4 * c, vh[r]);
        }
        if (tid < TMAX * (W / 4)) {
            const int t = tid >> 3, c = tid & 7;
            const size_t o = ((size_t)t * B + b) * W + 4 * c;
            const float4 vz = *reinterpret_cast<const float4*>(s_z + 4 * tid);
            const float4 cv = *reinterpret_cast<const float4*>(s_w + 4 * tid);         // slot t: what the sender read at step t
            if (t < t_done) { st_wt4(tp.z + o, vz); st_wt4(tp.zr + o, cv); }
        }
        if (tid == 0) __hip_atomic_store(&tp.tstar[b], tstar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the write-through stores above have completed
    __syncthreads();
    if (tid == 0) st_ll(tp.gamell, (size_t)b, (float)tstar, epoch);           // pair A: "the rows of sample b are through"
    MMG_STAMP(120); if (tid == 0) { MMG_GT(2048 + 16 * b + 3); }
    // ================================================================ epilogue 1: what the STATISTICS roles read (reward, hit,
    // log-likelihood / neg-entropy sums; pair B goes out below, once these stores have completed) and the small tape arrays whose
    // LDS slots the reverse pass overwrites (plain stores: k_wgrad / the host read them after the launch)
    // ---- output selection / reward / top-k (wave 3)
    if (wave == 3) {
        const float o = (lane < 32) ? s_y[tstar * 32 + lane] : -3.0e38f;
        const float mx = dpp_wave_max(o);
        const float e = (lane < Dr) ? __expf(o - mx) : 0.f;
        const float lse = mx + flog(dpp_wave_sum(e));
        const float dt = __shfl(o, tgt, 64) - lse;
        const float ld = o - lse;
        const float smv = __expf(ld);
        if (lane < 32) s_sm[lane] = (lane < Dr) ? smv : 0.f;
        if (lane < Dr) {
            tp.outp[(size_t)b * Dr + lane] = o;
            tp.dist[(size_t)b * Dr + lane] = ld;
            tp.sm[(size_t)b * Dr + lane] = smv;
        }
        const float above = dpp_wave_sum((lane < Dr && ld > dt) ? 1.f : 0.f);
        if (lane == 0) {
            st_wt(&tp.logs[b], dt);
            __hip_atomic_store(&tp.hit[b], (above < (float)dm.top_k) ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_gm[0] = dt;
        }
    }
    // ---- log-likelihood / neg-entropy sums of all steps (model.py:908-922): step tt = tid / 16, bits 2 jj, 2 jj + 1
    {
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
        if (jj == 0 && tt < t_done) { st_wt(&tp.lp_z[(size_t)tt * B + b], lz); st_wt(&tp.ne_z[(size_t)tt * B + b], nz); }
        if (jj == 0 && tt < w_done) { st_wt(&tp.lp_w[(size_t)tt * B + b], lw); st_wt(&tp.ne_w[(size_t)tt * B + b], nw); }
    }
    if (tid < t_done) {
        const float p = s_ps[tid], sb = s_sb[tid];
        const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
        st_wt(&tp.lp_s[(size_t)tid * B + b], sb * l1 + (1.f - sb) * l0);
        st_wt(&tp.ne_s[(size_t)tid * B + b], p * l1 + (1.f - p) * l0);
        tp.s[(size_t)tid * B + b] = sb; tp.ps[(size_t)tid * B + b] = p;
    }
    {   // g, pz, c, w, pw, softmax rows: their LDS slots become the seed bases / MFMA results of the reverse pass
        const float4 vgg = *reinterpret_cast<const float4*>(s_g + 4 * tid);
        { const int t = tid >> 4, c = tid & 15; if (t < w_done) *reinterpret_cast<float4*>(tp.g + ((size_t)t * B + b) * R + 4 * c) = vgg; }
        if (tid < TMAX * (W / 4)) {
            const int t = tid >> 3, c = tid & 7;
            const size_t o = ((size_t)t * B + b) * W + 4 * c;
            const float4 vpz = *reinterpret_cast<const float4*>(s_pz + 4 * tid);
            const float4 cv = *reinterpret_cast<const float4*>(s_w + 4 * tid);
            const float4 vw = *reinterpret_cast<const float4*>(s_w + W + 4 * tid), vpw = *reinterpret_cast<const float4*>(s_pw + 4 * tid);
            const float4 sg = *reinterpret_cast<const float4*>(s_sig + 4 * c);
            if (t < t_done) {
                *reinterpret_cast<float4*>(tp.pz + o) = vpz;
                // (component-wise: a ?: over two float4 values went through SCRATCH -- two stores and a dependent load in the epilogue)
                *reinterpret_cast<float4*>(tp.c + o) = make_float4(t == 0 ? sg.x : cv.x, t == 0 ? sg.y : cv.y, t == 0 ? sg.z : cv.z, t == 0 ? sg.w : cv.w);   // model.py:199
            }
            if (t < w_done) {
                *reinterpret_cast<float4*>(tp.w + o) = vw;
                *reinterpret_cast<float4*>(tp.pw + o) = vpw;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // epilogue 1's write-through stores have completed
    __syncthreads();                                     // (and wave 3's softmax row / reward are in LDS)
    if (tid == 0) st_ll(tp.gamell, (size_t)B + 16 + b, (float)tstar, epoch);  // pair B: "what the statistics roles read of sample b is through"
    if (tid == 0) { MMG_GT(2048 + 16 * b + 13); }
    // ================================================================ the reverse pass's weight fragments: requested NOW, in the order
    // of their use (sweep 1: W_w, binary_layer; y1^T; sweep 2: W_h; the recurrence: W_hh^T -- loads return in order, so the first
    // sweep starts on a third of the bytes), they arrive under epilogue 2.  `wrep` was written through by prep roles of this
    // launch ~20 us ago; their "through" pairs are awaited first.
    constexpr int K4 = NT / R;
    const int k4 = tid / K4, p4 = tid % K4;
    float whhT[3 * R / K4], wwF[W / 4], whF[R / 4], wbF[4][W / 4], y1T[R / K4];
    unsigned long long u_rep;
    // (a load's first use drains every load AND store in flight -- the waitcnt pass of the compiler treats mixed loads / stores as
    //  out of order on this chip -- so between here and the first sweep NOTHING reads a global value: y2 / s weights and the
    //  description matrix come from LDS, parked there by the forward prologue)
    const float wsk = s_w2s[R + k4];
    const float w2_mine = s_w2s[min(tid, R - 1)];
    constexpr int NTILE_V = (V + 15) / 16;
    {
        static_assert(3 * R / K4 == 48 && R / K4 == 16 && W / 4 == 8 && R / 4 == 16 && MMG_REPACK_F4 == 30, "layout of tape.wrep");
        // The repack blocks' "through" pairs FIRST (written ~20 us ago: one trip), the plain fragment loads only behind them: a
        // workgroup that fetched tape.wrep before the repack was through would park the PREVIOUS minibatch's fragments in this
        // XCD's L2 for every later sample role of the XCD, whose own pair check then passes.  Nothing is in flight
        // here (the vmcnt(0) above), and the sweeps these fragments feed run in the shadow of the statistics chain.
        u_rep = ld_ll(tp.gamell, (size_t)B + (lane & 7));
        for (int spins = 0; __any(!ll_fresh(u_rep, epoch)); ) {
            u_rep = ld_ll(tp.gamell, (size_t)B + (lane & 7));
            if (++spins > (1 << 16)) { if (lane == 0) __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 9u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        const float4* wr = reinterpret_cast<const float4*>(tp.wrep) + tid;
        float4 q[MMG_REPACK_F4];
#pragma unroll
        for (int j = 16; j < 18; ++j) q[j] = wr[j * NT];
#pragma unroll
        for (int j = 22; j < 30; ++j) q[j] = wr[j * NT];
#pragma unroll
        for (int j = 12; j < 16; ++j) q[j] = wr[j * NT];
#pragma unroll
        for (int j = 18; j < 22; ++j) q[j] = wr[j * NT];
#pragma unroll
        for (int j = 0; j < 12; ++j) q[j] = wr[j * NT];
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
    const float Lrew = s_gm[0];
    // ================================================================ epilogue 2: the bulk of the tape (a, GRU gates, class logits, masks,
    // dbar; k_wgrad / the host read them after the launch): plain coalesced stores HERE, under the latency of the fragment loads
    // just issued
    {
        if (tid <= t_done) tp.mask[(size_t)tid * B + b] = (uint8_t)(s_mask[tid] != 0.f);
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
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = tid + NT * r, t = i >> 5, d = i & 31;
            const float yv = s_y[i];
            if (t < t_done && d < Dr) tp.y[((size_t)t * B + b) * Dr + d] = yv;
        }
        // dbar = softmax(y_t) . desc of this sample's steps (model.py:442-449; read by k_wgrad's w_d job only): [16 steps, 32] x
        // [32, V] on the matrix cores, A from the LDS softmax rows (rows >= w_done: zeros), B = the caller's description matrix
        {
            float a[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) a[ks] = s_pi[fi * 32 + 4 * ks + fq];
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                const int nt = wv + 4 * uu;
                if (nt >= NTILE_V) break;                                            // (wave-uniform)
                const int n = nt * 16 + fi;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) { const int d = 4 * ks + fq; acc = mfma16(a[ks], (d < Dr && n < V) ? s_desc[min(d, Dr - 1) * V + min(n, V - 1)] : 0.f, acc); }
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int t = 4 * fq + r; if (t < w_done && n < V) tp.dbar[((size_t)t * B + b) * V + n] = acc[r]; }
            }
        }
    }
    MMG_STAMP(6); if (tid == 0) { MMG_GT(2048 + 16 * b + 4); }
    // ================================================================ reverse pass (kernels_fast.h: k_bwd_conv_fast, MERGED + MERGE_DC)
    // on the forward's LDS slots: t_w = s_w slots 1.., t_pw = s_pw (contiguous), t_z | t_pz, t_g, t_gru, t_h, t_a
#ifdef MMG_TIMING
#define MMG_GSTAMP(slot) do { if (b == 0 && tid == 0) tp.dbg[128 + (slot)] = (long long)wall_clock64(); } while (0)
#else
#define MMG_GSTAMP(slot) do {} while (0)
#endif
    MMG_GSTAMP(0);
    float* const t_w = s_w + W; float* const t_pw = s_pw; float* const t_z = s_z; float* const t_pz = s_pz;
    static_assert(L::pw == L::w + (TMAX + 1) * W && L::pz == L::z + TMAX * W, "message slots are contiguous pairs of [16][W] tiles");
    float* const t_g = s_g; const float* const t_gru = s_gru; const float* const t_h = s_h; const float* const t_a = s_a;
    float* const s_coef = lds + G::coef; float* const s_dy = lds + G::dy; float* const s_dA = lds + G::dA; float* const s_dAy = lds + G::dAy;
    float* const s_dls = lds + G::dls; float* const s_cf = lds + G::cf; float* const s_dhin = lds + G::dhin; float* const s_statv = lds + G::statv;
    float* const s_dghb = lds + G::dgh; float* const s_sbas = lds + G::sb;
    const int Tm1 = T - 1;
    const size_t so = (size_t)min(tid, Tm1) * B + b;
    const float rs_ = s_sb[min(tid, TMAX - 1)], rps_ = s_ps[min(tid, TMAX - 1)];
    LossCoef lc; lc.cw = s_coef; lc.ce = s_coef + 3 * T; lc.cb = s_coef + 6 * T;
    // ---- output step t*: NLL seed dy, A* (kept by the forward pass), dA; dy / A* released to the class roles
    const float dy_mine = (tid < Dr) ? (s_sm[min(tid, 31)] - (tid == tgt ? 1.f : 0.f)) / (float)dm.Bg : 0.f;
    {
        const int t = tstar;
        if (tid < 64) {
            if (lane < Dr) __hip_atomic_store(&tp.dy[(size_t)b * Dr + lane], dy_mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane < 32) s_dy[lane] = dy_mine;
            const float dsum = dpp_wave_sum(dy_mine);
            if (lane == 0) tp.dysum[b] = dsum;
        } else if (tid < 64 + R) {
            tp.hstar[(size_t)b * R + tid - 64] = t_h[(t + 1) * R + tid - 64];
        }
        __syncthreads();
        if (tid < R) {
            const float a = s_Astep[t * R + tid];
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) acc += (d < Dr && a + s_cd[d * R + tid] > 0.f) ? s_dy[d] : 0.f;
            const float v = acc * w2_mine;
            s_dA[tid] = v; tp.dA[(size_t)b * R + tid] = v;
            __hip_atomic_store(&tp.Astar[(size_t)b * R + tid], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- everything that does not depend on the carried dh, for ALL steps at once, on the two seed BASES (kernels_fast.h)
    MMG_GSTAMP(8); if (tid == 0) { MMG_GT(2048 + 16 * b + 5); }
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
        if (!(t < tstar)) { a1 = 0.f; a2 = 0.f; }                  // receiver message: active while m_{t+1} == 1
        if (!(t <= tstar)) { b1 = 0.f; b2 = 0.f; }
        s1w[u] = a1; s2w[u] = a2; s1z[u] = b1; s2z[u] = b2;
        const int j = i - t * W;
        s_sbas[t * G::SW + j] = a1; s_sbas[(TMAX + t) * G::SW + j] = a2; s_sbas[(2 * TMAX + t) * G::SW + j] = b1; s_sbas[(3 * TMAX + t) * G::SW + j] = b2;
    }
    __syncthreads();
    MMG_GSTAMP(9);
    f32x4 g1 = {0.f, 0.f, 0.f, 0.f}, g2 = {0.f, 0.f, 0.f, 0.f}, p1[4], p2[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { p1[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; p2[nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < W / 4; ++ks) {
        const int o = fi * G::SW + 4 * ks + fq;
        const float aw1 = s_sbas[o], aw2 = s_sbas[TMAX * G::SW + o], az1 = s_sbas[2 * TMAX * G::SW + o], az2 = s_sbas[3 * TMAX * G::SW + o];
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
    {   // W_y1h^T dA enters dh at the output step only (unit k4, reduction slice p4)
        float accy = 0.f;
#pragma unroll
        for (int i = 0; i < R / K4; ++i) accy = fmaf(y1T[i], s_dA[p4 * (R / K4) + i], accy);
        accy = lane_group_sum<K4>(accy);
        if (p4 == 0) s_dAy[k4] = accy;
    }
    float* const s_G1 = lds + G::sg; float* const s_G2 = lds + G::sg + TMAX * G::SR;     // [16][SR] each
    float* const s_H2 = t_z;                            // [16][R] over t_z | t_pz (their contents went out in epilogue 0 / 1)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s_G1[(4 * fq + r) * G::SR + unit] = g1[r]; s_G2[(4 * fq + r) * G::SR + unit] = g2[r]; }
    __syncthreads();
    if (tid == 0) {           // wave 0 wrote dy and A* with write-through stores (output step, above): they only have to have completed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // before the class roles' counter moves -- HERE, where the fragment loads are in anyway
        __hip_atomic_fetch_add(tp.sync + MMG_SYNC_ARR(1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    MMG_GSTAMP(10); if (tid == 0) { MMG_GT(2048 + 16 * b + 6); }
    const int i_st0 = min(tid, 27 * T - 1), i_st1 = min(tid + NT, 27 * T - 1);
    unsigned long long u_st0 = 0, u_st1 = 0, u_bs = 0, u_br = 0;
    auto load_stat_pairs = [&]() {
        u_st0 = ld_ll(tp.statll, i_st0); u_st1 = ld_ll(tp.statll, i_st1);
        u_bs = ld_ll(tp.statll, statll_bs(dm) + so); u_br = ld_ll(tp.statll, statll_br(dm) + so);
    };
    load_stat_pairs(); asm volatile("" ::: "memory");      // (the loads stay ahead of the sweep's LDS reads)
    {
        f32x4 h1 = {0.f, 0.f, 0.f, 0.f}, h2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < R / 4; ++ks) {
            h1 = mfma16(s_G1[fi * G::SR + 4 * ks + fq], whF[ks], h1);
            h2 = mfma16(s_G2[fi * G::SR + 4 * ks + fq], whF[ks], h2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { s_dhin[(4 * fq + r) * R + unit] = h1[r]; s_H2[(4 * fq + r) * R + unit] = h2[r]; }
    }
    MMG_GSTAMP(11); if (tid == 0) { MMG_GT(2048 + 16 * b + 7); }
    float rbs_, rbr_;
    {                                                    // the statistics roles of this launch publish stats, bs, br
        const bool need_b = min(tid, Tm1) <= tstar;      // (the statistics roles write the baseline scores of live rows only)
        for (int spins = 0;; ) {
            const bool fresh = ll_fresh(u_st0, epoch) && ll_fresh(u_st1, epoch) && (!need_b || (ll_fresh(u_bs, epoch) && ll_fresh(u_br, epoch)));
            if (!__any(!fresh)) break;
            if (++spins > (1 << 18)) { if (lane == 0) __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            load_stat_pairs();
        }
        s_statv[i_st0] = ll_value(u_st0); s_statv[i_st1] = ll_value(u_st1);
        __syncthreads();
        MMG_GSTAMP(12); if (tid == 0) { MMG_GT(2048 + 16 * b + 8); }
        const CoefRegs creg = coef_lds_regs(dm, s_statv);
        rbs_ = ll_value(u_bs); rbr_ = ll_value(u_br);
        coef_compute(dm, creg, lc);
    }
    MMG_GSTAMP(2); if (tid == 0) { MMG_GT(2048 + 16 * b + 9); }
    if (tid < TMAX) {
        // per-step scalars of the three streams (model.py:908-922): wh = (L - baseline) cw, ce; stop-bit and MSE seeds
        const int t = min(tid, Tm1);
        const bool on = tid <= tstar;
        s_cf[tid] = on ? (Lrew - rbr_) * lc.cw[T + t] : 0.f;          s_cf[TMAX + tid] = on ? lc.ce[T + t] : 0.f;          // receiver message
        s_cf[2 * TMAX + tid] = on ? (Lrew - rbs_) * lc.cw[2 * T + t] : 0.f; s_cf[3 * TMAX + tid] = on ? lc.ce[2 * T + t] : 0.f;   // sender message
        float dls = 0.f;
        if (on) dls = bit_seed_fast(rs_, rps_, (Lrew - rbr_) * lc.cw[t], lc.ce[t]);
        s_dls[tid] = dls;
        if (tid <= tstar) {
            const size_t row = (size_t)tid * B + b;
            tp.dls[row] = dls;
            tp.dbs[row] = lc.cb[t] * (rbs_ - Lrew);                  // MSE seeds (model.py:971-988)
            tp.dbr[row] = lc.cb[t] * (rbr_ - Lrew);
        }
    }
    __syncthreads();
    MMG_GSTAMP(3); if (tid == 0) { MMG_GT(2048 + 16 * b + 10); }
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
    // ---- the recurrence: two phases and ONE barrier per step (kernels_fast.h)
    auto step_in = [&](int t) {
        float v = fmaf(s_cf[t], s_dhin[t * R + k4], s_cf[TMAX + t] * s_H2[t * R + k4]) + wsk * s_dls[t];
        if (t == tstar) v += s_dAy[k4];
        return v;
    };
    float dh_c = 0.f;
    for (int t = tstar; t >= 0; --t) {
        const size_t row = (size_t)t * B + b;
        float* const dgb = s_dghb + (t & 1) * 3 * R;
        MMG_GSTAMP(16 + 2 * t);
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
            const float vi = (p4 == 0) ? drp : (p4 == 1) ? dup : dnp;
            const float vh = (p4 == 2) ? dnp * rr : vi;
            const int idx = p4 * R + k4;
            if (p4 < 3) { gi[idx] = vi; gh[idx] = vh; dgb[idx] = vh; }
        }
        __syncthreads(); MMG_GSTAMP(16 + 2 * t + 1);
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
    MMG_GSTAMP(4); if (tid == 0) { MMG_GT(2048 + 16 * b + 11); tp.dbg2[2048 + 16 * b + 12] = (long long)tstar; }
}

}  // namespace mmg
