// kernels_tile.h -- the any-shape path of the exchange recurrence as SAMPLE-TILE kernels on the matrix cores.
//
// A workgroup owns a tile of 16 samples (one MFMA M dimension) and runs their conversation: every layer of the step is
// a [16, K] x [K, N] product on v_mfma_f32_16x16x4_f32 (exact fp32), the activations of the tile live in LDS, weights
// and class rows (Cd, desc) are fetched ONCE PER TILE-STEP -- not once per sample-step as the per-sample kernels of
// kernels_fwd.h / kernels_bwd.h do (BASELINE config 5: 13 GB of class rows per minibatch from L2; config 4: 2.4 MB of
// weights per sample-step).  Used for every shape the register-resident kernels (kernels_fast.h) do not cover.
//
//   k_conv_tile     forward conversation of one tile: all T steps, or ONE step with the sender computed by
//                   k_send_s1 / k_send_s2 over the whole batch (shapes whose sender MLP is too large for 4..128 CUs:
//                   config 4, H*W = 262 144 -- there the step is three launches, each filling the chip)
//   k_send_s1/s2    sender MLP of one step over all samples: a = tanh(h_x + c W_c^T + b), z ~ Bernoulli(sigmoid(a W_b^T + b))
//   k_bwd_tile      reverse-time pass of one tile (receiver BPTT; seeds; output-step class gradient)
//   k_send_bwd      sender backward over the (step, sample) rows: not recurrent (the sender's input is detached)
//   k_dC_tile       class-side reduction of the y head over (class tile x sample chunks)
// Reference math: model.py:193-238 (Sender), 333-342 + 411-477 (Receiver), SURVEY.md Appendix A.
#pragma once
#include "device_utils.h"
#include "kernels_fwd.h"
#include "kernels_bwd.h"
#include "kernels_fast.h"
#include "layout.h"

namespace mmg {

#define MMG_TM 16                                   // samples per tile = MFMA M
__host__ __device__ inline int ld16(int n) { return ((n + 15) & ~15) + 4; }      // LDS row stride of a [16][n] activation tile

// 4 consecutive floats of a weight row, BRANCH-FREE (a branch around a load makes hipcc wait vmcnt(0) at the join: one
// memory round trip per k-group instead of a pipelined stream).  Columns beyond ncols are clamped, not zeroed: every caller
// multiplies them with an activation operand that is zero there (LDS tiles are zero-padded) or never reads the result.
// VEC: rows 16-byte aligned and ncols % 4 == 0.
template <bool VEC>
__device__ __forceinline__ float4 ldrow4c(const float* __restrict__ row, int k, int ncols) {
    if (VEC) return *reinterpret_cast<const float4*>(row + min(k, ncols - 4));
    const int km = ncols - 1;
    float4 v;
    v.x = row[min(k, km)]; v.y = row[min(k + 1, km)]; v.z = row[min(k + 2, km)]; v.w = row[min(k + 3, km)];
    return v;
}

// number of K parts a product with ntiles output tiles is split into so that all nw waves work
__host__ __device__ inline int tile_kparts(int ntiles, int nw) {
    int kp = 1;
    while (ntiles * kp * 2 <= nw && kp < 4) kp *= 2;     // (at most 4 parts: the staging area is kparts x 16 x N floats)
    return kp;
}
// floats of the raw accumulator staging area of a product with N outputs
__host__ __device__ inline int tile_raw_floats(int N, int nw) { return MMG_TM * ld16(N) * tile_kparts((N + 15) >> 4, nw); }
__host__ __device__ inline int tile_raw_floats_nn(int N, int nw) { return MMG_TM * ld16(N) * tile_kparts((N + 63) >> 6, nw); }

// ---------------------------------------------------------------------------------------------
// raw[kp][m][n] = sum_{k in part kp} A[m][k] * Wm[n*ldw + k]        ("NT": PyTorch [out,in] weights, k contiguous)
// A: LDS [16][lda], zero-padded to a multiple of 16 columns.  Work items (n-tile, k-part) go round-robin over the waves.
// Fragments: lane (i = l & 15, q = l >> 4) reads float4 A[i][kg + 4q ..] and float4 Wm[n0 + i][kg + 4q ..] and issues four
// MFMAs -- the k index inside a group of 16 is permuted identically for both operands, which leaves the sum unchanged.
// Loads go out in batches of 4 (two output tiles) or 8 k-groups before the first MFMA of the batch.
// No barrier inside; callers __syncthreads() before reading raw.
// ---------------------------------------------------------------------------------------------
template <bool VEC>
__device__ __forceinline__ void tgemm_nt_body(const float* A, int lda, const float* __restrict__ Wm, int ldw, int N, int K,
                                              float* raw, int wave, int nw) {
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int ntiles = (N + 15) >> 4, kparts = tile_kparts(ntiles, nw), ldr = ld16(N);
    const int kgroups = (K + 15) >> 4, per = (kgroups + kparts - 1) / kparts;
    const float* arow = A + i * lda + q * 4;
    const int nitems = ntiles * kparts;
    for (int it = wave; it < nitems; it += 2 * nw) {           // two items per pass: both weight fragments in flight together
        const int it1 = it + nw;
        const bool has1 = it1 < nitems;
        const int tn = it / kparts, kp = it - tn * kparts;
        const int tn1 = (has1 ? it1 : it) / kparts, kp1 = (has1 ? it1 : it) - tn1 * kparts;
        const float* w0 = Wm + (size_t)min(tn * 16 + i, N - 1) * ldw;
        const float* w1 = Wm + (size_t)min(tn1 * 16 + i, N - 1) * ldw;
        const int g0 = kp * per, g1 = min(kgroups, g0 + per), h0 = kp1 * per, h1 = has1 ? min(kgroups, h0 + per) : h0;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < per; s += 8) {                     // 8 k-groups (one float4 each) per item per round trip
            float4 b0[8], b1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                b0[u] = ldrow4c<VEC>(w0, min(g0 + s + u, kgroups - 1) * 16 + q * 4, K);
                b1[u] = ldrow4c<VEC>(w1, min(h0 + s + u, kgroups - 1) * 16 + q * 4, K);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (g0 + s + u < g1) {
                    const float4 a = *reinterpret_cast<const float4*>(arow + (g0 + s + u) * 16);
                    acc0 = mfma16(a.x, b0[u].x, acc0); acc0 = mfma16(a.y, b0[u].y, acc0);
                    acc0 = mfma16(a.z, b0[u].z, acc0); acc0 = mfma16(a.w, b0[u].w, acc0);
                }
                if (h0 + s + u < h1) {
                    const float4 a = *reinterpret_cast<const float4*>(arow + (h0 + s + u) * 16);
                    acc1 = mfma16(a.x, b1[u].x, acc1); acc1 = mfma16(a.y, b1[u].y, acc1);
                    acc1 = mfma16(a.z, b1[u].z, acc1); acc1 = mfma16(a.w, b1[u].w, acc1);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            raw[((kp * MMG_TM) + q * 4 + r) * ldr + tn * 16 + i] = acc0[r];
            if (has1) raw[((kp1 * MMG_TM) + q * 4 + r) * ldr + tn1 * 16 + i] = acc1[r];
        }
    }
}
__device__ __forceinline__ void tgemm_nt_raw(const float* A, int lda, const float* __restrict__ Wm, int ldw, int N, int K,
                                             float* raw, int wave, int nw) {
    tgemm_nt_body<true>(A, lda, Wm, ldw, N, K, raw, wave, nw);      // (host: the tile path needs H, W, R, V multiples of 4)
}

// ---------------------------------------------------------------------------------------------
// raw[kp][m][n] = sum_{k in part kp} A[m][k] * Bm[k*ldb + n]        ("NN": n contiguous -- desc [D, V]; transposed weight
// products of the backward pass, dX = dY . W with W the PyTorch [out,in] matrix).  Work items (group of 64 columns, k-part).
// Lane (i, q) reads float4 Bm[k][g*64 + 4i ..] for its four k = kg + 4q + c and feeds four n-tiles: accumulator j holds
// column g*64 + 4i + j (a column permutation inside the group, undone when raw is written).  Two k-groups (8 loads) in flight.
// ---------------------------------------------------------------------------------------------
template <bool VEC>
__device__ __forceinline__ void tgemm_nn_body(const float* A, int lda, const float* __restrict__ Bm, int ldb, int N, int K,
                                              float* raw, int wave, int nw) {
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int ngroups = (N + 63) >> 6, kparts = tile_kparts(ngroups, nw), ldr = ld16(N);
    const int kgroups = (K + 15) >> 4, per = (kgroups + kparts - 1) / kparts;
    const float* arow = A + i * lda + q * 4;
    for (int it = wave; it < ngroups * kparts; it += nw) {
        const int g = it / kparts, kp = it - g * kparts;
        const int nb = g * 64 + 4 * i;
        const int g0 = kp * per, g1 = min(kgroups, g0 + per);
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kg0 = g0; kg0 < g1; kg0 += 4) {               // 16 row loads in flight
            float4 a[4], b[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kg = min(kg0 + u, kgroups - 1);
#pragma unroll
                for (int c = 0; c < 4; ++c) b[u][c] = ldrow4c<VEC>(Bm + (size_t)min(kg * 16 + q * 4 + c, K - 1) * ldb, nb, N);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(arow + min(kg0 + u, kgroups - 1) * 16);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (kg0 + u < g1) {
                    acc[0] = mfma16(a[u].x, b[u][0].x, acc[0]); acc[1] = mfma16(a[u].x, b[u][0].y, acc[1]); acc[2] = mfma16(a[u].x, b[u][0].z, acc[2]); acc[3] = mfma16(a[u].x, b[u][0].w, acc[3]);
                    acc[0] = mfma16(a[u].y, b[u][1].x, acc[0]); acc[1] = mfma16(a[u].y, b[u][1].y, acc[1]); acc[2] = mfma16(a[u].y, b[u][1].z, acc[2]); acc[3] = mfma16(a[u].y, b[u][1].w, acc[3]);
                    acc[0] = mfma16(a[u].z, b[u][2].x, acc[0]); acc[1] = mfma16(a[u].z, b[u][2].y, acc[1]); acc[2] = mfma16(a[u].z, b[u][2].z, acc[2]); acc[3] = mfma16(a[u].z, b[u][2].w, acc[3]);
                    acc[0] = mfma16(a[u].w, b[u][3].x, acc[0]); acc[1] = mfma16(a[u].w, b[u][3].y, acc[1]); acc[2] = mfma16(a[u].w, b[u][3].z, acc[2]); acc[3] = mfma16(a[u].w, b[u][3].w, acc[3]);
                }
            }
        }
        // (columns clamped by the loads hold duplicates of column N-4..N-1: never read back)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* dst = raw + ((kp * MMG_TM) + q * 4 + r) * ldr + nb;
            if (nb + 3 < ldr) *reinterpret_cast<float4*>(dst) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
        }
    }
}
__device__ __forceinline__ void tgemm_nn_raw(const float* A, int lda, const float* __restrict__ Bm, int ldb, int N, int K,
                                             float* raw, int wave, int nw) {
    tgemm_nn_body<true>(A, lda, Bm, ldb, N, K, raw, wave, nw);
}

// sum of the k-parts of a raw product at (m, n)
__device__ __forceinline__ float raw_sum(const float* raw, int ldr, int kparts, int m, int n) {
    float v = raw[m * ldr + n];
    for (int kp = 1; kp < kparts; ++kp) v += raw[(kp * MMG_TM + m) * ldr + n];
    return v;
}

// ---------------------------------------------------------------------------------------------
// Loops whose body needs a value from global memory: the loads of U iterations are issued together, branch-free (index
// clamped), before the first use -- one memory round trip per U iterations instead of one per iteration.
// ---------------------------------------------------------------------------------------------
template <int NT, int U, class Ld, class Use>
__device__ __forceinline__ void batched_for(int total, Ld ld, Use use) {
    if (total <= 0) return;
    for (int base = threadIdx.x; base < total; base += NT * U) {
        decltype(ld(0)) v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld(min(base + u * NT, total - 1));
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * NT < total) use(base + u * NT, v[u]);
    }
}

// ---------------------------------------------------------------------------------------------
// LDS plan of the forward tile kernel (float offsets); host and device compute it from the dimensions
// ---------------------------------------------------------------------------------------------
struct TileLds {
    int a, rawS, c, z, pz, h, gh, A, gw, g, y, dbar, raw0, raw1, misc, total;
    int bc, hw0, bb, sc, bih, bhh, bh, bw, ws, w2;          // per-column vectors (biases, N = 1 weights), zero padded
    int ldH, ldW, ldR, ld3R, ldD, ldV;
};
__host__ __device__ inline TileLds tile_lds(const Dims& d, int nw, bool with_sender) {
    TileLds L;
    L.ldH = ld16(d.H); L.ldW = ld16(d.W); L.ldR = ld16(d.R); L.ld3R = ld16(3 * d.R); L.ldD = ld16(d.D); L.ldV = ld16(d.V);
    int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 3) & ~3; return at; };
    L.c = take(MMG_TM * L.ldW); L.z = take(MMG_TM * L.ldW); L.pz = take(MMG_TM * L.ldW);
    L.h = take(MMG_TM * L.ldR); L.gh = take(MMG_TM * L.ld3R);
    L.A = take(MMG_TM * L.ldR); L.gw = take(MMG_TM * L.ldR); L.g = take(MMG_TM * L.ldR);
    // the class-logit tile (live P6..P8) shares its space with the sender's hidden tile and the staging area of its
    // first product (live P1..P2)
    {
        const int ysz = MMG_TM * L.ldD, ssz = with_sender ? MMG_TM * L.ldH + tile_raw_floats(d.H, nw) : 0;
        L.y = take(ysz > ssz ? ysz : ssz);
        L.a = L.y; L.rawS = L.y + MMG_TM * L.ldH;
    }
    L.dbar = take(MMG_TM * L.ldV);
    L.bc = take(with_sender ? L.ldH : 0); L.hw0 = take(with_sender ? L.ldH : 0); L.bb = take(with_sender ? L.ldW : 0); L.sc = take(with_sender ? L.ldW : 0);
    L.bih = take(L.ld3R); L.bhh = take(L.ld3R); L.bh = take(L.ldR); L.bw = take(L.ldW); L.ws = take(L.ldR); L.w2 = take(L.ldR);
    int r0 = tile_raw_floats(3 * d.R, nw);
    auto mx = [](int a, int b) { return a > b ? a : b; };
    r0 = mx(r0, tile_raw_floats(d.W, nw)); r0 = mx(r0, tile_raw_floats(d.R, nw)); r0 = mx(r0, tile_raw_floats_nn(d.V, nw));
    L.raw0 = take(r0);
    L.raw1 = take(mx(tile_raw_floats(3 * d.R, nw), tile_raw_floats(d.R, nw)));
    L.misc = take(256);
    L.total = o;
    return L;
}

// misc slots (floats): [0,16) m_t  [16,32) t* (-1: unknown)  [32,48) running stop-prob product  [48,64) stop bit of this step
//                      [64,80) take-output flag
#define TL_MT 0
#define TL_TSTAR 16
#define TL_SPROD 32
#define TL_SBIT 48
#define TL_TAKE 64
#define TL_LIVE 96                                   // [96,112): row is stored this step (valid sample, still in conversation)
#define TL_LIVE2 112                                 // [112,128): ... and its conversation goes on after this step (message rows)

struct F2 { float x, y; };

// ---------------------------------------------------------------------------------------------
// k_conv_tile: grid = ceil(B/16) tiles.  ar.phases bit 0: the sender runs inside (otherwise z / pz of the step are read
// from the tape, written by k_send_s2).  Steps [ar.t_begin, ar.t_end); conversation state (h, last message, masks, t*,
// selected logits) lives in the tape between launches.  alive[t]: number of tiles that still have a live sample when step
// t starts (written here, read by the per-step sender launches to skip steps nobody needs).
// Tape rows are stored for LIVE (step, sample) rows only when samples may stop early (include/mmg.h), exactly the rows the
// per-sample kernels store.
// ---------------------------------------------------------------------------------------------
#ifdef MMG_TIMING
#define MMG_TSTAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) tp.dbg[(slot)] = (long long)wall_clock64(); } while (0)
#else
#define MMG_TSTAMP(slot) do {} while (0)
#endif
template <int NT>
__global__ __launch_bounds__(NT) void k_conv_tile(Dims dm, Params P, Tape tp, ConvArgs ar) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    MMG_TSTAMP(0);
#ifdef MMG_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0) tp.dbg[4] = (long long)clock64();
#endif
    constexpr int nw = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = dm.B, H = dm.H, W = dm.W, R = dm.R, V = dm.V, D = dm.D, T = dm.T;
    const int b0 = blockIdx.x * MMG_TM, nb = min(MMG_TM, B - b0);
    const bool do_sen = (ar.phases & 1) != 0;
    const bool binary = dm.use_binary != 0, train = ar.train != 0;
    const bool may_stop = !ar.run_all && !dm.fixed && train;          // a finished tile stops computing
    const TileLds L = tile_lds(dm, nw, do_sen);
    float* s_a = smem + L.a; float* s_c = smem + L.c; float* s_z = smem + L.z; float* s_pz = smem + L.pz;
    float* s_h = smem + L.h; float* s_gh = smem + L.gh; float* s_A = smem + L.A; float* s_gw = smem + L.gw; float* s_g = smem + L.g;
    float* s_y = smem + L.y; float* s_dbar = smem + L.dbar; float* raw0 = smem + L.raw0; float* raw1 = smem + L.raw1;
    float* misc = smem + L.misc; float* s_w2 = smem + L.w2; float* rawS = smem + L.rawS;
    const int t0 = ar.t_begin;

    for (int i = tid; i < L.total; i += NT) smem[i] = 0.f;             // zero padding of every operand tile
    __syncthreads();
    // ---- conversation state and per-column vectors (every load of this prologue is in flight together)
    if (tid < MMG_TM) {
        const int b = min(b0 + tid, B - 1);
        misc[TL_MT + tid] = (t0 == 0) ? 1.f : tp.mstate[b];
        misc[TL_TSTAR + tid] = (t0 == 0) ? -1.f : (float)tp.tstar[b];
        misc[TL_SPROD + tid] = (t0 == 0) ? 1.f : tp.sprod[b];
        if (t0 == 0 && tid < nb) tp.mask[b0 + tid] = 1;                // stop_mask[0] = ones   model.py:775
    }
    {
        auto vec = [&](int off, const float* src, int n) { batched_for<NT, 2>(n, [&](int i) { return src[i]; }, [&](int i, float v) { smem[off + i] = v; }); };
        if (do_sen) {
            vec(L.bc, P.p[S_CODE_B], H); vec(L.hw0, tp.hw0, H); vec(L.bb, P.p[S_BIN_B], W);
            batched_for<NT, 2>(W, [&](int i) { return P.p[S_CODE_BIAS][i]; }, [&](int i, float v) { smem[L.sc + i] = fsigmoid(v); });
        }
        vec(L.bih, P.p[R_BIH], 3 * R); vec(L.bhh, P.p[R_BHH], 3 * R); vec(L.bh, P.p[R_WH_B], R); vec(L.bw, P.p[R_W_B], W);
        vec(L.ws, P.p[R_S_W], R); vec(L.w2, P.p[R_Y2_W], R);
    }
    if (t0 == 0) {
        for (int idx = tid; idx < nb * R; idx += NT) tp.h[(size_t)b0 * R + idx] = 0.f;          // h_{-1} = 0 (rows b0 .. b0+nb-1 are contiguous)
        for (int idx = tid; idx < MMG_TM * W; idx += NT) s_c[(idx / W) * L.ldW + idx % W] = dm.first_rec;   // model.py:786
    } else {
        batched_for<NT, 4>(MMG_TM * R, [&](int idx) { const int m = idx / R, r = idx - m * R; return tp.h[((size_t)t0 * B + min(b0 + m, B - 1)) * R + r]; },
                           [&](int idx, float v) { const int m = idx / R, r = idx - m * R; s_h[m * L.ldR + r] = v; });
        batched_for<NT, 8>(MMG_TM * W, [&](int idx) { const int m = idx / W, j = idx - m * W; return tp.w[((size_t)(t0 - 1) * B + min(b0 + m, B - 1)) * W + j]; },
                           [&](int idx, float v) { const int m = idx / W, j = idx - m * W; s_c[m * L.ldW + j] = v; });
    }
    // image features of the tile (constant over the conversation): element idx = tid + u*NT of the [16, H] tile in a register
    constexpr int UH = 16;                                              // H <= NT with the sender inside (host: tile_ext otherwise)
    float hxr[UH];
#pragma unroll
    for (int u = 0; u < UH; ++u) {
        const int idx = min(tid + u * NT, MMG_TM * H - 1);
        hxr[u] = do_sen ? tp.hx[(size_t)min(b0 + idx / H, B - 1) * H + idx % H] : 0.f;
    }
    const uint32_t mb_counter = tp.counter[0];
    const float b2 = P.p[R_Y2_B][0], b_s = P.p[R_S_B][0];
    __syncthreads();
    if (may_stop && t0 > 0) {                                           // finished in an earlier launch?
        bool any = false;
        for (int m = 0; m < nb; ++m) any = any || (misc[TL_MT + m] != 0.f);
        if (!any) return;
    }

    int t = t0;
    bool finished = false;
    MMG_TSTAMP(1);
    // A step is eight phases: [products of the phase] barrier [epilogue of the phase] barrier.  The product code exists ONCE
    // (a loop over the phase's job list) -- inlined at every call site the kernel was 125 KB of code, twice the instruction
    // cache, and every step streamed it from L2 again.
    struct Job { const float* A; const float* Wm; float* raw; int lda, ldw, N, K, nn; };
    for (; t < ar.t_end; ++t) {
        const size_t rowb = (size_t)t * B;
        if (tid < MMG_TM) misc[TL_LIVE + tid] = (tid < nb && (!may_stop || misc[TL_MT + tid] != 0.f)) ? 1.f : 0.f;
        bool alive = true;
        for (int ph = 0; ph < 8; ++ph) {
            MMG_TSTAMP(8 + 16 * (t - t0) + ph);
            // per-thread indices are re-derived inside every phase from an opaque copy of the thread id: otherwise the compiler
            // hoists every epilogue's (row, column, tape address) arithmetic out of the step loop and keeps hundreds of
            // registers live across it (the kernel then spills to scratch even at 256 registers per thread)
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));
            const int lane = tid & 63;
            // ---------------- products of the phase
            Job j0, j1;
            int nj = 0;
            auto add = [&](const float* A, int lda, const float* Wm, int ldw, int N, int K, float* raw, int nn) {
                Job& J = nj ? j1 : j0;
                J.A = A; J.lda = lda; J.Wm = Wm; J.ldw = ldw; J.N = N; J.K = K; J.raw = raw; J.nn = nn; ++nj;
            };
            switch (ph) {
            case 0:   // sender hidden (model.py:195-216) and the GRU's hidden-side product (independent of z)
                if (do_sen && t > 0) add(s_c, L.ldW, P.p[S_CODE_W], W, H, W, rawS, 0);
                add(s_h, L.ldR, P.p[R_WHH], R, 3 * R, R, raw1, 0); break;
            case 1: if (do_sen) add(s_a, L.ldH, P.p[S_BIN_W], H, W, H, raw0, 0); break;                 // sender logits, model.py:218
            case 2: add(s_z, L.ldW, P.p[R_WIH], W, 3 * R, W, raw0, 0); break;                           // GRU input side, model.py:340
            case 3: add(s_h, L.ldR, P.p[R_Y1_W], R + V, R, R, raw0, 0); add(s_h, L.ldR, P.p[R_WH_W], R, R, R, raw1, 0); break;   // A (App. A.2), w_h h
            case 5: add(s_y, L.ldD, ar.desc, V, V, D, raw0, 1); break;                                  // softmax . desc, model.py:442-449
            case 6: add(s_dbar, L.ldV, P.p[R_WD_W], V, R, V, raw0, 0); break;                           // w_d dbar, model.py:452
            case 7: add(s_g, L.ldR, P.p[R_W_W], R, W, R, raw0, 0); break;                               // receiver message logits, model.py:454
            default: break;
            }
            for (int j = 0; j < nj; ++j) {
                const float* jA = j ? j1.A : j0.A; const float* jW = j ? j1.Wm : j0.Wm; float* jr = j ? j1.raw : j0.raw;
                const int jlda = j ? j1.lda : j0.lda, jldw = j ? j1.ldw : j0.ldw, jN = j ? j1.N : j0.N, jK = j ? j1.K : j0.K;
                if ((j ? j1.nn : j0.nn)) tgemm_nn_raw(jA, jlda, jW, jldw, jN, jK, jr, wave, nw);
                else tgemm_nt_raw(jA, jlda, jW, jldw, jN, jK, jr, wave, nw);
            }
            if (ph == 2 && binary && tid < 256) {                        // log-likelihood / neg-entropy of the sender's bits, model.py:908-922
                const int m = tid >> 4, l16 = tid & 15;
                float lpv = 0.f, nev = 0.f;
                for (int j = l16; j < W; j += 16) {
                    const float p = s_pz[m * L.ldW + j], zz = s_z[m * L.ldW + j];
                    const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
                    lpv += zz * l1 + (1.f - zz) * l0; nev += p * l1 + (1.f - p) * l0;
                }
                lpv = dpp_group_sum<16>(lpv); nev = dpp_group_sum<16>(nev);
                if (l16 == 0 && misc[TL_LIVE + m] != 0.f) { tp.lp_z[rowb + b0 + m] = lpv; tp.ne_z[rowb + b0 + m] = nev; }
            }
            if (ph == 3 && tid < 256) {                                  // stop bit (model.py:414-427)
                const int m = tid >> 4, l16 = tid & 15;
                const float* s_ws = smem + L.ws;
                float acc = 0.f;
                for (int r = l16; r < R; r += 16) acc = fmaf(s_ws[r], s_h[m * L.ldR + r], acc);
                acc = dpp_group_sum<16>(acc);
                if (l16 == 0) {
                    const int b = min(b0 + m, B - 1);
                    const float p = fsigmoid(acc + b_s);
                    float sv;
                    if (train) {
                        const float u = ar.u_s ? ar.u_s[rowb + b] : philox_uniform(ar.seed, (uint32_t)(t * dm.Bg + dm.boff + b), mb_counter, 1u);
                        sv = (u < p) ? 1.f : 0.f;                                           // model.py:420
                    } else {
                        const float prod = dm.s_prob_prod ? misc[TL_SPROD + m] * p : p;    // model.py:423-426
                        misc[TL_SPROD + m] = prod;
                        sv = rintf(prod);                                                   // model.py:427
                    }
                    misc[TL_SBIT + m] = sv;
                    if (misc[TL_LIVE + m] != 0.f) {
                        tp.s[rowb + b] = sv; tp.ps[rowb + b] = p;
                        const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
                        tp.lp_s[rowb + b] = sv * l1 + (1.f - sv) * l0;
                        tp.ne_s[rowb + b] = p * l1 + (1.f - p) * l0;
                    }
                }
            }
            __syncthreads();
            // ---------------- epilogue of the phase
            if (ph == 0) {
                if (do_sen) {
                    const int kp = tile_kparts((H + 15) >> 4, nw);
                    const float* s_bc = smem + L.bc; const float* s_hw0 = smem + L.hw0;
#pragma unroll
                    for (int u = 0; u < UH; ++u) {
                        const int idx = tid + u * NT;
                        if (idx < MMG_TM * H) {
                            const int m = idx / H, n = idx - m * H;
                            const float hw = (t == 0) ? s_hw0[n] : raw_sum(rawS, L.ldH, kp, m, n) + s_bc[n];
                            const float av = ftanh(hxr[u] + hw);                            // model.py:216
                            s_a[m * L.ldH + n] = av;
                            if (misc[TL_LIVE + m] != 0.f) tp.a[(rowb + b0 + m) * H + n] = av;
                        }
                    }
                    // (s_a shares its space with the class-logit tile: restore the zero padding of its K dimension)
                    for (int idx = tid; idx < MMG_TM * (L.ldH - H); idx += NT) s_a[(idx / (L.ldH - H)) * L.ldH + H + idx % (L.ldH - H)] = 0.f;
                    const float* s_sc = smem + L.sc;
                    for (int idx = tid; idx < nb * W; idx += NT) {
                        const int m = idx / W, j = idx - m * W;
                        if (misc[TL_LIVE + m] == 0.f) continue;
                        const float cv = s_c[m * L.ldW + j];
                        tp.zr[(rowb + b0 + m) * W + j] = cv;                                // z_r of baseline_sen, model.py:836
                        tp.c[(rowb + b0 + m) * W + j] = (t == 0) ? s_sc[j] : cv;
                    }
                }
                const float* s_bhh = smem + L.bhh;
                const int kp = tile_kparts((3 * R + 15) >> 4, nw);
                for (int idx = tid; idx < MMG_TM * 3 * R; idx += NT) {
                    const int m = idx / (3 * R), n = idx - m * 3 * R;
                    s_gh[m * L.ld3R + n] = raw_sum(raw1, L.ld3R, kp, m, n) + s_bhh[n];
                }
            } else if (ph == 1) {
                if (do_sen) {                                            // sample the sender's message, model.py:218-236
                    const float* s_bb = smem + L.bb;
                    const int kp = tile_kparts((W + 15) >> 4, nw);
                    for (int idx = tid; idx < MMG_TM * W; idx += NT) {
                        const int m = idx / W, n = idx - m * W, b = min(b0 + m, B - 1);
                        const float lz = raw_sum(raw0, L.ldW, kp, m, n) + s_bb[n];
                        float zz = lz, pp = 0.f;
                        if (binary) {
                            pp = fsigmoid(lz);
                            if (train) {
                                const float u = ar.u_z ? ar.u_z[(rowb + b) * W + n]
                                                       : philox_uniform(ar.seed, (uint32_t)((t * dm.Bg + dm.boff + b) * W + n), mb_counter, 0u);
                                zz = (u < pp) ? 1.f : 0.f;                                  // model.py:227
                            } else zz = rintf(pp);                                          // model.py:229
                            if (misc[TL_LIVE + m] != 0.f) tp.pz[(rowb + b) * W + n] = pp;
                        }
                        s_z[m * L.ldW + n] = zz; s_pz[m * L.ldW + n] = pp;
                        if (misc[TL_LIVE + m] != 0.f) tp.z[(rowb + b) * W + n] = zz;
                    }
                } else {
                    batched_for<NT, 8>(MMG_TM * W, [&](int idx) {
                            const int m = idx / W, n = idx - m * W;
                            const size_t o = (rowb + min(b0 + m, B - 1)) * W + n;
                            return F2{tp.z[o], tp.pz[o]};
                        }, [&](int idx, F2 v) { const int m = idx / W, n = idx - m * W; s_z[m * L.ldW + n] = v.x; s_pz[m * L.ldW + n] = binary ? v.y : 0.f; });
                }
            } else if (ph == 2) {                                        // GRUCell (model.py:340); gate order r, u, n
                const float* s_bih = smem + L.bih;
                const int kp = tile_kparts((3 * R + 15) >> 4, nw);
                for (int idx = tid; idx < MMG_TM * R; idx += NT) {
                    const int m = idx / R, i = idx - m * R;
                    const float gir = raw_sum(raw0, L.ld3R, kp, m, i) + s_bih[i], giu = raw_sum(raw0, L.ld3R, kp, m, R + i) + s_bih[R + i];
                    const float gin = raw_sum(raw0, L.ld3R, kp, m, 2 * R + i) + s_bih[2 * R + i];
                    const float* gh = s_gh + m * L.ld3R;
                    const float rr = fsigmoid(gir + gh[i]), uu = fsigmoid(giu + gh[R + i]);
                    const float ghn = gh[2 * R + i];
                    const float nn = ftanh(gin + rr * ghn);
                    const float hv = nn + uu * (s_h[m * L.ldR + i] - nn);
                    s_h[m * L.ldR + i] = hv;
                    if (misc[TL_LIVE + m] != 0.f) {
                        float* gr = tp.gru + (rowb + b0 + m) * 4 * R;
                        gr[i] = rr; gr[R + i] = uu; gr[2 * R + i] = nn; gr[3 * R + i] = ghn;
                        tp.h[((size_t)(t + 1) * B + b0 + m) * R + i] = hv;
                    }
                }
            } else if (ph == 3) {
                const float* s_bh = smem + L.bh;
                const int kp = tile_kparts((R + 15) >> 4, nw);
                for (int idx = tid; idx < MMG_TM * R; idx += NT) {
                    const int m = idx / R, n = idx - m * R;
                    s_A[m * L.ldR + n] = raw_sum(raw0, L.ldR, kp, m, n);
                    s_gw[m * L.ldR + n] = raw_sum(raw1, L.ldR, kp, m, n) + s_bh[n];
                }
                if (tid < MMG_TM) {                                      // stop-mask bookkeeping (model.py:852) -- per sample
                    const int m = tid;
                    const float m_t = misc[TL_MT + m], sv = misc[TL_SBIT + m];
                    const float m_next = fminf(m_t, sv);
                    const bool unknown = misc[TL_TSTAR + m] < 0.f;
                    const bool take = dm.fixed ? (t == T - 1) : (unknown && (m_next == 0.f || t == T - 1));
                    misc[TL_TAKE + m] = take ? 1.f : 0.f;
                    if (take) misc[TL_TSTAR + m] = (float)t;
                    misc[TL_MT + m] = m_next;
                    if (misc[TL_LIVE + m] != 0.f) tp.mask[(size_t)(t + 1) * B + b0 + m] = (uint8_t)(m_next != 0.f);
                    misc[TL_LIVE2 + m] = (misc[TL_LIVE + m] != 0.f && (!may_stop || m_next != 0.f)) ? 1.f : 0.f;
                }
            } else if (ph == 4) {
                // ===== class logits  y[m][d] = b_y2 + sum_r w_y2[r] relu(A[m][r] + Cd[d][r])     (model.py:432-433)
                if (D * MMG_TM <= 8 * NT) {
                    // few classes: one (class, sample) pair per thread pass; the class row goes out as 8 float4 loads at a time
                    for (int idx = tid; idx < D * MMG_TM; idx += NT) {
                        const int d = idx >> 4, m = idx & 15;
                        const float* crow = tp.Cd + (size_t)d * R;
                        const float* arow = s_A + m * L.ldR;
                        float a0 = 0.f, a1 = 0.f;
                        for (int r0 = 0; r0 < R; r0 += 32) {
                            float4 cq[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) cq[u] = ldrow4c<true>(crow, r0 + 4 * u, R);
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int r = r0 + 4 * u;
                                if (r < R) {                                                // (w2 is zero beyond R: clamped class values are harmless)
                                    const float4 av = *reinterpret_cast<const float4*>(arow + r);
                                    const float4 wq = *reinterpret_cast<const float4*>(s_w2 + r);
                                    a0 = fmaf(wq.x, fmaxf(av.x + cq[u].x, 0.f), a0); a1 = fmaf(wq.y, fmaxf(av.y + cq[u].y, 0.f), a1);
                                    a0 = fmaf(wq.z, fmaxf(av.z + cq[u].z, 0.f), a0); a1 = fmaf(wq.w, fmaxf(av.w + cq[u].w, 0.f), a1);
                                }
                            }
                        }
                        const float yv = (a0 + a1) + b2;
                        s_y[m * L.ldD + d] = yv;
                        if (misc[TL_LIVE + m] != 0.f) tp.y[(rowb + b0 + m) * D + d] = yv;
                    }
                } else {
                    // many classes: a thread owns a class and keeps its 16 samples' partial sums in registers; the class row is
                    // read ONCE for the whole tile (negated transposed copy CdT[r][d]: consecutive lanes, consecutive addresses)
                    for (int d0 = 0; d0 < D; d0 += NT) {
                        const int d = d0 + tid;
                        const int dc = min(d, D - 1);
                        const float cyd = tp.cy[dc];
                        float acc[MMG_TM];
#pragma unroll
                        for (int m = 0; m < MMG_TM; ++m) acc[m] = 0.f;
                        float cv[16], cn[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) cv[u] = tp.CdT[(size_t)min(u, R - 1) * D + dc];
#pragma unroll 1
                        for (int r0 = 0; r0 < R; r0 += 16) {           // chunk r0 in registers, chunk r0 + 16 in flight
#pragma unroll
                            for (int u = 0; u < 16; ++u) cn[u] = tp.CdT[(size_t)min(r0 + 16 + u, R - 1) * D + dc];
#pragma unroll
                            for (int u = 0; u < 16; u += 4) {
                                const float4 wq = *reinterpret_cast<const float4*>(s_w2 + r0 + u);           // zero beyond R
#pragma unroll
                                for (int m = 0; m < MMG_TM; ++m) {
                                    const float4 av = *reinterpret_cast<const float4*>(s_A + m * L.ldR + r0 + u);    // LDS broadcast
                                    // relu(A + c) = max(A, -c) + c: CdT holds -c, the sum_r w2[r] c[r] part is the per-class constant cy
                                    acc[m] = fmaf(wq.x, fmaxf(av.x, cv[u]), acc[m]); acc[m] = fmaf(wq.y, fmaxf(av.y, cv[u + 1]), acc[m]);
                                    acc[m] = fmaf(wq.z, fmaxf(av.z, cv[u + 2]), acc[m]); acc[m] = fmaf(wq.w, fmaxf(av.w, cv[u + 3]), acc[m]);
                                }
                            }
#pragma unroll
                            for (int u = 0; u < 16; ++u) cv[u] = cn[u];
                        }
                        if (d < D) {
#pragma unroll
                            for (int m = 0; m < MMG_TM; ++m) {
                                const float yv = acc[m] + cyd;
                                s_y[m * L.ldD + d] = yv;
                                if (misc[TL_LIVE + m] != 0.f) tp.y[(rowb + b0 + m) * D + d] = yv;
                            }
                        }
                    }
                }
                for (int idx = tid; idx < MMG_TM * (L.ldD - D); idx += NT) s_y[(idx / (L.ldD - D)) * L.ldD + D + idx % (L.ldD - D)] = 0.f;   // K padding of the mixture product
                __syncthreads();
                // output step of a sample: its logits go to tape.outp (model.py:1261-1264); the tile may be done
                alive = false;
                for (int m = 0; m < nb; ++m) alive = alive || (misc[TL_MT + m] != 0.f);
                for (int m = 0; m < nb; ++m) {
                    if (misc[TL_TAKE + m] != 0.f)
                        for (int d = tid; d < D; d += NT) tp.outp[(size_t)(b0 + m) * D + d] = s_y[m * L.ldD + d];
                }
                if (tid == 0 && t + 1 < T && alive) atomicAdd(&tp.alive[t + 1], 1);
                if (may_stop && !alive) break;
                __syncthreads();                                        // the selected rows are copied before the softmax overwrites them
                // softmax(y) (detached, model.py:441): wave per sample row, in place
                for (int m = wave; m < MMG_TM; m += nw) {
                    float* yr = s_y + m * L.ldD;
                    float mx = -3.0e38f;
                    for (int d0 = lane; d0 < D; d0 += 64 * 8) {          // 8 LDS reads in flight per lane
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = yr[min(d0 + 64 * u, D - 1)];
#pragma unroll
                        for (int u = 0; u < 8; ++u) mx = fmaxf(mx, v[u]);
                    }
                    mx = dpp_wave_max(mx);
                    float se = 0.f;
                    for (int d0 = lane; d0 < D; d0 += 64 * 8) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = yr[min(d0 + 64 * u, D - 1)];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const float e = __expf(v[u] - mx);
                            if (d0 + 64 * u < D) { yr[d0 + 64 * u] = e; se += e; }
                        }
                    }
                    se = dpp_wave_sum(se);
                    const float inv = __builtin_amdgcn_rcpf(se);
                    for (int d0 = lane; d0 < D; d0 += 64 * 8) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = yr[min(d0 + 64 * u, D - 1)];
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (d0 + 64 * u < D) yr[d0 + 64 * u] = v[u] * inv;
                    }
                }
            } else if (ph == 5) {                                        // description mixture
                const int kp = tile_kparts((V + 63) >> 6, nw);
                for (int idx = tid; idx < MMG_TM * V; idx += NT) {
                    const int m = idx / V, v = idx - m * V;
                    const float dv = raw_sum(raw0, L.ldV, kp, m, v);
                    s_dbar[m * L.ldV + v] = dv;
                    if (misc[TL_LIVE2 + m] != 0.f) tp.dbar[(rowb + b0 + m) * V + v] = dv;
                }
            } else if (ph == 6) {                                        // h_w = tanh(w_h h + b_h + w_d dbar)   (model.py:452)
                const int kp = tile_kparts((R + 15) >> 4, nw);
                for (int idx = tid; idx < MMG_TM * R; idx += NT) {
                    const int m = idx / R, n = idx - m * R;
                    const float gv = ftanh(s_gw[m * L.ldR + n] + raw_sum(raw0, L.ldR, kp, m, n));
                    s_g[m * L.ldR + n] = gv;
                    if (misc[TL_LIVE2 + m] != 0.f) tp.g[(rowb + b0 + m) * R + n] = gv;
                }
            } else {                                                     // receiver message (model.py:454-475)
                const float* s_bw = smem + L.bw;
                const int kp = tile_kparts((W + 15) >> 4, nw);
                for (int idx = tid; idx < MMG_TM * W; idx += NT) {
                    const int m = idx / W, n = idx - m * W, b = min(b0 + m, B - 1);
                    const float lw = raw_sum(raw0, L.ldW, kp, m, n) + s_bw[n];
                    float wv = lw, pp = 0.f;
                    if (binary) {
                        pp = fsigmoid(lw);
                        if (train) {
                            const float u = ar.u_w ? ar.u_w[(rowb + b) * W + n]
                                                   : philox_uniform(ar.seed, (uint32_t)((t * dm.Bg + dm.boff + b) * W + n), mb_counter, 2u);
                            wv = (u < pp) ? 1.f : 0.f;                                      // model.py:460
                        } else wv = rintf(pp);                                              // model.py:462
                        if (misc[TL_LIVE2 + m] != 0.f) tp.pw[(rowb + b) * W + n] = pp;
                    }
                    s_c[m * L.ldW + n] = wv; s_pz[m * L.ldW + n] = pp;
                    if (misc[TL_LIVE2 + m] != 0.f) tp.w[(rowb + b) * W + n] = wv;
                }
                __syncthreads();
                if (binary && tid < 256) {
                    const int m = tid >> 4, l16 = tid & 15;
                    float lpv = 0.f, nev = 0.f;
                    for (int j = l16; j < W; j += 16) {
                        const float p = s_pz[m * L.ldW + j], wv = s_c[m * L.ldW + j];
                        const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
                        lpv += wv * l1 + (1.f - wv) * l0; nev += p * l1 + (1.f - p) * l0;
                    }
                    lpv = dpp_group_sum<16>(lpv); nev = dpp_group_sum<16>(nev);
                    if (l16 == 0 && misc[TL_LIVE2 + m] != 0.f) { tp.lp_w[rowb + b0 + m] = lpv; tp.ne_w[rowb + b0 + m] = nev; }
                }
            }
            __syncthreads();
        }
        if (may_stop && !alive) { ++t; finished = true; break; }
    }
    MMG_TSTAMP(2);
#ifdef MMG_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0) tp.dbg[5] = (long long)clock64();
#endif
    // ---- state hand-over to the next launch of this conversation
    if (tid < nb) {
        tp.tstar[b0 + tid] = (int)misc[TL_TSTAR + tid];
        tp.sprod[b0 + tid] = misc[TL_SPROD + tid];
        tp.mstate[b0 + tid] = misc[TL_MT + tid];
    }
    if (!(finished || t == T)) return;                                  // (per-step launches: the conversation goes on)
    // ---- output selection, log-softmax, reward, top-k (model.py:1264-1275, 1333-1339): wave per sample
    __syncthreads();                                                    // tape.outp rows written above are visible to the workgroup
    for (int m = wave; m < nb; m += nw) {
        const int b = b0 + m;
        const float* o = tp.outp + (size_t)b * D;
        float mx = -3.0e38f;
        for (int d0 = lane; d0 < D; d0 += 64 * 8) {             // 8 logits in flight per lane
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = o[min(d0 + 64 * u, D - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) mx = fmaxf(mx, v[u]);
        }
        mx = dpp_wave_max(mx);
        float se = 0.f;
        for (int d0 = lane; d0 < D; d0 += 64 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = o[min(d0 + 64 * u, D - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) se += (d0 + 64 * u < D) ? __expf(v[u] - mx) : 0.f;
        }
        const float lse = mx + flog(dpp_wave_sum(se));
        const int tgt = ar.target ? (int)ar.target[b] : -1;
        const float dt = (tgt >= 0) ? (o[max(tgt, 0)] - lse) : 0.f;
        float above = 0.f;
        for (int d0 = lane; d0 < D; d0 += 64 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = o[min(d0 + 64 * u, D - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int d = d0 + 64 * u;
                if (d < D) {
                    const float ld = v[u] - lse;
                    tp.dist[(size_t)b * D + d] = ld;
                    tp.sm[(size_t)b * D + d] = __expf(ld);
                    if (tgt >= 0 && ld > dt) above += 1.f;
                }
            }
        }
        above = dpp_wave_sum(above);
        if (lane == 0) {
            tp.logs[b] = dt;
            tp.hit[b] = (tgt >= 0 && above < (float)dm.top_k) ? 1 : 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Per-step sender launches (large sender MLP, few samples: BASELINE config 4).  One workgroup per 16x16 output tile,
// its four waves split K (gemm_nt_acc), so a step's [B, H] / [B, W] products spread over the whole chip.
//   k_send_s1: a_t = tanh(h_x + c_t W_c^T + b_c), c_t = w_{t-1} (t = 0: hw0 precomputed by k_prep)      model.py:195-216
//   k_send_s2: z_t ~ Bernoulli(sigmoid(a_t W_b^T + b_b)) (train) | round (eval) | logits (continuous)   model.py:218-236
// skip != 0: return when no tile has a live sample at step t (alive[t] == 0).
// ---------------------------------------------------------------------------------------------
// acc (this wave's K share) of out tile (tm, tn): X[M, K] . Wm[N, K]^T, both with k contiguous
template <bool VEC>
__device__ __forceinline__ f32x4 gemm_nt_acc(int tm, int tn, const float* __restrict__ X, int ldx, const float* __restrict__ Wm, int ldw,
                                             int M, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const float* xr = X + (size_t)min(tm * 16 + i, M - 1) * ldx;
    const float* wr = Wm + (size_t)min(tn * 16 + i, N - 1) * ldw;
    // (k beyond K: both operands are clamped to the same valid columns -- the products are masked below)
    const int kgroups = (K + 15) >> 4, per = (kgroups + 3) >> 2;
    const int g0 = wave * per, g1 = min(kgroups, g0 + per);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int gb = g0; gb < g1; gb += 8) {                  // 8 k-groups (16 float4 loads) in flight per pass
        float4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = min(gb + u, max(g1 - 1, 0)) * 16 + q * 4;
            a[u] = ldrow4c<VEC>(xr, k, K); b[u] = ldrow4c<VEC>(wr, k, K);
            if (k + 3 >= K) {                               // tail group of a K that is not a multiple of 16 (no load inside)
                a[u].x = (k < K) ? a[u].x : 0.f; a[u].y = (k + 1 < K) ? a[u].y : 0.f;
                a[u].z = (k + 2 < K) ? a[u].z : 0.f; a[u].w = (k + 3 < K) ? a[u].w : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            if (gb + u < g1) {
                acc0 = mfma16(a[u].x, b[u].x, acc0); acc0 = mfma16(a[u].y, b[u].y, acc0);
                acc0 = mfma16(a[u].z, b[u].z, acc0); acc0 = mfma16(a[u].w, b[u].w, acc0);
            }
            if (gb + u + 1 < g1) {
                acc1 = mfma16(a[u + 1].x, b[u + 1].x, acc1); acc1 = mfma16(a[u + 1].y, b[u + 1].y, acc1);
                acc1 = mfma16(a[u + 1].z, b[u + 1].z, acc1); acc1 = mfma16(a[u + 1].w, b[u + 1].w, acc1);
            }
        }
    }
    return acc0 + acc1;
}

__global__ __launch_bounds__(MMG_BLOCK) void k_send_s1(Dims dm, Params P, Tape tp, int t, int skip) {
    __shared__ float s_acc[4][16][17];
    if (skip && tp.alive[t] == 0) return;
    const int B = dm.B, H = dm.H, W = dm.W;
    const int tiles_n = (H + 15) >> 4;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    const size_t rowb = (size_t)t * B;
    if (t > 0) {
        const f32x4 acc = ((W & 3) == 0) ? gemm_nt_acc<true>(tm, tn, tp.w + (size_t)(t - 1) * B * W, W, P.p[S_CODE_W], W, B, H, W)
                                          : gemm_nt_acc<false>(tm, tn, tp.w + (size_t)(t - 1) * B * W, W, P.p[S_CODE_W], W, B, H, W);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_acc[wave][q * 4 + r][i] = acc[r];
    }
    __syncthreads();
    {
        const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
        const int b = tm * 16 + r, n = tn * 16 + c;
        if (b < B && n < H) {
            const float hw = (t == 0) ? tp.hw0[n]
                                      : (s_acc[0][r][c] + s_acc[1][r][c]) + (s_acc[2][r][c] + s_acc[3][r][c]) + P.p[S_CODE_B][n];
            tp.a[(rowb + b) * H + n] = ftanh(tp.hx[(size_t)b * H + n] + hw);
        }
    }
    if (tn == 0) {                                          // code input rows of this sample tile (tapes c, zr)
        for (int idx = threadIdx.x; idx < 16 * W; idx += MMG_BLOCK) {
            const int m = idx / W, j = idx - m * W, b = tm * 16 + m;
            if (b < B) {
                const float cv = (t == 0) ? dm.first_rec : tp.w[((size_t)(t - 1) * B + b) * W + j];
                tp.zr[(rowb + b) * W + j] = cv;
                tp.c[(rowb + b) * W + j] = (t == 0) ? fsigmoid(P.p[S_CODE_BIAS][j]) : cv;
            }
        }
    }
}

__global__ __launch_bounds__(MMG_BLOCK) void k_send_s2(Dims dm, Params P, Tape tp, ConvArgs ar, int t, int skip) {
    __shared__ float s_acc[4][16][17];
    if (skip && tp.alive[t] == 0) return;
    const int B = dm.B, H = dm.H, W = dm.W;
    const int tiles_n = (W + 15) >> 4;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
    const size_t rowb = (size_t)t * B;
    const f32x4 acc = ((H & 3) == 0) ? gemm_nt_acc<true>(tm, tn, tp.a + rowb * H, H, P.p[S_BIN_W], H, B, W, H)
                                      : gemm_nt_acc<false>(tm, tn, tp.a + rowb * H, H, P.p[S_BIN_W], H, B, W, H);
#pragma unroll
    for (int r = 0; r < 4; ++r) s_acc[wave][q * 4 + r][i] = acc[r];
    __syncthreads();
    const int r = threadIdx.x >> 4, c = threadIdx.x & 15;
    const int b = tm * 16 + r, n = tn * 16 + c;
    if (b < B && n < W) {
        const float lz = (s_acc[0][r][c] + s_acc[1][r][c]) + (s_acc[2][r][c] + s_acc[3][r][c]) + P.p[S_BIN_B][n];
        float zz = lz;
        if (dm.use_binary) {
            const float pp = fsigmoid(lz);
            if (ar.train) {
                const float u = ar.u_z ? ar.u_z[(rowb + b) * W + n]
                                       : philox_uniform(ar.seed, (uint32_t)((t * dm.Bg + dm.boff + b) * W + n), tp.counter[0], 0u);
                zz = (u < pp) ? 1.f : 0.f;
            } else zz = rintf(pp);
            tp.pz[(rowb + b) * W + n] = pp;
        }
        tp.z[(rowb + b) * W + n] = zz;
    }
}

}  // namespace mmg
