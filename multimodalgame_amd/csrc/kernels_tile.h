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
// (st_wt / st_wt4 / ld_cc / ld_cc2 / ld_cc4: device_utils.h)
// values of lanes l, l + 2, l + 4, l + 6 (the results of four neighbouring 2-lane groups) as one float4 in lane l
__device__ __forceinline__ float4 gather4_even(float v) {
    return make_float4(v, __shfl_down(v, 2), __shfl_down(v, 4), __shfl_down(v, 6));
}
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
// Two small "NT" products of one phase (sender hidden + GRU hidden side; the two heads on h) in ONE memory round trip: run one
// after the other, each pays its own weight-load latency (~3 us per product per step at 128 tiles).  Items of both jobs are
// numbered together, a wave takes up to four of them and issues ALL their weight loads (<= 4 k-groups each) before the first
// MFMA.  Same staging layout as tgemm_nt_body.  Needs <= 4 k-groups per item and <= 4 * nw items in total (nt_pair_fits).
// ---------------------------------------------------------------------------------------------
struct NtJob { const float* A; const float* Wm; float* raw; int lda, ldw, N, K; };
__device__ __forceinline__ bool nt_pair_fits(const NtJob& a, const NtJob& b, int nw) {
    const int ta = (a.N + 15) >> 4, tb = (b.N + 15) >> 4, ka = tile_kparts(ta, nw), kb = tile_kparts(tb, nw);
    const int ga = (a.K + 15) >> 4, gb = (b.K + 15) >> 4;
    return (ga + ka - 1) / ka <= 4 && (gb + kb - 1) / kb <= 4 && ta * ka + tb * kb <= 4 * nw;
}
__device__ __forceinline__ void tgemm_nt_pair(const NtJob& ja, const NtJob& jb, int wave, int nw) {
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int ta = (ja.N + 15) >> 4, ka = tile_kparts(ta, nw), na = ta * ka;
    const int tb = (jb.N + 15) >> 4, kb = tile_kparts(tb, nw), nb_ = tb * kb;
    float4 w[4][4];
    int s_tn[4], s_kp[4], s_g0[4], s_n[4];
    bool s_b[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        const int it = wave + sl * nw;
        const bool isb = it >= na;
        const int li = isb ? it - na : it;
        const bool has = isb ? (li < nb_) : true;
        const int kparts = isb ? kb : ka, N = isb ? jb.N : ja.N, K = isb ? jb.K : ja.K, ldw = isb ? jb.ldw : ja.ldw;
        const float* Wm = isb ? jb.Wm : ja.Wm;
        const int kgroups = (K + 15) >> 4, per = (kgroups + kparts - 1) / kparts;
        const int lc = has ? li : 0;
        const int tn = lc / kparts, kp = lc - tn * kparts, g0 = kp * per;
        s_b[sl] = isb; s_tn[sl] = tn; s_kp[sl] = kp; s_g0[sl] = g0; s_n[sl] = has ? max(0, min(kgroups, g0 + per) - g0) : 0;
        const float* wrow = Wm + (size_t)min(tn * 16 + i, N - 1) * ldw;
#pragma unroll
        for (int u = 0; u < 4; ++u) w[sl][u] = ldrow4c<true>(wrow, min(g0 + u, kgroups - 1) * 16 + q * 4, K);
    }
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
        if (s_n[sl] == 0) continue;
        const float* A = s_b[sl] ? jb.A : ja.A;
        const int lda = s_b[sl] ? jb.lda : ja.lda, N = s_b[sl] ? jb.N : ja.N;
        float* raw = s_b[sl] ? jb.raw : ja.raw;
        const int ldr = ld16(N);
        const float* arow = A + i * lda + q * 4 + s_g0[sl] * 16;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            if (u < s_n[sl]) {
                const float4 a = *reinterpret_cast<const float4*>(arow + u * 16);
                acc0 = mfma16(a.x, w[sl][u].x, acc0); acc0 = mfma16(a.y, w[sl][u].y, acc0);
                acc0 = mfma16(a.z, w[sl][u].z, acc0); acc0 = mfma16(a.w, w[sl][u].w, acc0);
            }
            if (u + 1 < s_n[sl]) {
                const float4 a = *reinterpret_cast<const float4*>(arow + (u + 1) * 16);
                acc1 = mfma16(a.x, w[sl][u + 1].x, acc1); acc1 = mfma16(a.y, w[sl][u + 1].y, acc1);
                acc1 = mfma16(a.z, w[sl][u + 1].z, acc1); acc1 = mfma16(a.w, w[sl][u + 1].w, acc1);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) raw[((s_kp[sl] * MMG_TM) + q * 4 + r) * ldr + s_tn[sl] * 16 + i] = acc0[r] + acc1[r];
    }
}

// ---------------------------------------------------------------------------------------------
// Register-resident weight fragments for loops that multiply by the SAME matrix every step (the persistent roles): a wave's
// work item (n-tile, k-part) of tgemm_nt is fixed, so its weight fragment -- one float4 per k-group -- is loaded once and
// kept in MAXKG registers; the per-step product then has no global load at all.  Item numbering and staging layout are
// tgemm_nt_body's (item `it` = tn * kparts + kp), so products from fragments and from memory are interchangeable.
// ---------------------------------------------------------------------------------------------
template <int MAXKG>
struct WFrag { float4 b[MAXKG]; int tn, kp, g0, n; };       // n = k-groups held (0: no item)

template <int MAXKG>
__device__ __forceinline__ void wfrag_load(WFrag<MAXKG>& f, const float* __restrict__ Wm, int ldw, int N, int K, int it, int nw) {
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int ntiles = (N + 15) >> 4, kparts = tile_kparts(ntiles, nw);
    const int kgroups = (K + 15) >> 4, per = (kgroups + kparts - 1) / kparts;
    const bool has = it < ntiles * kparts;
    f.tn = (has ? it : 0) / kparts; f.kp = (has ? it : 0) - f.tn * kparts;
    f.g0 = f.kp * per;
    f.n = has ? min(kgroups, f.g0 + per) - f.g0 : 0;
    const float* w0 = Wm + (size_t)min(f.tn * 16 + i, N - 1) * ldw;
#pragma unroll
    for (int u = 0; u < MAXKG; ++u) f.b[u] = ldrow4c<true>(w0, min(f.g0 + u, kgroups - 1) * 16 + q * 4, K);
}
// raw[kp][m][tile columns] = A[16, K-part] . fragment   (A: LDS tile, zero padded)
template <int MAXKG>
__device__ __forceinline__ void wfrag_mma(const WFrag<MAXKG>& f, const float* A, int lda, float* raw, int ldr) {
    if (f.n == 0) return;
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const float* arow = A + i * lda + q * 4 + f.g0 * 16;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < MAXKG; u += 2) {
        if (u < f.n) {
            const float4 a = *reinterpret_cast<const float4*>(arow + u * 16);
            acc0 = mfma16(a.x, f.b[u].x, acc0); acc0 = mfma16(a.y, f.b[u].y, acc0);
            acc0 = mfma16(a.z, f.b[u].z, acc0); acc0 = mfma16(a.w, f.b[u].w, acc0);
        }
        if (u + 1 < MAXKG && u + 1 < f.n) {
            const float4 a = *reinterpret_cast<const float4*>(arow + (u + 1) * 16);
            acc1 = mfma16(a.x, f.b[u + 1].x, acc1); acc1 = mfma16(a.y, f.b[u + 1].y, acc1);
            acc1 = mfma16(a.z, f.b[u + 1].z, acc1); acc1 = mfma16(a.w, f.b[u + 1].w, acc1);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) raw[((f.kp * MMG_TM) + q * 4 + r) * ldr + f.tn * 16 + i] = acc0[r] + acc1[r];
}
// does one pass of the waves (items `wave` and optionally `wave + nw`) cover the product, with <= MAXKG k-groups per item?
__device__ __forceinline__ bool wfrag_fits(int N, int K, int nw, int maxkg, int items_per_wave) {
    const int ntiles = (N + 15) >> 4, kparts = tile_kparts(ntiles, nw);
    const int kgroups = (K + 15) >> 4, per = (kgroups + kparts - 1) / kparts;
    return ntiles * kparts <= items_per_wave * nw && per <= maxkg;
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
// Many-class y head of one tile over the class slice [d_lo, d_lo + Dl)   (model.py:432-433):
//   y[m][d] = b_y2 + sum_r w_y2[r] relu(A[m][r] + Cd[d][r]) = cy[d] + sum_r w_y2[r] max(A[m][r], -Cd[d][r])
// 4 samples x 4 classes per thread in registers (a 4 x 4 x 4 block per step of the r loop: 48 VALU operations per five
// 16-byte operand reads), lanes along the class groups: the NEGATED transposed class table CdT[r][d] is read once per
// tile with coalesced float4 loads (next r-chunk in flight), the A rows come from LDS.  Logits go to s_y[m][d - d_lo] and,
// per row flag, to tape.y (live rows) and tape.outp (the row's output step).  d_lo and Dl are multiples of 4 (or Dl ends at D).
// ---------------------------------------------------------------------------------------------
template <int NT>
__device__ __forceinline__ void class_logits(const Dims& dm, const Tape& tp, const float* s_A, int ldR, const float* s_w2, float* s_y, int ldY,
                                             int d_lo, int Dl, size_t rowb, int b0, const float* live, const float* take, int tid, float b2v) {
    // (live: rows whose logits go to tape.y -- the caller passes the take flags here when only the output step's are kept)
    const int D = dm.D, R = dm.R;
    if (Dl * MMG_TM <= 4 * NT) {
        // a narrow slice (class helpers of a 16-tile batch hold ~70 classes): the 4 x 4 blocks would occupy a fraction of the
        // workgroup, each walking R in 16 dependent round trips.  One (class, sample) pair per thread pass instead, the
        // class's whole row Cd[d, 0..63] in flight at once (relu form, as the few-class path of the tile kernel).
        for (int idx = tid; idx < Dl * MMG_TM; idx += NT) {
            const int l = idx >> 4, m = idx & 15, d = d_lo + l;
            const float* crow = tp.Cd + (size_t)d * R;
            const float* arow = s_A + m * ldR;
            float a0 = 0.f, a1 = 0.f;
            for (int r0 = 0; r0 < R; r0 += 64) {
                float4 cq[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) cq[u] = ldrow4c<true>(crow, r0 + 4 * u, R);
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int r = r0 + 4 * u;
                    if (r < R) {                                                            // (w2 is zero beyond R)
                        const float4 av = *reinterpret_cast<const float4*>(arow + r);
                        const float4 wq = *reinterpret_cast<const float4*>(s_w2 + r);
                        a0 = fmaf(wq.x, fmax_nn(av.x + cq[u].x, 0.f), a0); a1 = fmaf(wq.y, fmax_nn(av.y + cq[u].y, 0.f), a1);
                        a0 = fmaf(wq.z, fmax_nn(av.z + cq[u].z, 0.f), a0); a1 = fmaf(wq.w, fmax_nn(av.w + cq[u].w, 0.f), a1);
                    }
                }
            }
            const float yv = (a0 + a1) + b2v;
            s_y[m * ldY + l] = yv;
            if (live[m] != 0.f) tp.y[(rowb + b0 + m) * D + d] = yv;
            if (take[m] != 0.f) st_wt(&tp.outp[(size_t)(b0 + m) * D + d], yv);
        }
        for (int idx = tid; idx < MMG_TM * (ldY - Dl); idx += NT) s_y[(idx / (ldY - Dl)) * ldY + Dl + idx % (ldY - Dl)] = 0.f;
        return;
    }
    const bool dvec = ((D & 3) == 0) && ((d_lo & 3) == 0);
    const int mg = tid & 3, DG = (Dl + 3) >> 2;
    for (int dg0 = 0; dg0 < DG; dg0 += NT / 4) {
        const int dg = dg0 + (tid >> 2), l0 = min(dg, DG - 1) * 4, d0 = d_lo + l0;
        auto ldc = [&](int r) -> float4 {
            const float* row = tp.CdT + (size_t)min(r, R - 1) * D;
            if (dvec) return *reinterpret_cast<const float4*>(row + d0);
            float4 v; v.x = row[min(d0, D - 1)]; v.y = row[min(d0 + 1, D - 1)]; v.z = row[min(d0 + 2, D - 1)]; v.w = row[min(d0 + 3, D - 1)];
            return v;
        };
        float acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) { acc[a][0] = acc[a][1] = acc[a][2] = acc[a][3] = 0.f; }
        float4 cv[4], cn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cv[u] = ldc(u);
#pragma unroll 1
        for (int r0 = 0; r0 < R; r0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) cn[u] = ldc(r0 + 4 + u);
            const float4 wq = *reinterpret_cast<const float4*>(s_w2 + r0);            // zero beyond R
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float4 av = *reinterpret_cast<const float4*>(s_A + (4 * mg + a) * ldR + r0);
                acc[a][0] = fmaf(wq.x, fmax_nn(av.x, cv[0].x), acc[a][0]); acc[a][1] = fmaf(wq.x, fmax_nn(av.x, cv[0].y), acc[a][1]);
                acc[a][2] = fmaf(wq.x, fmax_nn(av.x, cv[0].z), acc[a][2]); acc[a][3] = fmaf(wq.x, fmax_nn(av.x, cv[0].w), acc[a][3]);
                acc[a][0] = fmaf(wq.y, fmax_nn(av.y, cv[1].x), acc[a][0]); acc[a][1] = fmaf(wq.y, fmax_nn(av.y, cv[1].y), acc[a][1]);
                acc[a][2] = fmaf(wq.y, fmax_nn(av.y, cv[1].z), acc[a][2]); acc[a][3] = fmaf(wq.y, fmax_nn(av.y, cv[1].w), acc[a][3]);
                acc[a][0] = fmaf(wq.z, fmax_nn(av.z, cv[2].x), acc[a][0]); acc[a][1] = fmaf(wq.z, fmax_nn(av.z, cv[2].y), acc[a][1]);
                acc[a][2] = fmaf(wq.z, fmax_nn(av.z, cv[2].z), acc[a][2]); acc[a][3] = fmaf(wq.z, fmax_nn(av.z, cv[2].w), acc[a][3]);
                acc[a][0] = fmaf(wq.w, fmax_nn(av.w, cv[3].x), acc[a][0]); acc[a][1] = fmaf(wq.w, fmax_nn(av.w, cv[3].y), acc[a][1]);
                acc[a][2] = fmaf(wq.w, fmax_nn(av.w, cv[3].z), acc[a][2]); acc[a][3] = fmaf(wq.w, fmax_nn(av.w, cv[3].w), acc[a][3]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) cv[u] = cn[u];
        }
        if (dg < DG) {
            const float4 cy4 = dvec ? *reinterpret_cast<const float4*>(tp.cy + d0)
                                    : make_float4(tp.cy[min(d0, D - 1)], tp.cy[min(d0 + 1, D - 1)], tp.cy[min(d0 + 2, D - 1)], tp.cy[min(d0 + 3, D - 1)]);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int m = 4 * mg + a;
                const float y4[4] = {acc[a][0] + cy4.x, acc[a][1] + cy4.y, acc[a][2] + cy4.z, acc[a][3] + cy4.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (l0 + c < Dl) {
                        s_y[m * ldY + l0 + c] = y4[c];
                        if (live[m] != 0.f) tp.y[(rowb + b0 + m) * D + d0 + c] = y4[c];
                        if (take[m] != 0.f) st_wt(&tp.outp[(size_t)(b0 + m) * D + d0 + c], y4[c]);    // model.py:1261-1264 (write-through: a helper's slice is read back by the tile's owner)
                    }
                }
            }
        }
    }
    for (int idx = tid; idx < MMG_TM * (ldY - Dl); idx += NT) s_y[(idx / (ldY - Dl)) * ldY + Dl + idx % (ldY - Dl)] = 0.f;   // K padding of the mixture product
}

// softmax NUMERATORS of the slice in place, e = exp(y - slice max); stat[m] = slice max, stat[16 + m] = sum of e.
// (the mixture product runs on the numerators; normalisation -- and the combination of slices -- follows it)
template <int NT>
__device__ __forceinline__ void class_softmax_num(float* s_y, int ldY, int Dl, float* stat, int wave, int lane) {
    constexpr int nw = NT / 64;
    for (int m = wave; m < MMG_TM; m += nw) {
        float* yr = s_y + m * ldY;
        float mx = -3.0e38f;
        for (int d0 = lane; d0 < Dl; d0 += 64 * 8) {                          // 8 LDS reads in flight per lane
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = yr[min(d0 + 64 * u, Dl - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) mx = fmaxf(mx, v[u]);
        }
        mx = dpp_wave_max(mx);
        float se = 0.f;
        for (int d0 = lane; d0 < Dl; d0 += 64 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = yr[min(d0 + 64 * u, Dl - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float e = __expf(v[u] - mx);
                if (d0 + 64 * u < Dl) { yr[d0 + 64 * u] = e; se += e; }
            }
        }
        se = dpp_wave_sum(se);
        if (lane == 0) { stat[m] = mx; stat[MMG_TM + m] = se; }
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
// per: classes this workgroup holds logits for (D, or its slice of them when class helpers share the tile: k_conv_split)
__host__ __device__ inline TileLds tile_lds(const Dims& d, int nw, bool with_sender, int per = 0) {
    TileLds L;
    L.ldH = ld16(d.H); L.ldW = ld16(d.W); L.ldR = ld16(d.R); L.ld3R = ld16(3 * d.R); L.ldD = ld16(per > 0 ? per : d.D); L.ldV = ld16(d.V);
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
    L.misc = take(448);
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
#define TL_STAT 128                                   // [128,160): slice max | slice sum of the softmax numerators (many-class path)
#define TL_FAC 160                                    // [160,416): combination weights of up to 16 class slices
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
#define MMG_RSTAMP(cond, slot) do { if ((cond) && threadIdx.x == 0) { if ((slot) < 256) tp.dbg[(slot)] = (long long)wall_clock64(); else tp.dbg2[(slot) - 256] = (long long)wall_clock64(); } } while (0)
#else
#define MMG_RSTAMP(cond, slot) do {} while (0)
#endif
#ifdef MMG_TIMING
#define MMG_TSTAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) tp.dbg[(slot)] = (long long)wall_clock64(); } while (0)
#else
#define MMG_TSTAMP(slot) do {} while (0)
#endif
// ---- counters between the roles of k_conv_persist: monotonic (zeroed by k_prep), one per 256-byte block.
// kind 0: g published by the receiver role of a tile (value = steps done), 1: sender-hidden slices a_t (NS1 per step),
// 2: message columns + GRU-input partials (NS2 per step), 3: the tile's conversation is over, 4 + t (t < 16): per-sample receiver
// roles that have delivered step t or stopped before it.
__device__ __forceinline__ uint32_t* pf_ctr(const Tape& tp, int kind, int tile) { return tp.pflags + ((size_t)kind * 64 + tile) * 64; }
// producer: everything the consumers read was written with agent-scope (write-through) stores
__device__ __forceinline__ void pf_signal(uint32_t* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// consumer: returns false when the tile's conversation ended instead (done counter set); bounded spin -> error word.
// ACQ = false: the caller reads the payload with ld_cc* (agent-scope loads) and needs no acquire fence.
template <bool ACQ = true>
__device__ __forceinline__ bool pf_wait(uint32_t* ctr, uint32_t target, uint32_t* done, uint32_t* sync_err) {
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        int ok = -1, spins = 0;
        while (ok < 0) {
            if (done && __hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ok = 0;
            else if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) ok = 1;
            else {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > MMG_SPIN_LIMIT) { __hip_atomic_store(sync_err + MMG_SYNC_ERR, 100u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = 0; }
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const bool r = s_ok != 0;
    __syncthreads();                                    // (s_ok may be rewritten by the next wait)
    return r;
}

template <int NT, bool PERSIST, bool SPLIT = false>
__device__ __forceinline__ void conv_tile_body(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int tile_idx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    MMG_TSTAMP(0);
#ifdef MMG_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0) tp.dbg[4] = (long long)clock64();
#endif
    constexpr int nw = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: work-item bookkeeping of the products lives in scalar registers)
    const int B = dm.B, H = dm.H, W = dm.W, R = dm.R, V = dm.V, D = dm.D, T = dm.T;
    const int b0 = tile_idx * MMG_TM, nb = min(MMG_TM, B - b0);
    constexpr bool persist = PERSIST;                   // receiver role of k_conv_persist: sender roles run beside it in this launch
    const bool do_sen = !PERSIST && (ar.phases & 1) != 0;
    const bool binary = dm.use_binary != 0, train = ar.train != 0;
    const bool may_stop = !ar.run_all && !dm.fixed && train;          // a finished tile stops computing
    const bool bigD = !PERSIST && D * MMG_TM > 8 * NT;             // many classes: register-tiled y head, slices, normalisation after the mixture product
    const int Dl = SPLIT ? min(ar.per, D) : D;          // classes of this workgroup's slice [0, Dl) (class helpers take the rest)
    const TileLds L = tile_lds(dm, nw, do_sen, SPLIT ? ar.per : 0);
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
        // per-column vectors -> LDS as ONE batched gather over the concatenation of the nine segments (one memory round trip)
        const float* sp[10]; int sl[10], so[10], ns = 0;
        auto seg = [&](int off, const float* src, int n) { sp[ns] = src; sl[ns] = n; so[ns] = off; ++ns; };
        if (do_sen) { seg(L.bc, P.p[S_CODE_B], H); seg(L.hw0, tp.hw0, H); seg(L.bb, P.p[S_BIN_B], W); seg(L.sc, P.p[S_CODE_BIAS], W); }
        seg(L.bih, P.p[R_BIH], 3 * R); seg(L.bhh, P.p[R_BHH], 3 * R); seg(L.bh, P.p[R_WH_B], R); seg(L.bw, P.p[R_W_B], W);
        seg(L.ws, P.p[R_S_W], R); seg(L.w2, P.p[R_Y2_W], R);
        int total = 0;
        for (int k = 0; k < ns; ++k) total += sl[k];
        struct PV { float v; int off; };
        batched_for<NT, 4>(total, [&](int idx) {
                const float* src = sp[0]; int off = so[0], base = 0, acc = 0;
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    if (k < ns) { if (idx >= acc) { src = sp[k]; off = so[k]; base = acc; } acc += sl[k]; }
                }
                return PV{src[idx - base], off + idx - base};
            }, [&](int, PV q) { smem[q.off] = (do_sen && q.off >= L.sc && q.off < L.sc + L.ldW) ? fsigmoid(q.v) : q.v; });
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

    // receiver role of the persistent launch: the weights of its small per-step products as register fragments (wfrag_*)
    WFrag<2> fA, fGw; WFrag<4> fWd, fHh0, fHh1;
    fA.n = fGw.n = fWd.n = fHh0.n = fHh1.n = 0;
    const bool res_heads = PERSIST && wfrag_fits(R, R, nw, 2, 1), res_wd = PERSIST && wfrag_fits(R, V, nw, 4, 1);
    const bool res_hh = PERSIST && wfrag_fits(3 * R, R, nw, 4, 2);
    if (res_heads) { wfrag_load<2>(fA, P.p[R_Y1_W], R + V, R, R, wave, nw); wfrag_load<2>(fGw, P.p[R_WH_W], R, R, R, wave, nw); }
    if (res_wd) wfrag_load<4>(fWd, P.p[R_WD_W], V, R, V, wave, nw);
    if (res_hh) { wfrag_load<4>(fHh0, P.p[R_WHH], R, 3 * R, R, wave, nw); wfrag_load<4>(fHh1, P.p[R_WHH], R, 3 * R, R, wave + nw, nw); }
    int t = t0;
    bool finished = false;
    int published = 0;                                                  // class-split: steps whose A tile went out to the helpers
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
                if (res_hh) { wfrag_mma<4>(fHh0, s_h, L.ldR, raw1, L.ld3R); wfrag_mma<4>(fHh1, s_h, L.ldR, raw1, L.ld3R); }
                else add(s_h, L.ldR, P.p[R_WHH], R, 3 * R, R, raw1, 0);
                break;
            case 1: if (do_sen) add(s_a, L.ldH, P.p[S_BIN_W], H, W, H, raw0, 0); break;                 // sender logits, model.py:218
            case 2: if (!persist) add(s_z, L.ldW, P.p[R_WIH], W, 3 * R, W, raw0, 0); break;              // GRU input side, model.py:340 (persist: partials of the sender roles)
            case 3:                                                                                     // A (App. A.2), w_h h
                if (res_heads) { wfrag_mma<2>(fA, s_h, L.ldR, raw0, L.ldR); wfrag_mma<2>(fGw, s_h, L.ldR, raw1, L.ldR); }
                else { add(s_h, L.ldR, P.p[R_Y1_W], R + V, R, R, raw0, 0); add(s_h, L.ldR, P.p[R_WH_W], R, R, R, raw1, 0); }
                break;
            case 5: add(s_y, L.ldD, ar.desc, V, V, bigD ? Dl : D, raw0, 1); break;                      // softmax . desc, model.py:442-449 (bigD: numerators of the slice)
            case 6:                                                                                     // w_d dbar, model.py:452
                if (res_wd) wfrag_mma<4>(fWd, s_dbar, L.ldV, raw0, L.ldR);
                else add(s_dbar, L.ldV, P.p[R_WD_W], V, R, V, raw0, 0);
                break;
            case 7: if (!persist) add(s_g, L.ldR, P.p[R_W_W], R, W, R, raw0, 0); break;                  // receiver message logits, model.py:454 (persist: in the sender roles)
            default: break;
            }
            bool paired = false;
            if (nj == 2 && !j0.nn && !j1.nn) {                           // two small NT products: one memory round trip for both
                const NtJob pa = {j0.A, j0.Wm, j0.raw, j0.lda, j0.ldw, j0.N, j0.K}, pb = {j1.A, j1.Wm, j1.raw, j1.lda, j1.ldw, j1.N, j1.K};
                if (nt_pair_fits(pa, pb, nw)) { tgemm_nt_pair(pa, pb, wave, nw); paired = true; }
            }
            for (int j = 0; j < (paired ? 0 : nj); ++j) {
                const float* jA = j ? j1.A : j0.A; const float* jW = j ? j1.Wm : j0.Wm; float* jr = j ? j1.raw : j0.raw;
                const int jlda = j ? j1.lda : j0.lda, jldw = j ? j1.ldw : j0.ldw, jN = j ? j1.N : j0.N, jK = j ? j1.K : j0.K;
                if ((j ? j1.nn : j0.nn)) tgemm_nn_raw(jA, jlda, jW, jldw, jN, jK, jr, wave, nw);
                else tgemm_nt_raw(jA, jlda, jW, jldw, jN, jK, jr, wave, nw);
            }
            MMG_TSTAMP(16 + 16 * (t - t0) + ph);
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
                MMG_RSTAMP(tile_idx == 0 && t == 3, 170);
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
                    MMG_RSTAMP(tile_idx == 0 && t == 3, 171);
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
                MMG_RSTAMP(tile_idx == 0 && t == 3, 172);
                const float* s_bhh = smem + L.bhh;
                const int kp = tile_kparts((3 * R + 15) >> 4, nw);
                for (int idx = tid; idx < MMG_TM * 3 * R; idx += NT) {
                    const int m = idx / (3 * R), n = idx - m * 3 * R;
                    s_gh[m * L.ld3R + n] = raw_sum(raw1, L.ld3R, kp, m, n) + s_bhh[n];
                }
                MMG_RSTAMP(tile_idx == 0 && t == 3, 173);
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
                    if (persist) {
                        // the message of this step and the GRU's input-side product arrive from the sender roles of the tile:
                        // gi = sum over the roles' column slices of z_slice W_ih[:, slice]^T, added in role order
                        MMG_RSTAMP(tile_idx == 0 && t == 3, 220);
                        pf_wait(pf_ctr(tp, 2, tile_idx), (uint32_t)(ar.ns2 * (t + 1)), nullptr, tp.sync);
                        MMG_RSTAMP(tile_idx == 0 && t == 3, 221);
                        const int kp = tile_kparts((3 * R + 15) >> 4, nw);
                        for (int idx = tid; idx < MMG_TM * 3 * R; idx += NT) {
                            const int m = idx / (3 * R), n = idx - m * 3 * R;
                            const size_t o = (size_t)min(b0 + m, B - 1) * 3 * R + n;
                            float p8[8], acc = 0.f;
                            for (int k0 = 0; k0 < ar.ns2; k0 += 8) {
#pragma unroll
                                for (int u = 0; u < 8; ++u) p8[u] = tp.gip[(size_t)min(k0 + u, ar.ns2 - 1) * B * 3 * R + o];
#pragma unroll
                                for (int u = 0; u < 8; ++u) acc += (k0 + u < ar.ns2) ? p8[u] : 0.f;
                            }
                            raw0[m * L.ld3R + n] = acc;
                            for (int q = 1; q < kp; ++q) raw0[(q * MMG_TM + m) * L.ld3R + n] = 0.f;
                        }
                    }
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
                    if (persist && m < nb) st_wt(&tp.mstate[b0 + m], m_next);        // the sender roles store live rows only, too
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
                                    a0 = fmaf(wq.x, fmax_nn(av.x + cq[u].x, 0.f), a0); a1 = fmaf(wq.y, fmax_nn(av.y + cq[u].y, 0.f), a1);
                                    a0 = fmaf(wq.z, fmax_nn(av.z + cq[u].z, 0.f), a0); a1 = fmaf(wq.w, fmax_nn(av.w + cq[u].w, 0.f), a1);
                                }
                            }
                        }
                        const float yv = (a0 + a1) + b2;
                        s_y[m * L.ldD + d] = yv;
                        if (misc[(ar.y_last_only ? TL_TAKE : TL_LIVE) + m] != 0.f) tp.y[(rowb + b0 + m) * D + d] = yv;
                    }
                } else {
                    if (SPLIT) {
                        // A tile and row flags for the class helpers of this tile, then the owner's own slice
                        float* pub = tp.Apub + (size_t)tile_idx * (MMG_TM * R + 32);
                        for (int idx = tid; idx < MMG_TM * R; idx += NT) st_wt(pub + idx, s_A[(idx / R) * L.ldR + idx % R]);
                        if (tid < MMG_TM) { st_wt(pub + MMG_TM * R + tid, misc[TL_LIVE + tid]); st_wt(pub + MMG_TM * R + 16 + tid, misc[TL_TAKE + tid]); }
                        pf_signal(pf_ctr(tp, 0, tile_idx));
                        ++published;
                    }
                    class_logits<NT>(dm, tp, s_A, L.ldR, s_w2, s_y, L.ldD, 0, Dl, rowb, b0, misc + (ar.y_last_only ? TL_TAKE : TL_LIVE), misc + TL_TAKE, tid, b2);
                }
                if (!bigD) for (int idx = tid; idx < MMG_TM * (L.ldD - D); idx += NT) s_y[(idx / (L.ldD - D)) * L.ldD + D + idx % (L.ldD - D)] = 0.f;   // K padding of the mixture product
                __syncthreads();
                // output step of a sample: its logits go to tape.outp (model.py:1261-1264); the tile may be done
                alive = false;
                for (int m = 0; m < nb; ++m) alive = alive || (misc[TL_MT + m] != 0.f);
                if (!bigD) for (int m = 0; m < nb; ++m) {
                    if (misc[TL_TAKE + m] != 0.f)
                        for (int d = tid; d < D; d += NT) tp.outp[(size_t)(b0 + m) * D + d] = s_y[m * L.ldD + d];
                }
                if (tid == 0 && t + 1 < T && alive) atomicAdd(&tp.alive[t + 1], 1);
                if (may_stop && !alive) break;
                __syncthreads();                                        // the selected rows are copied before the softmax overwrites them
                if (bigD) class_softmax_num<NT>(s_y, L.ldD, Dl, misc + TL_STAT, wave, lane);
                else {
                // softmax(y) (detached, model.py:441): wave per sample row, in place
                for (int m = wave; m < MMG_TM; m += nw) {
                    float* yr = s_y + m * L.ldD;
                    float mx = -3.0e38f;
                    for (int d = lane; d < D; d += 64) mx = fmaxf(mx, yr[d]);
                    mx = dpp_wave_max(mx);
                    float se = 0.f;
                    for (int d = lane; d < D; d += 64) { const float e = __expf(yr[d] - mx); yr[d] = e; se += e; }
                    se = dpp_wave_sum(se);
                    const float inv = __builtin_amdgcn_rcpf(se);
                    for (int d = lane; d < D; d += 64) yr[d] *= inv;
                }
                }
            } else if (ph == 5) {                                        // description mixture
                const int kp = tile_kparts((V + 63) >> 6, nw);
                float* fac = misc + TL_FAC;                                  // [slot][16]: weight of each slice's partial, already divided by the total
                if (bigD) {
                    if (SPLIT) pf_wait(pf_ctr(tp, 2, tile_idx), (uint32_t)(ar.nhelp * (t + 1)), nullptr, tp.sync);
                    if (tid < MMG_TM) {
                        // slices are combined like a streaming softmax: weights exp(max_h - M) / sum_h s_h exp(max_h - M)
                        const int m = tid;
                        const float* cp = tp.cpart + ((size_t)tile_idx * (SPLIT ? ar.nhelp : 1) * MMG_TM + m) * (V + 2);
                        float mh[16], sh[16];
                        float M = misc[TL_STAT + m];
#pragma unroll
                        for (int q = 0; q < 15; ++q) {
                            const bool on = SPLIT && q < ar.nhelp;
                            mh[q] = on ? cp[(size_t)q * MMG_TM * (V + 2) + V] : -3.0e38f;
                            sh[q] = on ? cp[(size_t)q * MMG_TM * (V + 2) + V + 1] : 0.f;
                        }
#pragma unroll
                        for (int q = 0; q < 15; ++q) M = fmaxf(M, mh[q]);
                        float S = misc[TL_STAT + MMG_TM + m] * __expf(misc[TL_STAT + m] - M);
#pragma unroll
                        for (int q = 0; q < 15; ++q) S += sh[q] * __expf(mh[q] - M);
                        const float inv = 1.0f / S;
                        fac[m] = __expf(misc[TL_STAT + m] - M) * inv;
#pragma unroll
                        for (int q = 0; q < 15; ++q) fac[(q + 1) * MMG_TM + m] = (SPLIT && q < ar.nhelp) ? __expf(mh[q] - M) * inv : 0.f;
                    }
                    __syncthreads();
                }
                for (int idx = tid; idx < MMG_TM * V; idx += NT) {
                    const int m = idx / V, v = idx - m * V;
                    float dv = raw_sum(raw0, L.ldV, kp, m, v);
                    if (bigD) {
                        dv *= fac[m];
                        if (SPLIT) {
                            const float* cp = tp.cpart + ((size_t)tile_idx * ar.nhelp * MMG_TM + m) * (V + 2) + v;
                            float pv[15];
#pragma unroll
                            for (int q = 0; q < 15; ++q) pv[q] = cp[(size_t)min(q, ar.nhelp - 1) * MMG_TM * (V + 2)];
#pragma unroll
                            for (int q = 0; q < 15; ++q) dv = fmaf(pv[q], fac[(q + 1) * MMG_TM + m], dv);     // (weights beyond nhelp are zero)
                        }
                    }
                    s_dbar[m * L.ldV + v] = dv;
                    if (misc[TL_LIVE2 + m] != 0.f) tp.dbar[(rowb + b0 + m) * V + v] = dv;
                }
            } else if (ph == 6) {                                        // h_w = tanh(w_h h + b_h + w_d dbar)   (model.py:452)
                const int kp = tile_kparts((R + 15) >> 4, nw);
                for (int idx = tid; idx < MMG_TM * R; idx += NT) {
                    const int m = idx / R, n = idx - m * R;
                    const float gv = ftanh(s_gw[m * L.ldR + n] + raw_sum(raw0, L.ldR, kp, m, n));
                    s_g[m * L.ldR + n] = gv;
                    if (persist) { if (m < nb) st_wt(&tp.g[(rowb + b0 + m) * R + n], gv); }     // read by the sender roles (all rows)
                    else if (misc[TL_LIVE2 + m] != 0.f) tp.g[(rowb + b0 + m) * R + n] = gv;
                }
                MMG_RSTAMP(persist && tile_idx == 0 && t == 2, 222);
                if (persist) pf_signal(pf_ctr(tp, 0, tile_idx));            // g_t (and the stop masks) are out: the sender roles go on
            } else if (!persist) {                                       // receiver message (model.py:454-475)
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
    if (SPLIT && published > 0)                                         // the helpers' slices of the selected logits are in tape.outp
        pf_wait(pf_ctr(tp, 2, tile_idx), (uint32_t)(ar.nhelp * published), nullptr, tp.sync);
    if ((persist || SPLIT) && threadIdx.x == 0) __hip_atomic_store(pf_ctr(tp, 3, tile_idx), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // conversation over
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

template <int NT>
__global__ __launch_bounds__(NT) void k_conv_tile(Dims dm, Params P, Tape tp, ConvArgs ar) {
    conv_tile_body<NT, false>(dm, P, tp, ar, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// k_conv_split: many classes, fewer sample tiles than CUs (BASELINE config 5: 1000 classes; 16 tiles per GPU in the 8-GPU
// sharding, 128 on one GPU).  The class-dependent part of a step -- logits over D classes (VALU) and the description mixture
// (softmax . desc) -- is the bulk of the step and is independent per class: the workgroups the tiles leave idle become
// CLASS HELPERS.  Per step the tile's owner publishes its A tile (4 KB) and row flags, owner and helpers each take a slice of
// the classes (logits -> tape, slice max / sum of the softmax numerators, unnormalised mixture partial), the helpers publish
// 16 x (V + 2) floats, and the owner combines the slices like a streaming softmax.  Same counters as k_conv_persist.
// ---------------------------------------------------------------------------------------------
struct HelperLds { int A, w2, y, raw, flags, total; };
__host__ __device__ inline HelperLds helper_lds(const Dims& d, int nw, int per) {
    HelperLds L; int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 3) & ~3; return at; };
    L.A = take(MMG_TM * ld16(d.R)); L.w2 = take(ld16(d.R)); L.y = take(MMG_TM * ld16(per)); L.raw = take(tile_raw_floats_nn(d.V, nw)); L.flags = take(64);
    L.total = o;
    return L;
}

template <int NT>
__device__ __forceinline__ void class_helper_role(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int tile, const int hidx) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int nw = NT / 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int B = dm.B, R = dm.R, V = dm.V, D = dm.D, T = dm.T;
    const int b0 = tile * MMG_TM;
    const int d_lo = min(D, (hidx + 1) * ar.per), Dl = min(ar.per, D - d_lo);
    const HelperLds L = helper_lds(dm, nw, ar.per);
    const int ldR = ld16(R), ldY = ld16(ar.per), ldV = ld16(V);
    float* s_A = smem + L.A; float* s_w2 = smem + L.w2; float* s_y = smem + L.y; float* raw = smem + L.raw; float* flags = smem + L.flags;
    for (int i = tid; i < L.total; i += NT) smem[i] = 0.f;
    __syncthreads();
    for (int r = tid; r < R; r += NT) s_w2[r] = P.p[R_Y2_W][r];
    const float b2h = P.p[R_Y2_B][0];
    uint32_t* cA = pf_ctr(tp, 0, tile); uint32_t* cY = pf_ctr(tp, 2, tile); uint32_t* done = pf_ctr(tp, 3, tile);
    const float* pub = tp.Apub + (size_t)tile * (MMG_TM * R + 32);
    float* out = tp.cpart + ((size_t)tile * ar.nhelp + hidx) * MMG_TM * (V + 2);
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t rowb = (size_t)t * B;
        if (!pf_wait(cA, (uint32_t)(t + 1), done, tp.sync)) return;
        batched_for<NT, 2>(MMG_TM * R + 32, [&](int idx) { return pub[idx]; },
                           [&](int idx, float v) { if (idx < MMG_TM * R) s_A[(idx / R) * ldR + idx % R] = v; else flags[idx - MMG_TM * R] = v; });
        __syncthreads();
        if (Dl > 0) {
            class_logits<NT>(dm, tp, s_A, ldR, s_w2, s_y, ldY, d_lo, Dl, rowb, b0, ar.y_last_only ? flags + 16 : flags, flags + 16, tid, b2h);
            __syncthreads();
            class_softmax_num<NT>(s_y, ldY, Dl, flags + 32, wave, lane);
            __syncthreads();
            tgemm_nn_raw(s_y, ldY, ar.desc + (size_t)d_lo * V, V, V, Dl, raw, wave, nw);
            __syncthreads();
        }
        {
            const int kp = tile_kparts((V + 63) >> 6, nw);
            for (int idx = tid; idx < MMG_TM * (V + 2); idx += NT) {
                const int m = idx / (V + 2), v = idx - m * (V + 2);
                float val;
                if (v < V) val = Dl > 0 ? raw_sum(raw, ldV, kp, m, v) : 0.f;
                else if (v == V) val = Dl > 0 ? flags[32 + m] : -3.0e38f;
                else val = Dl > 0 ? flags[48 + m] : 0.f;
                st_wt(out + idx, val);
            }
        }
        pf_signal(cY);
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_conv_split(Dims dm, Params P, Tape tp, ConvArgs ar, int tiles) {
    const int blk = blockIdx.x;
    if (blk < tiles) { conv_tile_body<NT, false, true>(dm, P, tp, ar, blk); return; }
    const int r = blk - tiles;
    class_helper_role<NT>(dm, P, tp, ar, r / ar.nhelp, r % ar.nhelp);
}

// ---------------------------------------------------------------------------------------------
// k_conv_persist: the whole conversation of the "large sender, few samples" regime (BASELINE config 4: H*W = 262 144, four
// sample tiles) as ONE launch of co-resident workgroup roles that hand their results on through memory + counters
// (pf_signal / pf_wait), instead of three launches per exchange step.  Per tile of 16 samples:
//   receiver role (1)   conv_tile_body with persist set: GRU, heads, class logits, description mixture, g_t     -> counter 0
//   S1 roles (H / 64)   every role recomputes the tile's message w_{t-1} = sample(sigmoid(g W_w^T + b)) (64 KB of weights,
//                       bit-identical in all of them: Philox / injected uniforms are indexed by element) and then ITS
//                       64-unit slice of a_t = tanh(h_x + w W_c^T + b)                                            -> counter 1
//   S2 roles (W / 32)   32 message bits z_t = sample(sigmoid(a W_b^T + b)) and the partial GRU input product
//                       z_slice W_ih[:, slice]^T, which the receiver role adds up in role order                   -> counter 2
// So the 2 MB of sender weights are streamed by 24 CUs per tile instead of one, W_ih and W_w leave the receiver role (its
// per-step weights drop from 388 KB to 125 KB), and nothing returns to the host between steps.
// All roles must be resident together (host: tiles * (1 + NS1 + NS2) <= 240 workgroups of 512 threads); every wait is
// bounded (error word).  Tape rows follow the live-row contract of k_conv_tile.
// ---------------------------------------------------------------------------------------------
struct SRoleLds { int g, w, pw, a, zs, raw, vec, live, total; };
__host__ __device__ inline SRoleLds srole_lds(const Dims& d, int nw) {
    SRoleLds L; int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 3) & ~3; return at; };
    L.g = take(MMG_TM * ld16(d.R)); L.w = take(MMG_TM * ld16(d.W)); L.pw = take(MMG_TM * ld16(d.W));
    L.a = take(MMG_TM * ld16(d.H)); L.zs = take(MMG_TM * ld16(32));
    int r = tile_raw_floats(d.W, nw);
    auto mx = [](int a, int b) { return a > b ? a : b; };
    r = mx(r, tile_raw_floats(64, nw)); r = mx(r, tile_raw_floats(32, nw)); r = mx(r, tile_raw_floats(3 * d.R, nw));
    L.raw = take(r); L.vec = take(ld16(d.W) + 2 * 64 + 2 * 32 + 16); L.live = take(16);
    L.total = o;
    return L;
}

template <int NT>
__device__ __forceinline__ void s1_role(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int tile, const int j) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int nw = NT / 64;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int B = dm.B, H = dm.H, W = dm.W, R = dm.R, T = dm.T;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0), n0 = j * 64;
    const bool binary = dm.use_binary != 0, train = ar.train != 0;
    const bool may_stop = !ar.run_all && !dm.fixed && train;
    const SRoleLds L = srole_lds(dm, nw);
    const int ldR = ld16(R), ldW = ld16(W);
    float* s_g = smem + L.g; float* s_w = smem + L.w; float* s_pw = smem + L.pw; float* raw = smem + L.raw;
    float* s_bw = smem + L.vec; float* s_bc = s_bw + ldW; float* s_hw0 = s_bc + 64; float* s_live = smem + L.live;
    for (int i = tid; i < L.total; i += NT) smem[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < W; i += NT) s_bw[i] = P.p[R_W_B][i];
    if (tid < 64) { s_bc[tid] = P.p[S_CODE_B][n0 + tid]; s_hw0[tid] = tp.hw0[n0 + tid]; }
    if (tid < MMG_TM) s_live[tid] = tid < nb ? 1.f : 0.f;
    float hxr[2];                                                       // h_x[b, n0 .. n0+63] of the tile: 1024 values on 512 threads
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int idx = tid + u * NT, m = idx >> 6, n = idx & 63; hxr[u] = tp.hx[(size_t)min(b0 + m, B - 1) * H + n0 + n]; }
    const uint32_t mb_counter = tp.counter[0];
    uint32_t* cG = pf_ctr(tp, 0, tile); uint32_t* cA = pf_ctr(tp, 1, tile); uint32_t* done = pf_ctr(tp, 3, tile);
    // both weight matrices of this role as register fragments, loaded once (W_w: two items of 4 k-groups per wave, the W_c
    // slice: one item of 8): the step loop then reads no weight from memory.  Larger shapes keep streaming them.
    const bool msg_in = ar.rsample == 2;                                // the receiver roles publish the message, not g
    const bool res_w = !msg_in && wfrag_fits(W, R, nw, 4, 2), res_c = wfrag_fits(64, W, nw, 8, 1);
    WFrag<4> fw0, fw1; WFrag<8> fc;
    fw0.n = fw1.n = fc.n = 0;
    if (res_w) { wfrag_load<4>(fw0, P.p[R_W_W], R, W, R, wave, nw); wfrag_load<4>(fw1, P.p[R_W_W], R, W, R, wave + nw, nw); }
    if (res_c) wfrag_load<8>(fc, P.p[S_CODE_W] + (size_t)n0 * W, W, 64, W, wave, nw);
    __syncthreads();
    constexpr int UW = 8;                                               // message elements per thread (host: 16 * W <= UW * NT)
    for (int t = 0; t <= T; ++t) {
        const size_t rowb = (size_t)t * B, rowp = (size_t)(t > 0 ? t - 1 : 0) * B;
        int tid = threadIdx.x;                                          // (opaque per step: keeps the index arithmetic of the
        asm volatile("" : "+v"(tid));                                   //  epilogues from being hoisted out of the step loop)
        if (t >= 1) {
            // ---- the receiver's message of step t-1 (model.py:454-475), recomputed identically by every S1 role of the tile.
            // The uniforms of its Bernoulli draws do not depend on anything this step computes: drawn BEFORE the wait.
            float uw[UW];
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                const int idx = min(tid + u * NT, MMG_TM * W - 1), m = idx / W, n = idx - m * W, b = min(b0 + m, B - 1);
                uw[u] = (!(binary && train) || msg_in) ? 0.f
                        : ar.u_w ? ar.u_w[(rowp + b) * W + n]
                                 : philox_uniform(ar.seed, (uint32_t)(((t - 1) * dm.Bg + dm.boff + b) * W + n), mb_counter, 2u);
            }
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 200);
            // (per-sample receiver roles: one count per sample on the counter OF THE STEP, kind 4 + step -- a sample that has stopped
            //  counts once on every later step's counter.  Rounds 3-4 kept ONE running counter and let a stopped sample add all its
            //  remaining steps at once: with enough of them gone the sum passed nb * t before the live samples had delivered.)
            if (!pf_wait(ar.rsample ? pf_ctr(tp, 4 + (t - 1), tile) : cG, (uint32_t)(ar.rsample ? nb : t), done, tp.sync)) return;
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 201); MMG_RSTAMP(tile == 0 && j == 0 && t == 4, 206);
            if (msg_in) {
                // (per-sample receiver roles that form the message themselves: it arrives ready)
                batched_for<NT, 8>(MMG_TM * W, [&](int idx) { const int m = idx / W, n = idx - m * W; return tp.w[(rowp + min(b0 + m, B - 1)) * W + n]; },
                                   [&](int idx, float v) { const int m = idx / W, n = idx - m * W; s_w[m * ldW + n] = v; });
            } else
            batched_for<NT, 2>(MMG_TM * R, [&](int idx) { const int m = idx / R, r = idx - m * R; return tp.g[(rowp + min(b0 + m, B - 1)) * R + r]; },
                               [&](int idx, float v) { const int m = idx / R, r = idx - m * R; s_g[m * ldR + r] = v; });
            if (tid < MMG_TM) s_live[tid] = (tid < nb && (!may_stop || tp.mstate[min(b0 + tid, B - 1)] != 0.f)) ? 1.f : 0.f;
            __syncthreads();
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 202);
            if (!msg_in) {
            if (res_w) { wfrag_mma<4>(fw0, s_g, ldR, raw, ldW); wfrag_mma<4>(fw1, s_g, ldR, raw, ldW); }
            else tgemm_nt_raw(s_g, ldR, P.p[R_W_W], R, W, R, raw, wave, nw);
            __syncthreads();
            }
            if (!msg_in) {
                const int kp = tile_kparts((W + 15) >> 4, nw);
#pragma unroll
                for (int u = 0; u < UW; ++u) {
                    const int idx = tid + u * NT;
                    if (idx < MMG_TM * W) {
                        const int m = idx / W, n = idx - m * W;
                        const float lw = raw_sum(raw, ldW, kp, m, n) + s_bw[n];
                        float wv = lw, pp = 0.f;
                        if (binary) {
                            pp = fsigmoid(lw);
                            wv = train ? ((uw[u] < pp) ? 1.f : 0.f) : rintf(pp);            // model.py:460 / 462
                        }
                        s_w[m * ldW + n] = wv; s_pw[m * ldW + n] = pp;
                    }
                }
            }
            if (!msg_in) __syncthreads();
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 203);
            if (t < T) {
                // ---- this role's 64 units of the sender hidden state (model.py:195-216)
                if (res_c) wfrag_mma<8>(fc, s_w, ldW, raw, ld16(64));
                else tgemm_nt_raw(s_w, ldW, P.p[S_CODE_W] + (size_t)n0 * W, W, 64, W, raw, wave, nw);
                __syncthreads();
            }
        }
        if (t < T) {
            const int kp = tile_kparts(4, nw), ldr = ld16(64);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = tid + u * NT, m = idx >> 6, n = idx & 63;
                const float hw = (t == 0) ? s_hw0[n] : raw_sum(raw, ldr, kp, m, n) + s_bc[n];
                if (m < nb) st_wt(&tp.a[(rowb + b0 + m) * H + n0 + n], ftanh(hxr[u] + hw));      // every valid row: the S2 roles multiply whole tiles
            }
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 204);
            pf_signal(cA);
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 205);
        }
        // ---- off the critical path (the S2 roles are running): role 0 of the tile records the message and its statistics
        if (j == 0 && !msg_in) {
            if (t >= 1) {
                for (int idx = tid; idx < nb * W; idx += NT) {
                    const int m = idx / W, n = idx - m * W;
                    if (s_live[m] == 0.f) continue;
                    tp.w[(rowp + b0 + m) * W + n] = s_w[m * ldW + n];
                    if (binary) tp.pw[(rowp + b0 + m) * W + n] = s_pw[m * ldW + n];
                }
                if (binary && tid < 256) {
                    const int m = tid >> 4, l16 = tid & 15;
                    float lpv = 0.f, nev = 0.f;
                    for (int q = l16; q < W; q += 16) {
                        const float p = s_pw[m * ldW + q], wv = s_w[m * ldW + q];
                        const float l1 = flog(p + MMG_EPS), l0 = flog(1.f - p + MMG_EPS);
                        lpv += wv * l1 + (1.f - wv) * l0; nev += p * l1 + (1.f - p) * l0;
                    }
                    lpv = dpp_group_sum<16>(lpv); nev = dpp_group_sum<16>(nev);
                    if (l16 == 0 && s_live[m] != 0.f) { tp.lp_w[rowp + b0 + m] = lpv; tp.ne_w[rowp + b0 + m] = nev; }
                }
            }
            if (t < T) for (int idx = tid; idx < nb * W; idx += NT) {
                const int m = idx / W, q = idx - m * W;
                if (s_live[m] == 0.f) continue;
                const float cv = (t == 0) ? dm.first_rec : s_w[m * ldW + q];
                tp.zr[(rowb + b0 + m) * W + q] = cv;                                            // z_r of baseline_sen, model.py:836
                tp.c[(rowb + b0 + m) * W + q] = (t == 0) ? fsigmoid(P.p[S_CODE_BIAS][q]) : cv;
            }
            __syncthreads();                                                                    // s_w / s_pw are rewritten next step
        }
        if (t == T) return;
    }
}

template <int NT>
__device__ __forceinline__ void s2_role(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int tile, const int k) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int nw = NT / 64;
    const int tid = threadIdx.x, wave = tid >> 6;
    const int B = dm.B, H = dm.H, W = dm.W, R = dm.R, T = dm.T;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0), n0 = k * 32;
    const bool binary = dm.use_binary != 0, train = ar.train != 0;
    const bool may_stop = !ar.run_all && !dm.fixed && train;
    const SRoleLds L = srole_lds(dm, nw);
    const int ldH = ld16(H), ldZ = ld16(32), ld3R = ld16(3 * R);
    float* s_a = smem + L.a; float* s_zs = smem + L.zs; float* raw = smem + L.raw;
    float* s_bb = smem + L.vec; float* s_live = smem + L.live;
    for (int i = tid; i < L.total; i += NT) smem[i] = 0.f;
    __syncthreads();
    if (tid < 32) s_bb[tid] = P.p[S_BIN_B][n0 + tid];
    if (tid < MMG_TM) s_live[tid] = tid < nb ? 1.f : 0.f;
    const uint32_t mb_counter = tp.counter[0];
    uint32_t* cA = pf_ctr(tp, 1, tile); uint32_t* cZ = pf_ctr(tp, 2, tile); uint32_t* done = pf_ctr(tp, 3, tile);
    // register-resident weights (see s1_role): 32 rows of W_b (one item of 16 k-groups per wave) and the W_ih column slice
    const bool res_b = wfrag_fits(32, H, nw, 16, 1), res_i = wfrag_fits(3 * R, 32, nw, 2, 2);
    WFrag<16> fb; WFrag<2> fi0, fi1;
    fb.n = fi0.n = fi1.n = 0;
    if (res_b) wfrag_load<16>(fb, P.p[S_BIN_W] + (size_t)n0 * H, H, 32, H, wave, nw);
    if (res_i) { wfrag_load<2>(fi0, P.p[R_WIH] + n0, W, 3 * R, 32, wave, nw); wfrag_load<2>(fi1, P.p[R_WIH] + n0, W, 3 * R, 32, wave + nw, nw); }
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t rowb = (size_t)t * B;
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 210);
        if (!pf_wait(cA, (uint32_t)(ar.ns1 * (t + 1)), done, tp.sync)) return;
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 211);
        const int H4 = H >> 2;
        batched_for<NT, 8>(MMG_TM * H4, [&](int idx) { const int m = idx / H4, q = idx - m * H4; return reinterpret_cast<const float4*>(tp.a + (rowb + min(b0 + m, B - 1)) * H)[q]; },
                           [&](int idx, float4 v) { const int m = idx / H4, q = idx - m * H4; *reinterpret_cast<float4*>(s_a + m * ldH + 4 * q) = v; });
        if (t >= 1 && tid < MMG_TM) s_live[tid] = (tid < nb && (!may_stop || tp.mstate[min(b0 + tid, B - 1)] != 0.f)) ? 1.f : 0.f;
        __syncthreads();
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 212);
        if (res_b) wfrag_mma<16>(fb, s_a, ldH, raw, ldZ);                                        // 32 message logits, model.py:218
        else tgemm_nt_raw(s_a, ldH, P.p[S_BIN_W] + (size_t)n0 * H, H, 32, H, raw, wave, nw);
        __syncthreads();
        {
            const int kp = tile_kparts(2, nw);
            for (int idx = tid; idx < MMG_TM * 32; idx += NT) {
                const int m = idx >> 5, n = idx & 31, b = min(b0 + m, B - 1);
                const float lz = raw_sum(raw, ldZ, kp, m, n) + s_bb[n];
                float zz = lz, pp = 0.f;
                if (binary) {
                    pp = fsigmoid(lz);
                    if (train) {
                        const float u = ar.u_z ? ar.u_z[(rowb + b) * W + n0 + n]
                                               : philox_uniform(ar.seed, (uint32_t)((t * dm.Bg + dm.boff + b) * W + n0 + n), mb_counter, 0u);
                        zz = (u < pp) ? 1.f : 0.f;                                          // model.py:227
                    } else zz = rintf(pp);                                                  // model.py:229
                }
                s_zs[m * ldZ + n] = zz;
                if (m < nb && s_live[m] != 0.f) {                                           // (read back by the receiver role: write-through)
                    st_wt(&tp.z[(rowb + b) * W + n0 + n], zz);
                    if (binary) st_wt(&tp.pz[(rowb + b) * W + n0 + n], pp);
                }
            }
        }
        __syncthreads();
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 213);
        if (res_i) { wfrag_mma<2>(fi0, s_zs, ldZ, raw, ld3R); wfrag_mma<2>(fi1, s_zs, ldZ, raw, ld3R); }   // z_slice W_ih[:, slice]^T
        else tgemm_nt_raw(s_zs, ldZ, P.p[R_WIH] + n0, W, 3 * R, 32, raw, wave, nw);
        __syncthreads();
        {
            const int kp = tile_kparts((3 * R + 15) >> 4, nw);
            for (int idx = tid; idx < MMG_TM * 3 * R; idx += NT) {
                const int m = idx / (3 * R), n = idx - m * 3 * R;
                if (m < nb) st_wt(&tp.gip[((size_t)k * B + b0 + m) * 3 * R + n], raw_sum(raw, ld3R, kp, m, n));
            }
        }
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 214);
        pf_signal(cZ);
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 215);
    }
}

// ---------------------------------------------------------------------------------------------
// Fused sender roles (ar.rsample == 3; message width 256, per-sample receiver roles that publish the message):
//   sa_role j (H / 64 per tile): 64 units of the sender's hidden layer a_t AND its share of the message logits,
//       zpart_j = a_t[:, slice j] W_b[:, slice j]^T   [16, W]   (K = 64: both matrices of the role live in registers)
//   sb_role k (W / 16 per tile): adds the H / 64 partials for its 16 message bits, samples them, and forms its partial of the
//       GRU input product z_slice W_ih[:, slice]^T (K = 16).
// Against the s1 / s2 roles: no role reads the whole hidden tile (64 KB per role per step) and none runs a K = 1024 chain
// of dependent MFMAs; the hand-off S1 -> S2 carries 16 x 16 x 16 floats per consumer instead of 16 x 1024.
// ---------------------------------------------------------------------------------------------
// LL (rounds 5+): every hand-off of the fused roles is the payload itself as (value, tag) pairs (tape.pll_*), tag = (launch
// epoch << 4) | step: a consumer spins on the pairs it needs -- no store-completion wait and no counter at the producer, no second
// trip to memory at the consumer (scripts/micro/hop_latency.hip: 8 pairs per lane 0.9-1.2 us per hop as pairs against 1.5-2.1 us
// as payload + flag, and 190 polling workgroups cost the hop 0.07 us).  ONE slot per buffer: the ring receiver -> SA -> SB ->
// receiver orders the steps (a producer of step t + 1 has, through the ring, every consumer's step t behind it), and the slots
// stay hot -- per-step slots (36 MB a minibatch, every line cold) cost the hops 5 us.  A sample that stops publishes a zero
// message and a zero mask under the FINAL tag (step 15), which every later step accepts; a tile whose 16 masks are all zero is
// over, and its sender roles return.  (T <= 15; the tag wraps after 2^28 launches into slots rewritten every launch.)
__device__ __forceinline__ uint32_t pll_tag(uint32_t ep, int t) { return (ep << 4) | (uint32_t)t; }
__device__ __forceinline__ bool pll_fresh_or_final(unsigned long long u, uint32_t ep, int t) {
    const uint32_t g = (uint32_t)(u >> 32);
    return g == pll_tag(ep, t) || g == pll_tag(ep, 15);
}
#ifndef MMG_LL_SLEEP
#define MMG_LL_SLEEP 0          // s_sleep units (64 cycles) between two rounds of a pair poll
#endif
template <int NT, bool LL = false>
__device__ __forceinline__ void sa_role(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int tile, const int j) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int nw = NT / 64;
    const int wave = threadIdx.x >> 6;
    const int B = dm.B, H = dm.H, W = dm.W, T = dm.T;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0), n0 = j * 64;
    const bool train = ar.train != 0;
    const bool may_stop = !ar.run_all && !dm.fixed && train;
    const int ldW = ld16(W), ldA = ld16(64);
    float* s_w = smem; float* s_a = s_w + MMG_TM * ldW; float* raw = s_a + MMG_TM * ldA;      // raw: [16][ld16(W)] (>= [16][ld16(64)])
    float* s_bc = raw + MMG_TM * ldW; float* s_hw0 = s_bc + 64; float* s_live = s_hw0 + 64;
    {
        const int tid = threadIdx.x;
        for (int i = tid; i < MMG_TM * (ldW + ldA); i += NT) smem[i] = 0.f;
        if (tid < 64) { s_bc[tid] = P.p[S_CODE_B][n0 + tid]; s_hw0[tid] = tp.hw0[n0 + tid]; }
        if (tid < MMG_TM) s_live[tid] = tid < nb ? 1.f : 0.f;
    }
    float hxr[2];                                                       // h_x[b, n0 .. n0+63] of the tile: 1024 values on 512 threads
#pragma unroll
    for (int u = 0; u < 2; ++u) { const int idx = threadIdx.x + u * NT, m = idx >> 6, n = idx & 63; hxr[u] = tp.hx[(size_t)min(b0 + m, B - 1) * H + n0 + n]; }
    uint32_t* cG = pf_ctr(tp, 0, tile); uint32_t* cA = pf_ctr(tp, 1, tile); uint32_t* done = pf_ctr(tp, 3, tile);
    WFrag<8> fc; WFrag<4> fz0, fz1;                                     // W_c rows n0 ..: 4 n-tiles x 2 k-parts of 8 k-groups; W_b^T slice: 16 n-tiles of 4 k-groups
    wfrag_load<8>(fc, P.p[S_CODE_W] + (size_t)n0 * W, W, 64, W, wave, nw);
    wfrag_load<4>(fz0, P.p[S_BIN_W] + n0, H, W, 64, wave, nw);
    wfrag_load<4>(fz1, P.p[S_BIN_W] + n0, H, W, 64, wave + nw, nw);
    float* zp = tp.zpart + ((size_t)tile * (H / 64) + j) * MMG_TM * W;
    const uint32_t ep = LL ? tp.counter[3] : 0u;                       // (k_prep moved it before this launch; nothing moves it during)
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t rowb = (size_t)t * B, rowp = (size_t)(t > 0 ? t - 1 : 0) * B;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (t >= 1 && LL) {
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 200);
            // the tile's messages w_{t-1} [16][W] and stop masks: 8 pairs per thread (W = 256, 512 threads), spun on directly
            unsigned long long uw[8], um = 0;
            size_t iw[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int idx = tid + u * NT, m = idx / W, c = idx - m * W; iw[u] = (size_t)min(b0 + m, B - 1) * W + c; }
            for (int spins = 0;; ) {
                bool fresh = true;
#pragma unroll
                for (int u = 0; u < 8; ++u) { uw[u] = ld_ll(tp.pll_w, iw[u]); fresh = fresh && pll_fresh_or_final(uw[u], ep, t - 1); }
                if (tid < MMG_TM) { um = ld_ll(tp.pll_m, (size_t)min(b0 + tid, B - 1)); fresh = fresh && pll_fresh_or_final(um, ep, t - 1); }
                if (!__any(!fresh)) break;
                __builtin_amdgcn_s_sleep(MMG_LL_SLEEP);
                if (++spins > MMG_SPIN_LIMIT) { __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 101u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 201); MMG_RSTAMP(tile == 0 && j == 0 && t == 4, 206);
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int idx = tid + u * NT, m = idx / W, c = idx - m * W; if (m < MMG_TM) s_w[m * ldW + c] = ll_value(uw[u]); }
            const bool lv = tid < nb && (!may_stop || ll_value(um) != 0.f);
            if (tid < MMG_TM) s_live[tid] = lv ? 1.f : 0.f;
            if (!__syncthreads_or(tid < MMG_TM && lv)) return;          // every sample of the tile has stopped
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 202);
            wfrag_mma<8>(fc, s_w, ldW, raw, ldA);
            __syncthreads();
        } else if (t >= 1) {
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 200);
            if (!pf_wait<false>(pf_ctr(tp, 4 + (t - 1), tile), (uint32_t)nb, done, tp.sync)) return;      // (the step's own counter: s1_role)
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 201); MMG_RSTAMP(tile == 0 && j == 0 && t == 4, 206);
            {
                const int W4 = W >> 2;
                batched_for<NT, 2>(MMG_TM * W4, [&](int idx) { const int m = idx / W4, q = idx - m * W4; return ld_cc4(tp.w + (rowp + min(b0 + m, B - 1)) * W + 4 * q); },
                                   [&](int idx, float4 v) { const int m = idx / W4, q = idx - m * W4; *reinterpret_cast<float4*>(s_w + m * ldW + 4 * q) = v; });
            }
            if (tid < MMG_TM) s_live[tid] = (tid < nb && (!may_stop || ld_cc(&tp.mstate[min(b0 + tid, B - 1)]) != 0.f)) ? 1.f : 0.f;
            __syncthreads();
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 202);
            wfrag_mma<8>(fc, s_w, ldW, raw, ldA);                       // this role's 64 units of the sender hidden state (model.py:195-216)
            __syncthreads();
        }
        {
            const int kp = tile_kparts(4, nw);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int idx = tid + u * NT, m = idx >> 6, n = idx & 63;
                const float hw = (t == 0) ? s_hw0[n] : raw_sum(raw, ldA, kp, m, n) + s_bc[n];
                const float av = ftanh(hxr[u] + hw);
                s_a[m * ldA + n] = av;
                if (m < nb && s_live[m] != 0.f) tp.a[(rowb + b0 + m) * H + n0 + n] = av;
            }
        }
        __syncthreads();
        MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 203);
        wfrag_mma<4>(fz0, s_a, ldA, raw, ldW); wfrag_mma<4>(fz1, s_a, ldA, raw, ldW);      // partial message logits over this role's K slice
        __syncthreads();
        if (LL) {
            const size_t zb = ((size_t)tile * (H / 64) + j) * MMG_TM * W;                        // pair index of this role's [16][W] block
            for (int idx = tid; idx < MMG_TM * (W >> 1); idx += NT) {
                const int m = idx / (W >> 1), n = (idx - m * (W >> 1)) * 2;
                const float2 v = *reinterpret_cast<const float2*>(raw + m * ldW + n);
                st_ll2(tp.pll_zp, zb + (size_t)m * W + n, v.x, v.y, pll_tag(ep, t));
            }
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 204);
            MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 205);
            continue;                                                    // (raw is rewritten behind the next step's first barrier)
        }
        for (int idx = tid; idx < MMG_TM * (W >> 2); idx += NT) {
            const int m = idx / (W >> 2), n = (idx - m * (W >> 2)) * 4;
            st_wt4(zp + m * W + n, *reinterpret_cast<const float4*>(raw + m * ldW + n));
        }
        MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 204);
        pf_signal(cA);
        MMG_RSTAMP(tile == 0 && j == 0 && t == 3, 205);
    }
}

template <int NT, bool LL = false>
__device__ __forceinline__ void sb_role(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int tile, const int k) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int nw = NT / 64;
    const int wave = threadIdx.x >> 6;
    const int B = dm.B, H = dm.H, W = dm.W, R = dm.R, T = dm.T;
    const int b0 = tile * MMG_TM, nb = min(MMG_TM, B - b0), c0 = k * 16, ns1 = H / 64;
    const bool binary = dm.use_binary != 0, train = ar.train != 0;
    const bool may_stop = !ar.run_all && !dm.fixed && train;
    const int ldZ = ld16(16), ld3R = ld16(3 * R);
    float* s_zs = smem; float* raw = s_zs + MMG_TM * ldZ; float* s_live = raw + MMG_TM * ld3R;
    {
        const int tid = threadIdx.x;
        for (int i = tid; i < MMG_TM * ldZ; i += NT) smem[i] = 0.f;
        if (tid < MMG_TM) s_live[tid] = tid < nb ? 1.f : 0.f;
    }
    const uint32_t mb_counter = tp.counter[0];
    uint32_t* cA = pf_ctr(tp, 1, tile); uint32_t* cZ = pf_ctr(tp, 2, tile); uint32_t* done = pf_ctr(tp, 3, tile);
    WFrag<1> fi0, fi1;                                                   // W_ih[:, c0 .. c0+15]: 12 n-tiles, one k-group
    wfrag_load<1>(fi0, P.p[R_WIH] + c0, W, 3 * R, 16, wave, nw);
    wfrag_load<1>(fi1, P.p[R_WIH] + c0, W, 3 * R, 16, wave + nw, nw);
    // thread (output o = tid / 2 = (sample m, bit c), half hf): adds the partials of roles hf, hf + 2, ...
    const int o = threadIdx.x >> 1, hf = threadIdx.x & 1, om = o >> 4, oc = o & 15;
    const float bbv = P.p[S_BIN_B][c0 + oc];
    const float* zp = tp.zpart + (size_t)tile * ns1 * MMG_TM * W + (size_t)om * W + c0 + oc;
    const uint32_t ep = LL ? tp.counter[3] : 0u;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const size_t rowb = (size_t)t * B;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int bq = min(b0 + om, B - 1);
        float uz = 0.f;                                                  // (drawn before the wait)
        if (binary && train && hf == 0)
            uz = ar.u_z ? ar.u_z[(rowb + bq) * W + c0 + oc] : philox_uniform(ar.seed, (uint32_t)((t * dm.Bg + dm.boff + bq) * W + c0 + oc), mb_counter, 0u);
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 210);
        float p8[8], acc = 0.f;
        if (LL) {
            // (1) the masks after step t - 1 (long since there: the receiver roles published them before the SA roles even
            //     started this step) decide whether the tile still runs; (2) the SA roles' partials of this role's 16 bits
            if (t >= 1) {
                unsigned long long um = 0;
                if (tid < MMG_TM) {
                    for (int spins = 0;; ) {
                        um = ld_ll(tp.pll_m, (size_t)min(b0 + tid, B - 1));
                        if (pll_fresh_or_final(um, ep, t - 1)) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > MMG_SPIN_LIMIT) { __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 102u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                    }
                }
                const bool lv = tid < nb && (!may_stop || ll_value(um) != 0.f);
                if (tid < MMG_TM) s_live[tid] = lv ? 1.f : 0.f;
                if (!__syncthreads_or(tid < MMG_TM && lv)) return;
            }
            unsigned long long up[8];
            const size_t zb = ((size_t)tile * ns1) * MMG_TM * W + (size_t)om * W + c0 + oc;
            for (int spins = 0;; ) {
                bool fresh = true;
#pragma unroll
                for (int u = 0; u < 8; ++u) { up[u] = ld_ll(tp.pll_zp, zb + (size_t)min(hf + 2 * u, ns1 - 1) * MMG_TM * W); fresh = fresh && ll_fresh(up[u], pll_tag(ep, t)); }
                if (!__any(!fresh)) break;
                __builtin_amdgcn_s_sleep(MMG_LL_SLEEP);
                if (++spins > MMG_SPIN_LIMIT) { __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 103u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 211);
#pragma unroll
            for (int u = 0; u < 8; ++u) p8[u] = ll_value(up[u]);
        } else {
            if (!pf_wait<false>(cA, (uint32_t)(ns1 * (t + 1)), done, tp.sync)) return;
            MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 211);
#pragma unroll
            for (int u = 0; u < 8; ++u) p8[u] = ld_cc(zp + (size_t)min(hf + 2 * u, ns1 - 1) * MMG_TM * W);
            if (t >= 1 && tid < MMG_TM) s_live[tid] = (tid < nb && (!may_stop || ld_cc(&tp.mstate[min(b0 + tid, B - 1)]) != 0.f)) ? 1.f : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (hf + 2 * u < ns1) ? p8[u] : 0.f;
        acc = dpp_group_sum<2>(acc);
        if (!LL) __syncthreads();                                        // s_live
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 212);
        float zq = 0.f, pq = 0.f;
        if (hf == 0) {
            const float lz = acc + bbv;
            float zz = lz, pp = 0.f;
            if (binary) {
                pp = fsigmoid(lz);
                zz = train ? ((uz < pp) ? 1.f : 0.f) : rintf(pp);                               // model.py:227 / 229
            }
            s_zs[om * ldZ + oc] = zz;
            zq = zz; pq = pp;
        }
        if (LL) {
            // the receiver role of sample om reads (bit, probability) as ONE pair: the probability carries the bit in its sign
            // (binary mode; continuous messages are not read back).  The tape rows are plain stores (read after the launch).
            const bool lrow = om < nb && s_live[om] != 0.f;
            if (hf == 0 && lrow) {
                const size_t o = (rowb + b0 + om) * W + c0 + oc;
                if (binary) st_ll(tp.pll_z, (size_t)(b0 + om) * W + c0 + oc, (zq != 0.f) ? pq : -pq, pll_tag(ep, t));
                tp.z[o] = zq;
                if (binary) tp.pz[o] = pq;
            }
        } else {
            const float4 z4 = gather4_even(zq), p4v = gather4_even(pq);                         // (read back by the receiver roles: write-through)
            if ((tid & 7) == 0 && om < nb && s_live[om] != 0.f) {
                st_wt4(&tp.z[(rowb + b0 + om) * W + c0 + oc], z4);
                if (binary) st_wt4(&tp.pz[(rowb + b0 + om) * W + c0 + oc], p4v);
            }
        }
        __syncthreads();
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 213);
        wfrag_mma<1>(fi0, s_zs, ldZ, raw, ld3R); wfrag_mma<1>(fi1, s_zs, ldZ, raw, ld3R);       // z_slice W_ih[:, slice]^T
        __syncthreads();
        if (LL) {
            const size_t gb = ((size_t)k * B + b0) * 3 * R;                                      // pair index of row b0 of this role's [B][3R] block
            for (int idx = tid; idx < MMG_TM * (3 * R >> 1); idx += NT) {
                const int m = idx / (3 * R >> 1), n = (idx - m * (3 * R >> 1)) * 2;
                const float2 v = *reinterpret_cast<const float2*>(raw + m * ld3R + n);
                if (m < nb) st_ll2(tp.pll_gi, gb + (size_t)m * 3 * R + n, v.x, v.y, pll_tag(ep, t));
            }
            MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 214);
            MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 215);
            continue;                                                    // (raw / s_zs are rewritten behind the next step's barriers)
        }
        for (int idx = tid; idx < MMG_TM * (3 * R >> 2); idx += NT) {
            const int m = idx / (3 * R >> 2), n = (idx - m * (3 * R >> 2)) * 4;
            if (m < nb) st_wt4(&tp.gip[((size_t)k * B + b0 + m) * 3 * R + n], *reinterpret_cast<const float4*>(raw + m * ld3R + n));
        }
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 214);
        pf_signal(cZ);
        MMG_RSTAMP(tile == 0 && k == 0 && t == 3, 215);
    }
}

// ---------------------------------------------------------------------------------------------
// rs_role: the receiver of ONE sample as a role of k_conv_persist (agents with a large sender but the receiver shape of
// BASELINE configs 1-4: R = 64, V = 100, D <= 32).  A 16-sample tile role walks through eight barrier-separated phases of
// [16, K] x [K, N] products per step (~20 us); one workgroup per sample with every receiver weight in registers (the
// lane layouts of k_conversation_fast2, kernels_fast.h) does the same step in ~3 us, and 64 of them fit beside the sender
// roles.  The GRU's input-side product arrives as the S2 roles' partials (W_ih never enters this role); g_t goes out with
// write-through stores and one count per sample on the tile's g counter.  A sample that stops adds the counts of the steps
// it will not take, the last one of a tile raises the tile's done flag.
// ---------------------------------------------------------------------------------------------
// MW = 256: the role also forms the receiver's message w_t = Bernoulli(sigmoid(W_w g_t + b_w)) (2 lanes per message bit, W_w in
// registers) and publishes IT instead of g_t: the S1 roles then start from the message (no redundant 16-fold recompute of
// it, one product less on the per-step chain).  MW = 0: other widths, g_t goes out and the S1 roles form the message.
template <int NT, int R, int V, int D, int MW, bool LL = false>
__device__ __forceinline__ void rs_role(const Dims& dm, const Params& P, const Tape& tp, const ConvArgs& ar, const int b) {
    static_assert(!LL || MW == 256, "pair hand-offs: the fused sender roles' shape");
    static_assert(NT == 512 && R == 64 && D <= 32 && V <= 200 && (MW == 0 || MW == 256), "receiver shape of the register-resident kernels");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_h = smem; float* s_gi = s_h + R; float* s_gh = s_gi + 3 * R; float* s_A = s_gh + 3 * R;
    float* s_y = s_A + R; float* s_yout = s_y + 32; float* s_dbar = s_yout + 32; float* s_pi = s_dbar + 208;   // [8][32]
    float* s_misc = s_pi + 256; float* s_us = s_misc + 8; float* s_lpz = s_us + 16;                             // [2][8]
    float* s_g = s_lpz + 16; float* s_c = s_g + R; float* s_lpw = s_c + 256;                                      // message path (MW): g, last message, [2][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = dm.B, T = dm.T, W = dm.W, Dr = dm.D;
    const int tile = b / MMG_TM, nb_tile = min(MMG_TM, B - tile * MMG_TM);
    const bool binary = dm.use_binary != 0, train = ar.train != 0;
    const bool may_stop = !ar.run_all && !dm.fixed && train;
    const uint32_t mb_counter = tp.counter[0];
    const uint32_t gb = (uint32_t)(dm.boff + b);
    if (train && tid < T) s_us[tid] = ar.u_s ? ar.u_s[(size_t)tid * B + b] : philox_uniform(ar.seed, (uint32_t)(tid * dm.Bg + gb), mb_counter, 1u);
    // ---- weights -> registers (once)
    // GRU hidden side: row n3 = tid/2 (< 3R), half h3
    const int n3 = tid >> 1, h3 = tid & 1;
    const bool gru_lane = n3 < 3 * R;
    const int nr = gru_lane ? n3 : 0;
    constexpr int J3H = R / 8;
    float whh[4 * J3H];
#pragma unroll
    for (int j = 0; j < J3H; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(P.p[R_WHH] + (size_t)nr * R + (h3 + 2 * j) * 4);
        whh[4 * j] = v.x; whh[4 * j + 1] = v.y; whh[4 * j + 2] = v.z; whh[4 * j + 3] = v.w;
    }
    const float bih = P.p[R_BIH][nr], bhh = P.p[R_BHH][nr];
    // y1[:, :R], w_h, w_d: 8 lanes per row
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
    // y head: 16 lanes per class
    constexpr int LY = 16;
    const int dy = tid / LY, kpy = tid % LY;
    float cd[4], w2[4];
    {
        const float4 v = *reinterpret_cast<const float4*>(tp.Cd + (size_t)(dy < Dr ? dy : 0) * R + kpy * 4);
        cd[0] = v.x; cd[1] = v.y; cd[2] = v.z; cd[3] = v.w;
        const float4 u = *reinterpret_cast<const float4*>(P.p[R_Y2_W] + kpy * 4);
        w2[0] = u.x; w2[1] = u.y; w2[2] = u.z; w2[3] = u.w;
    }
    const float b2 = P.p[R_Y2_B][0];
    // description column v7 = tid/2 (< V), half h7 covers classes h7*DH .. h7*DH + DH-1
    constexpr int DH = (D + 1) / 2;
    const int v7 = tid >> 1, h7 = tid & 1;
    float dcol[DH];
#pragma unroll
    for (int j = 0; j < DH; ++j) { const int d = h7 * DH + j; dcol[j] = ar.desc[(size_t)min(d, Dr - 1) * V + min(v7, V - 1)]; if (!(v7 < V && d < Dr)) dcol[j] = 0.f; }

    // message head (MW): row nw = tid/2 of W_w, half hw (the GRU lanes' k pattern)
    const int nw = tid >> 1, hw = tid & 1;
    float ww[MW ? 4 * J3H : 1];
    float bw = 0.f, sig_cb = 0.f;
    if (MW) {
#pragma unroll
        for (int j = 0; j < (MW ? J3H : 0); ++j) {
            const float4 v = *reinterpret_cast<const float4*>(P.p[R_W_W] + (size_t)nw * R + (hw + 2 * j) * 4);
            ww[4 * j] = v.x; ww[4 * j + 1] = v.y; ww[4 * j + 2] = v.z; ww[4 * j + 3] = v.w;
        }
        bw = P.p[R_W_B][nw];
        sig_cb = fsigmoid(P.p[S_CODE_BIAS][nw]);
        if (hw == 0) s_c[nw] = dm.first_rec;                               // model.py:786
    }

    if (tid < R) { s_h[tid] = 0.f; tp.h[(size_t)b * R + tid] = 0.f; }
    if (tid == 0) { s_misc[0] = 1.f; s_misc[1] = -1.f; s_misc[2] = 1.f; tp.mask[b] = 1; }
    __syncthreads();
    uint32_t* cG = pf_ctr(tp, 0, tile); uint32_t* cZ = pf_ctr(tp, 2, tile); uint32_t* done = pf_ctr(tp, 3, tile);
    const int ns2 = ar.ns2;
    const uint32_t ep = LL ? tp.counter[3] : 0u;
    float stop_p = 0.5f, stop_bit = 0.f;
    int signalled = 0;
    int t = 0;
    for (; t < T; ++t) {
        const size_t row = (size_t)t * B + b;
        const float ghv = bhh + dpp_group_sum<2>(dot4<J3H>(whh, s_h + h3 * 4, 8));      // hidden-side product: before the wait
        float uwv = 0.f;
        if (MW) {
            if (hw == 0) {                                                 // the sender's input of this step (tape only), model.py:836
                const float cv = s_c[nw];
                tp.zr[row * W + nw] = cv;
                tp.c[row * W + nw] = (t == 0) ? sig_cb : cv;
            }
            if (binary && train) uwv = ar.u_w ? ar.u_w[row * W + nw] : philox_uniform(ar.seed, (uint32_t)((t * dm.Bg + gb) * W + nw), mb_counter, 2u);
        }
        // ===== the sender roles' message of this step: its GRU input-side product arrives as ns2 partials
        MMG_RSTAMP(b == 0 && t == 3, 250); MMG_RSTAMP(t == 3 && b < 64, 256 + 8192 + 8 * b + 0);
        if (!LL) { if (!pf_wait<false>(cZ, (uint32_t)(ns2 * (t + 1)), nullptr, tp.sync)) return; }
        {
            float p4[8];
            float zz = 0.f, pp = 0.5f;
            if (LL) {
                // the SB roles' partials of this sample's row (8 per lane) and, binary mode, the message itself: spun on directly
                unsigned long long ug[8], uzp = 0;
                const size_t gbase = (size_t)b * 3 * R + nr;
                for (int spins = 0;; ) {
                    bool fresh = true;
#pragma unroll
                    for (int u = 0; u < 8; ++u) { ug[u] = ld_ll(tp.pll_gi, gbase + (size_t)min(h3 + 2 * u, ns2 - 1) * B * 3 * R); fresh = fresh && ll_fresh(ug[u], pll_tag(ep, t)); }
                    if (binary && tid < W) { uzp = ld_ll(tp.pll_z, (size_t)b * W + tid); fresh = fresh && ll_fresh(uzp, pll_tag(ep, t)); }
                    if (!__any(!fresh)) break;
                    __builtin_amdgcn_s_sleep(MMG_LL_SLEEP);
                if (++spins > MMG_SPIN_LIMIT) { __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 104u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) p4[u] = ll_value(ug[u]);
                if (binary && tid < W) { const float v = ll_value(uzp); zz = (__builtin_bit_cast(unsigned, v) >> 31) ? 0.f : 1.f; pp = fabsf(v); }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) p4[u] = ld_cc(&tp.gip[((size_t)min(h3 + 2 * u, ns2 - 1) * B + b) * 3 * R + nr]);
                if (binary && tid < W) { zz = ld_cc(&tp.z[row * W + tid]); pp = ld_cc(&tp.pz[row * W + tid]); }
            }
            MMG_RSTAMP(b == 0 && t == 3, 251); MMG_RSTAMP(t == 3 && b < 64, 256 + 8192 + 8 * b + 1);
            float gp = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) gp += (h3 + 2 * u < ns2) ? p4[u] : 0.f;
            const float giv = bih + dpp_group_sum<2>(gp);
            if (gru_lane && h3 == 0) { s_gi[n3] = giv; s_gh[n3] = ghv; }
            if (binary && wave < 4) {                                      // log-likelihood / neg-entropy of the sender's bits, model.py:908-922
                float lpv = 0.f, nev = 0.f;
                if (tid < W) {
                    const float l1 = flog(pp + MMG_EPS), l0 = flog(1.f - pp + MMG_EPS);
                    lpv = zz * l1 + (1.f - zz) * l0; nev = pp * l1 + (1.f - pp) * l0;
                }
                lpv = dpp_wave_sum(lpv); nev = dpp_wave_sum(nev);
                if (lane == 0) { s_lpz[wave] = lpv; s_lpz[8 + wave] = nev; }
            }
        }
        __syncthreads();
        MMG_RSTAMP(b == 0 && t == 3, 252); MMG_RSTAMP(t == 3 && b < 64, 256 + 8192 + 8 * b + 2);
        // ===== GRU state update
        if (tid < R) {
            const float rr = fsigmoid(s_gi[tid] + s_gh[tid]);
            const float uu = fsigmoid(s_gi[R + tid] + s_gh[R + tid]);
            const float ghn = s_gh[2 * R + tid];
            const float nn = ftanh(s_gi[2 * R + tid] + rr * ghn);
            const float hv = nn + uu * (s_h[tid] - nn);
            float* gr = tp.gru + row * 4 * R;
            gr[tid] = rr; gr[R + tid] = uu; gr[2 * R + tid] = nn; gr[3 * R + tid] = ghn;
            tp.h[((size_t)(t + 1) * B + b) * R + tid] = hv;
            s_h[tid] = hv;
        } else if (binary && tid == R) {
            const int nwz = (W + 63) >> 6;
            float a = 0.f, c = 0.f;
            for (int q = 0; q < nwz; ++q) { a += s_lpz[q]; c += s_lpz[8 + q]; }
            tp.lp_z[row] = a; tp.ne_z[row] = c;
        }
        __syncthreads();
        // ===== heads on h
        float gpre_h;
        {
            const float accA = dpp_group_sum<L4>(dot4<J4>(wy1, s_h + kp4 * 4, 4 * L4));
            const float accH = dpp_group_sum<L4>(dot4<J4>(wh, s_h + kp4 * 4, 4 * L4));
            if (kp4 == 0) s_A[n4] = accA;
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
                tp.s[row] = sbit; tp.ps[row] = p;
                stop_p = p; stop_bit = sbit;
            }
        }
        __syncthreads();
        // ===== class logits
        {
            const float4 a4 = *reinterpret_cast<const float4*>(s_A + kpy * 4);
            float acc = w2[0] * fmax_nn(a4.x + cd[0], 0.f);
            acc = fmaf(w2[1], fmax_nn(a4.y + cd[1], 0.f), acc);
            acc = fmaf(w2[2], fmax_nn(a4.z + cd[2], 0.f), acc);
            acc = fmaf(w2[3], fmax_nn(a4.w + cd[3], 0.f), acc);
            acc = dpp_group_sum<LY>(acc);
            if (kpy == 0) {
                const float yv = (dy < Dr) ? acc + b2 : -3.0e38f;
                s_y[dy] = yv;
                if (dy < Dr) tp.y[row * Dr + dy] = yv;
            }
        }
        const float m_t = s_misc[0], sbit = s_misc[3];
        const float m_next = fminf(m_t, sbit);
        const bool first_stop = (m_next == 0.f) && (s_misc[1] < 0.f);
        const bool take_out = dm.fixed ? (t == T - 1) : (first_stop || ((t == T - 1) && (s_misc[1] < 0.f)));
        __syncthreads();
        if (take_out && tid < 32) s_yout[tid] = s_y[tid];
        if (tid == 0) {
            tp.mask[(size_t)(t + 1) * B + b] = (uint8_t)(m_next != 0.f);
            if (take_out) s_misc[1] = (float)t;
            s_misc[0] = m_next;
            if (LL) st_ll(tp.pll_m, (size_t)b, m_next, pll_tag(ep, (may_stop && m_next == 0.f) ? 15 : t));   // the sender roles store live rows only; all zero: the tile is over
            else st_wt(&tp.mstate[b], m_next);
        }
        if (wave == 7 && lane == 0) {
            const float l1 = flog(stop_p + MMG_EPS), l0 = flog(1.f - stop_p + MMG_EPS);
            tp.lp_s[row] = stop_bit * l1 + (1.f - stop_bit) * l0;
            tp.ne_s[row] = stop_p * l1 + (1.f - stop_p) * l0;
        }
        if (may_stop && m_next == 0.f) {
            if (LL) {
                // this sample takes no further step: a zero message under the FINAL tag, which every later step of its sender roles
                // accepts (the zero mask went out above, under the same tag)
                if (tid < W) st_ll(tp.pll_w, (size_t)b * W + tid, 0.f, pll_tag(ep, 15));
            }
            ++t; __syncthreads(); break;
        }
        // ===== softmax (per wave) -> wave-private LDS -> description mixture (2 lanes per column)
        {
            const float yv = (lane < 32) ? s_y[lane] : -3.0e38f;
            float mx = fmaxf(yv, dpp_f<MMG_DPP_QUAD_1032>(yv)); mx = fmaxf(mx, dpp_f<MMG_DPP_QUAD_2301>(mx));
            mx = fmaxf(mx, dpp_f<MMG_DPP_ROW_HALF_MIRROR>(mx)); mx = fmaxf(mx, dpp_f<MMG_DPP_ROW_MIRROR>(mx));
            const float m0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 0));
            const float m1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mx), 16));
            const float mxx = fmaxf(m0, m1);
            const float e = (lane < Dr) ? __expf(yv - mxx) : 0.f;
            const float rs = dpp_group_sum<16>(e);
            const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs), 0));
            const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs), 16));
            const float inv = __builtin_amdgcn_rcpf(s0 + s1);
            float* pi = s_pi + wave * 32;
            if (lane < 32) pi[lane] = e * inv;
            __builtin_amdgcn_wave_barrier();
            float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
            for (int j = 0; j + 2 < DH; j += 3) {
                q0 = fmaf(pi[h7 * DH + j], dcol[j], q0);
                q1 = fmaf(pi[h7 * DH + j + 1], dcol[j + 1], q1);
                q2 = fmaf(pi[h7 * DH + j + 2], dcol[j + 2], q2);
            }
#pragma unroll
            for (int j = DH - DH % 3; j < DH; ++j) q0 = fmaf(pi[min(h7 * DH + j, 31)], dcol[j], q0);
            const float acc = dpp_group_sum<2>((q0 + q1) + q2);
            if (h7 == 0 && v7 < V) { s_dbar[v7] = acc; tp.dbar[row * V + v7] = acc; }
        }
        __syncthreads();
        // ===== h_w = tanh(w_h h + b_h + w_d dbar) -> the sender roles
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
                if (MW) { s_g[n4] = gv; tp.g[row * R + n4] = gv; }
                else st_wt(&tp.g[row * R + n4], gv);
            }
        }
        MMG_RSTAMP(b == 0 && t == 3, 253); MMG_RSTAMP(t == 3 && b < 64, 256 + 8192 + 8 * b + 3);
        if (MW) {
            __syncthreads();
            // ===== receiver message (model.py:454-475) -> the S1 roles
            const float acc = dpp_group_sum<2>(dot4<J3H>(ww, s_g + hw * 4, 8));
            float lpv = 0.f, nev = 0.f, wq = 0.f;
            if (hw == 0) {
                const float lw = acc + bw;
                float wv = lw, pp = 0.f;
                if (binary) {
                    pp = fsigmoid(lw);
                    wv = train ? ((uwv < pp) ? 1.f : 0.f) : rintf(pp);
                    tp.pw[row * W + nw] = pp;
                    const float l1 = flog(pp + MMG_EPS), l0 = flog(1.f - pp + MMG_EPS);
                    lpv = wv * l1 + (1.f - wv) * l0; nev = pp * l1 + (1.f - pp) * l0;
                }
                s_c[nw] = wv;
                wq = wv;
            }
            if (LL) {
                if (hw == 0) { st_ll(tp.pll_w, (size_t)b * W + nw, wq, pll_tag(ep, t)); tp.w[row * W + nw] = wq; }   // pair: the SA roles; tape: a plain store
            } else {
                const float4 w4 = gather4_even(wq);                         // four message bits per 16-byte write-through store
                if ((tid & 7) == 0) st_wt4(&tp.w[row * W + nw], w4);
            }
            if (binary) {
                lpv = dpp_wave_sum(lpv); nev = dpp_wave_sum(nev);
                if (lane == 0) { s_lpw[wave] = lpv; s_lpw[8 + wave] = nev; }
            }
        }
        MMG_RSTAMP(b == 0 && t == 3, 254); MMG_RSTAMP(t == 3 && b < 64, 256 + 8192 + 8 * b + 4);
        if (LL) __syncthreads(); else pf_signal(pf_ctr(tp, 4 + t, tile));   // (LL: s_lpw is read below; the message went out as pairs)
        MMG_RSTAMP(b == 0 && t == 3, 255); MMG_RSTAMP(t == 3 && b < 64, 256 + 8192 + 8 * b + 5);
        ++signalled;
        if (MW && binary && tid == 0) {                                    // (after the signal's barrier: off the sender roles' path)
            float a = 0.f, c = 0.f;
            for (int q = 0; q < 8; ++q) { a += s_lpw[q]; c += s_lpw[8 + q]; }
            tp.lp_w[row] = a; tp.ne_w[row] = c;
        }
    }
    __syncthreads();
    // the counts of the steps this sample does not take; the tile's last sample ends the sender roles
    if (!LL && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (the final stop mask is out before the counts move)
        for (int t2 = signalled; t2 < T; ++t2) __hip_atomic_fetch_add(pf_ctr(tp, 4 + t2, tile), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t fin = __hip_atomic_fetch_add(done + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)fin == nb_tile - 1) __hip_atomic_store(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- output selection / reward / top-k (model.py:1264-1275, 1333-1339)
    const int tstar = dm.fixed ? (T - 1) : (int)s_misc[1];
    if (tid < 64) {
        const float o = (lane < 32) ? s_yout[lane] : -3.0e38f;
        const float mx = dpp_wave_max(o);
        const float e = (lane < Dr) ? __expf(o - mx) : 0.f;
        const float lse = mx + flog(dpp_wave_sum(e));
        const int tgt = ar.target ? (int)ar.target[b] : -1;
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
            tp.sprod[b] = s_misc[2];
            tp.logs[b] = dt;
            tp.hit[b] = (tgt >= 0 && above < (float)dm.top_k) ? 1 : 0;
        }
    }
}

// SAMPLE_ROLES: the launch of per-sample receiver roles (ar.rsample != 0) and the variant with one 16-sample receiver role per
// tile are two kernels -- the tile role needs every one of its 256 registers (and spills a few), the per-sample roles do not,
// and a workgroup should not carry the code of roles its launch never runs (62 k lines of ISA for the union)
template <int NT, bool SAMPLE_ROLES, bool LL = false>
__global__ __launch_bounds__(NT) void k_conv_persist(Dims dm, Params P, Tape tp, ConvArgs ar, int tiles) {
    int blk = blockIdx.x;
    if (SAMPLE_ROLES && LL) {                           // fused sender roles with (value, epoch) pair hand-offs (host: ar.rsample == 3 only)
        const int cb = ar.b_count, ctile0 = ar.b_begin / MMG_TM, ctiles = (cb + MMG_TM - 1) / MMG_TM;
        const int nrole = cb + ctiles * (ar.ns1 + ar.ns2);
        if (blk >= nrole) {
            if (threadIdx.x >= MMG_BLOCK) return;
            gemm_nt_tile(blk - nrole, tp.hx, dm.H, P.p[BS_L1_W], dm.H + dm.W, nullptr, tp.basehx, dm.K, dm.B, dm.K, dm.H);
            return;
        }
        if (blk < cb) {
            const int b = ar.b_begin + blk;
            if (dm.D == 30) rs_role<NT, 64, 100, 30, 256, true>(dm, P, tp, ar, b); else rs_role<NT, 64, 100, 32, 256, true>(dm, P, tp, ar, b);
            return;
        }
        const int r = blk - cb;
        if (r < ctiles * ar.ns2) sb_role<NT, true>(dm, P, tp, ar, ctile0 + r / ar.ns2, r % ar.ns2);
        else { const int q = r - ctiles * ar.ns2; sa_role<NT, true>(dm, P, tp, ar, ctile0 + q / ar.ns1, q % ar.ns1); }
        return;
    }
    if (SAMPLE_ROLES) {                                 // one receiver role per sample, then the sender roles of the tiles
        // this launch covers the samples [b_begin, b_begin + b_count) = the tiles ctile0 .. ctile0 + ctiles - 1
        const int cb = ar.b_count, ctile0 = ar.b_begin / MMG_TM, ctiles = (cb + MMG_TM - 1) / MMG_TM;
        const int nrole = cb + ctiles * (ar.ns1 + ar.ns2);
        if (blk >= nrole) {
            // trailing workgroups (dispatched after every role, onto CUs the roles leave idle): 16 x 16 tiles of
            // basehx = h_x . baseline_sen.linear1.weight[:, :H]^T, which k_baselines4 needs next and nothing here produces
            if (threadIdx.x >= MMG_BLOCK) return;        // gemm_nt_tile is a 4-wave routine
            gemm_nt_tile(blk - nrole, tp.hx, dm.H, P.p[BS_L1_W], dm.H + dm.W, nullptr, tp.basehx, dm.K, dm.B, dm.K, dm.H);
            return;
        }
        if (blk < cb) {
            const int b = ar.b_begin + blk;
            if (ar.rsample >= 2) { if (dm.D == 30) rs_role<NT, 64, 100, 30, 256>(dm, P, tp, ar, b); else rs_role<NT, 64, 100, 32, 256>(dm, P, tp, ar, b); }
            else { if (dm.D == 30) rs_role<NT, 64, 100, 30, 0>(dm, P, tp, ar, b); else rs_role<NT, 64, 100, 32, 0>(dm, P, tp, ar, b); }
            return;
        }
        const int r = blk - cb;
        if (ar.rsample == 3) {                          // fused sender roles: ns2 sb roles, then ns1 sa roles per tile
            if (r < ctiles * ar.ns2) sb_role<NT>(dm, P, tp, ar, ctile0 + r / ar.ns2, r % ar.ns2);
            else { const int q = r - ctiles * ar.ns2; sa_role<NT>(dm, P, tp, ar, ctile0 + q / ar.ns1, q % ar.ns1); }
        } else {
            if (r < ctiles * ar.ns2) s2_role<NT>(dm, P, tp, ar, ctile0 + r / ar.ns2, r % ar.ns2);
            else { const int q = r - ctiles * ar.ns2; s1_role<NT>(dm, P, tp, ar, ctile0 + q / ar.ns1, q % ar.ns1); }
        }
        return;
    }
    // (a per-XCD placement of a tile's roles -- all hand-offs inside one L2 -- measured SLOWER at config 4, 575 against 527 us per
    //  minibatch: a load that follows a write-through store in the same L2 took 5-8 us instead of 2.3; deleted in round 6)
    if (blk < tiles) { conv_tile_body<NT, true>(dm, P, tp, ar, blk); return; }
    const int r = blk - tiles;
    if (r < tiles * ar.ns2) { s2_role<NT>(dm, P, tp, ar, r / ar.ns2, r % ar.ns2); return; }
    const int q = r - tiles * ar.ns2;
    s1_role<NT>(dm, P, tp, ar, q / ar.ns1, q % ar.ns1);
}

// (scripts/isa_stats.py -D MMG_ROLE_DIAG: every role of k_conv_persist as a kernel of its own -- diag_kernels.h)

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

namespace mmg {

// ---------------------------------------------------------------------------------------------
// k_bwd_tile: reverse-time pass of one 16-sample tile (same math and tape contract as k_bwd_conv, kernels_bwd.h):
// output step: dy = (softmax - onehot)/B, A* = W_y1[:, :R] h*, dA (class-side sum over the D rows of Cd, streamed once per
// tile); then per step, back to front: REINFORCE / entropy seeds of the receiver's message and stop heads (App. A.4),
// dg = dlw W_w, dh += dgpre W_h + dls w_s (+ dA W_y1h at the sample's output step), GRU cell backward, dh_{t-1} = dh u +
// dgh W_hh -- the transposed products as [16, K] x [K, N] MFMA tiles ("NN" form: the PyTorch [out,in] matrix IS [K, N]).
// The sender's backward is not recurrent (its input is detached) and runs over all (step, sample) rows in k_send_bwd.
// Rows of steps a sample never took are zero-filled when zero_dead != 0 (no live-row list for k_wgrad), else untouched.
// ---------------------------------------------------------------------------------------------
struct BwdLds { int dh, dlw, dgp, dgh, dAm, dA, A, hs, dy, ws, w2, coef, raw0, raw1, misc, total;
                int whh, wh, wy1, ww;          // LDS copies of W_hh / w_h / y1[:, :R] / W_w in [K][N + 4] layout (-1: streamed from L2)
                int ldW, ldR, ld3R, ldD; };
__host__ __device__ inline BwdLds bwd_tile_lds(const Dims& d, int nw) {
    BwdLds L;
    L.ldW = ld16(d.W); L.ldR = ld16(d.R); L.ld3R = ld16(3 * d.R); L.ldD = ((d.D + 31) & ~31) + 4;      // (dA streams classes 32 at a time)
    int o = 0;
    auto take = [&](int n) { const int at = o; o += (n + 3) & ~3; return at; };
    L.dh = take(MMG_TM * L.ldR); L.dlw = -1; L.dgp = -1; L.dgh = take(MMG_TM * L.ld3R);
    L.dAm = take(MMG_TM * L.ldR); L.dA = take(MMG_TM * L.ldR); L.A = take(MMG_TM * L.ldR); L.hs = take(MMG_TM * L.ldR);
    L.ws = take(L.ldR); L.w2 = take(L.ldR); L.coef = -1;
    const int r = tile_raw_floats_nn(d.R, nw) > tile_raw_floats(d.R, nw) ? tile_raw_floats_nn(d.R, nw) : tile_raw_floats(d.R, nw);
    L.raw0 = take(r); L.raw1 = take(r);
    L.misc = take(128);
    // the dy tile (output-step prelude only) shares its space with the [K][N + 4] copy of W_hh the time loop multiplies by
    const int stride = d.R + 4, budget = 160 * 256 - 64;
    const int dysz = MMG_TM * L.ldD, hhsz = 3 * d.R * stride;
    L.whh = -1; L.wh = -1; L.wy1 = -1; L.ww = -1;
    const bool hh = o + (dysz > hhsz ? dysz : hhsz) <= budget;
    L.dy = take(hh ? (dysz > hhsz ? dysz : hhsz) : dysz);
    if (hh) L.whh = L.dy;
    L.total = o;
    if (L.total > budget && !((3 * d.R) & 15) && 3 * MMG_TM * L.ldR >= MMG_TM * L.ld3R) {
        // wide receivers (R = 256: 204 KB): the gate-gradient tile dgh lives in the reverse-time loop only, the dA | A | h* tiles
        // (contiguous) in the output-step prelude only -- they share their space.  Every element of dgh's K = 3R columns is
        // rewritten each step before the product reads it, and 3R is a multiple of 16 (no K padding to keep zero).
        BwdLds M = L;
        int o2 = 0;
        auto take2 = [&](int n) { const int at = o2; o2 += (n + 3) & ~3; return at; };
        M.dh = take2(MMG_TM * L.ldR);
        M.dAm = take2(MMG_TM * L.ldR); M.dA = take2(MMG_TM * L.ldR); M.A = take2(MMG_TM * L.ldR); M.hs = take2(MMG_TM * L.ldR);
        M.dgh = M.dA;
        M.ws = take2(L.ldR); M.w2 = take2(L.ldR);
        M.raw0 = take2(r); M.raw1 = take2(r);
        M.misc = take2(128);
        M.dy = take2(dysz);
        M.whh = -1;
        M.total = o2;
        return M;
    }
    return L;
}
#define BL_TSTAR 0      // [0,16) t* of the tile's samples   [16,32) reward L   [32,48) dls of this step   [48,64) live this step
#define BL_L 16
#define BL_DLS 32
#define BL_LIVE 48

// URP: forward-tape values a thread prefetches per step: 16 * R <= URP * NT (host picks the smallest instantiation)
template <int NT, int URP>
__global__ __launch_bounds__(NT) void k_bwd_tile(Dims dm, Params P, Tape tp, const int64_t* __restrict__ target, int zero_dead, int make_map) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    MMG_RSTAMP(blockIdx.x == 0, 224);
    constexpr int nw = NT / 64;
    const int B = dm.B, W = dm.W, R = dm.R, V = dm.V, D = dm.D, T = dm.T;
    const int b0 = blockIdx.x * MMG_TM, nb = min(MMG_TM, B - b0);
    const bool binary = dm.use_binary != 0;
    const BwdLds L = bwd_tile_lds(dm, nw);
    float* s_dh = smem + L.dh; float* s_dgh = smem + L.dgh;
    float* s_dAm = smem + L.dAm; float* s_dA = smem + L.dA; float* s_A = smem + L.A; float* s_hs = smem + L.hs; float* s_dy = smem + L.dy;
    float* s_ws = smem + L.ws; float* s_w2 = smem + L.w2; float* raw0 = smem + L.raw0; float* raw1 = smem + L.raw1; float* misc = smem + L.misc;
    {
        const int tid = threadIdx.x;
        for (int i = tid; i < L.total; i += NT) smem[i] = 0.f;
        __syncthreads();
        if (make_map && blockIdx.x == 0 && tid < 64) build_row_map(dm, tp);       // live (step, sample) rows for k_wgrad / k_send_bwd
        if (tid < MMG_TM) {
            const int b = min(b0 + tid, B - 1);
            misc[BL_TSTAR + tid] = (tid < nb) ? (float)tp.tstar[b] : -1.f;          // padded rows: never live
            misc[BL_L + tid] = tp.logs[b];
        }
        for (int r = tid; r < R; r += NT) { s_ws[r] = P.p[R_S_W][r]; s_w2[r] = P.p[R_Y2_W][r]; }
    }
    __syncthreads();
    MMG_RSTAMP(blockIdx.x == 0, 225);
    int tmax = 0;
    for (int m = 0; m < nb; ++m) tmax = max(tmax, (int)misc[BL_TSTAR + m]);
    const int wave = threadIdx.x >> 6;

    // ---------------- output step (model.py:1264-1275): dy, h*, A*, dA
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        for (int m = wave; m < nb; m += nw) {                              // dNLL/d outp, wave per sample
            const int b = b0 + m, tgt = (int)target[b];
            float dsum = 0.f;
            for (int d0 = lane; d0 < D; d0 += 64 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = tp.sm[(size_t)b * D + min(d0 + 64 * u, D - 1)];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int d = d0 + 64 * u;
                    if (d < D) {
                        const float dv = (v[u] - (d == tgt ? 1.f : 0.f)) / (float)dm.Bg;
                        s_dy[m * L.ldD + d] = dv; tp.dy[(size_t)b * D + d] = dv; dsum += dv;
                    }
                }
            }
            dsum = dpp_wave_sum(dsum);
            if (lane == 0) tp.dysum[b] = dsum;
        }
        batched_for<NT, 4>(MMG_TM * R, [&](int idx) {
                const int m = idx / R, r = idx - m * R;
                const int ts = max((int)misc[BL_TSTAR + m], 0);
                return tp.h[((size_t)(ts + 1) * B + min(b0 + m, B - 1)) * R + r];
            }, [&](int idx, float v) {
                const int m = idx / R, r = idx - m * R;
                s_hs[m * L.ldR + r] = v;
                if (m < nb) tp.hstar[(size_t)(b0 + m) * R + r] = v;
            });
        __syncthreads();
        MMG_RSTAMP(blockIdx.x == 0, 226);
        for (int idx = tid; idx < MMG_TM * D; idx += NT) {                 // transposed copy: 16 samples of a class = one 64-byte line
            const int m = idx & 15, d = idx >> 4;
            if (m < nb) tp.dyT[(size_t)d * B + b0 + m] = s_dy[m * L.ldD + d];
        }
        tgemm_nt_raw(s_hs, L.ldR, P.p[R_Y1_W], R + V, R, R, raw0, wave, nw);
        __syncthreads();
        {
            const int kp = tile_kparts((R + 15) >> 4, nw);
            for (int idx = tid; idx < MMG_TM * R; idx += NT) {
                const int m = idx / R, r = idx - m * R;
                const float a = raw_sum(raw0, L.ldR, kp, m, r);
                s_A[m * L.ldR + r] = a;
                if (m < nb) tp.Astar[(size_t)(b0 + m) * R + r] = a;
            }
        }
        __syncthreads();
        MMG_RSTAMP(blockIdx.x == 0, 227);
        // dA[m][r] = w2[r] sum_d dy[m][d] 1[A*[m][r] + Cd[d][r] > 0].  Thread (r, class slice): lanes along r (class rows read
        // coalesced), the classes split over the NT / RL thread groups, all 16 samples of the tile in registers -- every class
        // row is read once per tile and 32 of them are in flight per thread; the slices are then added through LDS.
        {
            const int RL = (R < NT) ? R : NT, NSL = NT / RL;                // lanes along r, class slices
            const int sl = tid / RL, r1 = tid - sl * RL;
            const int per = (((D + NSL - 1) / NSL) + 3) & ~3, d_lo = min(D, sl * per), d_hi = min(D, d_lo + per);
            float* part = raw0;                                             // [NSL][16][RL] partials (raw0 | raw1 are contiguous)
            for (int rb = 0; rb < R; rb += RL) {
                const int r = rb + r1;
                // indicator 1[A + c > 0] as ONE instruction: clamp((c + A) * 2^100) to [0, 1] (a fused multiply-add with the
                // clamp output modifier) -- exactly 0 or 1 unless 0 < |A + c| < 2^-100, which a sum of two O(1) floats never is
                constexpr float BIG = 1.2676506e30f;                            // 2^100
                float acc[MMG_TM], av[MMG_TM];
#pragma unroll
                for (int m = 0; m < MMG_TM; ++m) { acc[m] = 0.f; av[m] = s_A[m * L.ldR + min(r, R - 1)] * BIG; }
                if (sl < NSL && r < R) {
                    for (int d0 = d_lo; d0 < d_hi; d0 += 32) {
                        float cv[32];
#pragma unroll
                        for (int u = 0; u < 32; ++u) cv[u] = tp.Cd[(size_t)min(d0 + u, d_hi - 1) * R + r];
                        // (beyond d_hi the clamped class row repeats; its dy factor is masked to zero)
#pragma unroll 2
                        for (int u = 0; u < 32; u += 4) {
#pragma unroll
                            for (int m = 0; m < MMG_TM; ++m) {
                                // (d_lo and the 32-class chunks are multiples of 4, rows of s_dy are 16-byte aligned and padded past D:
                                //  one 16-byte LDS read per four classes -- with scalar reads this loop is bound by LDS instructions)
                                const float4 dq = *reinterpret_cast<const float4*>(s_dy + m * L.ldD + d0 + u);
                                const float y0 = (d0 + u < d_hi) ? dq.x : 0.f, y1 = (d0 + u + 1 < d_hi) ? dq.y : 0.f;
                                const float y2 = (d0 + u + 2 < d_hi) ? dq.z : 0.f, y3 = (d0 + u + 3 < d_hi) ? dq.w : 0.f;
                                acc[m] = fmaf(__builtin_amdgcn_fmed3f(fmaf(cv[u], BIG, av[m]), 0.f, 1.f), y0, acc[m]);
                                acc[m] = fmaf(__builtin_amdgcn_fmed3f(fmaf(cv[u + 1], BIG, av[m]), 0.f, 1.f), y1, acc[m]);
                                acc[m] = fmaf(__builtin_amdgcn_fmed3f(fmaf(cv[u + 2], BIG, av[m]), 0.f, 1.f), y2, acc[m]);
                                acc[m] = fmaf(__builtin_amdgcn_fmed3f(fmaf(cv[u + 3], BIG, av[m]), 0.f, 1.f), y3, acc[m]);
                            }
                        }
                    }
                }
                // combine the class slices: NSL * 16 * RL floats through LDS in chunks that fit the two staging areas
                const int cap = (2 * (L.raw1 - L.raw0)) / (MMG_TM * RL);        // slices per pass
                for (int s0 = 0; s0 < NSL; s0 += cap) {
                    __syncthreads();
                    if (sl >= s0 && sl < s0 + cap && sl < NSL) {
#pragma unroll
                        for (int m = 0; m < MMG_TM; ++m) part[((sl - s0) * MMG_TM + m) * RL + r1] = acc[m];
                    }
                    __syncthreads();
                    for (int idx = tid; idx < MMG_TM * RL; idx += NT) {
                        const int m = idx / RL, rr = idx - m * RL;
                        if (rb + rr < R) {
                            float v = (s0 == 0) ? 0.f : s_dA[m * L.ldR + rb + rr];
                            for (int q = 0; q < min(cap, NSL - s0); ++q) v += part[(q * MMG_TM + m) * RL + rr];
                            s_dA[m * L.ldR + rb + rr] = v;
                        }
                    }
                }
                __syncthreads();
            }
            for (int idx = tid; idx < MMG_TM * R; idx += NT) {
                const int m = idx / R, r = idx - m * R;
                const float v = s_dA[m * L.ldR + r] * s_w2[r];
                s_dA[m * L.ldR + r] = v;
                if (m < nb) tp.dA[(size_t)(b0 + m) * R + r] = v;
            }
        }
        __syncthreads();
    }

    MMG_RSTAMP(blockIdx.x == 0, 228);
    // ---------------- dAy = dA W_y1h: enters dh once, at the sample's output step (one product per tile, not one per step)
    {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        tgemm_nn_raw(s_dA, L.ldR, P.p[R_Y1_W], R + V, R, R, raw0, wave, nw);
        __syncthreads();
        const int kp = tile_kparts((R + 63) >> 6, nw);
        for (int idx = tid; idx < MMG_TM * R; idx += NT) {
            const int m = idx / R, i = idx - m * R;
            s_dAm[m * L.ldR + i] = raw_sum(raw0, L.ldR, kp, m, i);
        }
        __syncthreads();
    }
    // ---------------- W_hh in [K][N + 4] layout in LDS (the dy tile is dead now): the loop's only product reads no weight from L2
    if (L.whh >= 0) {
        const int stride = R + 4, n4 = R >> 2;
        batched_for<NT, 8>(3 * R * n4, [&](int idx) { const int k = idx / n4, q = idx - k * n4; return *reinterpret_cast<const float4*>(P.p[R_WHH] + (size_t)k * R + 4 * q); },
                           [&](int idx, float4 v) { const int k = idx / n4, q = idx - k * n4; *reinterpret_cast<float4*>(smem + L.whh + k * stride + 4 * q) = v; });
        __syncthreads();
    }
    MMG_RSTAMP(blockIdx.x == 0, 229);
    // ---------------- reverse time.  Everything that does not depend on the carried dh was formed for all (step, sample)
    // rows by k_bwd_pre (dhin = dgpre W_h + dls w_s); a step is: [dh, GRU cell backward] barrier [dgh W_hh] barrier.
    bool have_prod = false;                                             // raw0 holds dgh_{t+1} W_hh
    for (int t = zero_dead ? T - 1 : tmax; t >= 0; --t) {
        const size_t rowb = (size_t)t * B;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        MMG_RSTAMP(blockIdx.x == 0 && t < 4, 232 + 4 * t);
        // ---- this step's forward tape (GRU gates, h_{t-1}) and dhin, ONE round trip
        float fru[URP][4], fh[URP], fin[URP];
#pragma unroll
        for (int u = 0; u < URP; ++u) {
            const int idx = min(tid + u * NT, MMG_TM * R - 1), m = idx / R, i = idx - m * R, b = min(b0 + m, B - 1);
            const float* gr = tp.gru + (rowb + b) * 4 * R;
            fru[u][0] = gr[i]; fru[u][1] = gr[R + i]; fru[u][2] = gr[2 * R + i]; fru[u][3] = gr[3 * R + i];
            fh[u] = tp.h[(rowb + b) * R + i];
            fin[u] = binary ? tp.dhin[(rowb + b) * R + i] : 0.f;
        }
        {
            // dh_t = dh_{t+1} u_{t+1} + dgh_{t+1} W_hh + dhin_t (+ dAy at the output step); GRU cell backward (model.py:340)
            const int kp = tile_kparts((R + 63) >> 6, nw);
#pragma unroll
            for (int u = 0; u < URP; ++u) {
                const int idx = tid + u * NT;
                if (idx < MMG_TM * R) {
                    const int m = idx / R, i = idx - m * R;
                    const float ts = misc[BL_TSTAR + m];
                    const bool live = (float)t <= ts;
                    const float rr = fru[u][0], uu = fru[u][1], nn = fru[u][2], ghn = fru[u][3];
                    float dh = s_dh[m * L.ldR + i] + (have_prod ? raw_sum(raw0, L.ldR, kp, m, i) : 0.f);
                    dh += ((float)t == ts) ? s_dAm[m * L.ldR + i] : 0.f;
                    dh += (binary && live) ? fin[u] : 0.f;              // (rows of steps a sample never took hold no dhin)
                    if (!live) dh = 0.f;
                    const float dn = dh * (1.f - uu), du = dh * (fh[u] - nn);
                    const float dnp = dn * (1.f - nn * nn), dup = du * uu * (1.f - uu);
                    const float drp = dnp * ghn * rr * (1.f - rr);
                    float* dg = s_dgh + m * L.ld3R;
                    dg[i] = live ? drp : 0.f; dg[R + i] = live ? dup : 0.f; dg[2 * R + i] = live ? dnp * rr : 0.f;
                    s_dh[m * L.ldR + i] = live ? dh * uu : 0.f;
                    if (m < nb && (live || zero_dead)) {
                        float* gi = tp.dgi + (rowb + b0 + m) * 3 * R; float* gh = tp.dgh + (rowb + b0 + m) * 3 * R;
                        gi[i] = live ? drp : 0.f; gi[R + i] = live ? dup : 0.f; gi[2 * R + i] = live ? dnp : 0.f;
                        gh[i] = live ? drp : 0.f; gh[R + i] = live ? dup : 0.f; gh[2 * R + i] = live ? dnp * rr : 0.f;
                    }
                }
            }
        }
        __syncthreads();
        MMG_RSTAMP(blockIdx.x == 0 && t < 4, 232 + 4 * t + 1);
        if (t > 0) {                                                    // dh_{t-1} receives dgh W_hh
            if (L.whh >= 0) tgemm_nn_raw(s_dgh, L.ld3R, smem + L.whh, R + 4, R, 3 * R, raw0, wave, nw);
            else tgemm_nn_raw(s_dgh, L.ld3R, P.p[R_WHH], R, R, 3 * R, raw0, wave, nw);
            have_prod = true;
            __syncthreads();
        }
    }
    MMG_RSTAMP(blockIdx.x == 0, 230);
}

// ---------------------------------------------------------------------------------------------
// k_bwd_pre (binary mode): the part of the receiver's BPTT that does NOT depend on the carried dh, for all (step, sample)
// rows at once instead of inside the reverse-time loop of k_bwd_tile -- REINFORCE / entropy seeds of the stop bit and the
// receiver's message (App. A.4), MSE seeds of the baselines (model.py:971-988), dgpre = (dlw W_w)(1 - g^2) and
//   dhin = dgpre W_h + dls w_s        (what step t adds to dh besides the recurrence and the output-step term)
// grid = T x ceil(B/16): a workgroup owns 16 samples of one step.  Rows of steps a sample never took: skipped (live-row
// contract) or zero-filled (zero_dead).
// ---------------------------------------------------------------------------------------------
// staging area of bwd_pre_body's two products: [k-parts][16][ld16(N)] with N = R, or -- column bands (pre_bands = 4 at R = 256:
// a 64-column band of dgpre per workgroup) -- N = 64 in four k-parts
__host__ __device__ inline int bwd_pre_raw_floats(int R, int nw) {
    const int a = tile_raw_floats_nn(R, nw), b = tile_raw_floats_nn(64, nw);
    return a > b ? a : b;
}
__host__ __device__ inline int bwd_pre_lds_floats(const Dims& d) {
    return MMG_TM * ld16(d.W) + MMG_TM * ld16(d.R) + bwd_pre_raw_floats(d.R, MMG_BLOCK / 64) + 7 * 64 + 64 + ld16(d.R);
}
// weight fragment of an "NN" product with N <= 64 output columns, loaded ahead of its use: wave w owns k-part w (all four
// n-tiles), lane (i, q) holds float4 Bm[16 kg + 4 q + c][4 i ..] for its <= MAXKG k-groups (tgemm_nn_body's layout)
template <int MAXKG>
struct NFrag { float4 b[MAXKG][4]; int kp, g0, n; };
template <int MAXKG>
__device__ __forceinline__ bool nfrag_fits(int N, int K, int nw) {
    const int kgroups = (K + 15) >> 4, kparts = tile_kparts(1, nw);
    return N <= 64 && kparts == nw && (kgroups + kparts - 1) / kparts <= MAXKG;
}
template <int MAXKG>
__device__ __forceinline__ void nfrag_load(NFrag<MAXKG>& f, const float* __restrict__ Bm, int ldb, int N, int K, int wave, int nw) {
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const int kgroups = (K + 15) >> 4, kparts = tile_kparts(1, nw), per = (kgroups + kparts - 1) / kparts;
    f.kp = wave; f.g0 = wave * per; f.n = max(0, min(kgroups, f.g0 + per) - f.g0);
#pragma unroll
    for (int u = 0; u < MAXKG; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            f.b[u][c] = ldrow4c<true>(Bm + (size_t)min(min(f.g0 + u, kgroups - 1) * 16 + q * 4 + c, K - 1) * ldb, 4 * i, N);
}
template <int MAXKG>
__device__ __forceinline__ void nfrag_mma(const NFrag<MAXKG>& f, const float* A, int lda, float* raw, int ldr) {
    const int lane = threadIdx.x & 63, i = lane & 15, q = lane >> 4;
    const float* arow = A + i * lda + q * 4;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < MAXKG; ++u) {
        if (u < f.n) {
            const float4 a = *reinterpret_cast<const float4*>(arow + (f.g0 + u) * 16);
            acc[0] = mfma16(a.x, f.b[u][0].x, acc[0]); acc[1] = mfma16(a.x, f.b[u][0].y, acc[1]); acc[2] = mfma16(a.x, f.b[u][0].z, acc[2]); acc[3] = mfma16(a.x, f.b[u][0].w, acc[3]);
            acc[0] = mfma16(a.y, f.b[u][1].x, acc[0]); acc[1] = mfma16(a.y, f.b[u][1].y, acc[1]); acc[2] = mfma16(a.y, f.b[u][1].z, acc[2]); acc[3] = mfma16(a.y, f.b[u][1].w, acc[3]);
            acc[0] = mfma16(a.z, f.b[u][2].x, acc[0]); acc[1] = mfma16(a.z, f.b[u][2].y, acc[1]); acc[2] = mfma16(a.z, f.b[u][2].z, acc[2]); acc[3] = mfma16(a.z, f.b[u][2].w, acc[3]);
            acc[0] = mfma16(a.w, f.b[u][3].x, acc[0]); acc[1] = mfma16(a.w, f.b[u][3].y, acc[1]); acc[2] = mfma16(a.w, f.b[u][3].z, acc[2]); acc[3] = mfma16(a.w, f.b[u][3].w, acc[3]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float* dst = raw + ((f.kp * MMG_TM) + q * 4 + r) * ldr + 4 * i;
        if (4 * i + 3 < ldr) *reinterpret_cast<float4*>(dst) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    }
}

// UR: g values a thread holds per step: 16 * R <= UR * NT (host: 8 up to R = 128, 16 up to R = 256)
// pre_bands > 1 (wide receivers, R = 256: each product streams 256 KB per workgroup otherwise): the workgroup of (step, tile) is
// split over column bands of dgpre -- band c forms dgpre[:, band] = (dlw W_w[:, band])(1 - g^2) and the PARTIAL product
// dgpre[:, band] W_h[band, :] into tape.dhin[band] (band 0 adds dls w_s); k_rc_bwd adds the partials
template <int UR>
__device__ __forceinline__ void bwd_pre_body(const Dims& dm, const Params& P, const Tape& tp, const int zero_dead, const int blk_in, const int pre_bands = 1) {
    const int band = blk_in % pre_bands, blk = blk_in / pre_bands;
    const int Rb = dm.R / pre_bands, c0 = band * Rb;
    if (blk_in == 0 && threadIdx.x < 64) tp.rcflags[(size_t)threadIdx.x * 64] = 0u;       // hand-off counters of k_rc_bwd's roles (the launch after this one)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = MMG_BLOCK, nw = NT / 64;
    const int B = dm.B, W = dm.W, R = dm.R, T = dm.T;
    const int ldW = ld16(W), ldR = ld16(R);
    const int tiles = (B + MMG_TM - 1) / MMG_TM;
    const int t = blk / tiles, b0 = (blk - t * tiles) * MMG_TM, nb = min(MMG_TM, B - b0);
    float* s_dlw = smem; float* s_dgp = s_dlw + MMG_TM * ldW; float* raw = s_dgp + MMG_TM * ldR;
    float* s_coef = raw + bwd_pre_raw_floats(R, nw); float* misc = s_coef + 7 * 64; float* s_ws = misc + 64;
    // misc: [0,16) t*   [16,32) reward L   [32,48) baseline_rec score of the row   [48,64) dls
    LossCoef lc; lc.cw = s_coef; lc.ce = s_coef + 3 * T; lc.cb = s_coef + 6 * T;
    const int tid = threadIdx.x, wave = tid >> 6;
    const size_t rowb = (size_t)t * B;
    // ---- every global load of the workgroup goes out NOW, in one round trip: statistics, row scalars, the message / g
    // tiles of the step and (where they fit: N <= 64, <= 4 k-groups per wave) both weight fragments
    const CoefRegs creg = coef_load(dm, tp.stats);
    const int bm = min(b0 + min(tid, MMG_TM - 1), B - 1);
    const float r_ts = (float)tp.tstar[bm], r_L = tp.logs[bm];
    const float r_br = tp.br[rowb + bm], r_bs = tp.bs[rowb + bm];
    const float r_s = dm.fixed ? 0.f : tp.s[rowb + bm], r_ps = dm.fixed ? 0.5f : tp.ps[rowb + bm];
    constexpr int UW = 16;                                              // 16 * W <= UW * NT (W <= 256); 16 * R <= UR * NT
    F2 rw[UW]; float rg[UR];
#pragma unroll
    for (int u = 0; u < UW; ++u) {
        const int idx = min(tid + u * NT, MMG_TM * W - 1), m = idx / W, j = idx - m * W;
        const size_t o = (rowb + min(b0 + m, B - 1)) * W + j;
        rw[u] = F2{tp.w[o], tp.pw[o]};
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
        const int idx = min(tid + u * NT, MMG_TM * R - 1), m = idx / R, r = idx - m * R;
        rg[u] = tp.g[(rowb + min(b0 + m, B - 1)) * R + r];
    }
    const bool fr1 = nfrag_fits<4>(R, W, nw), fr2 = nfrag_fits<2>(R, R, nw);
    NFrag<4> fw; NFrag<2> fh;
    fw.n = fh.n = 0;
    if (fr1) nfrag_load<4>(fw, P.p[R_W_W], R, R, W, wave, nw);
    if (fr2) nfrag_load<2>(fh, P.p[R_WH_W], R, R, R, wave, nw);
    const float r_ws = P.p[R_S_W][min(tid, R - 1)];

    for (int i = tid; i < MMG_TM * (ldW + ldR); i += NT) smem[i] = 0.f;
    if (tid < MMG_TM) { misc[tid] = (tid < nb) ? r_ts : -1.f; misc[16 + tid] = r_L; }
    for (int r = tid; r < ldR; r += NT) s_ws[r] = 0.f;
    coef_compute(dm, creg, lc);                                             // (ends with a barrier)
    if (tid < R) s_ws[tid] = r_ws;
    bool any = false;
    for (int m = 0; m < nb; ++m) any = any || ((float)t <= misc[m]);
    if (!any && !zero_dead) return;
    if (tid < MMG_TM) {
        const int m = tid;
        const bool live = (float)t <= misc[m];
        float dls = 0.f, dbs = 0.f, dbr = 0.f;
        if (live) {
            const float Lr = misc[16 + m];
            if (!dm.fixed) dls = bit_seed_fast(r_s, r_ps, (Lr - r_br) * lc.cw[t], lc.ce[t]);
            dbs = lc.cb[t] * (r_bs - Lr); dbr = lc.cb[t] * (r_br - Lr);          // MSE seeds, model.py:971-988
        }
        misc[32 + m] = r_br; misc[48 + m] = dls;
        if (band == 0 && m < nb && (live || zero_dead)) { tp.dls[rowb + b0 + m] = dls; tp.dbs[rowb + b0 + m] = dbs; tp.dbr[rowb + b0 + m] = dbr; }
    }
    __syncthreads();
    // seeds of the receiver-message stream (active while m_{t+1} == 1, i.e. t < t*)
#pragma unroll
    for (int u = 0; u < UW; ++u) {
        const int idx = tid + u * NT;
        if (idx < MMG_TM * W) {
            const int m = idx / W, j = idx - m * W;
            const bool act = (float)t < misc[m];
            const float sv = act ? bit_seed_fast(rw[u].x, rw[u].y, (misc[16 + m] - misc[32 + m]) * lc.cw[T + t], lc.ce[T + t]) : 0.f;
            s_dlw[m * ldW + j] = sv;
            if (band == 0 && m < nb && (zero_dead || (float)t <= misc[m])) tp.dlw[(rowb + b0 + m) * W + j] = sv;
        }
    }
    __syncthreads();
    if (pre_bands > 1) {
        tgemm_nn_raw(s_dlw, ldW, P.p[R_W_W] + c0, R, Rb, W, raw, wave, nw);   // dg[:, band] = dlw W_w[:, band]
        __syncthreads();
        const int kpb = tile_kparts((Rb + 63) >> 6, nw), ldb = ld16(Rb);
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const int idx = tid + u * NT;
            if (idx < MMG_TM * R) {
                const int m = idx / R, r = idx - m * R;
                if (r >= c0 && r < c0 + Rb) {
                    const bool act = (float)t < misc[m];
                    const float v = act ? raw_sum(raw, ldb, kpb, m, r - c0) * (1.f - rg[u] * rg[u]) : 0.f;
                    s_dgp[m * ldR + r] = v;
                    if (m < nb && (zero_dead || (float)t <= misc[m])) tp.dgpre[(rowb + b0 + m) * R + r] = v;
                }
            }
        }
        __syncthreads();
        tgemm_nn_raw(s_dgp + c0, ldR, P.p[R_WH_W] + (size_t)c0 * R, R, R, Rb, raw, wave, nw);      // dgpre[:, band] W_h[band, :]
        __syncthreads();
        const int kp2 = tile_kparts((R + 63) >> 6, nw);
        float* dhin = tp.dhin + (size_t)band * T * B * R;
        for (int idx = tid; idx < MMG_TM * R; idx += NT) {
            const int m = idx / R, i = idx - m * R;
            if (m < nb && (float)t <= misc[m]) dhin[(rowb + b0 + m) * R + i] = raw_sum(raw, ldR, kp2, m, i) + (band == 0 ? misc[48 + m] * s_ws[i] : 0.f);
        }
        return;
    }
    if (fr1) nfrag_mma<4>(fw, s_dlw, ldW, raw, ldR);                          // dg = dlw W_w
    else tgemm_nn_raw(s_dlw, ldW, P.p[R_W_W], R, R, W, raw, wave, nw);
    __syncthreads();
    const int kp = tile_kparts((R + 63) >> 6, nw);
#pragma unroll
    for (int u = 0; u < UR; ++u) {
        const int idx = tid + u * NT;
        if (idx < MMG_TM * R) {
            const int m = idx / R, r = idx - m * R;
            const bool act = (float)t < misc[m];
            const float v = act ? raw_sum(raw, ldR, kp, m, r) * (1.f - rg[u] * rg[u]) : 0.f;
            s_dgp[m * ldR + r] = v;
            if (m < nb && (zero_dead || (float)t <= misc[m])) tp.dgpre[(rowb + b0 + m) * R + r] = v;
        }
    }
    __syncthreads();
    if (fr2) nfrag_mma<2>(fh, s_dgp, ldR, raw, ldR);                          // dgpre W_h
    else tgemm_nn_raw(s_dgp, ldR, P.p[R_WH_W], R, R, R, raw, wave, nw);
    __syncthreads();
    for (int idx = tid; idx < MMG_TM * R; idx += NT) {
        const int m = idx / R, i = idx - m * R;
        if (m < nb && (float)t <= misc[m]) tp.dhin[(rowb + b0 + m) * R + i] = raw_sum(raw, ldR, kp, m, i) + misc[48 + m] * s_ws[i];
    }
}

template <int UR>
__global__ __launch_bounds__(MMG_BLOCK) void k_bwd_pre(Dims dm, Params P, Tape tp, int zero_dead) {
    bwd_pre_body<UR>(dm, P, tp, zero_dead, (int)blockIdx.x);
}
__device__ __forceinline__ void send_bwd_body(const Dims& dm, const Params& P, const Tape& tp, const int* __restrict__ rmap, const int bx, const int by);
// k_bwd_pre and the sender's backward (k_send_bwd over all T * B rows, dead row blocks return) in ONE launch: neither depends
// on the other, both are "statistics -> seeds -> one or two products" latency chains of ~20 us -- side by side instead of
// one after the other.  Blocks [0, npre): bwd_pre_body; then nbands blocks per 16-row block: send_bwd_body.
template <int UR>
__global__ __launch_bounds__(MMG_BLOCK) void k_bwd_pre_send(Dims dm, Params P, Tape tp, int zero_dead, int npre, int nbands, int pre_bands) {
    const int blk = blockIdx.x;
    if (blk < npre) { bwd_pre_body<UR>(dm, P, tp, zero_dead, blk, pre_bands); return; }
    const int sb = blk - npre;
    send_bwd_body(dm, P, tp, nullptr, sb / nbands, sb % nbands);
}

// ---------------------------------------------------------------------------------------------
// k_send_bwd: sender backward over (step, sample) rows (binary mode): dlz = REINFORCE/entropy seed of the sender's bits,
// dpre = (dlz W_b) (1 - a^2).  grid (ceil(rows/16), ceil(H/64)): a workgroup owns 16 rows and 64 columns of H; rows come from the live-row list (rmap) or are all T*B rows (dead rows zero-filled).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void send_bwd_body(const Dims& dm, const Params& P, const Tape& tp, const int* __restrict__ rmap, const int bx, const int by) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = MMG_BLOCK, nw = NT / 64;
    const int B = dm.B, H = dm.H, W = dm.W, T = dm.T;
    const int ldW = ld16(W);
    float* s_dlz = smem;                                   // [16][ldW]
    float* s_coef = smem + MMG_TM * ldW;                   // 7*64
    int* s_row = reinterpret_cast<int*>(s_coef + 7 * 64);  // [16] tape row of each tile row, -1: none
    float* raw = s_coef + 7 * 64 + 16;                     // [16][ld16(64 * bands)]
    LossCoef lc; lc.cw = s_coef; lc.ce = s_coef + 3 * T; lc.cb = s_coef + 6 * T;
    const int tid = threadIdx.x, wave = tid >> 6;
    MMG_RSTAMP(bx == 0 && by == 0, 240);
    const int nrows = T * B;                               // (the live-row list ends with -1 entries: no dependent load of its length)
    const int r0 = bx * MMG_TM;
    if (r0 >= nrows) return;
    if (rmap && rmap[r0] < 0) return;
    if (!rmap) {                                           // all T * B rows: a block without a live row has nothing to do
        const int row = min(r0 + (int)(threadIdx.x & 15), nrows - 1);
        if (!__syncthreads_or((row / B) <= tp.tstar[row % B])) return;
    }
    // ---- every global load of the workgroup in ONE round trip (after the row list): statistics, this band's weight
    // fragment, the rows' message bits / probabilities, their scalars and their slice of the hidden tile
    const int n0 = by * 64, Nb = min(64, H - n0);
    const CoefRegs creg = coef_load(dm, tp.stats);
    const bool fr = nfrag_fits<4>(Nb, W, nw);
    NFrag<4> fb;
    fb.n = 0;
    if (fr) nfrag_load<4>(fb, P.p[S_BIN_W] + n0, H, Nb, W, wave, nw);
    // tape row of tile row m (every thread derives the rows it touches itself: no LDS hop before the loads go out)
    auto row_of = [&](int m) { return (r0 + m < nrows) ? (rmap ? rmap[r0 + m] : r0 + m) : -1; };
    constexpr int UW = 16, UA = 4;                                      // 16 * W <= UW * NT (W <= 256), 16 * 64 = UA * NT
    F2 rz[UW]; float ra[UA];
    int rrow[UW];
#pragma unroll
    for (int u = 0; u < UW; ++u) {
        const int idx = min(tid + u * NT, MMG_TM * W - 1), m = idx / W, j = idx - m * W;
        rrow[u] = row_of(m);
        const size_t o = (size_t)max(rrow[u], 0) * W + j;
        rz[u] = F2{tp.z[o], tp.pz[o]};
    }
    float rlg[UW], rbs[UW]; int rts[UW];
#pragma unroll
    for (int u = 0; u < UW; ++u) {
        const int row = max(rrow[u], 0), b = row % B;
        rts[u] = rmap ? T : tp.tstar[b]; rlg[u] = tp.logs[b]; rbs[u] = tp.bs[row];      // (listed rows are live)
    }
    int arow[UA];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int idx = min(tid + u * NT, MMG_TM * Nb - 1), m = idx / Nb, n = idx - m * Nb;
        arow[u] = row_of(m);
        ra[u] = tp.a[(size_t)max(arow[u], 0) * H + n0 + n];
    }
    MMG_RSTAMP(bx == 0 && by == 0, 241);
    for (int i = tid; i < MMG_TM * ldW; i += NT) s_dlz[i] = 0.f;
    coef_compute(dm, creg, lc);                                             // (ends with a barrier)
    MMG_RSTAMP(bx == 0 && by == 0, 242);
#pragma unroll
    for (int u = 0; u < UW; ++u) {
        const int idx = tid + u * NT;
        if (idx < MMG_TM * W && rrow[u] >= 0) {
            const int m = idx / W, j = idx - m * W, row = rrow[u], t = row / B;
            float sv = 0.f;
            if (t <= rts[u]) sv = bit_seed_fast(rz[u].x, rz[u].y, (rlg[u] - rbs[u]) * lc.cw[2 * T + t], lc.ce[2 * T + t]);
            s_dlz[m * ldW + j] = sv;
            if (by == 0) tp.dlz[(size_t)row * W + j] = sv;
        }
    }
    __syncthreads();
    MMG_RSTAMP(bx == 0 && by == 0, 243);
    // this workgroup's 64 columns of H; its four waves split K
    if (fr) nfrag_mma<4>(fb, s_dlz, ldW, raw, ld16(Nb));
    else tgemm_nn_raw(s_dlz, ldW, P.p[S_BIN_W] + n0, H, Nb, W, raw, wave, nw);
    __syncthreads();
    MMG_RSTAMP(bx == 0 && by == 0, 244);
    const int ldr = ld16(Nb), kp = tile_kparts((Nb + 63) >> 6, nw);
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int idx = tid + u * NT;
        if (idx < MMG_TM * Nb && arow[u] >= 0) {
            const int m = idx / Nb, n = idx - m * Nb;
            tp.dpre[(size_t)arow[u] * H + n0 + n] = raw_sum(raw, ldr, kp, m, n) * (1.f - ra[u] * ra[u]);
        }
    }
    MMG_RSTAMP(bx == 0 && by == 0, 245);
}

__global__ __launch_bounds__(MMG_BLOCK) void k_send_bwd(Dims dm, Params P, Tape tp, const int* __restrict__ rmap, const int* __restrict__ rcount) {
    (void)rcount;
    send_bwd_body(dm, P, tp, rmap, (int)blockIdx.x, (int)blockIdx.y);
}

}  // namespace mmg

namespace mmg {
// dhx[b, :] = sum over the sample's live steps of dpre[t, b, :]  (image_layer's gradient reduces over B rows instead of
// over all live (step, sample) rows).  One float4 per thread, all of the sample's steps in flight.
// Trailing blocks: u0[h] = sum_b dpre[t = 0, b, h] -- code_bias only sees step 0, where the code input sigmoid(code_bias) is
// the same for every sample, so its gradient is dsig * W_c^T u0: a row-weighted column sum over code_layer.weight (k_wgrad).
__device__ __forceinline__ void dhx_body(const Dims& dm, const Tape& tp, const int nblk_dhx, const int blk) {
    const int H4 = dm.H >> 2;
    if (blk >= nblk_dhx) {
        __shared__ float4 s_p[4][64];
        const int h4 = (blk - nblk_dhx) * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
        const int hc = min(h4, H4 - 1);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b0 = part; b0 < dm.B; b0 += 4 * 16) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = reinterpret_cast<const float4*>(tp.dpre + (size_t)min(b0 + 4 * u, dm.B - 1) * dm.H)[hc];
#pragma unroll
            for (int u = 0; u < 16; ++u) if (b0 + 4 * u < dm.B) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        s_p[part][threadIdx.x & 63] = acc;
        __syncthreads();
        if (part == 0 && h4 < H4) {
            const float4 a = s_p[0][threadIdx.x], b = s_p[1][threadIdx.x], c = s_p[2][threadIdx.x], d = s_p[3][threadIdx.x];
            reinterpret_cast<float4*>(tp.u0)[h4] = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
        }
        return;
    }
    const int idx = blk * MMG_BLOCK + threadIdx.x;
    if (idx >= dm.B * H4) return;
    const int b = idx / H4, h4 = idx - b * H4;
    const int ts = dm.use_binary ? tp.tstar[b] : -1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 <= ts; t0 += 16) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = reinterpret_cast<const float4*>(tp.dpre + ((size_t)min(t0 + u, ts) * dm.B + b) * dm.H)[h4];
#pragma unroll
        for (int u = 0; u < 16; ++u) if (t0 + u <= ts) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    reinterpret_cast<float4*>(tp.dhx + (size_t)b * dm.H)[h4] = acc;
}
__global__ __launch_bounds__(MMG_BLOCK) void k_dhx(Dims dm, Tape tp, int nblk_dhx) { dhx_body(dm, tp, nblk_dhx, (int)blockIdx.x); }
}  // namespace mmg

namespace mmg {
// ---------------------------------------------------------------------------------------------
// k_dC_tile: class-side reduction of the y head for many samples / classes (same outputs as k_dC, kernels_bwd.h):
//   dC[d, r]  = w_y2[r] sum_b dy[b, d] 1[A*[b, r] + Cd[d, r] > 0]      Py2[d, r] = sum_b dy[b, d] relu(A*[b, r] + Cd[d, r])
// grid (ceil(D / CPB), NSB): a workgroup owns CPB classes x all R units and one slice of the samples; lanes run along r
// (A* rows coalesced), dy comes from its transposed copy dyT[d, b] (one 64-byte line per 16 samples, written by k_bwd_tile),
// 32 samples in flight per thread.  NSB > 1: partial sums to part[slice], combined in fixed order by the last-launched
// pass (blockIdx.y == 0 of a second launch with `combine` set) -- deterministic, no float atomics.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MMG_BLOCK) void k_dC_tile(Dims dm, Params P, Tape tp, int nsb, int combine) {
    const int R = dm.R, B = dm.B, D = dm.D;
    const int RL = R < MMG_BLOCK ? R : MMG_BLOCK;                 // lanes along r
    const int CPB = MMG_BLOCK / RL;                                // classes per workgroup
    const int c = threadIdx.x / RL, rl = threadIdx.x - c * RL;
    const int d = blockIdx.x * CPB + c;
    if (c >= CPB || d >= D) return;
    if (combine) {                                                 // sum the sample slices in order
        for (int r = rl; r < R; r += RL) {
            float a = 0.f, p = 0.f;
            for (int s = 0; s < nsb; ++s) { a += tp.dCpart[((size_t)s * 2 * D + d) * R + r]; p += tp.dCpart[((size_t)s * 2 * D + D + d) * R + r]; }
            tp.dC[(size_t)d * R + r] = a * P.p[R_Y2_W][r];
            tp.Py2[(size_t)d * R + r] = p;
        }
        return;
    }
    const int slice = blockIdx.y, per = (B + nsb - 1) / nsb;
    const int bb0 = slice * per, bb1 = min(B, bb0 + per);
    const float* dyt = tp.dyT + (size_t)d * B;
    for (int r = rl; r < R; r += RL) {
        const float cv = tp.Cd[(size_t)d * R + r];
        float dc0 = 0.f, dc1 = 0.f, py0 = 0.f, py1 = 0.f;
        for (int b0 = bb0; b0 < bb1; b0 += 32) {
            float yv[32], av[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int b = min(b0 + u, bb1 - 1);
                yv[u] = dyt[b]; av[u] = tp.Astar[(size_t)b * R + r];
            }
#pragma unroll
            for (int u = 0; u < 32; u += 2) {
                const float p0 = av[u] + cv, p1 = av[u + 1] + cv;
                if (b0 + u < bb1 && p0 > 0.f) { dc0 += yv[u]; py0 = fmaf(yv[u], p0, py0); }
                if (b0 + u + 1 < bb1 && p1 > 0.f) { dc1 += yv[u + 1]; py1 = fmaf(yv[u + 1], p1, py1); }
            }
        }
        if (nsb == 1) {
            tp.dC[(size_t)d * R + r] = (dc0 + dc1) * P.p[R_Y2_W][r];
            tp.Py2[(size_t)d * R + r] = py0 + py1;
        } else {
            tp.dCpart[((size_t)slice * 2 * D + d) * R + r] = dc0 + dc1;
            tp.dCpart[((size_t)slice * 2 * D + D + d) * R + r] = py0 + py1;
        }
    }
}
}  // namespace mmg

namespace mmg {
// ---------------------------------------------------------------------------------------------
// k_bwd_sample: the receiver's reverse-time pass of ONE sample per workgroup for the receiver shape of the register-resident
// kernels (R = 64, V = 100, D <= 32) with ANY message / sender width -- the sample-tile path's counterpart of rs_role.
// Everything that depends on the message width was formed for all (step, sample) rows by k_bwd_pre (dhin) and the sender's
// backward runs in k_send_bwd, so what is left is k_bwd_conv_fast's output step (dy, A*, dA) and its two-phase recurrence
// with W_hh^T in registers: ~0.65 us per step instead of ~3 us per step of a 16-sample tile (k_bwd_tile), on B CUs instead
// of B / 16.  Same tape contract as k_bwd_tile (zero_dead / live rows; block 0 builds the live-row list).
// ---------------------------------------------------------------------------------------------
template <int R, int V, int D>
__global__ __launch_bounds__(256, 1) void k_bwd_sample(Dims dm, Params P, Tape tp, const int64_t* __restrict__ target, int zero_dead, int make_map,
                                                       int nblk_dhx) {
    constexpr int NT = 256, K4 = NT / R, TMAX = 16;
    static_assert(R == 64 && D <= 32 && K4 == 4, "receiver shape of the register-resident kernels");
    __shared__ __attribute__((aligned(16))) float s_dh[R], s_dgh[3 * R], s_dy[32], s_A[R], s_dA[R], s_dAy[R];
    __shared__ __attribute__((aligned(16))) float s_dhin[TMAX * R], t_gru[TMAX * 4 * R], t_h[(TMAX + 1) * R];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int B = dm.B, T = dm.T, Dr = dm.D;
    const bool binary = dm.use_binary != 0;
    if (b > B) {                                                             // trailing workgroups: dhx = sum_t dpre, u0 (k_dhx) -- dpre is complete
        dhx_body(dm, tp, nblk_dhx, b - B - 1);                               // (the sender's backward ran in the launch before this one)
        return;
    }
    if (b == B) {                                                            // extra workgroup: live (step, sample) rows for k_wgrad / k_send_bwd
        if (make_map && tid < 64) build_row_map(dm, tp);
        return;
    }
    const int tstar = tp.tstar[b];
    const int tgt = (int)target[b];
    const int k4 = tid / K4, p4 = tid % K4;
    float whhT[3 * R / K4], y1T[R / K4], y1r[R / K4];
#pragma unroll
    for (int i = 0; i < 3 * R / K4; ++i) whhT[i] = P.p[R_WHH][(size_t)(p4 * (3 * R / K4) + i) * R + k4];
#pragma unroll
    for (int i = 0; i < R / K4; ++i) y1T[i] = P.p[R_Y1_W][(size_t)(p4 * (R / K4) + i) * (R + V) + k4];
#pragma unroll
    for (int i = 0; i < R / K4; ++i) y1r[i] = P.p[R_Y1_W][(size_t)k4 * (R + V) + p4 * (R / K4) + i];
    const float sm_mine = (tid < Dr) ? tp.sm[(size_t)b * Dr + tid] : 0.f;
    const float w2_mine = (tid < R) ? P.p[R_Y2_W][tid] : 0.f;
    float cdcol[D];
#pragma unroll
    for (int d = 0; d < D; ++d) cdcol[d] = (tid < R) ? tp.Cd[(size_t)min(d, Dr - 1) * R + tid] : 0.f;   // (dy is zero beyond Dr)
    // this sample's GRU tape, h and dhin for ALL steps in one round trip (clamped, not guarded)
    constexpr int NU_ = TMAX * 4 * R / NT, NH_ = ((TMAX + 1) * R + NT - 1) / NT, NI_ = TMAX * R / NT;
    float ru_[NU_], rh_[NH_], ri_[NI_];
    const int Tm1 = T - 1;
#pragma unroll
    for (int u = 0; u < NU_; ++u) { const int i = tid + NT * u, t = min(i / (4 * R), min(tstar, Tm1)), j = i % (4 * R); ru_[u] = tp.gru[((size_t)t * B + b) * 4 * R + j]; }
#pragma unroll
    for (int u = 0; u < NH_; ++u) { const int i = tid + NT * u, t = min(i / R, min(tstar + 1, T)), j = i % R; rh_[u] = tp.h[((size_t)t * B + b) * R + j]; }
#pragma unroll
    for (int u = 0; u < NI_; ++u) { const int i = tid + NT * u, t = min(i / R, min(tstar, Tm1)), j = i % R; ri_[u] = binary ? tp.dhin[((size_t)t * B + b) * R + j] : 0.f; }
#pragma unroll
    for (int u = 0; u < NU_; ++u) t_gru[tid + NT * u] = ru_[u];
#pragma unroll
    for (int u = 0; u < NH_; ++u) { const int i = tid + NT * u; if (i < (TMAX + 1) * R) t_h[i] = rh_[u]; }
#pragma unroll
    for (int u = 0; u < NI_; ++u) s_dhin[tid + NT * u] = ri_[u];
    const float dy_mine = (tid < Dr) ? (sm_mine - (tid == tgt ? 1.f : 0.f)) / (float)dm.Bg : 0.f;
    if (tid < R) s_dh[tid] = 0.f;
    __syncthreads();
    // ---- output step t* (model.py:1264-1275): dy, A* = y1[:, :R] h*, dA, dAy = W_y1h^T dA
    {
        const int t = tstar;
        if (tid < 64) {
            if (lane < Dr) { tp.dy[(size_t)b * Dr + lane] = dy_mine; tp.dyT[(size_t)lane * B + b] = dy_mine; }
            if (lane < 32) s_dy[lane] = dy_mine;
            const float dsum = dpp_wave_sum(dy_mine);
            if (lane == 0) tp.dysum[b] = dsum;
        } else if (tid < 64 + R) {
            tp.hstar[(size_t)b * R + tid - 64] = t_h[(t + 1) * R + tid - 64];
        }
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < R / K4; ++i) acc = fmaf(y1r[i], t_h[(t + 1) * R + p4 * (R / K4) + i], acc);
        acc = lane_group_sum<K4>(acc);
        if (p4 == 0) s_A[k4] = acc;
        __syncthreads();
        if (tid < R) {
            const float a = s_A[tid];
            float dacc = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) dacc += (a + cdcol[d] > 0.f) ? s_dy[d] : 0.f;
            const float v = dacc * w2_mine;
            s_dA[tid] = v; tp.dA[(size_t)b * R + tid] = v; tp.Astar[(size_t)b * R + tid] = a;
        }
        __syncthreads();
        float accy = 0.f;
#pragma unroll
        for (int i = 0; i < R / K4; ++i) accy = fmaf(y1T[i], s_dA[p4 * (R / K4) + i], accy);
        accy = lane_group_sum<K4>(accy);
        if (p4 == 0) s_dAy[k4] = accy;
    }
    // rows of steps this sample never took (only when k_wgrad has no live-row list)
    for (int t = zero_dead ? tstar + 1 : T; t < T; ++t) {
        const size_t row = (size_t)t * B + b;
        if (tid < 3 * R) { tp.dgi[row * 3 * R + tid] = 0.f; tp.dgh[row * 3 * R + tid] = 0.f; }
    }
    __syncthreads();
    // ---- the recurrence: [cell backward] barrier [W_hh^T dgh] barrier
    for (int t = tstar; t >= 0; --t) {
        const size_t row = (size_t)t * B + b;
        const float din = s_dhin[t * R + k4] + ((t == tstar) ? s_dAy[k4] : 0.f);
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

namespace mmg {
// acc += X[16 rows, K] . Wt[16 cols, K]^T for one MFMA tile; lane (i, q) owns row i of both operands (k contiguous),
// 8 k-groups (16 float4 loads) in flight; columns beyond K are clamped on both sides and masked on X.
template <bool VEC>
__device__ __forceinline__ void seg_mfma_batched(f32x4& acc, const float* __restrict__ xrow, const float* __restrict__ wrow, int K, int q) {
    const int kgroups = (K + 15) >> 4;
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int g0 = 0; g0 < kgroups; g0 += 8) {
        float4 a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = min(g0 + u, kgroups - 1) * 16 + q * 4;
            a[u] = ldrow4c<VEC>(xrow, k, K); b[u] = ldrow4c<VEC>(wrow, k, K);
            if (k + 3 >= K) {
                a[u].x = (k < K) ? a[u].x : 0.f; a[u].y = (k + 1 < K) ? a[u].y : 0.f;
                a[u].z = (k + 2 < K) ? a[u].z : 0.f; a[u].w = (k + 3 < K) ? a[u].w : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            if (g0 + u < kgroups) {
                acc = mfma16(a[u].x, b[u].x, acc); acc = mfma16(a[u].y, b[u].y, acc);
                acc = mfma16(a[u].z, b[u].z, acc); acc = mfma16(a[u].w, b[u].w, acc);
            }
            if (g0 + u + 1 < kgroups) {
                acc1 = mfma16(a[u + 1].x, b[u + 1].x, acc1); acc1 = mfma16(a[u + 1].y, b[u + 1].y, acc1);
                acc1 = mfma16(a[u + 1].z, b[u + 1].z, acc1); acc1 = mfma16(a[u + 1].w, b[u + 1].w, acc1);
            }
        }
    }
    acc += acc1;
}

// ---------------------------------------------------------------------------------------------
// k_baselines4: k_baselines3 (kernels_fwd.h: both baselines over the LIVE (step, sample) rows only, one MFMA pass per
// 16 rows x 64 hidden units, no time loop) for any message / state width: the operand rows are streamed with batched float4
// loads instead of living in four register fragments.  Needs B <= 64, W / R / H multiples of 4 and tape.basehx
// (h_x . linear1.weight[:, :H]^T of baseline_sen, a k_gemm_nt launch).  grid (ceil(T*B/16), ceil(K/64), 2).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MMG_BLOCK) void k_baselines4(Dims dm, Params P, Tape tp) {
    __shared__ int s_rid[16];
    __shared__ float s_part[4][16];
    const int B = dm.B, H = dm.H, W = dm.W, R = dm.R, K = dm.K, T = dm.T;
    const int which = blockIdx.z, lo = blockIdx.x * 16, byi = blockIdx.y, npb = gridDim.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int ts = tp.tstar[min(lane, B - 1)];
    const int n = (byi * 4 + wave) * 16 + i;
    const bool nv = n < K;
    const float* W1 = which ? P.p[BS_L1_W] : P.p[BR_L1_W];
    const int ldw = which ? H + W : W + R;
    const float* wrow = W1 + (size_t)(nv ? n : 0) * ldw;
    const float bias = nv ? (which ? P.p[BS_L1_B][n] : P.p[BR_L1_B][n]) : 0.f;
    const float w2 = nv ? (which ? P.p[BS_L2_W][n] : P.p[BR_L2_W][n]) : 0.f;
    if (wave == 0) {                                     // entries [lo, lo + 16) of the live-row list
        if (lane < 16) s_rid[lane] = -1;
        int base = 0;
        for (int t = 0; t < T && base < lo + 16; ++t) {
            const bool act = (lane < B) && (t <= ts);
            const unsigned long long m = __ballot(act);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (act && pos >= lo && pos < lo + 16) s_rid[pos - lo] = t * B + lane;
            base += __popcll(m);
        }
    }
    __syncthreads();
    if (s_rid[0] < 0) return;                            // window beyond the live rows
    const int rid = s_rid[i];
    const size_t rr = (size_t)(rid >= 0 ? rid : s_rid[0]);      // (padding rows repeat a live row; their results are dropped)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int orow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        orow[r] = s_rid[q * 4 + r];
        if (which) acc[r] = (orow[r] >= 0) ? tp.basehx[(size_t)(orow[r] % B) * K + min(n, K - 1)] : 0.f;
    }
    seg_mfma_batched<true>(acc, (which ? tp.zr : tp.z) + rr * W, wrow + (which ? H : 0), W, q);
    if (!which) seg_mfma_batched<true>(acc, tp.h + (rr + B) * R, wrow + W, R, q);
    float* hid = which ? tp.hid_s : tp.hid_r;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = fmaxf(acc[r] + bias, 0.f);                                // model.py:514
        if (orow[r] >= 0 && nv) hid[(size_t)orow[r] * K + n] = v; else v = 0.f;
        v = dpp_group_sum<16>(v * w2);
        if (i == 0) s_part[wave][q * 4 + r] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16 && s_rid[threadIdx.x] >= 0) {
        float* part = which ? tp.bs_part : tp.br_part;
        part[(size_t)s_rid[threadIdx.x] * npb + byi] =
            (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
    }
}
}  // namespace mmg
