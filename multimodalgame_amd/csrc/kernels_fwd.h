// kernels_fwd.h -- forward kernels of the exchange path.
//   k_prep          per-minibatch constants of the parameters (Cd, hw0, dsig)
//   k_gemm_nt       out[M,N] = X[M,K] . W[N,K]^T + bias   (fp32 MFMA 16x16x4; h_x and baseline_sen's
//                   h_x part -- computed ONCE per minibatch instead of once per step as model.py:195)
//   k_conversation  one workgroup per sample runs the whole T-step conversation (model.py:801-866)
//   k_baselines     both baselines over all (step, sample) rows, fp32 MFMA (model.py:835-843)
#pragma once
#include "device_utils.h"
#include "layout.h"

namespace mmg {

// ---------------------------------------------------------------------------------------------
// k_prep: grid = D + 1 blocks.  Block d < D: Cd[d,:] = W_y1[:, R:] . desc[d] + b_y1  (the
// description half of y1 applied once per class instead of once per (sample, class) row of
// build_inp, model.py:412/432 -- SURVEY.md Appendix A.2).  Block D: hw0 = code_layer(sigmoid(
// code_bias)) (model.py:199-200) and dsig = sigmoid'(code_bias).
// ---------------------------------------------------------------------------------------------
// LLO: the output ALSO goes out as (value, epoch) pairs at ll[m * ldo + n] (consumer roles of the SAME launch spin on them);
// LLA: the X operand is such a pair array written by other roles of this launch (X = the pair array, ldx in pairs)
template <bool LLO = false, bool LLA = false>
__device__ __forceinline__ void gemm_nt_tile(int tile, const float* __restrict__ X, int ldx, const float* __restrict__ Wm, int ldw,
                                             const float* __restrict__ bias, float* __restrict__ out, int ldo, int M, int N, int K,
                                             float* ll = nullptr, uint32_t epoch = 0u, uint32_t* sync = nullptr);

// blocks [0, D]: parameter-only constants; blocks (D, D + hx_tiles]: tiles of h_x = image_layer(x)
// (model.py:195) -- independent work sharing one launch.
// cpb > 1 (many classes): a class block owns `cpb` consecutive classes and keeps its y1 / w_d weight row in registers across
// them -- one block per class re-reads all 2 R weight rows (51 KB) from L2 per class: 51 MB per launch at D = 1000.
// ROLE: the blocks are workgroup roles of the conversation's launch (k_conversation_fast3): what the sample roles read (h_x, hw0,
// Cd, Dd) ALSO goes out as (value, launch epoch) pairs (tape.prepll, device_utils.h: st_ll) the consumers spin on; the epoch is
// the minibatch counter of this launch, which the launch's closing role bumps (prep_closing_role) once every consumer has
// arrived -- every role has read it by then -- instead of the first hw0 block; only the counters of the backward launch's roles are zeroed here.
#define PREP_CTR_ARRIVE 193
#define PREP_CTR(tp, k) ((tp).pflags + (size_t)(k) * 64)
// pair offsets inside tape.prepll
__host__ __device__ inline size_t prepll_hx(const Dims&) { return 0; }
__host__ __device__ inline size_t prepll_hw0(const Dims& d) { return (size_t)d.B * d.H; }
__host__ __device__ inline size_t prepll_cd(const Dims& d) { return (size_t)d.B * d.H + d.H; }
__host__ __device__ inline size_t prepll_dd(const Dims& d) { return (size_t)d.B * d.H + d.H + (size_t)d.D * d.R; }
// Trailing blocks of k_prep on the agents of BASELINE configs 1-3 / 5 (H = 256, W = 32, R = 64): the transposed weight fragments
// k_bwd_conv_fast keeps in registers, repacked so that each lane finds ITS values as 30 consecutive-by-lane float4s -- as the matrices
// lie in the parameter buffer they are 120 strided dword loads per thread, and a wave holds at most 63 loads in flight (the backward's
// prologue is ~230 loads: 3.7 memory latencies).  wrep[(j * 256 + tid) * 4 + c]:
//   j  0..11  W_hh^T   [(p4 * 48 + i) * R + k4]          i = 4 j + c          (k4 = tid / 4, p4 = tid % 4)
//   j 12..15  y1[:, :R]^T [(p4 * 16 + i) * (R + V) + k4]  i = 4 (j - 12) + c
//   j 16..17  W_w      [(4 ks + fq) * R + 16 wv + fi]     ks = 4 (j - 16) + c  (wv = tid / 64, fi = lane % 16, fq = lane / 16)
//   j 18..21  W_h      [(4 ks + fq) * R + 16 wv + fi]     ks = 4 (j - 18) + c
//   j 22..29  binary_layer [(4 ks + fq) * H + 64 wv + 16 nt + fi]   nt * 8 + ks = 4 (j - 22) + c
#define MMG_REPACK_BLOCKS 8
#define MMG_REPACK_F4 30
__host__ __device__ inline bool prep_has_repack(const Dims& d) { return d.H == 256 && d.W == 32 && d.R == 64; }
// GAME (kernels_game.h: the fragments are read by sample roles of the SAME launch, ~20 us later): write-through stores, and once
// they have completed the block publishes the pair gamell[B + rb] -- the consumers check it beside their loads
template <bool GAME = false>
__device__ __forceinline__ void prep_repack(const Dims& dm, const Params& P, const Tape& tp, const int rb, const uint32_t epoch = 0u) {
    constexpr int H = 256, W = 32, R = 64, NTH = 256;
    const int ldy = R + dm.V;
    for (int idx = rb * NTH + (int)threadIdx.x; idx < MMG_REPACK_F4 * NTH; idx += MMG_REPACK_BLOCKS * NTH) {
        const int j = idx / NTH, t = idx % NTH;
        const int k4 = t >> 2, p4 = t & 3, wv = t >> 6, fi = t & 15, fq = (t & 63) >> 4;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (j < 12) v[c] = P.p[R_WHH][(size_t)(p4 * 48 + 4 * j + c) * R + k4];
            else if (j < 16) v[c] = P.p[R_Y1_W][(size_t)(p4 * 16 + 4 * (j - 12) + c) * ldy + k4];
            else if (j < 18) v[c] = P.p[R_W_W][(size_t)(4 * (4 * (j - 16) + c) + fq) * R + 16 * wv + fi];
            else if (j < 22) v[c] = P.p[R_WH_W][(size_t)(4 * (4 * (j - 18) + c) + fq) * R + 16 * wv + fi];
            else { const int f = 4 * (j - 22) + c, nt = f >> 3, ks = f & 7; v[c] = P.p[S_BIN_W][(size_t)(4 * ks + fq) * H + 64 * wv + 16 * nt + fi]; }
        }
        if (GAME) st_wt4(tp.wrep + (size_t)idx * 4, make_float4(v[0], v[1], v[2], v[3]));
        else *reinterpret_cast<float4*>(tp.wrep + (size_t)idx * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (GAME) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) st_ll(tp.gamell, (size_t)dm.B + rb, 1.f, epoch);
    }
    (void)W;
}
template <bool ROLE, bool GAME = false>
__device__ __forceinline__ void prep_body(const Dims& dm, const Params& P, const Tape& tp, const float* __restrict__ desc,
                                          const float* __restrict__ x, const int cpb, const int blk, float* smem, const int bump_mb) {
    const int tid = threadIdx.x;
    const uint32_t epoch = ROLE ? tp.counter[3] + 1u : 0u;
    const int HB = (dm.H + 63) / 64;                 // blocks [nC, nC + HB): 64 rows of hw0 each
    const int nC = (dm.D + cpb - 1) / cpb;           // class blocks
    if (blk >= nC + HB) {
        const int nhx = x ? ((dm.B + 15) / 16) * ((dm.H + 15) / 16) : 0;
        if (blk >= nC + HB + nhx) { prep_repack<GAME>(dm, P, tp, blk - nC - HB - nhx, epoch); return; }
        gemm_nt_tile<ROLE, false>(blk - nC - HB, x, dm.F, P.p[S_IMG_W], dm.F, P.p[S_IMG_B], tp.hx, dm.H, dm.B, dm.H, dm.F, tp.prepll, epoch);
        return;
    }
    if (blk < nC && cpb > 1) {
        // (host: cpb > 1 only with R <= 64, V <= 128, V % 4 == 0) threads [0, R): Cd rows, [R, 2R): Dd rows
        const int R = dm.R, V = dm.V, ld = dm.R + dm.V, D = dm.D;
        const int d0 = blk * cpb, nd = min(cpb, D - d0);
        float* s_desc = smem;                       // [cpb][V]
        float* s_cd = smem + cpb * V;               // [cpb][R]
        const bool isC = tid < R, isD = tid >= R && tid < 2 * R;
        const int r = isC ? tid : (isD ? tid - R : 0);
        const float* wrow = isD ? P.p[R_WD_W] + (size_t)r * V : P.p[R_Y1_W] + (size_t)r * ld + R;
        float4 wreg[32];                            // (issued BEFORE the description rows: their copy loop waits for its loads)
#pragma unroll
        for (int j = 0; j < 32; ++j) wreg[j] = (4 * j < V) ? *reinterpret_cast<const float4*>(wrow + 4 * min(j, V / 4 - 1)) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float by = isC ? P.p[R_Y1_B][r] : 0.f;
        for (int i = tid; i < nd * V; i += blockDim.x) { const float dv = desc[(size_t)d0 * V + i]; s_desc[i] = dv; tp.descc[(size_t)d0 * V + i] = dv; }
        __syncthreads();
        if (isC || isD) {
            for (int c = 0; c < nd; ++c) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    if (4 * j < V) {
                        const float4 dv = *reinterpret_cast<const float4*>(s_desc + c * V + 4 * j);
                        a0 = fmaf(wreg[j].x, dv.x, a0); a1 = fmaf(wreg[j].y, dv.y, a1); a2 = fmaf(wreg[j].z, dv.z, a2); a3 = fmaf(wreg[j].w, dv.w, a3);
                    }
                }
                const float v = (a0 + a1) + (a2 + a3) + by;
                const int d = d0 + c;
                if (isC) { tp.Cd[(size_t)d * R + r] = v; if (ROLE) st_ll(tp.prepll, prepll_cd(dm) + (size_t)d * R + r, v, epoch); tp.CdT[(size_t)r * D + d] = -v; s_cd[c * R + r] = v; if (d < 32) tp.cd32[((size_t)(d >> 2) * R + r) * 4 + (d & 3)] = v; }
                else { tp.Dd[(size_t)d * R + r] = v; if (ROLE) st_ll(tp.prepll, prepll_dd(dm) + (size_t)d * R + r, v, epoch); }
            }
        }
        __syncthreads();
        {   // cy[d] = b_y2 + sum_r w2[r] Cd[d][r]: wave w takes the classes w, w + 4, ...
            const int lane = tid & 63, wave = tid >> 6;
            const float w2 = (lane < R) ? P.p[R_Y2_W][lane] : 0.f;
            for (int c = wave; c < nd; c += 4) {
                const float t = dpp_wave_sum((lane < R) ? w2 * s_cd[c * R + lane] : 0.f);
                if (lane == 0) tp.cy[d0 + c] = t + P.p[R_Y2_B][0];
            }
        }
        return;
    }
    if (blk < nC) {
        // Cd[d, r] = b_y1[r] + sum_v W_y1[r, R+v] * desc[d, v]: thread r owns output r and issues all of its
        // row loads at once (one memory round trip per block instead of one per row pass)
        const int d = blk;
        const int R = dm.R, V = dm.V, ld = dm.R + dm.V;
        float* s_desc = smem;                       // [V]
        for (int v = tid; v < V; v += blockDim.x) { const float dv = desc[(size_t)d * V + v]; s_desc[v] = dv; tp.descc[(size_t)d * V + v] = dv; }
        __syncthreads();
        const float* by1 = P.p[R_Y1_B];
        const bool vec = ((ld & 3) == 0) && ((R & 3) == 0) && ((V & 3) == 0);
        float cy_part = 0.f;
        // threads [R, 2R) (when the block has them): Dd[d, r] = sum_v w_d[r, v] desc[d, v] -- the query head's description
        // product folded onto the classes, formed beside Cd (kernels_fast3.h: g = tanh(w_h h + b_h + softmax(y) . Dd))
        for (int rr = tid; rr < 2 * R; rr += blockDim.x) {
            if (rr < R) continue;
            const int r = rr - R;
            const float* wrow = P.p[R_WD_W] + (size_t)r * V;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            if ((V & 3) == 0) {
#pragma unroll 8
                for (int v = 0; v < V; v += 4) {
                    const float4 wv = *reinterpret_cast<const float4*>(wrow + v);
                    const float4 dv = *reinterpret_cast<const float4*>(s_desc + v);
                    a0 = fmaf(wv.x, dv.x, a0); a1 = fmaf(wv.y, dv.y, a1); a2 = fmaf(wv.z, dv.z, a2); a3 = fmaf(wv.w, dv.w, a3);
                }
            } else {
                for (int v = 0; v < V; ++v) a0 = fmaf(wrow[v], s_desc[v], a0);
            }
            tp.Dd[(size_t)d * R + r] = (a0 + a1) + (a2 + a3);
            if (ROLE) st_ll(tp.prepll, prepll_dd(dm) + (size_t)d * R + r, (a0 + a1) + (a2 + a3), epoch);
        }
        for (int r = tid; r < R; r += blockDim.x) {
            const float* wrow = P.p[R_Y1_W] + (size_t)r * ld + R;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            if (vec) {
#pragma unroll 8
                for (int v = 0; v < V; v += 4) {
                    const float4 wv = *reinterpret_cast<const float4*>(wrow + v);
                    const float4 dv = *reinterpret_cast<const float4*>(s_desc + v);
                    a0 = fmaf(wv.x, dv.x, a0); a1 = fmaf(wv.y, dv.y, a1); a2 = fmaf(wv.z, dv.z, a2); a3 = fmaf(wv.w, dv.w, a3);
                }
            } else {
                for (int v = 0; v < V; ++v) a0 = fmaf(wrow[v], s_desc[v], a0);
            }
            const float cdv = (a0 + a1) + (a2 + a3) + by1[r];
            if (GAME) st_wt(&tp.Cd[(size_t)d * R + r], cdv); else tp.Cd[(size_t)d * R + r] = cdv;      // (GAME: class roles of the same launch read it)
            if (ROLE) st_ll(tp.prepll, prepll_cd(dm) + (size_t)d * R + r, cdv, epoch);
            if (d < 32) tp.cd32[((size_t)(d >> 2) * R + r) * 4 + (d & 3)] = cdv;
            tp.CdT[(size_t)r * dm.D + d] = -cdv;      // NEGATED: relu(A + c) = max(A, -c) + c (kernels_tile.h, many-class y head)
            cy_part = fmaf(P.p[R_Y2_W][r], cdv, cy_part);
        }
        cy_part = block_sum(cy_part, smem + ((V + 3) & ~3));
        if (tid == 0) tp.cy[d] = cy_part + P.p[R_Y2_B][0];
    } else {
        float* s_sig = smem;                        // [W]
        const float* cb = P.p[S_CODE_BIAS];
        const int W = dm.W;
        for (int j = tid; j < W; j += blockDim.x) {
            const float sg = sigmoidf_(cb[j]);
            s_sig[j] = sg;
            if (blk == nC) tp.dsig[j] = sg * (1.f - sg);
        }
        const bool first = blk == nC;
        if (first && tid < dm.T + 2) tp.alive[tid] = (tid == 0) ? 1 : 0;     // per-step live-tile counts (kernels_tile.h)
        if (first) for (int i = tid; i < (ROLE ? 128 : 20 * 64); i += blockDim.x) tp.pflags[(size_t)i * 64] = 0u;   // role counters of k_conv_persist / the backward launch
        if (first && mc_shape(dm.H, dm.W, dm.R, dm.V, dm.D, dm.T))                                    // ... and of k_conversation_mc
            for (int i = tid; i < 2 * ((dm.B + 15) / 16); i += blockDim.x) tp.mcflags[(size_t)i * 64] = 0u;
        if (first && tid == 0) {
            // minibatch counter = the Philox stream of a TRAINING conversation: evaluation passes draw nothing and leave it alone (an
            // eval_dev on the training engine must not move the sampling stream -- data-parallel ranks evaluate on rank 0 only);
            // the launch epoch of the (value, epoch) pair hand-offs moves with every launch
            if (!ROLE) { if (bump_mb) tp.counter[0] += 1u; tp.counter[3] += 1u; }
            if (tp.counter[2] > tp.counter[1]) tp.counter[1] = tp.counter[2];   // optimizer step bumped by k_opt
        }
        __syncthreads();
        // hw0 = code_layer(sigmoid(code_bias)): four lanes per row, interleaved float4 slices (64 contiguous bytes per row step),
        // 8 loads in flight per lane
        const float* bc = P.p[S_CODE_B];
        const int n = (blk - nC) * 64 + (tid >> 2), p4 = tid & 3;
        const int nc = min(n, dm.H - 1);
        const float* wrow = P.p[S_CODE_W] + (size_t)nc * W;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if ((W & 3) == 0) {
            const int W4 = W >> 2;
            for (int k0 = p4; k0 < W4; k0 += 32) {
                float4 wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = *reinterpret_cast<const float4*>(wrow + 4 * min(k0 + 4 * u, W4 - 1));
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (k0 + 4 * u < W4) {
                        const float4 sv = *reinterpret_cast<const float4*>(s_sig + 4 * (k0 + 4 * u));
                        a0 = fmaf(wv[u].x, sv.x, a0); a1 = fmaf(wv[u].y, sv.y, a1); a2 = fmaf(wv[u].z, sv.z, a2); a3 = fmaf(wv[u].w, sv.w, a3);
                    }
                }
            }
        } else {
            for (int k = p4; k < W; k += 4) a0 = fmaf(wrow[k], s_sig[k], a0);
        }
        const float tot = dpp_group_sum<4>((a0 + a1) + (a2 + a3));
        if (p4 == 0 && n < dm.H) { tp.hw0[n] = tot + bc[n]; if (ROLE) st_ll(tp.prepll, prepll_hw0(dm) + n, tot + bc[n], epoch); }
    }
}
// Consumers of a launch with prep roles count themselves once they hold what they waited for (a fire-and-forget increment);
// the launch's LAST workgroup is the closing role: it waits for all of them -- every role has read the minibatch counter by
// then -- bumps it and re-arms the arrival counter.  (Done by the last consumer to arrive instead, through a returning atomic:
// the closing work lands at the end of a conversation, on the launch's critical path when that one is the longest.)
__device__ __forceinline__ void prep_consumer_arrive(const Tape& tp) {
    __hip_atomic_fetch_add(PREP_CTR(tp, PREP_CTR_ARRIVE), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void prep_closing_role(const Tape& tp, const uint32_t consumers, const int bump_mb) {
    if (threadIdx.x != 0) return;
    int spins = 0;
    while (__hip_atomic_load(PREP_CTR(tp, PREP_CTR_ARRIVE), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < consumers) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > MMG_SPIN_LIMIT) { __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 5u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __hip_atomic_store(PREP_CTR(tp, PREP_CTR_ARRIVE), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bump_mb) tp.counter[0] += 1u;
    tp.counter[3] += 1u;
}
__host__ __device__ inline int prep_blocks(const Dims& d, int cpb, bool with_hx) {
    return (d.D + cpb - 1) / cpb + (d.H + 63) / 64 + (with_hx ? ((d.B + 15) / 16) * ((d.H + 15) / 16) : 0) + (prep_has_repack(d) ? MMG_REPACK_BLOCKS : 0);
}

__global__ __launch_bounds__(MMG_BLOCK) void k_prep(Dims dm, Params P, Tape tp, const float* __restrict__ desc,
                                                    const float* __restrict__ x, int cpb, int bump_mb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    prep_body<false>(dm, P, tp, desc, x, cpb, (int)blockIdx.x, smem, bump_mb);
}



// ---------------------------------------------------------------------------------------------
// k_gemm_nt: out[m, n] = sum_k X[m*ldx + k] * Wm[n*ldw + k] + bias[n].  One workgroup per 16x16
// output tile; its 4 waves split K and combine through LDS.  Fragments are read straight from
// global memory (both operands have K contiguous): with K % 16 == 0 each lane fetches one float4
// per operand per four MFMAs (the k index inside the group of 16 is permuted identically for A
// and B, which leaves the sum unchanged).
// ---------------------------------------------------------------------------------------------
template <bool LLO, bool LLA>
__device__ __forceinline__ void gemm_nt_tile(int tile, const float* __restrict__ X, int ldx,
                                                       const float* __restrict__ Wm, int ldw,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ out, int ldo, int M, int N, int K, float* ll, uint32_t epoch, uint32_t* sync) {
    __shared__ float s_acc[4][16][17];
    const int tiles_n = (N + 15) >> 4;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int m = tm * 16 + i, n = tn * 16 + i;
    const bool mv = m < M, nv = n < N;
    const float* xr = X + (size_t)(mv ? m : 0) * ldx;
    const float* wr = Wm + (size_t)(nv ? n : 0) * ldw;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const bool vec = ((K & 15) == 0) && ((ldx & 3) == 0) && ((ldw & 3) == 0) &&
                     ((((uintptr_t)X) & 15) == 0) && ((((uintptr_t)Wm) & 15) == 0);
    if (vec) {
        const int kchunks = K >> 4;                       // groups of 16 k
        const int per = (kchunks + 3) >> 2;
        const int c0 = wave * per, c1 = min(kchunks, c0 + per);
        for (int cb0 = c0; cb0 < c1; cb0 += 8) {            // 8 chunks (16 float4 loads) in flight per pass
            float4 a[8], b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = (cb0 + u) * 16 + q * 4;
                const bool kv = (cb0 + u) < c1;
                a[u] = (mv && kv) ? (LLA ? ll_wait4(X, (size_t)(mv ? m : 0) * ldx + k, epoch, sync) : *reinterpret_cast<const float4*>(xr + k)) : make_float4(0.f, 0.f, 0.f, 0.f);
                b[u] = (nv && kv) ? *reinterpret_cast<const float4*>(wr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (cb0 + u < c1) {
                    acc = mfma16(a[u].x, b[u].x, acc); acc = mfma16(a[u].y, b[u].y, acc);
                    acc = mfma16(a[u].z, b[u].z, acc); acc = mfma16(a[u].w, b[u].w, acc);
                }
            }
        }
    } else {
        const int ksteps = (K + 3) >> 2;
        const int per = (ksteps + 3) >> 2;
        const int s0 = wave * per, s1 = min(ksteps, s0 + per);
        for (int st = s0; st < s1; ++st) {
            const int k = st * 4 + q;
            const float a = (mv && k < K) ? xr[k] : 0.f;
            const float b = (nv && k < K) ? wr[k] : 0.f;
            acc = mfma16(a, b, acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) s_acc[wave][q * 4 + r][i] = acc[r];
    __syncthreads();
    {
        const int r = threadIdx.x >> 4, cidx = threadIdx.x & 15;       // 256 threads = 16x16 outputs
        const int mo = tm * 16 + r, no = tn * 16 + cidx;
        if (mo < M && no < N) {
            float v = s_acc[0][r][cidx] + s_acc[1][r][cidx] + s_acc[2][r][cidx] + s_acc[3][r][cidx];
            v += bias ? bias[no] : 0.f;
            out[(size_t)mo * ldo + no] = v;
            if (LLO) st_ll(ll, (size_t)mo * ldo + no, v, epoch);
        }
    }
}

__global__ __launch_bounds__(MMG_BLOCK) void k_gemm_nt(const float* __restrict__ X, int ldx,
                                                       const float* __restrict__ Wm, int ldw,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ out, int ldo, int M, int N, int K) {
    gemm_nt_tile(blockIdx.x, X, ldx, Wm, ldw, bias, out, ldo, M, N, K);
}

// ---------------------------------------------------------------------------------------------
// k_conversation
// ---------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* x;            // unused here (h_x precomputed) but kept for agent-level calls
    const int64_t* target;     // [B] or NULL
    const float* desc;         // [D,V]
    const float* u_z;          // [T,B,W] or NULL
    const float* u_s;          // [T,B]   or NULL
    const float* u_w;          // [T,B,W] or NULL
    uint64_t seed;
    int train;                 // sample (1) or round (0)
    int run_all;               // 1: every sample runs all steps
    int t_begin, t_end;        // step window (whole conversation: 0, T)
    int phases;                // bit 0 sender, bit 1 receiver
    // agent-level I/O (NULL in the fused path: state lives in the tape)
    const float* w_in;         // sender input code for t_begin > 0   [B,W]
    float* h_state;            // receiver GRU state in/out           [B,R]
    float* sprod_state;        // receiver running stop product       [B]
    int sprod_first;
    int rsample;               // k_conv_persist: one receiver role per SAMPLE (register-resident weights) instead of one per tile
    int b_begin, b_count;      // ... for the samples [b_begin, b_begin + b_count) (large batches run as consecutive launches)
    int persist, ns1, ns2;     // kernels_tile.h, k_conv_persist: sender roles per sample tile (0: not persistent)
    int nhelp, per;            // kernels_tile.h, k_conv_split: class helpers per sample tile, classes per slice
    int y_last_only;           // Fixed-mode training step (mmg_train_step): tape.y keeps the output step's logits only
    int lean;                  // training-minimal call in continuous mode: tape arrays the receiver-only backward never reads are not stored
    int nprep, prep_cpb;       // k_conversation_fast3: leading workgroups that run k_prep's blocks as roles of the launch (0: k_prep ran before), classes per class block
    int nbase;                 // ... and trailing basehx tiles
    int l2_handoff;            // kernels_mc3.h / kernels_mc3p.h: the tile's members share an XCD (probed at mmg_create): pairs as plain stores, inside its L2
};

struct ConvSmem {
    float *hx, *a, *c, *z, *w, *lp, *ne, *h, *hn, *A, *g, *g2, *gi, *gh, *y, *yout, *dbar, *red, *misc;
};

__device__ __forceinline__ int pad4(int n) { return (n + 3) & ~3; }

__host__ __device__ inline int conv_smem_floats(const Dims& d, int nthreads) {
    auto p4 = [](int n) { return (n + 3) & ~3; };
    return 2 * p4(d.H) + 5 * p4(d.W) + 5 * p4(d.R) + 2 * p4(3 * d.R) + 2 * p4(d.D) + p4(d.V) + 4 * nthreads + 32;
}

// NT = 256 (many samples: occupancy) or 512 with deeper load batches (few samples, large weight matrices: the weights
// stream from L2 every step, so what counts is bytes in flight per CU).
// k_xcc_probe: which XCD runs workgroup i of a launch?  mmg_create checks the rule the XCD-aware launches rely on -- workgroup
// i goes to XCD i % 8 -- before it lets a hand-off stay inside one XCD's L2 (st_ll_l2).
__global__ __launch_bounds__(64) void k_xcc_probe(uint32_t* __restrict__ out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

template <int NT>
__global__ __launch_bounds__(NT) void k_conversation(Dims dm, Params P, Tape tp, ConvArgs ar) {
    constexpr int GU = NT == 512 ? 8 : 4;               // row passes per load batch of gemv_rows
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int B = dm.B, H = dm.H, W = dm.W, R = dm.R, V = dm.V, D = dm.D, T = dm.T;
    ConvSmem s;
    {
        float* p = smem;
        s.hx = p; p += pad4(H);  s.a = p; p += pad4(H);
        s.c = p; p += pad4(W);   s.z = p; p += pad4(W);  s.w = p; p += pad4(W);
        s.lp = p; p += pad4(W);  s.ne = p; p += pad4(W);
        s.h = p; p += pad4(R);   s.hn = p; p += pad4(R); s.A = p; p += pad4(R);
        s.g = p; p += pad4(R);   s.g2 = p; p += pad4(R);
        s.gi = p; p += pad4(3 * R); s.gh = p; p += pad4(3 * R);
        s.y = p; p += pad4(D);   s.yout = p; p += pad4(D);
        s.dbar = p; p += pad4(V);
        s.red = p; p += 4 * NT;
        s.misc = p;
    }
    float* s_ne = s.ne;

    const bool do_sen = ar.phases & 1, do_rec = ar.phases & 2;
    const bool binary = dm.use_binary != 0;
    const bool train = ar.train != 0;

    // ---- conversation state ----
    if (do_sen) for (int i = tid; i < H; i += nt) s.hx[i] = tp.hx[(size_t)b * H + i];
    if (do_rec) {
        for (int i = tid; i < R; i += nt)
            s.h[i] = ar.h_state ? ar.h_state[(size_t)b * R + i] : (ar.t_begin == 0 ? 0.f : tp.h[((size_t)ar.t_begin * B + b) * R + i]);
        if (ar.t_begin == 0 && !ar.h_state) for (int i = tid; i < R; i += nt) tp.h[(size_t)b * R + i] = 0.f;
    }
    for (int j = tid; j < W; j += nt) {
        float wv = dm.first_rec;                                   // model.py:786
        if (ar.t_begin > 0) wv = ar.w_in ? ar.w_in[(size_t)b * W + j] : tp.w[((size_t)(ar.t_begin - 1) * B + b) * W + j];
        s.w[j] = wv;
    }
    if (tid == 0) {
        s.misc[0] = 1.f;                                           // running stop mask m_t
        s.misc[1] = -1.f;                                          // t* (not yet known)
        s.misc[2] = (ar.sprod_state && !ar.sprod_first) ? ar.sprod_state[b] : 1.f;   // running prod of p_s
        if (ar.t_begin == 0 && do_rec) tp.mask[b] = 1;             // stop_mask[0] = ones   model.py:775
    }
    __syncthreads();

    const uint32_t mb_counter = tp.counter[0];
    const uint32_t gb = (uint32_t)(dm.boff + b);                   // index inside the GLOBAL minibatch

    int t = ar.t_begin;
    for (; t < ar.t_end; ++t) {
        const size_t row = (size_t)t * B + b;
        // ================= Sender (model.py:193-238) =================
        if (do_sen) {
            // code input: sigmoid(code_bias) at t == 0 (hw0 precomputed), else the receiver's last message
            for (int j = tid; j < W; j += nt) {
                const float cv = s.w[j];
                s.c[j] = cv;
                tp.zr[row * W + j] = cv;                           // z_r fed to baseline_sen (model.py:836)
                tp.c[row * W + j] = (t == 0) ? sigmoidf_(P.p[S_CODE_BIAS][j]) : cv;
            }
            __syncthreads();
            if (t > 0) {
                // (epilogues only park the sums in LDS: a bias / uniform load inside them would be one dependent
                //  global round trip per row pass, executed by a single lane of each group)
                gemv_rows<GU>(P.p[S_CODE_W], W, H, W, s.c, [&](int n, float acc) { s.a[n] = acc; });
                __syncthreads();
            }
            const float* bc = P.p[S_CODE_B];
            for (int i = tid; i < H; i += nt) {
                const float hw = (t == 0) ? tp.hw0[i] : s.a[i] + bc[i];
                const float av = tanhf(s.hx[i] + hw);              // model.py:216
                s.a[i] = av;
                tp.a[row * H + i] = av;
            }
            __syncthreads();
            {
                const float* bb = P.p[S_BIN_B];
                const float* uz = ar.u_z ? ar.u_z + row * W : nullptr;
                gemv_rows<GU>(P.p[S_BIN_W], H, W, H, s.a, [&](int n, float acc) { s.z[n] = acc; });
                __syncthreads();
                for (int n = tid; n < W; n += nt) {
                    const float lz = s.z[n] + bb[n];
                    float zz = lz, lpv = 0.f, nev = 0.f;
                    if (binary) {
                        const float p = sigmoidf_(lz);             // model.py:223
                        if (train) {
                            const float u = uz ? uz[n] : philox_uniform(ar.seed, (uint32_t)((t * dm.Bg + gb) * W + n), mb_counter, 0u);
                            zz = (u < p) ? 1.f : 0.f;              // model.py:227
                        } else {
                            zz = rintf(p);                         // model.py:229 (torch.round = half-to-even)
                        }
                        tp.pz[row * W + n] = p;
                        const float l1 = logf(p + MMG_EPS), l0 = logf(1.f - p + MMG_EPS);
                        lpv = zz * l1 + (1.f - zz) * l0;           // model.py:908-910
                        nev = p * l1 + (1.f - p) * l0;             // model.py:919-922
                    }
                    s.z[n] = zz; s.lp[n] = lpv; s_ne[n] = nev;
                    tp.z[row * W + n] = zz;
                }
            }
            __syncthreads();
            if (binary) {
                float lpv = 0.f, nev = 0.f;
                for (int j = tid; j < W; j += nt) { lpv += s.lp[j]; nev += s_ne[j]; }
                lpv = block_sum(lpv, s.misc + 8);
                nev = block_sum(nev, s.misc + 16);
                if (tid == 0) { tp.lp_z[row] = lpv; tp.ne_z[row] = nev; }
            }
        } else {
            // receiver-only call: the message comes from the tape slot of this step
            for (int j = tid; j < W; j += nt) s.z[j] = tp.z[row * W + j];
            __syncthreads();
        }
        if (!do_rec) continue;

        // ================= Receiver (model.py:333-342, 411-477) =================
        {   // GRUCell (model.py:340); gate order r, z(u), n
            const float* bih = P.p[R_BIH]; const float* bhh = P.p[R_BHH];
            gemv_rows<GU>(P.p[R_WIH], W, 3 * R, W, s.z, [&](int n, float acc) { s.gi[n] = acc; });
            gemv_rows<GU>(P.p[R_WHH], R, 3 * R, R, s.h, [&](int n, float acc) { s.gh[n] = acc; });
            __syncthreads();
            for (int i = tid; i < 3 * R; i += nt) { s.gi[i] += bih[i]; s.gh[i] += bhh[i]; }
        }
        __syncthreads();
        for (int i = tid; i < R; i += nt) {
            const float rr = sigmoidf_(s.gi[i] + s.gh[i]);
            const float uu = sigmoidf_(s.gi[R + i] + s.gh[R + i]);
            const float ghn = s.gh[2 * R + i];
            const float nn = tanhf(s.gi[2 * R + i] + rr * ghn);
            const float hv = nn + uu * (s.h[i] - nn);
            s.hn[i] = hv;
            float* gr = tp.gru + row * 4 * R;
            gr[i] = rr; gr[R + i] = uu; gr[2 * R + i] = nn; gr[3 * R + i] = ghn;
            tp.h[((size_t)(t + 1) * B + b) * R + i] = hv;
        }
        __syncthreads();
        {   // stop head (model.py:414-427) and the h-half of y1 (Appendix A.2)
            const float bsv = P.p[R_S_B][0];
            gemv_rows<GU>(P.p[R_S_W], R, 1, R, s.hn, [&](int n, float acc) {
                const float p = sigmoidf_(acc + bsv);
                float sv;
                if (train) {
                    const float u = ar.u_s ? ar.u_s[row] : philox_uniform(ar.seed, (uint32_t)(t * dm.Bg + gb), mb_counter, 1u);
                    sv = (u < p) ? 1.f : 0.f;                      // model.py:420
                } else {
                    const float prod = dm.s_prob_prod ? s.misc[2] * p : p;   // model.py:423-426 (misc[2] == 1 on a fresh conversation)
                    s.misc[2] = prod;
                    sv = rintf(prod);                              // model.py:427
                }
                tp.s[row] = sv; tp.ps[row] = p;
                const float l1 = logf(p + MMG_EPS), l0 = logf(1.f - p + MMG_EPS);
                tp.lp_s[row] = sv * l1 + (1.f - sv) * l0;
                tp.ne_s[row] = p * l1 + (1.f - p) * l0;
                s.misc[3] = sv;
            });
            gemv_rows<GU>(P.p[R_Y1_W], R + V, R, R, s.hn, [&](int n, float acc) { s.A[n] = acc; });
        }
        __syncthreads();
        {   // y[d] = b_y2 + sum_r w_y2[r] * relu(A[r] + Cd[d,r])      (model.py:432-433)
            const float* w2 = P.p[R_Y2_W];
            const float b2 = P.p[R_Y2_B][0];
            const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
            const bool vec = (R & 3) == 0;
            const int items = vec ? (R >> 2) : R;
            int G = 1; while (G < 64 && G < items) G <<= 1;
            const int rpw = 64 / G, sub = lane / G, gl = lane - sub * G;
            constexpr int U = 4;                                    // class passes loaded together (cf. gemv_rows)
            const int stride = nw * rpw;
            for (int d0 = wave * rpw + sub; d0 < D + sub; d0 += stride * U) {
                float acc[U];
                const float* crow[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { acc[u] = 0.f; crow[u] = tp.Cd + (size_t)min(d0 + u * stride, D - 1) * R; }
                if (vec) {
                    for (int k = gl; k < items; k += G) {
                        float4 cv[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) cv[u] = reinterpret_cast<const float4*>(crow[u])[k];
                        const float4 av = reinterpret_cast<const float4*>(s.A)[k];
                        const float4 wv = reinterpret_cast<const float4*>(w2)[k];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            acc[u] = fmaf(wv.x, fmax_nn(av.x + cv[u].x, 0.f), acc[u]); acc[u] = fmaf(wv.y, fmax_nn(av.y + cv[u].y, 0.f), acc[u]);
                            acc[u] = fmaf(wv.z, fmax_nn(av.z + cv[u].z, 0.f), acc[u]); acc[u] = fmaf(wv.w, fmax_nn(av.w + cv[u].w, 0.f), acc[u]);
                        }
                    }
                } else {
                    for (int k = gl; k < items; k += G) {
                        float cv[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) cv[u] = crow[u][k];
#pragma unroll
                        for (int u = 0; u < U; ++u) acc[u] = fmaf(w2[k], fmax_nn(s.A[k] + cv[u], 0.f), acc[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int d = d0 + u * stride;
                    const float r = group_sum(acc[u], G);
                    if (gl == 0 && d < D) { const float yv = r + b2; s.y[d] = yv; tp.y[row * D + d] = yv; }
                }
            }
        }
        __syncthreads();
        {   // stop-mask bookkeeping (model.py:852) -- uniform decision for the whole workgroup
            const float m_t = s.misc[0], sv = s.misc[3];
            const float m_next = fminf(m_t, sv);
            const bool first_stop = (m_next == 0.f) && (s.misc[1] < 0.f);
            const bool last = (t == T - 1);
            if (first_stop || (last && s.misc[1] < 0.f)) {
                for (int d = tid; d < D; d += nt) s.yout[d] = s.y[d];   // the output step (model.py:1261-1264)
            }
            __syncthreads();
            if (tid == 0) {
                tp.mask[(size_t)(t + 1) * B + b] = (uint8_t)(m_next != 0.f);
                if (first_stop || (last && s.misc[1] < 0.f)) s.misc[1] = (float)t;
                s.misc[0] = m_next;
            }
            __syncthreads();
            // a sample whose conversation has ended contributes nothing further to any loss (its
            // query w_t is never read: the receiver-message stream is active only while m_{t+1} == 1)
            if (!ar.run_all && !dm.fixed && train && m_next == 0.f) { ++t; break; }
        }
        // softmax(y) (detached, model.py:441) and the description mixture (model.py:442-449)
        float mx = -3.4e38f;
        for (int d = tid; d < D; d += nt) mx = fmaxf(mx, s.y[d]);
        mx = block_max(mx, s.misc + 8);
        float se = 0.f;
        for (int d = tid; d < D; d += nt) { const float e = expf(s.y[d] - mx); s.y[d] = e; se += e; }
        se = block_sum(se, s.misc + 16);
        const float inv = 1.f / se;
        for (int d = tid; d < D; d += nt) s.y[d] *= inv;
        __syncthreads();
        gemv_t(ar.desc, V, D, V, s.y, s.dbar, s.red, false);
        for (int v = tid; v < V; v += nt) tp.dbar[row * V + v] = s.dbar[v];
        {   // h_w = tanh(w_h(h) + w_d(dbar))   (model.py:452)
            gemv_rows<GU>(P.p[R_WH_W], R, R, R, s.hn, [&](int n, float acc) { s.g[n] = acc; });
            gemv_rows<GU>(P.p[R_WD_W], V, R, V, s.dbar, [&](int n, float acc) { s.g2[n] = acc; });
        }
        __syncthreads();
        for (int i = tid; i < R; i += nt) {
            const float gv = tanhf((s.g[i] + P.p[R_WH_B][i]) + s.g2[i]);
            s.g[i] = gv;
            tp.g[row * R + i] = gv;
            s.h[i] = s.hn[i];                                       // advance the GRU state
        }
        __syncthreads();
        {   // receiver message (model.py:454-475)
            const float* bw = P.p[R_W_B];
            const float* uw = ar.u_w ? ar.u_w + row * W : nullptr;
            gemv_rows<GU>(P.p[R_W_W], R, W, R, s.g, [&](int n, float acc) { s.w[n] = acc; });
            __syncthreads();
            for (int n = tid; n < W; n += nt) {
                const float lw = s.w[n] + bw[n];
                float wv = lw, lpv = 0.f, nev = 0.f;
                if (binary) {
                    const float p = sigmoidf_(lw);
                    if (train) {
                        const float u = uw ? uw[n] : philox_uniform(ar.seed, (uint32_t)((t * dm.Bg + gb) * W + n), mb_counter, 2u);
                        wv = (u < p) ? 1.f : 0.f;                  // model.py:460
                    } else {
                        wv = rintf(p);                             // model.py:462
                    }
                    tp.pw[row * W + n] = p;
                    const float l1 = logf(p + MMG_EPS), l0 = logf(1.f - p + MMG_EPS);
                    lpv = wv * l1 + (1.f - wv) * l0;
                    nev = p * l1 + (1.f - p) * l0;
                }
                s.w[n] = wv; s.lp[n] = lpv; s_ne[n] = nev;
                tp.w[row * W + n] = wv;
            }
        }
        __syncthreads();
        if (binary) {
            float lpv = 0.f, nev = 0.f;
            for (int j = tid; j < W; j += nt) { lpv += s.lp[j]; nev += s_ne[j]; }
            lpv = block_sum(lpv, s.misc + 8);
            nev = block_sum(nev, s.misc + 16);
            if (tid == 0) { tp.lp_w[row] = lpv; tp.ne_w[row] = nev; }
        }
        __syncthreads();
    }
    if (!do_rec) return;

    // ---- agent-level state hand-back ----
    if (ar.h_state) for (int i = tid; i < R; i += nt) ar.h_state[(size_t)b * R + i] = s.h[i];
    if (ar.sprod_state && tid == 0) ar.sprod_state[b] = s.misc[2];
    if (ar.t_end != T && s.misc[1] < 0.f) return;     // partial window without an output step
    if (ar.t_begin != 0) return;

    // ---- output selection, log-softmax, reward, top-k (model.py:1264-1275, 1333-1339) ----
    const int tstar = dm.fixed ? (T - 1) : (int)s.misc[1];
    if (dm.fixed) {
        __syncthreads();
        for (int d = tid; d < D; d += nt) s.yout[d] = tp.y[((size_t)(T - 1) * B + b) * D + d];
        __syncthreads();
    }
    if (tid == 0) tp.tstar[b] = tstar;
    float mx = -3.4e38f;
    for (int d = tid; d < D; d += nt) mx = fmaxf(mx, s.yout[d]);
    mx = block_max(mx, s.misc + 8);
    float se = 0.f;
    for (int d = tid; d < D; d += nt) se += expf(s.yout[d] - mx);
    se = block_sum(se, s.misc + 16);
    const float lse = mx + logf(se);
    const int tgt = ar.target ? (int)ar.target[b] : -1;
    const float dt = (tgt >= 0) ? (s.yout[tgt] - lse) : 0.f;
    float above = 0.f;
    for (int d = tid; d < D; d += nt) {
        const float o = s.yout[d], ld = o - lse;
        tp.outp[(size_t)b * D + d] = o;
        tp.dist[(size_t)b * D + d] = ld;
        tp.sm[(size_t)b * D + d] = expf(ld);
        if (tgt >= 0 && ld > dt) above += 1.f;
    }
    above = block_sum(above, s.misc + 8);
    if (tid == 0) {
        tp.logs[b] = dt;
        tp.hit[b] = (tgt >= 0 && above < (float)dm.top_k) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// k_baselines: grid (ceil(rows/16), 2).  blockIdx.y == 0: baseline_rec over [z_t || h_{t+1}]
// (model.py:842-843), == 1: baseline_sen over [h_x || z_r] (model.py:835-836) with the h_x part
// (Gs, incl. bias) added per sample.  Each workgroup owns 16 (step, sample) rows; its 4 waves take
// every 4th 16-wide tile of the K hidden units (fp32 MFMA), apply relu, store the hidden tile for
// the backward pass and reduce hidden . w2 to the score.
// ---------------------------------------------------------------------------------------------
struct BasArgs {
    int rows;                 // T*B (fused) or B (agent-level)
    const float* x1; int ld1, k1, mod1; // first K segment of the input (row index taken modulo mod1 if > 0)
    const float* x2; int ld2, k2;     // second K segment (NULL if none)
    const float* addend; int add_mod; // per-row pre-activation addend [rows % add_mod, K] (NULL if none)
    const float* W1; int ldw, col0;   // linear1.weight, row stride, first column used
    const float* b1;                  // NULL when folded into the addend
    const float* W2; const float* b2;
    float* hid;                       // [rows, K] or NULL
    float* score;                     // [rows]
};

__device__ __forceinline__ void bas_accumulate(f32x4& acc, const float* __restrict__ xrow, bool xv,
                                               const float* __restrict__ wrow, bool wv, int K, int q) {
    const bool vec = ((K & 15) == 0) && ((((uintptr_t)xrow) & 15) == 0) && ((((uintptr_t)wrow) & 15) == 0);
    // NB: `vec` must be wave-uniform: row strides are checked by the caller, bases differ only by rows
    if (vec) {
        for (int k = q * 4; k < K; k += 16) {
            float4 a = xv ? *reinterpret_cast<const float4*>(xrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 b = wv ? *reinterpret_cast<const float4*>(wrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc = mfma16(a.x, b.x, acc); acc = mfma16(a.y, b.y, acc);
            acc = mfma16(a.z, b.z, acc); acc = mfma16(a.w, b.w, acc);
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 4) {
            const int k = k0 + q;
            const float a = (xv && k < K) ? xrow[k] : 0.f;
            const float b = (wv && k < K) ? wrow[k] : 0.f;
            acc = mfma16(a, b, acc);
        }
    }
}

__global__ __launch_bounds__(MMG_BLOCK) void k_baselines(int K, BasArgs rec, BasArgs sen) {
    __shared__ float s_part[4][16];
    const BasArgs& A = (blockIdx.y == 0) ? rec : sen;
    const int row0 = blockIdx.x * 16;
    if (row0 >= A.rows) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int xrow = row0 + i;
    const bool xv = xrow < A.rows;
    const size_t xr = xv ? xrow : 0;
    // alignment of the vector path must hold for every row: require ld % 4 == 0, else force scalar by
    // passing a K that fails the (K & 15) test -- done here by checking the strides once.
    const bool ok1 = ((A.ld1 & 3) == 0) && ((A.ldw & 3) == 0) && ((A.col0 & 3) == 0);
    const bool ok2 = A.x2 && ((A.ld2 & 3) == 0) && ((A.ldw & 3) == 0) && (((A.col0 + A.k1) & 3) == 0);
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    const int ntiles = (K + 15) >> 4;
    for (int tn = wave; tn < ntiles; tn += 4) {
        const int n = tn * 16 + i;
        const bool nv = n < K;
        const float* wrow = A.W1 + (size_t)(nv ? n : 0) * A.ldw + A.col0;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const size_t xr1 = A.mod1 > 0 ? xr % (size_t)A.mod1 : xr;
        if (ok1) bas_accumulate(acc, A.x1 + xr1 * A.ld1, xv, wrow, nv, A.k1, q);
        else {
            for (int k0 = 0; k0 < A.k1; k0 += 4) {
                const int k = k0 + q;
                acc = mfma16((xv && k < A.k1) ? A.x1[xr1 * A.ld1 + k] : 0.f, (nv && k < A.k1) ? wrow[k] : 0.f, acc);
            }
        }
        if (A.x2) {
            if (ok2) bas_accumulate(acc, A.x2 + xr * A.ld2, xv, wrow + A.k1, nv, A.k2, q);
            else {
                for (int k0 = 0; k0 < A.k2; k0 += 4) {
                    const int k = k0 + q;
                    acc = mfma16((xv && k < A.k2) ? A.x2[xr * A.ld2 + k] : 0.f, (nv && k < A.k2) ? wrow[A.k1 + k] : 0.f, acc);
                }
            }
        }
        // epilogue: this lane holds rows q*4 + r, column i of the tile
        const int no = tn * 16 + i;
        const float bias = (no < K && A.b1) ? A.b1[no] : 0.f;
        const float w2 = (no < K) ? A.W2[no] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ro = row0 + q * 4 + r;
            if (ro < A.rows && no < K) {
                float v = acc[r] + bias;
                if (A.addend) v += A.addend[(size_t)(ro % A.add_mod) * K + no];
                v = fmaxf(v, 0.f);                                  // model.py:514
                if (A.hid) A.hid[(size_t)ro * K + no] = v;
                part[r] = fmaf(v, w2, part[r]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float v = part[r];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (i == 0) s_part[wave][q * 4 + r] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int ro = row0 + threadIdx.x;
        if (ro < A.rows)
            A.score[ro] = s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] +
                          s_part[3][threadIdx.x] + A.b2[0];         // model.py:515
    }
}

// ---------------------------------------------------------------------------------------------
// k_baselines2 (fused training path): grid (ceil(B/16), ceil(K/64), 2).  A workgroup owns 16 samples
// and 4 sixteen-wide tiles of the K hidden units (one per wave) and walks the steps t itself:
//   baseline_sen: the h_x . W1[:, :H]^T product (K = H, per SAMPLE, not per step) is accumulated once
//                 and re-used as the MFMA C operand of every step's z_r . W1[:, H:]^T product;
//   baseline_rec: [z_t || h_{t+1}] . W1^T per step.
// (base_ready: that product was already formed, as tape.basehx, by idle workgroups of the conversation launch.)
// Epilogue per step: relu, store the hidden tile (tape, for the backward pass), reduce hidden . w2
// over the block's 64 hidden units and write that PARTIAL score to part[t, b, blockIdx.y]; k_stats
// adds the ceil(K/64) partials and linear2.bias.  Steps beyond every sample's own last step are skipped.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void seg_accumulate(f32x4& acc, const float* __restrict__ xrow, bool xv,
                                               const float* __restrict__ wrow, bool wv, int K, int q, bool vec) {
    if (vec) {
        for (int k = q * 4; k < K; k += 16) {
            float4 a = xv ? *reinterpret_cast<const float4*>(xrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 b = wv ? *reinterpret_cast<const float4*>(wrow + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc = mfma16(a.x, b.x, acc); acc = mfma16(a.y, b.y, acc);
            acc = mfma16(a.z, b.z, acc); acc = mfma16(a.w, b.w, acc);
        }
    } else {
        for (int k0 = 0; k0 < K; k0 += 4) {
            const int k = k0 + q;
            acc = mfma16((xv && k < K) ? xrow[k] : 0.f, (wv && k < K) ? wrow[k] : 0.f, acc);
        }
    }
}

// fragment helpers for the vector path: a lane's share of one K segment is float4 chunks at k = q*4 + 16*j
template <int MAXQ>
__device__ __forceinline__ void frag_load(float4 (&f)[MAXQ], const float* __restrict__ row, bool valid, int K, int q) {
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) {
        const int k = q * 4 + 16 * j;
        f[j] = (valid && k < K) ? *reinterpret_cast<const float4*>(row + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int MAXQ>
__device__ __forceinline__ void frag_mfma(f32x4& acc, const float4 (&a)[MAXQ], const float4 (&b)[MAXQ], int K, int q) {
#pragma unroll
    for (int j = 0; j < MAXQ; ++j) {
        if (q * 4 + 16 * j < K || j * 16 < K) {          // wave-uniform bound: j*16 < K
            acc = mfma16(a[j].x, b[j].x, acc); acc = mfma16(a[j].y, b[j].y, acc);
            acc = mfma16(a[j].z, b[j].z, acc); acc = mfma16(a[j].w, b[j].w, acc);
        }
    }
}

__global__ __launch_bounds__(MMG_BLOCK) void k_baselines2(Dims dm, Params P, Tape tp, int skip_inactive, int base_ready) {
    // s_part[t][wave][row]: partial scores of this block's 64 hidden units, combined once at the end
    __shared__ float s_part[64][4][16];
    __shared__ int s_tmax;
    const int B = dm.B, H = dm.H, W = dm.W, R = dm.R, K = dm.K, T = dm.T;
    const int which = blockIdx.z & 1;                   // 0: baseline_rec, 1: baseline_sen
    const int tpart = blockIdx.z >> 1, tparts = gridDim.z >> 1;   // the steps are split over `tparts` workgroups
    const int b0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int npb = gridDim.y;
    const int tn = blockIdx.y * 4 + wave;
    const int n = tn * 16 + i;
    const bool nv = n < K;
    const int bx = b0 + i;                              // sample whose input row this lane loads
    const bool xv = bx < B;
    if (threadIdx.x == 0) {
        int tm = 0;
        for (int k = 0; k < 16 && b0 + k < B; ++k) tm = max(tm, tp.tstar[b0 + k]);
        s_tmax = (skip_inactive && !dm.fixed) ? tm : T - 1;
    }
    __syncthreads();
    const int tlast = s_tmax;
    const int tper = (tlast + tparts) / tparts;          // ceil((tlast + 1) / tparts)
    const int tbeg = tpart * tper, tmax = min(tlast, tbeg + tper - 1);
    if (tbeg > tlast) return;
#ifdef MMG_TIMING
#define MMG_B2STAMP(slot) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) tp.dbg[64 + 32 * which + (slot)] = (long long)wall_clock64(); } while (0)
#else
#define MMG_B2STAMP(slot) do {} while (0)
#endif
    MMG_B2STAMP(0);
    const float* W1 = which ? P.p[BS_L1_W] : P.p[BR_L1_W];
    const int ldw = which ? H + W : W + R;
    const float* wrow = W1 + (size_t)(nv ? n : 0) * ldw;
    const float bias = nv ? (which ? P.p[BS_L1_B][n] : P.p[BR_L1_B][n]) : 0.f;
    const float w2 = nv ? (which ? P.p[BS_L2_W][n] : P.p[BR_L2_W][n]) : 0.f;
    float* hid = which ? tp.hid_s : tp.hid_r;
    float* part = which ? tp.bs_part : tp.br_part;
    const bool vecH = ((H & 15) == 0) && ((ldw & 3) == 0);
    // register-fragment path: message and state segments of at most 64 floats, 16-byte aligned rows
    const bool frag = ((W & 15) == 0) && (W <= 64) && ((R & 15) == 0) && (R <= 64) && ((ldw & 3) == 0) && ((H & 3) == 0);
    const bool vecW = ((W & 15) == 0) && ((ldw & 3) == 0) && ((H & 3) == 0);
    const bool vecR = ((R & 15) == 0) && ((ldw & 3) == 0) && ((W & 3) == 0);
    f32x4 base = {0.f, 0.f, 0.f, 0.f};
    if (which && base_ready) {                           // formed by spare workgroups of the conversation launch (kernels_fast.h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int bo = b0 + q * 4 + r;
            base[r] = tp.basehx[(size_t)min(bo, B - 1) * K + min(n, K - 1)];
        }
    } else if (which) {
        if (vecH && H == 256) {                          // every operand load in flight at once, two MFMA chains
            const float* xr = tp.hx + (size_t)(xv ? bx : 0) * H;
            float4 a[16], bq[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float4 av = *reinterpret_cast<const float4*>(xr + u * 16 + q * 4);
                const float4 bv = *reinterpret_cast<const float4*>(wrow + u * 16 + q * 4);
                a[u] = make_float4(xv ? av.x : 0.f, xv ? av.y : 0.f, xv ? av.z : 0.f, xv ? av.w : 0.f);
                bq[u] = make_float4(nv ? bv.x : 0.f, nv ? bv.y : 0.f, nv ? bv.z : 0.f, nv ? bv.w : 0.f);
            }
            f32x4 b1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 16; u += 2) {
                base = mfma16(a[u].x, bq[u].x, base); b1 = mfma16(a[u + 1].x, bq[u + 1].x, b1);
                base = mfma16(a[u].y, bq[u].y, base); b1 = mfma16(a[u + 1].y, bq[u + 1].y, b1);
                base = mfma16(a[u].z, bq[u].z, base); b1 = mfma16(a[u + 1].z, bq[u + 1].z, b1);
                base = mfma16(a[u].w, bq[u].w, base); b1 = mfma16(a[u + 1].w, bq[u + 1].w, b1);
            }
            base += b1;
        } else {
            seg_accumulate(base, tp.hx + (size_t)(xv ? bx : 0) * H, xv, wrow, nv, H, q, vecH);
        }
    }
    const size_t xs = (size_t)(xv ? bx : 0);
    MMG_B2STAMP(1);

    if (frag) {
        // weights of the per-step segments stay in registers for the whole walk over t
        float4 w_msg[4], w_st[4];
        float4 xm[4][4], xt[4][4];                                        // ring of 4 steps of inputs: 3 steps in flight
        frag_load(w_msg, wrow + (which ? H : 0), nv, W, q);
        if (!which) frag_load(w_st, wrow + W, nv, R, q);
        const float* msg = which ? tp.zr : tp.z;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (tbeg + u <= tmax) {
                frag_load(xm[u], msg + ((size_t)(tbeg + u) * B + xs) * W, xv, W, q);
                if (!which) frag_load(xt[u], tp.h + ((size_t)(tbeg + u + 1) * B + xs) * R, xv, R, q);
            }
        }
        for (int t0 = tbeg; t0 <= tmax; t0 += 4) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int t = t0 + u;
            if (t > tmax) break;
            MMG_B2STAMP(2 + t);
            if (t + 3 <= tmax) {                                          // keep three steps of inputs in flight
                frag_load(xm[(u + 3) & 3], msg + ((size_t)(t + 3) * B + xs) * W, xv, W, q);
                if (!which) frag_load(xt[(u + 3) & 3], tp.h + ((size_t)(t + 4) * B + xs) * R, xv, R, q);
            }
            f32x4 acc = base, acc2 = {0.f, 0.f, 0.f, 0.f};
            frag_mfma(acc, xm[u], w_msg, W, q);
            if (!which) { frag_mfma(acc2, xt[u], w_st, R, q); acc += acc2; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int bo = b0 + q * 4 + r;
                float v = fmaxf(acc[r] + bias, 0.f);                        // model.py:514
                if (bo < B && nv) hid[((size_t)t * B + bo) * K + n] = v; else v = 0.f;
                v = dpp_group_sum<16>(v * w2);
                if (i == 0) s_part[t][wave][q * 4 + r] = v;
            }
          }
        }
    } else {
        for (int t = tbeg; t <= tmax; ++t) {
            const size_t xrow = (size_t)t * B + xs;
            f32x4 acc = base;
            if (which) {
                seg_accumulate(acc, tp.zr + xrow * W, xv, wrow + H, nv, W, q, vecW);
            } else {
                seg_accumulate(acc, tp.z + xrow * W, xv, wrow, nv, W, q, vecW);
                seg_accumulate(acc, tp.h + ((size_t)(t + 1) * B + xs) * R, xv, wrow + W, nv, R, q, vecR);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int bo = b0 + q * 4 + r;
                float v = fmaxf(acc[r] + bias, 0.f);
                if (bo < B && nv) hid[((size_t)t * B + bo) * K + n] = v; else v = 0.f;
                v = dpp_group_sum<16>(v * w2);
                if (i == 0) s_part[t][wave][q * 4 + r] = v;
            }
        }
    }
    MMG_B2STAMP(20);
    __syncthreads();
    for (int idx = threadIdx.x; idx < (tmax - tbeg + 1) * 16; idx += MMG_BLOCK) {
        const int t = tbeg + (idx >> 4), r = idx & 15;
        if (b0 + r < B)
            part[((size_t)t * B + b0 + r) * npb + blockIdx.y] =
                (s_part[t][0][r] + s_part[t][1][r]) + (s_part[t][2][r] + s_part[t][3][r]);
    }
    MMG_B2STAMP(21);
}

// ---------------------------------------------------------------------------------------------
// k_baselines3: both baselines over the LIVE (step, sample) rows only -- rows (t, b) with t <= t*(b), in (t, b) order,
// the same list build_row_map (kernels_bwd.h) leaves for k_wgrad.  With early stopping most of the T*B rows are dead
// (config 2: ~140 of 640 live), and k_baselines2 walks every step up to the longest conversation of its 16 samples.
// grid (ceil(T*B/16), ceil(K/64), 2): a workgroup owns 16 live rows x 64 hidden units of one baseline -- one MFMA pass,
// no time loop; workgroups whose 16-row window lies beyond the live rows return after counting.
// Needs B <= 64 (one lane per sample when listing rows), W and R multiples of 16 up to 64, tape.basehx.
// Same operand order as k_baselines2, so hidden tiles and partial scores are bit-identical to it.
// ---------------------------------------------------------------------------------------------
// ROLE: the body runs as a workgroup role of the backward launch (kernels_fast.h): partial scores go out as write-through
// stores and the role counts itself on the (baseline, hidden block) counter `done` -- only windows inside the live rows do;
// the statistics roles derive the number of such windows from t* themselves.
template <bool ROLE>
__device__ __forceinline__ void baselines3_body(const Dims& dm, const Params& P, const Tape& tp, int window, int byi, int which, int npb) {
    __shared__ int s_rid[16];
    __shared__ float s_part[4][16];
    const int B = dm.B, H = dm.H, W = dm.W, R = dm.R, K = dm.K, T = dm.T;
    const int lo = window * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, q = lane >> 4;
    const int ts = tp.tstar[min(lane, B - 1)];           // (first: the row list depends on it)
    const int n = (byi * 4 + wave) * 16 + i;
    const bool nv = n < K;
    const float* W1 = which ? P.p[BS_L1_W] : P.p[BR_L1_W];
    const int ldw = which ? H + W : W + R;
    const float* wrow = W1 + (size_t)(nv ? n : 0) * ldw;
    const float bias = nv ? (which ? P.p[BS_L1_B][n] : P.p[BR_L1_B][n]) : 0.f;
    const float w2 = nv ? (which ? P.p[BS_L2_W][n] : P.p[BR_L2_W][n]) : 0.f;
    float4 w_msg[4], w_st[4];
    frag_load(w_msg, wrow + (which ? H : 0), nv, W, q);
    if (!which) frag_load(w_st, wrow + W, nv, R, q);
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
    const bool xv = rid >= 0;
    const size_t rr = (size_t)(xv ? rid : 0);
    float4 xm[4], xt[4];
    frag_load(xm, (which ? tp.zr : tp.z) + rr * W, xv, W, q);
    if (!which) frag_load(xt, tp.h + (rr + B) * R, xv, R, q);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
    int orow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        orow[r] = s_rid[q * 4 + r];
        if (which) acc[r] = (orow[r] >= 0) ? tp.basehx[(size_t)(orow[r] % B) * K + min(n, K - 1)] : 0.f;
    }
    frag_mfma(acc, xm, w_msg, W, q);
    if (!which) { frag_mfma(acc2, xt, w_st, R, q); acc += acc2; }
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
        const float v = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
        if (ROLE) {                // roles of a larger launch: its statistics roles spin on (value, epoch) pairs (kernels_bwd.h: combine_score_ll)
            part[(size_t)s_rid[threadIdx.x] * npb + byi] = v;
            st_ll(tp.partll, ((size_t)(which ? 0 : 1) * T * B + (size_t)s_rid[threadIdx.x]) * npb + byi, v, tp.counter[3]);
        } else part[(size_t)s_rid[threadIdx.x] * npb + byi] = v;
    }
}

__global__ __launch_bounds__(MMG_BLOCK) void k_baselines3(Dims dm, Params P, Tape tp) {
    baselines3_body<false>(dm, P, tp, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, (int)gridDim.y);
}

}  // namespace mmg
