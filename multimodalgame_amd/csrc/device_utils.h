// device_utils.h -- wave64 helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// clamp(a * k + c) to [0, 1] on two floats at once: v_pk_fma_f32 with the clamp output modifier (k: one scalar for both halves)
__device__ __forceinline__ f32x2 pk_fma_clamp(f32x2 a, float k, f32x2 c) {
    f32x2 r;
    const f32x2 k2 = {k, k};
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(k2), "v"(c));
    return r;
}
// Pointers a kernel reads out of a TABLE in memory (k_wgrad's job table) are generic to the compiler, which then emits flat_load
// with 64-bit address arithmetic per access; these casts state what the host guarantees (device memory): global_load, half the
// address instructions.
typedef __attribute__((address_space(1))) const float gfloat;
typedef __attribute__((address_space(1))) const f32x4 gf32x4;
__device__ __forceinline__ gfloat* as_global(const float* p) { return (gfloat*)p; }
typedef __attribute__((address_space(1))) float gfloat_w;
__device__ __forceinline__ gfloat_w* as_global_w(float* p) { return (gfloat_w*)p; }
__device__ __forceinline__ float4 ldg4(gfloat* p) {
    const f32x4 v = *(gf32x4*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

#define MMG_BLOCK 256          // 4 wave64 per workgroup everywhere
#define MMG_EPS 1e-8f          // the reference's log(p + 1e-8)   model.py:908

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// max(a, b) of the class-logit inner loops as ONE v_max_f32, by inline assembly: IEEE fmaxf canonicalises both operands when
// they come straight from memory (three v_max per relu), and so does every builtin the compiler understands -- the former
// v_med3_f32(a, b, +inf) formulation was folded back into max + two canonicalising v_max (ISA of round 4: 291 v_max in
// k_conversation_mc for 96 relus).  Operands are finite by construction (sums of products of finite weights / activations).
__device__ __forceinline__ float fmax_nn(float a, float b) {
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- DPP cross-lane adds: a VALU operand modifier, no LDS round trip (ds_bpermute costs ~60+ cycles
// on a dependent chain).  quad_perm swaps inside quads, row_half_mirror / row_mirror fold 8 / 16 lanes.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
#define MMG_DPP_QUAD_1032 0xB1
#define MMG_DPP_QUAD_2301 0x4E
#define MMG_DPP_ROW_HALF_MIRROR 0x141
#define MMG_DPP_ROW_MIRROR 0x140

// sum over aligned groups of N lanes (N = 2, 4, 8, 16): every lane of the group gets the total
template <int N>
__device__ __forceinline__ float dpp_group_sum(float v) {
    static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16, "DPP group sums cover up to one row of 16 lanes");
    if (N >= 2) v += dpp_f<MMG_DPP_QUAD_1032>(v);
    if (N >= 4) v += dpp_f<MMG_DPP_QUAD_2301>(v);
    if (N >= 8) v += dpp_f<MMG_DPP_ROW_HALF_MIRROR>(v);
    if (N >= 16) v += dpp_f<MMG_DPP_ROW_MIRROR>(v);
    return v;
}
// Full-wave reductions: rows of 16 by the mirror steps above, then the classic GFX9 row broadcasts (lane 15 of a row into
// the next row, lane 31 into rows 2-3) leave the total in lane 63, read back through an SGPR -- no ds_bpermute.
#define MMG_DPP_ROW_BCAST15 0x142
#define MMG_DPP_ROW_BCAST31 0x143
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_rows_f(float v, float old) {    // rows outside ROWS keep `old`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false));
}
__device__ __forceinline__ float dpp_wave_sum(float v) {            // all 64 lanes get the total (wave-uniform)
    v = dpp_group_sum<16>(v);
    v += dpp_rows_f<MMG_DPP_ROW_BCAST15, 0xA>(v, 0.f);
    v += dpp_rows_f<MMG_DPP_ROW_BCAST31, 0xC>(v, 0.f);
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float dpp_wave_max(float v) {
    v = fmaxf(v, dpp_f<MMG_DPP_QUAD_1032>(v)); v = fmaxf(v, dpp_f<MMG_DPP_QUAD_2301>(v));
    v = fmaxf(v, dpp_f<MMG_DPP_ROW_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f<MMG_DPP_ROW_MIRROR>(v));
    v = fmaxf(v, dpp_rows_f<MMG_DPP_ROW_BCAST15, 0xA>(v, v));
    v = fmaxf(v, dpp_rows_f<MMG_DPP_ROW_BCAST31, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_d(double v) {                  // 64-bit value moved as two DPP dwords; other rows: 0.0
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, ROWS, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, ROWS, 0xF, false);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ double dpp_wave_sum_d(double v) {        // same tree as dpp_wave_sum, in f64
    v += dpp_d<MMG_DPP_QUAD_1032, 0xF>(v); v += dpp_d<MMG_DPP_QUAD_2301, 0xF>(v);
    v += dpp_d<MMG_DPP_ROW_HALF_MIRROR, 0xF>(v); v += dpp_d<MMG_DPP_ROW_MIRROR, 0xF>(v);
    v += dpp_d<MMG_DPP_ROW_BCAST15, 0xA>(v); v += dpp_d<MMG_DPP_ROW_BCAST31, 0xC>(v);
    const long long bits = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)bits, 63), hi = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), 63);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}

// wave-level sum over aligned groups of G lanes (G power of two, <= 64, wave-uniform): lane 0 of every group -- in fact
// every lane of it -- gets the group's total.  DPP only (see above); all 64 lanes must be active.
__device__ __forceinline__ float group_sum(float v, int G) {
    if (G >= 2) v += dpp_f<MMG_DPP_QUAD_1032>(v);
    if (G >= 4) v += dpp_f<MMG_DPP_QUAD_2301>(v);
    if (G >= 8) v += dpp_f<MMG_DPP_ROW_HALF_MIRROR>(v);
    if (G >= 16) v += dpp_f<MMG_DPP_ROW_MIRROR>(v);
    if (G >= 32) {
        v += dpp_rows_f<MMG_DPP_ROW_BCAST15, 0xA>(v, 0.f);            // rows 1 and 3 now hold the sums of lanes 0-31 / 32-63
        const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
        const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
        v = (G == 64) ? lo + hi : (((threadIdx.x & 63) < 32) ? lo : hi);
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) { return group_sum(v, 64); }
__device__ __forceinline__ float wave_max(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// block-wide reductions through a small LDS scratch (>= 8 floats); every thread gets the result.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// ---------------------------------------------------------------------------------------------
// Row-major GEMV, lanes along K:  for n < N:  epi(n, sum_k W[n*ld + k] * v[k])
// W is the PyTorch [out,in] weight in global memory (L2-resident), v lives in LDS (16-byte
// aligned).  G = lanes per row (power of two covering K/4 float4 items), 64/G rows per wave pass,
// so each wave-wide load instruction covers contiguous rows.  The epilogue runs on one lane per
// row.  Falls back to scalar loads when the row stride / base are not 16-byte aligned.
// No barrier inside; callers __syncthreads() before consuming what the epilogue stored.
// ---------------------------------------------------------------------------------------------
template <int U = 4, class Epi>
__device__ __forceinline__ void gemv_rows(const float* __restrict__ Wm, int ld, int N, int K,
                                          const float* v, Epi epi) {
    // U row passes are loaded together (row indices clamped, not predicated, so the loads stay branch-free and the
    // compiler keeps counted waits): U float4 per lane in flight instead of one L2 round trip per pass.  Each row is
    // still summed by the same lanes in the same order as a one-pass-at-a-time loop would.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const bool vec = ((ld & 3) == 0) && ((K & 3) == 0) && ((((uintptr_t)Wm) & 15) == 0);
    const int items = vec ? (K >> 2) : K;
    int G = 1;
    while (G < 64 && G < items) G <<= 1;
    const int rpw = 64 / G, sub = lane / G, gl = lane - sub * G;
    const int stride = nw * rpw;                        // rows the workgroup covers per pass
    for (int n0 = wave * rpw + sub; n0 < N + sub; n0 += stride * U) {    // (n0 - sub) is wave-uniform
        float acc[U];
        size_t roff[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { acc[u] = 0.f; roff[u] = (size_t)min(n0 + u * stride, N - 1) * ld; }
        if (vec) {
            const float4* v4 = reinterpret_cast<const float4*>(v);
            for (int k = gl; k < items; k += G) {
                float4 a[U];
#pragma unroll
                for (int u = 0; u < U; ++u) a[u] = reinterpret_cast<const float4*>(Wm + roff[u])[k];
                const float4 bq = v4[k];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc[u] = fmaf(a[u].x, bq.x, acc[u]); acc[u] = fmaf(a[u].y, bq.y, acc[u]);
                    acc[u] = fmaf(a[u].z, bq.z, acc[u]); acc[u] = fmaf(a[u].w, bq.w, acc[u]);
                }
            }
        } else {
            for (int k = gl; k < items; k += G) {
                float a[U];
#pragma unroll
                for (int u = 0; u < U; ++u) a[u] = Wm[roff[u] + k];
                const float bq = v[k];
#pragma unroll
                for (int u = 0; u < U; ++u) acc[u] = fmaf(a[u], bq, acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = n0 + u * stride;
            const float r = group_sum(acc[u], G);
            if (gl == 0 && n < N) epi(n, r);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Transposed GEMV, lanes along the output:  out[k] (+)= sum_n W[n*ld + k] * d[n],  k < K.
// d and out in LDS; `scratch` >= 4 * blockDim.x floats, 16-byte aligned.  Ends with a barrier (out is ready).
// ---------------------------------------------------------------------------------------------
template <int UN = 8>
__device__ __forceinline__ void gemv_t(const float* __restrict__ Wm, int ld, int N, int K,
                                       const float* d, float* out, float* scratch, bool accumulate) {
    // Rows n are walked UN at a time with every load of the group issued before the first FMA;
    // the FMA order per output is the plain n = 0, 1, 2, ... order.
    const int tid = threadIdx.x, nt = blockDim.x;
    const bool vec = ((ld & 3) == 0) && ((K & 3) == 0) && ((((uintptr_t)Wm) & 15) == 0) && ((((uintptr_t)out) & 15) == 0);
    if (vec && (K >> 2) >= nt) {                        // wide outputs: a lane owns 4 consecutive outputs per pass
        const int K4 = K >> 2;
        for (int k4 = tid; k4 < K4; k4 += nt) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int n = 0; n < N; n += UN) {
                float4 w[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) w[u] = reinterpret_cast<const float4*>(Wm + (size_t)min(n + u, N - 1) * ld)[k4];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const float dv = (n + u < N) ? d[min(n + u, N - 1)] : 0.f;
                    acc.x = fmaf(w[u].x, dv, acc.x); acc.y = fmaf(w[u].y, dv, acc.y);
                    acc.z = fmaf(w[u].z, dv, acc.z); acc.w = fmaf(w[u].w, dv, acc.w);
                }
            }
            float4* o4 = reinterpret_cast<float4*>(out) + k4;
            if (accumulate) { const float4 o = *o4; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
            *o4 = acc;
        }
        __syncthreads();
        return;
    }
    if (K >= nt) {
        for (int k = tid; k < K; k += nt) {
            float acc = 0.f;
            for (int n = 0; n < N; n += UN) {
                float w[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) w[u] = Wm[(size_t)min(n + u, N - 1) * ld + k];
#pragma unroll
                for (int u = 0; u < UN; ++u) acc = fmaf(w[u], (n + u < N) ? d[min(n + u, N - 1)] : 0.f, acc);
            }
            out[k] = accumulate ? out[k] + acc : acc;
        }
        __syncthreads();
        return;
    }
    if (vec) {                                          // narrow outputs: K/4 lanes per part, nt / (K/4) parts over n
        const int K4 = K >> 2, parts = nt / K4;
        const int part = tid / K4, k4 = tid - part * K4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (part < parts) {
            for (int n = part; n < N; n += parts * UN) {
                float4 w[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) w[u] = reinterpret_cast<const float4*>(Wm + (size_t)min(n + u * parts, N - 1) * ld)[k4];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const float dv = (n + u * parts < N) ? d[min(n + u * parts, N - 1)] : 0.f;
                    acc.x = fmaf(w[u].x, dv, acc.x); acc.y = fmaf(w[u].y, dv, acc.y);
                    acc.z = fmaf(w[u].z, dv, acc.z); acc.w = fmaf(w[u].w, dv, acc.w);
                }
            }
            reinterpret_cast<float4*>(scratch)[tid] = acc;
        }
        __syncthreads();
        if (tid < K4) {
            float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p = 0; p < parts; ++p) {
                const float4 q = reinterpret_cast<const float4*>(scratch)[p * K4 + tid];
                sacc.x += q.x; sacc.y += q.y; sacc.z += q.z; sacc.w += q.w;
            }
            float4* o4 = reinterpret_cast<float4*>(out) + tid;
            if (accumulate) { const float4 o = *o4; sacc.x += o.x; sacc.y += o.y; sacc.z += o.z; sacc.w += o.w; }
            *o4 = sacc;
        }
        __syncthreads();
        return;
    }
    const int parts = nt / K;
    const int part = tid / K, k = tid - part * K;
    if (part < parts) {
        float acc = 0.f;
        for (int n = part; n < N; n += parts * UN) {
            float w[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) w[u] = Wm[(size_t)min(n + u * parts, N - 1) * ld + k];
#pragma unroll
            for (int u = 0; u < UN; ++u) acc = fmaf(w[u], (n + u * parts < N) ? d[min(n + u * parts, N - 1)] : 0.f, acc);
        }
        scratch[tid] = acc;
    }
    __syncthreads();
    if (tid < K) {
        float sacc = 0.f;
        for (int p = 0; p < parts; ++p) sacc += scratch[p * K + tid];
        out[tid] = accumulate ? out[tid] + sacc : sacc;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), one call per Bernoulli draw: counter = (element index,
// minibatch counter, stream id, 0), key = 64-bit seed.  Returns a uniform in [0,1) with 24 bits.
// tests/philox_ref.py holds the numpy restatement used to feed the oracle the same numbers.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * x0, p1 = (uint64_t)0xCD9E8D57u * x2;
        const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0, y1 = (uint32_t)p1;
        const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1, y3 = (uint32_t)p0;
        x0 = y0; x1 = y1; x2 = y2; x3 = y3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(x0 >> 8) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------------------------------------
// Dependencies between workgroup ROLES of one launch.  Several small kernels of the minibatch are latency
// bound (4-7 us each, of which only 1-3 us is work); merging producer and consumer roles into one grid lets the
// consumer's prologue (weights -> registers) run while the producer works, and replaces a kernel boundary by one
// device-scope release/acquire pair.  Producers always have LOWER workgroup ids than their consumers and never
// wait on a higher id, so with the in-order workgroup dispatch of the hardware the lowest unfinished workgroup can
// always run: no deadlock.  The spin is bounded all the same: on expiry the error word sync[511] is set and the
// workgroup continues (wrong numbers flagged, never a hung GPU).
//   sync[128k] = arrivals of dependency k, sync[128k+64] = consumers that have passed it (the last one re-arms both).
// ---------------------------------------------------------------------------------------------
#define MMG_SYNC_ERR 511
#define MMG_SYNC_ERR_REMOTE 1001u   // k_opt: another rank of the data-parallel job reported a timed-out dependency
// every counter in its own 256-byte block (pollers of one dependency do not queue behind the increments of another)
#define MMG_SYNC_ARR(dep) (128 * (dep))
#define MMG_SYNC_PASS(dep) (128 * (dep) + 64)
#define MMG_SPIN_LIMIT (1 << 22)
__device__ __forceinline__ void role_signal(uint32_t* sync, int dep) {
    // A device-scope release is an L2 write-back (buffer_wbl2) and those serialise across the chip (~0.1 us each: 971
    // signalling workgroups once cost 100 us).  So: every wave only waits for its own stores to reach L2 (vmcnt),
    // and ONE wave per workgroup then releases at device scope on behalf of all of them.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (a workgroup-scope fence does not wait for global stores)
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(sync + MMG_SYNC_ARR(dep), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// Same for producers that wrote everything the consumers will read with device-scope (write-through) stores
// (__hip_atomic_store, agent scope): nothing to write back, the stores only have to have completed.
__device__ __forceinline__ void role_signal_wt(uint32_t* sync, int dep) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(sync + MMG_SYNC_ARR(dep), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ACQ = false: the caller reads what the producers wrote with agent-scope loads (__hip_atomic_load, relaxed) -- coherent by
// themselves -- and skips the acquire fence (a buffer_inv of the L2: measured at config 4, 40 us per conversation of 7 steps).
// The consumers count themselves; the last one re-arms both counters for the next launch.  This is a RETURNING atomic (a memory
// round trip, ~1 us): role_wait<.., REARM = false> leaves it to the caller, who runs role_rearm (one thread) where nothing waits
// for it -- at the end of the role.
__device__ __forceinline__ void role_rearm(uint32_t* sync, int dep, uint32_t consumers) {
    const uint32_t passed = __hip_atomic_fetch_add(sync + MMG_SYNC_PASS(dep), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (passed + 1 == consumers) {                      // everyone has seen the final count
        __hip_atomic_store(sync + MMG_SYNC_ARR(dep), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + MMG_SYNC_PASS(dep), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int SLEEP = 1, bool ACQ = true, bool REARM = true>
__device__ __forceinline__ void role_wait(uint32_t* sync, int dep, uint32_t producers, uint32_t consumers) {
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(sync + MMG_SYNC_ARR(dep), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < producers) {
            __builtin_amdgcn_s_sleep(SLEEP);
            if (++spins > MMG_SPIN_LIMIT) { __hip_atomic_store(sync + MMG_SYNC_ERR, (uint32_t)(dep + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        if (REARM) role_rearm(sync, dep, consumers);
    }
    __syncthreads();
    if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop stale cache lines before reading what the producers wrote
}

// agent-scope (write-through) store: payload another workgroup of the same launch reads after a counter hand-off
__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... 16 bytes at once (p 16-byte aligned): the agent-scope store of gfx942 / gfx950 is a global store with sc1 set
__device__ __forceinline__ void st_wt4(float* p, float4 v) {
    const f32x4 x = {v.x, v.y, v.z, v.w};
    // (the trailing s_nop: on gfx9-family chips a VMEM store of more than 64 bits must not be followed directly by a VALU write of
    //  its data VGPRs -- the compiler's hazard recogniser inserts the wait state for its own stores, not behind inline assembly.
    //  Found in round 5: kernels_game.h read back address bits in the first two floats of such a store.)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(x) : "memory");
}
// ... the same 16 bytes as two 8-byte agent-scope stores through the compiler's own atomics (no inline assembly)
__device__ __forceinline__ void st_wt4c(float* p, float4 v) {
    const unsigned long long lo = ((unsigned long long)__builtin_bit_cast(unsigned, v.y) << 32) | __builtin_bit_cast(unsigned, v.x);
    const unsigned long long hi = ((unsigned long long)__builtin_bit_cast(unsigned, v.w) << 32) | __builtin_bit_cast(unsigned, v.z);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// agent-scope (sc1) loads of such a payload: coherent across the XCDs' L2s by themselves, so the consumer needs NO acquire fence
// (an agent-scope acquire is a buffer_inv of the whole L2: everything the role reads afterwards comes from memory again)
__device__ __forceinline__ float ld_cc(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float2 ld_cc2(const float* p) {
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__builtin_bit_cast(float, (unsigned)u), __builtin_bit_cast(float, (unsigned)(u >> 32)));
}
__device__ __forceinline__ float4 ld_cc4(const float* p) { const float2 a = ld_cc2(p), b = ld_cc2(p + 2); return make_float4(a.x, a.y, b.x, b.y); }

// (value, epoch) pairs: ONE aligned 8-byte write-through store / agent-scope load each, so a reader of the same launch sees the
// value together with its tag or not at all -- the consumer spins on the payload itself (one memory round trip once it has
// landed; no counter, no store-completion wait at the producer, nothing to re-arm: an epoch is used by one launch only)
__device__ __forceinline__ void st_ll(float* ll, size_t i, float v, uint32_t epoch) {
    const unsigned long long u = ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(ll) + i, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// two consecutive pairs (i even) in ONE 16-byte write-through store: (v0, epoch, v1, epoch); each 8-byte pair lands whole
__device__ __forceinline__ void st_ll2(float* ll, size_t i, float v0, float v1, uint32_t epoch) {
    const float e = __builtin_bit_cast(float, epoch);
    st_wt4(ll + 2 * i, make_float4(v0, e, v1, e));
}
// ... the same pair kept INSIDE the XCD's L2 (round 6): a plain 8-byte store goes through the CU's write-through L1 into the L2 of
// ITS XCD and stays there (written back when the line is evicted or the launch ends); the agent-scope (sc1) loads of consumers on
// the SAME XCD hit it there (scripts/micro/hop_latency.hip: "a plain store is visible to sc1 loads of the same XCD only").  Only
// for hand-offs whose producer and consumers the launch places on one XCD, and only when mmg_create's probe (k_xcc_probe)
// confirmed that placement rule on this device: the payload then never reaches the HBM-side counters.
__device__ __forceinline__ void st_ll_l2(float* ll, size_t i, float v, uint32_t epoch) {
    const unsigned long long u = ((unsigned long long)epoch << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(ll) + i, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the XCD (XCC) this wave runs on: HW_REG_XCC_ID (id 20), bits [3:0]
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 0xfu; }
__device__ __forceinline__ unsigned long long ld_ll(const float* ll, size_t i) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(ll) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ll_value(unsigned long long u) { return __builtin_bit_cast(float, (unsigned)u); }
__device__ __forceinline__ bool ll_fresh(unsigned long long u, uint32_t epoch) { return (uint32_t)(u >> 32) == epoch; }

// four consecutive pairs as one float4, spinning (bounded) until all four carry this launch's epoch.  On expiry the error word
// sync[MMG_SYNC_ERR] is set like every other hand-off of the library does (k_opt then skips the update on every rank and every
// later training call fails); the budget matches the sample roles' (1 << 16 rounds of 18 loads there, 4 loads a round here).
#define MMG_LL_ERR_BASEHX 6u
__device__ __forceinline__ float4 ll_wait4(const float* ll, size_t i, uint32_t epoch, uint32_t* sync) {
    unsigned long long u0, u1, u2, u3;
    int spins = 0;
    for (;;) {
        u0 = ld_ll(ll, i); u1 = ld_ll(ll, i + 1); u2 = ld_ll(ll, i + 2); u3 = ld_ll(ll, i + 3);
        if (ll_fresh(u0, epoch) && ll_fresh(u1, epoch) && ll_fresh(u2, epoch) && ll_fresh(u3, epoch)) break;
        if (++spins > (1 << 18)) { if (sync) __hip_atomic_store(sync + MMG_SYNC_ERR, MMG_LL_ERR_BASEHX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    return make_float4(ll_value(u0), ll_value(u1), ll_value(u2), ll_value(u3));
}

// one 16x16x4 fp32 MFMA step:  D += A(16x4) * B(4x16);  lane l holds A[l&15][l>>4], B[l>>4][l&15],
// D[(l>>4)*4 + reg][l&15]   (cdna_hip_programming.md §3)
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

}  // namespace mmg
