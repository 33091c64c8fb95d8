// Hand-off latency between two workgroups of ONE launch on MI355X: same XCD against different XCDs, by store / load cache policy.
// A ping-pong of (value, tag) pairs, 64-bit each: workgroup 0 stores tag i, its partner polls until it sees it and answers.
// Reported: round trip / 2 = one hop (store issue -> value in the consumer's register).   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;

template <int ST> __device__ __forceinline__ void st64(u64* p, u64 v) {
    if (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
}
// (issue only: the caller waits once for all loads of a poll round)
template <int LD> __device__ __forceinline__ u64 ld64(const u64* p) {
    u64 v;
    if (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (LD == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (LD == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf); }

// payload: NPL pairs per lane of 64 lanes (the hand-off's size).  FLAG = false: the poller spins on ALL of them (the payload is
// its own flag).  FLAG = true: the producer waits for its payload stores to complete, then bumps a flag word; the consumer polls the
// flag with one lane and then loads the payload (what pf_signal / pf_wait of kernels_tile.h do).
template <int ST, int LD, int NPL, bool FLAG, int MODE = 0>
__global__ void k_pingpong(u64* buf, int partner, int iters, int npl, long long* out, int* xcc, int* err, int noise) {
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    if (threadIdx.x >= 64 && (b == 0 || b == partner)) return;
    if (lane == 0) xcc[b] = xcc_id();
    if (b != 0 && b != partner) {
        if (!noise) return;
        const u64* reg = buf + 2 * 64 * 64 + 128 + (size_t)(b % 16) * 8 * 64;      // 16 regions: the pollers of a tile share their lines
        for (int r = 0; r < 4000000; ++r) {
            u64 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = ld64<LD>(reg + k * 64 + lane);
            wait_loads();
            u64 x = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) x |= v[k];
            if (x != 0) break;                       // workgroup 0 writes every region when it is through
            if (noise == 4) __builtin_amdgcn_s_sleep(4); else if (noise == 16) __builtin_amdgcn_s_sleep(16);
        }
        return;
    }
    u64* ping = buf; u64* pong = buf + 64 * 64;            // [npl][64]
    const bool me0 = b == 0;
    long long t0 = 0;
    for (int i = 1; i <= iters; ++i) {
        if (i == 11 && me0) t0 = wall_clock64();
        u64* fping = buf + 2 * 64 * 64; u64* fpong = fping + 64;
        auto publish = [&](u64* dst, u64* flag) {
#pragma unroll
            for (int k = 0; k < NPL; ++k) if (MODE != 2 || k == 0) st64<ST>(dst + k * 64 + lane, ((u64)i << 32) | (u64)(k + lane));
            if (FLAG) { wait_loads(); if (lane == 0) st64<ST>(flag, (u64)i << 32); }
        };
        if (me0) publish(ping, fping);
        const u64* src = me0 ? pong : ping; const u64* fsrc = me0 ? fpong : fping;
        int spins = 0;
        if (FLAG) {
            for (;;) {
                u64 f = 0;
                if (lane == 0) { f = ld64<LD>(fsrc); wait_loads(); }
                if (__shfl(f, 0) >> 32 == (u64)i) break;
                if (++spins > 200000) { if (lane == 0) *err = i; return; }
            }
        }
        for (;;) {
            u64 v[NPL];
#pragma unroll
            for (int k = 0; k < NPL; ++k) if (MODE != 1 || k == NPL - 1) v[k] = ld64<LD>(src + k * 64 + lane);
            wait_loads();
            bool fresh = true;
#pragma unroll
            for (int k = 0; k < NPL; ++k) if (MODE == 0 || (MODE == 1 && k == NPL - 1) || (MODE == 2 && k == 0)) fresh = fresh && (v[k] >> 32) == (u64)i;
            if (MODE == 2) { u64 x = 0;
#pragma unroll
                for (int k = 1; k < NPL; ++k) x ^= v[k];
                if (x == 0x123456789abcdefull) *err = -1; }
            if (!__any(!fresh)) break;
            if (++spins > 200000) { if (lane == 0) *err = i; return; }
        }
        if (!me0) publish(pong, fpong);
    }
    if (me0 && lane == 0) out[0] = wall_clock64() - t0;
    if (me0 && noise) for (int k = 0; k < 16 * 8; ++k) st64<1>(buf + 2 * 64 * 64 + 128 + (size_t)k * 64 + lane, 1ull);
}

template <int ST, int LD, int NPL, bool FLAG, int MODE = 0>
static void run(const char* name, int partner, int noise = 0, int grid = 64) {
    const int npl = NPL;
    u64* buf; long long* out; int *xcc, *err;
    hipMalloc(&buf, (2 * 64 * 64 + 128 + 16 * 8 * 64) * sizeof(u64)); hipMemset(buf, 0, (2 * 64 * 64 + 128 + 16 * 8 * 64) * sizeof(u64));
    hipMalloc(&out, 8); hipMalloc(&xcc, 64 * 4); hipMalloc(&err, 4); hipMemset(err, 0, 4); hipMemset(out, 0, 8);
    const int iters = 510;
    hipLaunchKernelGGL((k_pingpong<ST, LD, NPL, FLAG, MODE>), dim3(grid), dim3(64 * (noise ? 8 : 1)), 0, 0, buf, partner, iters, npl, out, xcc, err, noise);
    hipDeviceSynchronize();
    long long o; int e; std::vector<int> x(64);
    hipMemcpy(&o, out, 8, hipMemcpyDeviceToHost); hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); hipMemcpy(x.data(), xcc, 64 * 4, hipMemcpyDeviceToHost);
    if (e) printf("%-44s partner %2d (XCD %d -> %d) %2d pairs/lane%s: NEVER SEEN (stuck at round %d)\n", name, partner, x[0], x[partner], npl, FLAG ? " + flag" : "", e);
    else printf("%-44s partner %2d (XCD %d -> %d) %2d pairs/lane%s: %.2f us per hop\n", name, partner, x[0], x[partner], npl, FLAG ? " + flag" : "", o * 10.0 / 1e3 / 500 / 2);
    hipFree(buf); hipFree(out); hipFree(xcc); hipFree(err);
}

template <int NPL, bool FLAG> static void sweep() {
    for (int partner : {8, 1}) {        // workgroup i runs on XCD i % 8: partner 8 shares workgroup 0's XCD, partner 1 does not
        run<1, 1, NPL, FLAG>("store sc1 / load sc1 (the library's pairs)", partner);
        run<2, 2, NPL, FLAG>("store sc0 sc1 / load sc0 sc1 (system scope)", partner);
        run<0, 1, NPL, FLAG>("plain store / load sc1", partner);
        if (NPL == 1 && !FLAG) {
            run<0, 0, NPL, FLAG>("plain store / load sc0", partner);
            run<1, 0, NPL, FLAG>("store sc1 / load sc0", partner);
        }
    }
}
int main(int argc, char** argv) {
    if (argc == 2) {     // the curve over the payload size (pairs per lane), both XCD placements, the library's policy
        for (int partner : {8, 1}) {
            run<1, 1, 1, false>("sc1 / sc1", partner); run<1, 1, 2, false>("sc1 / sc1", partner); run<1, 1, 4, false>("sc1 / sc1", partner);
            run<1, 1, 8, false>("sc1 / sc1", partner); run<1, 1, 12, false>("sc1 / sc1", partner); run<1, 1, 16, false>("sc1 / sc1", partner);
            run<1, 1, 20, false>("sc1 / sc1", partner); run<1, 1, 24, false>("sc1 / sc1", partner); run<1, 1, 32, false>("sc1 / sc1", partner);
            run<1, 1, 16, true>("sc1 / sc1", partner); run<1, 1, 24, true>("sc1 / sc1", partner);
        }
        return 0;
    }
    if (argc > 3) {     // 190 workgroups of 8 waves poll 8 pairs per lane all the time (what the roles of k_conv_persist would do between steps)
        run<1, 1, 8, false>("quiet", 1); run<1, 1, 8, false>("190 polling workgroups", 1, 1, 192); run<1, 1, 8, false>("... with s_sleep 4 between rounds", 1, 4, 192);
        run<1, 1, 8, false>("... with s_sleep 16 between rounds", 1, 16, 192); run<1, 1, 8, true>("190 polling workgroups", 1, 1, 192);
        return 0;
    }
    if (argc > 2) {
        for (int partner : {1}) {
            run<1, 1, 12, false, 1>("12 stores, 1 load", partner); run<1, 1, 16, false, 1>("16 stores, 1 load", partner); run<1, 1, 24, false, 1>("24 stores, 1 load", partner); run<1, 1, 32, false, 1>("32 stores, 1 load", partner);
            run<1, 1, 12, false, 2>("1 store, 12 loads", partner); run<1, 1, 16, false, 2>("1 store, 16 loads", partner); run<1, 1, 24, false, 2>("1 store, 24 loads", partner); run<1, 1, 32, false, 2>("1 store, 32 loads", partner);
        }
        return 0;
    }
    sweep<1, false>(); sweep<8, false>(); sweep<32, false>(); sweep<8, true>(); sweep<32, true>();
    return 0;
}
