// kernels_bwd.h -- loss statistics, hand-written backward and the optimizer of the exchange path.
//   k_stats     per-rank partial sums of every batch statistic (one workgroup)
//   k_bwd_conv  one workgroup per sample: REINFORCE / entropy / NLL / MSE gradient seeds
//                and reverse-time propagation through the receiver GRU;
//               writes every pre-activation gradient to the tape
//   k_dC        reduction over samples of the class-side gradient of the y head
//   k_wgrad     all weight gradients as grouped fp32-MFMA  dW = delta^T . input  tiles plus
//               bias gradients as column sums -- deterministic (no float atomics)
//   k_gradnorm / k_opt   per-agent clip_grad_norm(1.0) + RMSprop / Adam / SGD on the flat buffers
#pragma once
#include <type_traits>
#include "device_utils.h"
#include "layout.h"

namespace mmg {

// ---------------------------------------------------------------------------------------------
// k_stats.  Active sets (model.py:1256-1262) expressed through t*(b), the step whose logits are
// the sample's output: m_t[b] == 1  <=>  t <= t*(b);  m_{t+1}[b] == 1 (after the forced final zero,
// model.py:870)  <=>  t < t*(b).  Fixed exchange has t* = T-1 for every sample, which reproduces the
// unmasked sums over T (T-1 for the receiver messages, rec_feats[:-1] at model.py:1286).
// ---------------------------------------------------------------------------------------------
// CC: the partials were written by other workgroups of THIS launch (write-through stores): agent-scope loads
template <bool CC = false>
__device__ __forceinline__ float combine_score(const float* part, size_t row, int npb, float b2) {
    float v = 0.f;
    if (npb <= 8) {                                                     // all partial loads in flight at once
        float p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = (j < npb) ? (CC ? __hip_atomic_load(&part[row * npb + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : part[row * npb + j]) : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) v += p[j];
    } else {
        for (int j = 0; j < npb; ++j) v += part[row * npb + j];
    }
    return v + b2;                                                      // model.py:515
}

// the partials as (value, epoch) pairs written by baseline roles of THIS launch (tape.partll): the wave spins until every lane
// that needs its row (live rows only are ever written) holds npb <= 8 fresh pairs
__host__ __device__ inline size_t partll_at(const Dims& d, int sen_side, size_t row, int npb) { return ((size_t)(sen_side ? 0 : 1) * d.T * d.B + row) * npb; }
__device__ __forceinline__ float combine_score_ll(const Dims& dm, const Tape& tp, int sen_side, size_t row, int npb, float b2, uint32_t epoch, bool need) {
    const size_t at = partll_at(dm, sen_side, row, npb);
    unsigned long long u[8];
    for (int spins = 0;; ) {
        bool fresh = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) { u[j] = ld_ll(tp.partll, at + min(j, npb - 1)); fresh = fresh && ll_fresh(u[j], epoch); }
        if (!__any(need && !fresh)) break;
        if (++spins > (1 << 16)) { __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) v += (j < npb) ? ll_value(u[j]) : 0.f;
    return v + b2;
}

// from_parts: baseline scores arrive as per-64-hidden-unit partials (k_baselines2); the blocks of the
// two baseline kinds also materialise bs / br on the tape (exchange() returns them, k_bwd_conv reads them).
// one wave reduces the (stream, step) pairs first, first + stride, ...
// WT: results go out as device-scope (write-through) stores, so a role of a larger launch can publish them with a plain
// counter increment instead of a device-scope release (= L2 write-back); see role_signal_wt.
// ... and what the sample roles of the same launch need of them ALSO goes out as (value, epoch) pairs (tape.statll; device_utils.h:
// st_ll): the consumer spins on the payload -- one trip through memory instead of counter poll + loads (statll_* give the pair index)
__host__ __device__ inline size_t statll_stream(int T, int k, int t) { return (size_t)(k * T + t) * 9; }      // n | lo, hi of sums 1..4
__host__ __device__ inline size_t statll_bs(const Dims& d) { return (size_t)27 * d.T; }
__host__ __device__ inline size_t statll_br(const Dims& d) { return (size_t)27 * d.T + (size_t)d.T * d.B; }
__device__ __forceinline__ void st_ll_d(float* ll, size_t i, double v, uint32_t epoch) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    st_ll(ll, i, __builtin_bit_cast(float, (unsigned)u), epoch); st_ll(ll, i + 1, __builtin_bit_cast(float, (unsigned)(u >> 32)), epoch);
}
// PLL: the partial baseline scores are pairs of this launch's baseline roles (combine_score_ll; npb <= 8)
// LLIN (kernels_game.h: the conversation runs in THIS launch too, B <= 64): the wave first spins on the samples' "forward pass
// written through" pairs (tape.gamell, value = t*), then reads rewards, hits and log-likelihood sums with agent-scope loads
__device__ __forceinline__ int game_wait_done(const Tape& tp, int B, uint32_t epoch, int first = 0) {
    const int lane = threadIdx.x & 63;
    unsigned long long u = 0;
    for (int spins = 0;; ) {
        u = ld_ll(tp.gamell, (size_t)first + min(lane, B - 1));
        if (!__any(!ll_fresh(u, epoch))) break;
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 20)) { if (lane == 0) __hip_atomic_store(tp.sync + MMG_SYNC_ERR, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    return (int)ll_value(u);
}
template <bool WT = false, bool CC = false, bool PLL = false, bool LLIN = false>
__device__ __forceinline__ void stats_pairs(const Dims& dm, const Params& P, const Tape& tp, int from_parts, int first, int stride, uint32_t epoch = 0u) {
    auto put_d = [](double* p, double v) { if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v; };
    auto put_f = [](float* p, float v) { if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v; };
    const int T = dm.T, B = dm.B;
    const int npb = (dm.K + 63) / 64;
    const float b2s = P.p[BS_L2_B][0], b2r = P.p[BR_L2_B][0];
    // grid = 5T + 2 blocks of one wave: every (stream, step) pair reduces concurrently
    const int lane = threadIdx.x & 63;
    const int npairs = 5 * T + 2;
    auto ldf = [](const float* p) { return LLIN ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    const int ts_ll = LLIN ? game_wait_done(tp, B, epoch, B + 16) : 0;        // pair B of every sample (kernels_game.h)
#ifdef MMG_TIMING
    if (LLIN && lane == 0) tp.dbg2[3072 + 4 * first + 1] = (long long)wall_clock64();
#endif
    for (int p = first; p < npairs; p += stride) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
        if (p >= 5 * T) {                       // sum of rewards (-> NLL) and top-k hits
            const int which = p - 5 * T;
            for (int b = lane; b < B; b += 64) a0 += which == 0 ? (double)ldf(&tp.logs[b]) : (double)(LLIN ? __hip_atomic_load(&tp.hit[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tp.hit[b]);
            a0 = dpp_wave_sum_d(a0);
            if (lane == 0) put_d(&tp.stats[stat_glob(T, which)], a0);
            continue;
        }
        const int kind = p / T, t = p - kind * T;
        const bool enabled = dm.use_binary && !(kind == 0 && dm.fixed);
        if (enabled) {
            for (int b = lane; b < B; b += 64) {
                // every load is issued unconditionally (one memory round trip); the activity test masks afterwards
                const size_t row = (size_t)t * B + b;
                const int ts = LLIN ? ts_ll : tp.tstar[b];
                const float L = ldf(&tp.logs[b]);
                const bool sen_side = (kind == 2) || (kind == 4);
                // the array is chosen FIRST (wave-uniform pointer selects), then ONE load each: a ternary over loads is a chain
                // of branches, and every join drains vmcnt -- three or four dependent memory round trips instead of one
                const float* lp_ptr = (kind == 0) ? tp.lp_s : (kind == 1) ? tp.lp_w : tp.lp_z;
                const float* ne_ptr = (kind == 0) ? tp.ne_s : (kind == 1) ? tp.ne_w : tp.ne_z;
                const float lp_raw = ldf(&lp_ptr[row]), ne_raw = ldf(&ne_ptr[row]);
                const float beta_all = PLL ? combine_score_ll(dm, tp, sen_side, row, npb, sen_side ? b2s : b2r, epoch, t <= ts)
                                     : from_parts ? combine_score<CC>(sen_side ? tp.bs_part : tp.br_part, row, npb, sen_side ? b2s : b2r)
                                                  : (sen_side ? tp.bs : tp.br)[row];
                const float lp_all = (kind < 3) ? lp_raw : 0.f;
                const float ne_all = (kind < 3) ? ne_raw : 0.f;
                const bool act = (kind == 1) ? (t < ts) : (t <= ts);
                if (from_parts && kind >= 3 && t <= ts) {
                    put_f(kind == 3 ? &tp.br[row] : &tp.bs[row], beta_all);
                    if (WT) st_ll(tp.statll, (kind == 3 ? statll_br(dm) : statll_bs(dm)) + row, beta_all, epoch);
                }
                if (!act) continue;
                if (kind < 3) {
                    const float beta = beta_all, lp = lp_all, ne = ne_all;
                    const double wv = (double)(L - beta);          // model.py:912
                    a0 += 1.0; a1 += wv; a2 += wv * wv; a3 += wv * (double)lp; a4 += (double)ne;
                } else {
                    const double dv = (double)(beta_all - L);      // model.py:972
                    a0 += dv * dv;
                }
            }
        }
        a0 = dpp_wave_sum_d(a0); a1 = dpp_wave_sum_d(a1); a2 = dpp_wave_sum_d(a2); a3 = dpp_wave_sum_d(a3); a4 = dpp_wave_sum_d(a4);
        if (lane == 0) {
            if (kind < 3) {
                double* st = tp.stats + stat_stream(T, kind, t, 0);
                put_d(st, a0); put_d(st + 1, a1); put_d(st + 2, a2); put_d(st + 3, a3); put_d(st + 4, a4);
                if (WT) {
                    const size_t i = statll_stream(T, kind, t);
                    st_ll(tp.statll, i, (float)a0, epoch);              // a count: exact in fp32
                    st_ll_d(tp.statll, i + 1, a1, epoch); st_ll_d(tp.statll, i + 3, a2, epoch);
                    st_ll_d(tp.statll, i + 5, a3, epoch); st_ll_d(tp.statll, i + 7, a4, epoch);
                }
            } else {
                put_d(&tp.stats[stat_bas(T, kind - 3, t)], a0);
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_stats(Dims dm, Params P, Tape tp, int from_parts) {
    stats_pairs(dm, P, tp, from_parts, blockIdx.x, gridDim.x);
}

// ---------------------------------------------------------------------------------------------
// Coefficients of the multistep losses from the (globally reduced) statistics.
//   cw[k][t] = c_t / n_t / max(1, std_t)   multiplies -(L - beta) * dlogp     (model.py:912-916)
//   ce[k][t] = c_t / n_t * lambda_k        multiplies d negent                (model.py:919-926)
//   cb[t]    = c_t / n_t * 2               multiplies (beta - L)              (model.py:972)
// with c_t = n_t / sum_t n_t (Adaptive, model.py:960-961) or 1 / len(list) (Fixed, model.py:967).
// ---------------------------------------------------------------------------------------------
struct LossCoef { float* cw; float* ce; float* cb; };   // LDS: cw[3*T], ce[3*T], cb[T]

__device__ __forceinline__ void loss_coefficients(const Dims& dm, const double* st, LossCoef lc, float* losses_out,
                                                  double* totals = nullptr, bool write_nll = true) {
    // one thread per (stream, step): threads [0,3T) -> cw/ce, [3T,4T) -> cb; block-level sums for the
    // logged losses go through LDS (lc.cw/ce/cb double as staging for the partial losses afterwards)
    const int T = dm.T, tid = threadIdx.x;
    __shared__ double s_part[5][64];
    if (losses_out) {
        for (int i = tid; i < 5 * 64; i += blockDim.x) s_part[i / 64][i % 64] = 0.0;
        __syncthreads();
    }
    for (int i = tid; i < 4 * T; i += blockDim.x) {
        const int k = i / T, t = i - k * T;
        if (k < 3) {
            double nsum = 0;
            for (int tt = 0; tt < T; ++tt) nsum += st[stat_stream(T, k, tt, 0)];
            const int len = (k == 1) ? T - 1 : T;
            const float lam = (k == 0) ? dm.es : (k == 1) ? dm.erec : dm.esen;
            const bool has = (k == 0) ? dm.has_es : (k == 1) ? dm.has_erec : dm.has_esen;
            const double* s5 = st + stat_stream(T, k, t, 0);
            const double n = s5[0];
            float cw = 0.f, ce = 0.f;
            if (n > 0 && nsum > 0) {
                const double c_over_n = dm.fixed ? 1.0 / ((double)len * n) : 1.0 / nsum;
                double denom = 1.0;
                if (n > 1) {                                            // model.py:914-915
                    const double mean = s5[1] / n;
                    double var = (s5[2] - n * mean * mean) / (n - 1.0);
                    if (var < 0) var = 0;
                    const double sd = sqrt(var);
                    denom = sd > 1.0 ? sd : 1.0;
                }
                cw = (float)(c_over_n / denom);
                ce = has ? (float)(c_over_n * (double)lam) : 0.f;
                if (t < 64 && losses_out) s_part[k][t] = -(double)cw * s5[3] + (double)ce * s5[4];
            }
            lc.cw[k * T + t] = cw; lc.ce[k * T + t] = ce;
        } else {
            double nsum = 0;
            for (int tt = 0; tt < T; ++tt) nsum += st[stat_stream(T, 2, tt, 0)];
            const double n = st[stat_stream(T, 2, t, 0)];
            float cb = 0.f;
            if (n > 0 && nsum > 0) {
                const double c_over_n = dm.fixed ? 1.0 / ((double)T * n) : 1.0 / nsum;
                cb = (float)(2.0 * c_over_n);
                if (t < 64 && losses_out) { s_part[3][t] = c_over_n * st[stat_bas(T, 0, t)]; s_part[4][t] = c_over_n * st[stat_bas(T, 1, t)]; }
            }
            lc.cb[t] = cb;
        }
    }
    __syncthreads();
    if (losses_out && tid == 0) {
        double acc[5] = {0, 0, 0, 0, 0};
        int nsteps = 0;
        for (int t = 0; t < T; ++t) {
            for (int k = 0; k < 5; ++k) acc[k] += (t < 64) ? s_part[k][t] : 0.0;
            nsteps += (st[stat_stream(T, 2, t, 0)] > 0) ? 1 : 0;
        }
        if (!dm.use_binary) nsteps = T;
        if (write_nll) losses_out[0] = (float)(-st[stat_glob(T, 0)] / (double)dm.Bg);   // NLL (model.py:1271); k_wgrad<OPT>: written by the norm role
        losses_out[1] = (float)acc[0];                                   // loss_binary_s
        losses_out[2] = (float)acc[1];                                   // loss_binary_rec
        losses_out[3] = (float)acc[2];                                   // loss_binary_sen
        losses_out[4] = (float)acc[3];                                   // loss_bas_rec
        losses_out[5] = (float)acc[4];                                   // loss_bas_sen
        losses_out[6] = (float)nsteps;                                   // exchange steps the reference executes
        losses_out[7] = (float)st[stat_glob(T, 1)];                      // top-k hits
        if (totals) {
            double live = 0.0;                                           // sum_t |{b: step t is live}| = sample-steps
            for (int t = 0; t < T; ++t) live += st[stat_stream(T, 2, t, 0)];
            if (!dm.use_binary) live = (double)T * (double)dm.Bg;
            totals[0] += (double)nsteps; totals[1] += st[stat_glob(T, 1)]; totals[2] += 1.0; totals[3] += live;
        }
    }
    __syncthreads();
}

// Split form of loss_coefficients for kernels that want the statistics loads in flight with everything else:
// coef_load() (call first) fetches this thread's share, coef_compute() turns it into cw / ce / cb in LDS.
struct CoefRegs { double s5[5]; double nsum; };

// CC: the statistics were written by other workgroups of THIS launch (write-through stores): agent-scope loads, so that the
// caller's role_wait needs no acquire fence (= no L2 invalidate on the recurrence's critical path)
template <bool CC = false>
__device__ __forceinline__ CoefRegs coef_load(const Dims& dm, const double* __restrict__ st) {
    // thread i < 4T handles (k = i / T, t = i % T); k == 3 are the baseline coefficients (sender-stream counts)
    auto ld = [](const double* p) { return CC ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    CoefRegs c;
    const int T = dm.T;
    const int i = min((int)threadIdx.x, 4 * T - 1);
    const int k = i / T, t = i - k * T;
    const int ks = (k < 3) ? k : 2;
#pragma unroll
    for (int j = 0; j < 5; ++j) c.s5[j] = ld(&st[stat_stream(T, ks, t, j)]);
    double nsum = 0;
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
        const double v = ld(&st[stat_stream(T, ks, min(tt, T - 1), 0)]);
        nsum += (tt < T) ? v : 0.0;
    }
    c.nsum = nsum;
    return c;
}

// The same share from the VALUES of the statistics roles' (value, epoch) pair table, staged in LDS by the workgroup
// (layout statll_stream: per (stream, step) the count as a float, then the four f64 sums as low / high halves)
__device__ __forceinline__ CoefRegs coef_lds_regs(const Dims& dm, const float* sv) {
    CoefRegs r;
    const int T = dm.T;
    const int i = min((int)threadIdx.x, 4 * T - 1);
    const int k = i / T, t = i - k * T;
    const int ks = (k < 3) ? k : 2;
    const float* e = sv + (ks * T + t) * 9;
    r.s5[0] = (double)e[0];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        r.s5[1 + j] = __builtin_bit_cast(double, (unsigned long long)__builtin_bit_cast(unsigned, e[1 + 2 * j]) |
                                                 ((unsigned long long)__builtin_bit_cast(unsigned, e[2 + 2 * j]) << 32));
    float nsum = 0.f;                                    // (counts: exact in fp32)
    const float* e0 = sv + ks * T * 9;
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) { const float v = e0[min(tt, T - 1) * 9]; nsum += (tt < T) ? v : 0.f; }
    r.nsum = (double)nsum;
    return r;
}

// (on the recurrence's critical path in the register-resident backward: f64 only where it matters -- the variance's cancellation;
//  the counts are small integers and the results are fp32 coefficients, so their reciprocals, the square root and the quotient run
//  in fp32: four f64 divisions and an f64 sqrt -- software sequences of ~20 dependent f64 instructions each -- were 0.6 us here)
__device__ __forceinline__ void coef_compute(const Dims& dm, const CoefRegs& c, LossCoef lc) {
    const int T = dm.T, i = threadIdx.x;
    if (i < 4 * T) {
        const int k = i / T, t = i - k * T;
        const double n = c.s5[0];
        const float nf = (float)n, nsf = (float)c.nsum;
        if (k < 3) {
            const int len = (k == 1) ? T - 1 : T;
            const float lam = (k == 0) ? dm.es : (k == 1) ? dm.erec : dm.esen;
            const bool has = (k == 0) ? dm.has_es : (k == 1) ? dm.has_erec : dm.has_esen;
            float cw = 0.f, ce = 0.f;
            if (n > 0 && c.nsum > 0) {
                const float c_over_n = dm.fixed ? 1.f / ((float)len * nf) : 1.f / nsf;
                float denom = 1.f;
                if (n > 1) {                                            // model.py:914-915
                    double rn = (double)(1.f / nf);
                    rn = rn * (2.0 - n * rn);                           // one Newton step: 1 / n to f64 precision (n is a count)
                    const double mean = c.s5[1] * rn;
                    double var = (c.s5[2] - n * mean * mean) * (double)(1.f / (nf - 1.f));
                    if (var < 0) var = 0;
                    const float sd = sqrtf((float)var);
                    denom = sd > 1.f ? sd : 1.f;
                }
                cw = c_over_n / denom;
                ce = has ? c_over_n * lam : 0.f;
            }
            lc.cw[k * T + t] = cw; lc.ce[k * T + t] = ce;
        } else {
            float cb = 0.f;
            if (n > 0 && c.nsum > 0) cb = 2.f * (dm.fixed ? 1.f / ((float)T * nf) : 1.f / nsf);
            lc.cb[t] = cb;
        }
    }
    __syncthreads();
}

// d loss / d logit of one Bernoulli unit: REINFORCE term + entropy term (Appendix A.4)
__device__ __forceinline__ float bit_seed(float q, float p, float wh, float ce) {
    const float pe = p + MMG_EPS, qe = 1.f - p + MMG_EPS;
    float dLdp = -wh * (q / pe - (1.f - q) / qe);
    if (ce != 0.f) dLdp += ce * (logf(pe) + p / pe - logf(qe) - (1.f - p) / qe);
    return dLdp * p * (1.f - p);
}

__host__ __device__ inline int bwd_smem_floats(const Dims& d) {
    auto p4 = [](int n) { return (n + 3) & ~3; };
    return 7 * d.T + 3 * p4(d.H) + 3 * p4(d.W) + 6 * p4(d.R) + 2 * p4(3 * d.R) + p4(d.D) + 4 * MMG_BLOCK + 32;
}

// MANY: thousands of samples -- occupancy matters more than load depth (<= 128 VGPRs, 4 workgroups per CU).
template <bool MANY>
__global__ __launch_bounds__(MMG_BLOCK, MANY ? 4 : 1) void k_bwd_conv(Dims dm, Params P, Tape tp, const int64_t* __restrict__ target) {
    constexpr int TU = MANY ? 2 : 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int B = dm.B, H = dm.H, W = dm.W, R = dm.R, V = dm.V, D = dm.D, T = dm.T;
    auto p4 = [](int n) { return (n + 3) & ~3; };
    float* p = smem;
    LossCoef lc; lc.cw = p; p += 3 * T; lc.ce = p; p += 3 * T; lc.cb = p; p += T;
    p = smem + ((7 * T + 3) & ~3);
    float* s_dhx = p; p += p4(H);   float* s_da = p; p += p4(H);   float* s_dpre = p; p += p4(H);
    float* s_dlz = p; p += p4(W);   float* s_dlw = p; p += p4(W);  float* s_dc0 = p; p += p4(W);
    float* s_dh = p; p += p4(R);    float* s_dhn = p; p += p4(R);  float* s_dg = p; p += p4(R);
    float* s_A = p; p += p4(R);     float* s_dA = p; p += p4(R);   float* s_hs = p; p += p4(R);
    float* s_dgi = p; p += p4(3 * R); float* s_dgh = p; p += p4(3 * R);
    float* s_dy = p; p += p4(D);
    float* s_red = p; p += 4 * MMG_BLOCK;
    float* s_misc = p;

    loss_coefficients(dm, tp.stats, lc, nullptr, nullptr);        // logged losses: spare block of k_wgrad

    const bool binary = dm.use_binary != 0;
    const int tstar = tp.tstar[b];
    const float L = tp.logs[b];
    const float* cw_s = lc.cw, *cw_r = lc.cw + T, *cw_z = lc.cw + 2 * T;
    const float* ce_s = lc.ce, *ce_r = lc.ce + T, *ce_z = lc.ce + 2 * T;

    for (int i = tid; i < R; i += nt) s_dh[i] = 0.f;
    for (int i = tid; i < H; i += nt) s_dhx[i] = 0.f;
    __syncthreads();

    // ---------------- zero the gradient tapes of steps this sample never took ----------------
    for (int t = tstar + 1; t < T; ++t) {
        const size_t row = (size_t)t * B + b;
        for (int i = tid; i < W; i += nt) { tp.dlz[row * W + i] = 0.f; tp.dlw[row * W + i] = 0.f; }
        for (int i = tid; i < H; i += nt) tp.dpre[row * H + i] = 0.f;
        for (int i = tid; i < R; i += nt) tp.dgpre[row * R + i] = 0.f;
        for (int i = tid; i < 3 * R; i += nt) { tp.dgi[row * 3 * R + i] = 0.f; tp.dgh[row * 3 * R + i] = 0.f; }
        if (tid == 0) { tp.dls[row] = 0.f; tp.dbs[row] = 0.f; tp.dbr[row] = 0.f; }
    }

    // ---------------- reverse time ----------------
    for (int t = tstar; t >= 0; --t) {
        const size_t row = (size_t)t * B + b;
        const float* hn = tp.h + ((size_t)(t + 1) * B + b) * R;    // h after step t
        const float* hp = tp.h + ((size_t)t * B + b) * R;          // h before step t

        // ---- receiver message head (stream 1: active while m_{t+1} == 1) ----
        const bool act_next = binary && (t < tstar);
        if (act_next) {
            const float wh = (L - tp.br[row]) * cw_r[t];
            for (int j = tid; j < W; j += nt) {
                const float v = bit_seed(tp.w[row * W + j], tp.pw[row * W + j], wh, ce_r[t]);
                s_dlw[j] = v; tp.dlw[row * W + j] = v;
            }
            __syncthreads();
            gemv_t<TU>(P.p[R_W_W], R, W, R, s_dlw, s_dg, s_red, false);            // dg = W_w^T dlw
            for (int i = tid; i < R; i += nt) {
                const float g = tp.g[row * R + i];
                const float v = s_dg[i] * (1.f - g * g);
                s_dg[i] = v; tp.dgpre[row * R + i] = v;
            }
            __syncthreads();
            gemv_t<TU>(P.p[R_WH_W], R, R, R, s_dg, s_dh, s_red, true);             // dh += W_h^T dgpre
        } else {
            for (int j = tid; j < W; j += nt) tp.dlw[row * W + j] = 0.f;
            for (int i = tid; i < R; i += nt) tp.dgpre[row * R + i] = 0.f;
        }
        // ---- stop head (stream 0, Adaptive only) ----
        if (binary && !dm.fixed) {
            const float wh = (L - tp.br[row]) * cw_s[t];
            const float dls = bit_seed(tp.s[row], tp.ps[row], wh, ce_s[t]);
            if (tid == 0) tp.dls[row] = dls;
            const float* ws = P.p[R_S_W];
            for (int i = tid; i < R; i += nt) s_dh[i] += ws[i] * dls;
        } else if (tid == 0) {
            tp.dls[row] = 0.f;
        }
        __syncthreads();
        // ---- class logits at the output step (NLL, model.py:1271) ----
        if (t == tstar) {
            const int tgt = (int)target[b];
            float dsum = 0.f;
            for (int d = tid; d < D; d += nt) {
                const float v = (tp.sm[(size_t)b * D + d] - (d == tgt ? 1.f : 0.f)) / (float)dm.Bg;
                s_dy[d] = v; tp.dy[(size_t)b * D + d] = v; dsum += v;
            }
            dsum = block_sum(dsum, s_misc);
            if (tid == 0) tp.dysum[b] = dsum;
            for (int i = tid; i < R; i += nt) { const float v = hn[i]; s_hs[i] = v; tp.hstar[(size_t)b * R + i] = v; }
            __syncthreads();
            gemv_rows<(MANY ? 2 : 4)>(P.p[R_Y1_W], R + V, R, R, s_hs, [&](int n, float acc) { s_A[n] = acc; });
            __syncthreads();
            const float* w2 = P.p[R_Y2_W];
            for (int i = tid; i < R; i += nt) {
                const float a = s_A[i];
                float acc = 0.f;
                for (int d = 0; d < D; d += 8) {                               // 8 class rows of Cd in flight, same add order
                    float cv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) cv[u] = tp.Cd[(size_t)min(d + u, D - 1) * R + i];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += (d + u < D && a + cv[u] > 0.f) ? s_dy[min(d + u, D - 1)] : 0.f;
                }
                const float v = acc * w2[i];
                s_dA[i] = v; tp.dA[(size_t)b * R + i] = v; tp.Astar[(size_t)b * R + i] = a;
            }
            __syncthreads();
            gemv_t<TU>(P.p[R_Y1_W], R + V, R, R, s_dA, s_dh, s_red, true);         // dh += W_y1h^T dA
        }
        // ---- GRU cell backward ----
        {
            const float* gr = tp.gru + row * 4 * R;
            for (int i = tid; i < R; i += nt) {
                const float rr = gr[i], uu = gr[R + i], nn = gr[2 * R + i], ghn = gr[3 * R + i];
                const float dh = s_dh[i];
                const float dn = dh * (1.f - uu), du = dh * (hp[i] - nn);
                const float dnp = dn * (1.f - nn * nn), dup = du * uu * (1.f - uu);
                const float drp = dnp * ghn * rr * (1.f - rr);
                s_dgi[i] = drp; s_dgi[R + i] = dup; s_dgi[2 * R + i] = dnp;
                s_dgh[i] = drp; s_dgh[R + i] = dup; s_dgh[2 * R + i] = dnp * rr;
                s_dhn[i] = dh * uu;
            }
            __syncthreads();
            for (int i = tid; i < 3 * R; i += nt) { tp.dgi[row * 3 * R + i] = s_dgi[i]; tp.dgh[row * 3 * R + i] = s_dgh[i]; }
            gemv_t<TU>(P.p[R_WHH], R, 3 * R, R, s_dgh, s_dhn, s_red, true);        // dh_{t-1} += W_hh^T dgh
            for (int i = tid; i < R; i += nt) s_dh[i] = s_dhn[i];
            __syncthreads();
        }
        // ---- sender (stream 2) ----
        if (binary) {
            const float wh = (L - tp.bs[row]) * cw_z[t];
            for (int j = tid; j < W; j += nt) {
                const float v = bit_seed(tp.z[row * W + j], tp.pz[row * W + j], wh, ce_z[t]);
                s_dlz[j] = v; tp.dlz[row * W + j] = v;
            }
            __syncthreads();
            gemv_t<TU>(P.p[S_BIN_W], H, W, H, s_dlz, s_da, s_red, false);          // da = W_b^T dlz
            for (int i = tid; i < H; i += nt) {
                const float a = tp.a[row * H + i];
                const float v = s_da[i] * (1.f - a * a);
                s_dpre[i] = v; tp.dpre[row * H + i] = v; s_dhx[i] += v;
            }
            __syncthreads();
            if (t == 0) {                                                      // code_bias path (model.py:199)
                gemv_t<TU>(P.p[S_CODE_W], W, H, W, s_dpre, s_dc0, s_red, false);
                for (int j = tid; j < W; j += nt) tp.dc0[(size_t)b * W + j] = s_dc0[j];
            }
            // ---- baselines (MSE, model.py:971-988) ----
            const float dbs = lc.cb[t] * (tp.bs[row] - L), dbr = lc.cb[t] * (tp.br[row] - L);
            // d hidden = dbeta * w2 * 1[hidden > 0] is formed on the fly by k_wgrad (virtual operand)
            if (tid == 0) { tp.dbs[row] = dbs; tp.dbr[row] = dbr; }
        } else {
            for (int i = tid; i < W; i += nt) tp.dlz[row * W + i] = 0.f;
            for (int i = tid; i < H; i += nt) tp.dpre[row * H + i] = 0.f;
            if (tid == 0) { tp.dbs[row] = 0.f; tp.dbr[row] = 0.f; }
        }
        __syncthreads();
    }
    for (int i = tid; i < H; i += nt) tp.dhx[(size_t)b * H + i] = s_dhx[i];
    if (!binary) for (int j = tid; j < W; j += nt) tp.dc0[(size_t)b * W + j] = 0.f;
}

// Active rows.  Every gradient tape row (t, b) with t > t*(b) is zero, and with early stopping that is most of them
// (config 2: ~140 of 640 rows are live).  One wave lists the live rows in (t, b) order; k_wgrad then reduces over
// that list instead of over all T*B rows.  t*(b) comes from the conversation launch, so this runs anywhere after it.
// CC: t* was written by other workgroups of THIS launch (write-through stores): agent-scope loads
template <bool CC = false>
__device__ __forceinline__ void build_row_map(const Dims& dm, const Tape& tp) {
    const int lane = threadIdx.x & 63, B = dm.B, T = dm.T;
    int base = 0;
    for (int t = 0; t < T; ++t) {
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + lane;
            const int tsb = CC ? __hip_atomic_load(&tp.tstar[min(b, B - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : tp.tstar[min(b, B - 1)];
            const bool act = (b < B) && (t <= tsb);
            const unsigned long long m = __ballot(act);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (act) tp.rmap[pos] = t * B + b;
            base += __popcll(m);
        }
    }
    if (lane == 0) tp.rcount[0] = base;
    for (int i = base + lane; i < T * B; i += 64) tp.rmap[i] = -1;       // (readers that index the list directly need no count)
}

// ---------------------------------------------------------------------------------------------
// k_dC: grid = D.  dC[d,r] = sum_b dy[b,d] * w_y2[r] * 1[A*[b,r] + Cd[d,r] > 0]   (-> y1.weight[:,R:], y1.bias)
//                  Py2[d,r] = sum_b dy[b,d] * relu(A*[b,r] + Cd[d,r])              (-> y2.weight)
// ---------------------------------------------------------------------------------------------
// CC: dy / A* were written by other workgroups of this launch (write-through stores): read them with agent-scope loads
template <bool CC = false>
__device__ __forceinline__ void dC_class(const Dims& dm, const Params& P, const Tape& tp, int d, float* s_c, float* s_p) {
    auto ld = [](const float* p) { return CC ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    const int tid = threadIdx.x, R = dm.R, B = dm.B, D = dm.D;
    const int cols = R < 64 ? R : 64;                 // up to 64 columns per pass, >= 4 sample groups
    const int groups = MMG_BLOCK / cols;
    const int g = tid / cols, rr = tid - g * cols;
    const float* w2 = P.p[R_Y2_W];
    for (int r0 = 0; r0 < R; r0 += cols) {
        const int r = r0 + rr;
        float dc0 = 0.f, dc1 = 0.f, py0 = 0.f, py1 = 0.f;
        if (g < groups && r < R) {
            const float cv = ld(&tp.Cd[(size_t)d * R + r]);         // (CC: possibly written by a prep role of this launch, kernels_game.h)
            for (int b = g; b < B; b += 16 * groups) {              // 16 samples (32 loads) in flight per thread
                float yv[16], pv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int bb = b + u * groups;
                    const bool ok = bb < B;
                    yv[u] = ok ? ld(&tp.dy[(size_t)bb * D + d]) : 0.f;
                    pv[u] = ok ? ld(&tp.Astar[(size_t)bb * R + r]) + cv : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    if (pv[u] > 0.f) { dc0 += yv[u]; py0 = fmaf(yv[u], pv[u], py0); }
                    if (pv[u + 1] > 0.f) { dc1 += yv[u + 1]; py1 = fmaf(yv[u + 1], pv[u + 1], py1); }
                }
            }
        }
        s_c[tid] = (dc0 + dc1) * ((g < groups && r < R) ? w2[r] : 0.f);
        s_p[tid] = py0 + py1;
        __syncthreads();
        if (tid < cols && r0 + tid < R) {
            float a = 0.f, c = 0.f;
            for (int k = 0; k < groups; ++k) { a += s_c[k * cols + tid]; c += s_p[k * cols + tid]; }
            tp.dC[(size_t)d * R + r0 + tid] = a;
            tp.Py2[(size_t)d * R + r0 + tid] = c;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(MMG_BLOCK) void k_dC(Dims dm, Params P, Tape tp) {
    __shared__ float s_c[MMG_BLOCK], s_p[MMG_BLOCK];
    dC_class(dm, P, tp, blockIdx.x, s_c, s_p);
}

// ---------------------------------------------------------------------------------------------
// k_wgrad: job table in the tape ("tables").  GEMM job: C[n,k] = sum_row A[row*lda + n] *
// Bm[(row % bmod)*ldb + k]  -- both operands are tape arrays with the reduction dimension
// (step, sample) as the slow axis, exactly the MFMA 16x16x4 fragment shape (A: 4 rows x 16 n,
// B: 4 rows x 16 k per instruction).  One wave per 16x16 tile of dW.  Column job: dst[c] =
// scale[c] * sum_row src[row*ld + c].
// ---------------------------------------------------------------------------------------------
enum { SRC_STATIC = 0, SRC_X = 1, SRC_DESC = 2 };
// vhid != NULL: the A operand is virtual,  A[row, n] = A[row] * vw2[n] * 1[vhid[row*lda + n] > 0]
// (the baselines' d hidden = d score * linear2.weight * relu', never materialised)
struct GemmJob { const float* A; const float* Bm; float* C; const float* vhid; const float* vw2;
                 int lda, ldb, ldc, rows, N, K, bmod, bsrc, tile_begin, tiles_k, compact, nsplit; };   // compact: rows are (step, sample) rows; nsplit: row slices per output tile
// vbeta != NULL: virtual source,  src'[row, c] = vbeta[row] * vw2[c] * 1[src[row*ld + c] > 0]
// wrow != NULL: row-weighted sum,  dst[c] = sum_row wrow[row] * src[row*ld + c]   (the N = 1 "GEMMs": d score^T . hidden)
struct ColJob { const float* src; float* dst; const float* scale; const float* vbeta; const float* vw2; const float* wrow;
                int ld, rows, cols, blk_begin, compact, special; };   // special: the code_bias job (see k_wgrad)
#define MMG_MAX_GEMM 40
#define MMG_MAX_COL 40
struct NormPlan { int64_t begin[MMG_GN_BLOCKS], end[MMG_GN_BLOCKS]; int agent[MMG_GN_BLOCKS]; };
#define MMG_MAX_WBLOCKS 16384
struct JobTable {
    int n_gemm, n_col, gemm_tiles, gemm_blocks, col_blocks, special_block, special_job, pad2;   // special_*: the code_bias job's workgroup / entry of c[] (-1: none)
    GemmJob g[MMG_MAX_GEMM];
    ColJob c[MMG_MAX_COL];
    NormPlan np;
    int g_begin[64], c_begin[64];               // compact copies of tile_begin / blk_begin (lane-parallel lookup), INT_MAX padded
    int n_wblocks;                              // blocks of k_wgrad (= entries of its part[] output)
    signed char wblock_agent[MMG_MAX_WBLOCKS];  // agent whose gradient block i of k_wgrad writes
};

// load 4 consecutive floats of one row with bounds / alignment handling (zeros outside)
__device__ __forceinline__ float4 load4_guard(const float* __restrict__ base_, size_t row_off, int col, int ncols, bool row_ok, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!row_ok) return v;
    gfloat* p = as_global(base_) + row_off + col;          // (device memory: see as_global)
    if (vec_ok && col + 3 < ncols) return ldg4(p);
    if (col < ncols) v.x = p[0];
    if (col + 1 < ncols) v.y = p[1];
    if (col + 2 < ncols) v.z = p[2];
    if (col + 3 < ncols) v.w = p[3];
    return v;
}

// The launch geometry of k_wgrad travels in the kernel arguments (with the pointers) instead of behind the JobTable pointer: a
// block knows its role without a memory trip.  (Round 6 also tried the jobs' first tiles there -- job lookup by 40 scalar
// compares instead of a lane-parallel load + ballot: no gain at config 2, and the 80 extra SGPRs cost the tile-walking launches
// of config 4 8 us, 35 -> 43: reverted.)
struct WgHead { int gemm_tiles, n_wblocks, special_block, special_job; };

struct OptArgs {
    int optim_type, only_receiver, from_wgrad, bump_step, bump_mb;
    float lr;
    int64_t agent_begin[5];
    int64_t total;
};

// OPT (k_wgrad<true>): the clip + optimizer step of model.py:1310-1330 INSIDE the weight-gradient launch.  Every workgroup still
// holds the gradient elements it just stored in registers; what it lacks is the four per-agent clip coefficients, which need the
// squared norm of ALL blocks.  So: each block publishes its sum of squares as a (value, epoch) pair (tape.gnll) and requests its
// parameter / optimizer-state elements; FOUR extra workgroups (one norm role per agent) spin on their agent's pairs, add them in
// k_opt's fixed order, and publish the agent's coefficient (64 replicas: no word has more than ~16 pollers); the blocks spin on
// their replica, update their elements and store parameters + state.  No k_opt launch (6 us), and the gradients are never re-read.
// The spare block closes the launch (NLL / non-finite guard, error word, counters) once the four totals are out.
// A block's phase 2 starts only after EVERY block has published, i.e. finished reading -- some jobs read parameters
// (linear2.weight, code_layer.weight).  All blocks of the launch must be co-resident (the host checks the occupancy budget).
struct WgOpt {
    OptArgs oa;
    float* params; float* state; const float* grads; float* gnll; float* coefll;
    uint32_t* counter; uint32_t* err_host;
};
#define MMG_COEF_REPL 64
// one gradient element's update (k_opt's arithmetic); w, s1, s2 were requested before the wait for the coefficient
__device__ __forceinline__ void opt_update_one(const OptArgs& oa, float* params, float* state, int64_t idx, float g, float coef,
                                               float w, float s1, float s2, uint32_t step) {
    const float gv = g * coef;
    if (oa.optim_type == MMG_OPT_RMSPROP) {
        const float sv = 0.99f * s1 + (1.f - 0.99f) * gv * gv;
        w -= oa.lr * gv / (sqrtf(sv) + 1e-8f);
        state[idx] = sv;
    } else if (oa.optim_type == MMG_OPT_ADAM) {
        const float b1 = 0.9f, b2 = 0.999f;
        const float bc1 = 1.f - powf(b1, (float)step), bc2s = sqrtf(1.f - powf(b2, (float)step));
        const float mv = s1 + (gv - s1) * (1.f - b1), vv = b2 * s2 + (1.f - b2) * gv * gv;
        w -= (oa.lr / bc1) * mv / (sqrtf(vv) / bc2s + 1e-8f);
        state[idx] = mv; state[oa.total + idx] = vv;
    } else {
        w -= oa.lr * gv;
    }
    params[idx] = w;
}
// the wave's clip coefficient of `agent`: spins on this block's replica of the norm role's pairs; < 0: the update is skipped
__device__ __forceinline__ float opt_wait_coef(const WgOpt& wo, int agent, uint32_t epoch, const uint32_t* sync) {
    unsigned long long u;
    for (int spins = 0;; ) {
        u = ld_ll(wo.coefll, (size_t)agent * MMG_COEF_REPL + (blockIdx.x & (MMG_COEF_REPL - 1)));
        if (ll_fresh(u, epoch)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 20)) { __hip_atomic_store(const_cast<uint32_t*>(sync) + MMG_SYNC_ERR, 10u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return -1.f; }
    }
    return ll_value(u);
}

template <bool OPT>
__global__ __launch_bounds__(MMG_BLOCK) void k_wgrad(const JobTable* __restrict__ jt, const float* __restrict__ x,
                                                     const float* __restrict__ desc, float* __restrict__ part,
                                                     Dims dm, const double* __restrict__ stats, float* __restrict__ losses,
                                                     double* __restrict__ totals, const int* __restrict__ rmap,
                                                     const int* __restrict__ rcount, float* __restrict__ wpart, uint32_t* __restrict__ wcnt,
                                                     const uint32_t* __restrict__ sync, float* __restrict__ grad_tail, WgOpt wo, int gt_stride, WgHead hd
#ifdef MMG_TIMING
                                                     , long long* __restrict__ dbg2
#endif
                                                     ) {
    // gt_stride > 0 (agents with more output tiles than resident workgroups: config 4's 3 815): the first gt_stride workgroups
    // (a multiple of 8: the XCD of a tile stays that of its workgroup) walk the GEMM tiles with that stride -- the live-row list
    // and the job table are fetched once per workgroup instead of once per tile, and no tile waits for a slot; the other
    // workgroups (column sums, the special and the spare block) follow them.  bid: the block index of the un-strided launch.
    const int bid = (gt_stride > 0 && (int)blockIdx.x >= gt_stride) ? (int)blockIdx.x - gt_stride + hd.gemm_tiles : (int)blockIdx.x;
    const uint32_t oepoch = OPT ? wo.counter[3] + 1u : 0u;          // (the norm role bumps the counters when every block has read them)
    const uint32_t ostep = OPT ? wo.counter[1] + 1u : 0u;
    if (OPT && bid > hd.n_wblocks) {
        // ---- a norm role: ONE PER AGENT (round 6; one workgroup for all four polled 8 pairs per lane, ~0.9 us per poll round).  It
        // spins on the sums of squares of ITS agent's blocks only -- two or three of the eight pair slots of a lane -- adds them in
        // k_opt's order (the other agents' entries are skipped where k_opt adds zeros: bit for bit the same sum) and publishes the
        // agent's clip coefficient (64 replicas) and its total (the closing role below reads the four totals).  An agent's
        // blocks update as soon as THEIR coefficient is out: the receiver's and sender's do not wait for the baselines' tiles.
        __shared__ float s_ss[4];
        const int agent = bid - hd.n_wblocks - 1;
        const int n = hd.n_wblocks;
        float ss = 0.f;
        for (int k0 = threadIdx.x; k0 < n; k0 += 8 * MMG_BLOCK) {
            unsigned long long u[8]; bool mine[8], any[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + q * MMG_BLOCK;
                mine[q] = (k < n) && ((int)jt->wblock_agent[min(k, n - 1)] == agent);
                any[q] = __any(mine[q]);
                u[q] = 0ull;
            }
            for (int spins = 0;; ) {
                bool fresh = true;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (any[q]) {                            // (wave-uniform: slots without a block of this agent cost nothing)
                        const unsigned long long v = ld_ll(wo.gnll, (size_t)min(k0 + q * MMG_BLOCK, n - 1));
                        if (mine[q]) { u[q] = v; fresh = fresh && ll_fresh(v, oepoch); }
                    }
                if (!__any(!fresh)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 20)) { __hip_atomic_store(const_cast<uint32_t*>(sync) + MMG_SYNC_ERR, 11u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) ss += mine[q] ? ll_value(u[q]) : 0.f;
        }
        const float wsum = dpp_wave_sum(ss);
        if ((threadIdx.x & 63) == 0) s_ss[threadIdx.x >> 6] = wsum;
        __syncthreads();
        // an in-launch dependency wait of this minibatch timed out (sync[MMG_SYNC_ERR]): parameters and optimizer state stay
        // untouched (coefficient -1) -- exactly k_opt's contract; the closing role posts the word to the host
        const uint32_t err = __hip_atomic_load(sync + MMG_SYNC_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < MMG_COEF_REPL + 1) {
            const float tot = (s_ss[0] + s_ss[1]) + (s_ss[2] + s_ss[3]);
            const float coef = 1.0f / (sqrtf(tot) + 1e-6f);               // max_norm = 1 (model.py:1310)
            if (threadIdx.x < MMG_COEF_REPL) st_ll(wo.coefll, (size_t)agent * MMG_COEF_REPL + threadIdx.x, err != 0u ? -1.f : (coef < 1.f ? coef : 1.f), oepoch);
            else st_ll(wo.coefll, (size_t)4 * MMG_COEF_REPL + agent, tot, oepoch);
        }
        return;
    }
#ifdef MMG_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 4096) dbg2[2 * blockIdx.x] = (long long)wall_clock64();
#define MMG_WG_END() do { __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 4096) dbg2[2 * blockIdx.x + 1] = (long long)wall_clock64(); } while (0)
#else
#define MMG_WG_END() do {} while (0)
#endif
    // One workgroup per 16x16 tile of dW = A^T . Bm (A: [rows, N] gradient tape, Bm: [rows, K] input
    // tape, reduction over the (step, sample) rows).  Rows are consumed in chunks of 64: every thread
    // fetches 4 consecutive columns of one row of each operand (16-byte coalesced loads, 64 B per row),
    // the chunk is staged in LDS (double-buffered, one barrier per chunk, next chunk's loads in flight
    // during the MFMAs), wave w multiplies rows 16w..16w+15 of the chunk (4 x mfma_f32_16x16x4), and the
    // four partial tiles are combined through LDS.  Every block also leaves the sum of squares of what it
    // wrote in part[blockIdx.x] (clip_grad_norm partials, summed per agent by k_opt).
    constexpr int CH = 64;
    // row strides of 16 / 48 floats: an MFMA operand read touches rows r..r+3 x 16 columns, i.e. per 32-lane half two rows
    // whose banks are i and 16 + i -- conflict-free (strides 17 / 33 measured 30 % bank-conflict cycles: SQ_LDS_BANK_CONFLICT);
    // the staging stores are one aligned 16-byte write per operand slice
    __shared__ __attribute__((aligned(16))) float s_a[2][CH][16];
    __shared__ __attribute__((aligned(16))) float s_b[2][CH][48];   // 32 output columns per block: the staged A chunk feeds two k-tiles
    float (*s_acc)[16][33] = reinterpret_cast<float (*)[16][33]>(&s_b[0][0][0]);   // [4][16][33] reused after the row loop
    float* s_part = &s_b[0][0][0];                                                   // column-sum staging
    __shared__ float s_red[8];
    if (bid == hd.special_block) {
        // (found by ONE scalar load, ahead of the live-row list and the job lookup every other block starts with: this is the launch's
        //  longest block -- 9.0 us next to ~7 -- and those were a dependent memory round trip in front of its two)
        const ColJob& C = jt->c[hd.special_job];
        // code_bias gradient: dst[j] = scale[j] * sum_h wrow[h*cols + j] * v[h],  v[h] = sum_{b < rows} src[b*ld + h]
        // (src = dpre rows of step 0, wrow = code_layer.weight [ld, cols], scale = sigmoid'(code_bias))
        float* s_v = &s_b[0][0][0];                        // ld <= 4224 floats of staging
        const int Hh = C.ld, Wc = C.cols, nb = C.rows;
        const int p8 = threadIdx.x & 7, jj = threadIdx.x >> 3;
        // this runs in ONE workgroup next to ~1000 short ones: every load that does not depend on v is issued first
        // (the weight slice of the first 32 output columns), then the rows in batches of 64 -- two memory round trips
        float wreg[32];
        const bool wfast = (Hh == 256) && (Wc <= 32);
        if (wfast) {
#pragma unroll
            for (int i = 0; i < 32; ++i) wreg[i] = C.wrow[(size_t)(p8 + 8 * i) * Wc + min(jj, Wc - 1)];
        }
        for (int h0 = threadIdx.x; h0 < Hh; h0 += MMG_BLOCK) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int b0 = 0; b0 < nb; b0 += 64) {
                float v[64];
#pragma unroll
                for (int u = 0; u < 64; ++u) v[u] = C.src[(size_t)min(b0 + u, nb - 1) * Hh + h0];
#pragma unroll
                for (int u = 0; u < 64; u += 4) {
                    a0 += (b0 + u < nb) ? v[u] : 0.f; a1 += (b0 + u + 1 < nb) ? v[u + 1] : 0.f;
                    a2 += (b0 + u + 2 < nb) ? v[u + 2] : 0.f; a3 += (b0 + u + 3 < nb) ? v[u + 3] : 0.f;
                }
            }
            s_v[h0] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        float sq = 0.f;
        for (int j0 = 0; j0 < Wc; j0 += MMG_BLOCK / 8) {   // 8 lanes per output column
            const int j = j0 + jj;
            float acc = 0.f;
            if (wfast) {
#pragma unroll
                for (int i = 0; i < 32; ++i) acc = fmaf(wreg[i], s_v[p8 + 8 * i], acc);
            } else if (j < Wc) {
                for (int h0 = p8; h0 < Hh; h0 += 8) acc = fmaf(C.wrow[(size_t)h0 * Wc + j], s_v[h0], acc);
            }
            acc = dpp_group_sum<8>(acc);
            if (j < Wc && p8 == 0) { const float gsum = acc * C.scale[j]; C.dst[j] = gsum; sq = fmaf(gsum, gsum, sq); }
        }
        sq = block_sum(sq, s_red);
        if (threadIdx.x == 0) part[bid] = sq;
        if (OPT) {
            if (threadIdx.x == 0) st_ll(wo.gnll, bid, sq, oepoch);
#ifdef MMG_TIMING
        if (threadIdx.x == 0 && bid < 2000) dbg2[8192 + 4 * bid + 2] = (long long)wall_clock64();
#endif
#ifdef MMG_TIMING
            if (threadIdx.x == 0 && bid < 2000) dbg2[8192 + 4 * bid + 2] = (long long)wall_clock64();
#endif
            const float coef = opt_wait_coef(wo, jt->wblock_agent[bid], oepoch, sync);
            // (one workgroup, <= 32 elements: read back what it stored -- same thread, same address)
            for (int j0 = 0; j0 < Wc && coef >= 0.f; j0 += MMG_BLOCK / 8) {
                const int j = j0 + jj;
                if (j < Wc && p8 == 0) {
                    const int64_t idx = (C.dst - wo.grads) + j;
                    opt_update_one(wo.oa, wo.params, wo.state, idx, C.dst[j], coef, wo.params[idx], wo.state[idx],
                                   wo.oa.optim_type == MMG_OPT_ADAM ? wo.state[wo.oa.total + idx] : 0.f, ostep);
                }
            }
        }
        MMG_WG_END();
        return;
    }
    // rmap != NULL: jobs whose rows are (step, sample) rows reduce over the compacted list of live rows only
    // (build_row_map); the list is copied to LDS first thing, in the shadow of the job lookup below.
    constexpr int MAXMAP = 2048;
    __shared__ unsigned short s_map[MAXMAP];
    const bool use_map = rmap != nullptr;
    int nact = 0;
    if (use_map) {
        const int tb = dm.T * dm.B;
        for (int i2 = threadIdx.x; i2 < tb; i2 += MMG_BLOCK) s_map[i2] = (unsigned short)rmap[i2];
        nact = rcount[0];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (bid == hd.n_wblocks) {
        // spare block: the six logged loss scalars, the semantic step count and the running totals (off every
        // critical path: this launch lasts ~17 us, the bookkeeping ~3)
        float* s_lc = &s_b[0][0][0];                       // (7 * 64 floats of the staging area: this block stages nothing -- keeps the
                                                           //  kernel at four workgroups per CU, which k_wgrad<true> needs for co-residency)
        LossCoef lc; lc.cw = s_lc; lc.ce = s_lc + 3 * dm.T; lc.cb = s_lc + 6 * dm.T;
        loss_coefficients(dm, stats, lc, losses, totals, !OPT);
        // the quad behind the gradients (include/mmg.h: mmg_grad_floats): [0] = 1.0 when a dependency wait of this minibatch
        // timed out on THIS rank.  The data-parallel all-reduce sums it with the gradients, so every rank's k_opt sees that
        // some rank's contribution is built from stale data and all of them skip the update together.
        // [1], [2]: this rank's sum of rewards and top-k hits.  Continuous messages couple the shards through nothing else
        // (loss = NLL mean over the GLOBAL batch, model.py:1297-1305): the data-parallel step then needs no statistics
        // all-reduce -- the two sums travel with the gradients and k_gradnorm rewrites the logged NLL / hit count from them.
        if (threadIdx.x == 0) {
            grad_tail[0] = (__hip_atomic_load(sync + MMG_SYNC_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1.f : 0.f;
            grad_tail[1] = (float)stats[stat_glob(dm.T, 0)];
            grad_tail[2] = (float)stats[stat_glob(dm.T, 1)];
            grad_tail[3] = 0.f;
        }
        if (OPT) {
            // ---- the CLOSING role of the launch: once the four norm roles have published their totals, every block of the launch
            // has published its sum of squares, i.e. has read the launch epoch / optimizer step counters -- they may move now.
            // NON-FINITE GUARD: the class-logit ReLU is v_max_f32 (device_utils.h: fmax_nn), which reads a NaN pre-activation as
            // "unit off" where torch's relu propagates it -- a NaN in W_y1h or in the GRU state would leave the NLL of that step
            // finite (log D) while every reference loss is NaN.  The backward pass does carry it, so a non-finite gradient norm of
            // ANY agent makes the logged NLL NaN in the same step, as the reference's would be (scripts/nonfinite_probe.py).
            __shared__ float s_tot[4];
            if (threadIdx.x < 4) {
                unsigned long long u;
                for (int spins = 0;; ) {
                    u = ld_ll(wo.coefll, (size_t)4 * MMG_COEF_REPL + threadIdx.x);
                    if (ll_fresh(u, oepoch)) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1 << 20)) { __hip_atomic_store(const_cast<uint32_t*>(sync) + MMG_SYNC_ERR, 11u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
                s_tot[threadIdx.x] = ll_value(u);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                bool bad = false;
#pragma unroll
                for (int a = 0; a < 4; ++a) bad = bad || !(s_tot[a] == s_tot[a]) || s_tot[a] > 3.0e38f;
                losses[0] = bad ? __builtin_nanf("") : (float)(-stats[stat_glob(dm.T, 0)] / (double)dm.Bg);     // NLL (model.py:1271)
                const uint32_t err = __hip_atomic_load(sync + MMG_SYNC_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (wo.err_host && (err != 0u || *wo.err_host == 0u)) *wo.err_host = err;
                if (err == 0u) wo.counter[2] = ostep;                           // committed to counter[1] by the next k_prep
                if (wo.oa.bump_mb) { wo.counter[0] += 1u; wo.counter[3] += 1u; }
            }
        }
        return;
    }
    if (bid < hd.gemm_tiles) {
      __shared__ int s_arrived;
      for (int vt = bid; vt < hd.gemm_tiles; vt += (gt_stride > 0 ? gt_stride : hd.gemm_tiles)) {
        // XCD-aware tile order: workgroup b is dispatched to XCD b % 8, and each XCD has its own 4 MB L2.  Giving
        // every XCD a CONTIGUOUS range of tiles (= one or two jobs) keeps the operand tapes it re-reads
        // (16-32 tiles share each of them) inside its L2; with the default order every XCD streams all ~7 MB
        // of tapes through its L2 and the kernel is bound by L2-miss traffic.  (bijective remap, T1)
        const int nwg = hd.gemm_tiles, xq = nwg >> 3, xr = nwg & 7;
        const int xcd = vt & 7, slot = vt >> 3;
        const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + slot;
        // job lookup: lane l compares the l-th job's first tile, one ballot (no serial scalar loads)
        const int j = __popcll(__ballot(jt->g_begin[lane] <= tile)) - 1;
        const GemmJob& G = jt->g[j];
        // many rows, few output tiles (thousands of samples): the rows of a tile are split over nsplit workgroups whose raw
        // partial tiles the last of them to arrive adds in slice order
        const int ns = G.nsplit > 1 ? G.nsplit : 1;
        const int lts = tile - G.tile_begin, sp = lts % ns, lt = lts / ns;
        const int tn = lt / G.tiles_k, tk = lt - tn * G.tiles_k;
        const int n0 = tn * 16, k0 = tk * 32;
        const int i = lane & 15, q = lane >> 4;
        const float* Bbase = (G.bsrc == SRC_X) ? x : (G.bsrc == SRC_DESC) ? desc : G.Bm;
        const bool virt = G.vhid != nullptr;
        const float* Abase = virt ? G.vhid : G.A;
        const bool cmp = use_map && G.compact;
        const int lda = G.lda, ldb = G.ldb, bmod = G.bmod, N = G.N, K = G.K;
        // The live-row list (s_map, nact) is a memory trip to what the previous launch just wrote -- ~2-3 us -- and so is the first
        // operand chunk.  Round 6: the two trips overlap.  Jobs over plain rows (image_layer, the class jobs) never needed the list;
        // and the compacted list BEGINS with the B rows of step 0 in order (build_row_map: every sample is live at t = 0), so
        // with B >= 64 chunk 0 of a live-row job is rows 0..63 -- both kinds issue their first loads BEFORE the barrier that
        // waits for the list (`early`), the rest of the prefetch ring behind it.  Nothing below may read nact ahead of that barrier.
        const bool ident0 = cmp && dm.B >= 64 && sp == 0;
        int rows = cmp ? 0 : G.rows;                       // (live-row jobs: set behind the barrier)
        const bool veca = ((lda & 3) == 0) && ((((uintptr_t)Abase) & 15) == 0);
        const bool vecb = ((ldb & 3) == 0) && ((((uintptr_t)Bbase) & 15) == 0);
        // loader roles: A row la, columns lac..+3 (64 rows x 16 cols); B row lb0 / lb0+32, columns lbc..+3 (64 x 32)
        const int la = threadIdx.x >> 2, lac = (threadIdx.x & 3) * 4;
        const int lb0 = threadIdx.x >> 3, lbc = (threadIdx.x & 7) * 4;
        float4 vw = make_float4(0.f, 0.f, 0.f, 0.f);
        if (virt) vw = load4_guard(G.vw2, 0, n0 + lac, N, true, false);
        int nchunks_all = 0, cps = 0, cbeg = 0, cend = 0, nchunks = 0, rows_end = 0, rlast = 0;
        auto set_rows = [&](int r) {
            rows = r;
            nchunks_all = (rows + CH - 1) / CH; cps = (nchunks_all + ns - 1) / ns;
            cbeg = sp * cps; cend = min(nchunks_all, cbeg + cps); nchunks = max(cend - cbeg, 0);
            rows_end = min(rows, cend * CH);                     // rows of this slice: [cbeg * CH, rows_end)
            rlast = rows - 1;
        };
        if (!cmp) set_rows(G.rows);
        constexpr int DEPTH = 3;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        // Branch-free steady state: every load is unconditional (indices clamped, values masked afterwards), so the
        // compiler keeps COUNTED s_waitcnt vmcnt(N).  With any branch around a load it falls back to vmcnt(0) at the
        // join, which drains the prefetches and costs one full memory round trip (~1.5 us) per chunk.
        // interior tiles: one 16-byte load per operand slice; edge tiles (N or K tail, unaligned rows): four clamped
        // scalar loads -- the same pipeline either way.
        // Tile tails may over-read: columns beyond N (A) / K (B) only feed output elements that are never stored, and
        // every operand except x lives inside the workspace (the job tables follow the last operand array).
        const bool interior = veca && vecb && ((G.bsrc != SRC_X) || (k0 + 32 <= K));
        // branch-free "row % bmod" (identity when unused) WITHOUT a runtime division (~9 instructions apiece, two per chunk and
        // thread: the loop around the eight MFMAs of a chunk is issue-bound): row < 2^16, so q = (row * ceil(2^32 / bmod)) >> 32 is
        // floor(row / bmod) or one more -- one conditional correction
        const int bmodv = bmod ? bmod : 0x7fffffff;
        const uint32_t bmagic = bmod > 1 ? (uint32_t)((0x100000000ull + (unsigned)bmod - 1ull) / (unsigned)bmod) : 0u;
        const bool bmod_one = bmod == 1;                     // (its magic number is 2^32: every row is row 0)
        (void)bmodv;
        auto rowmod = [&](int r) -> int {
            const int q = (int)__umulhi((uint32_t)r, bmagic);
            const int m = r - q * bmod;                      // (bmod = 0: q = 0, m = r)
            return bmod_one ? 0 : (m < 0 ? m + bmod : m);
        };
        // (global address space stated: pointers out of the job table are generic to the compiler -- flat loads with 64-bit address
        //  arithmetic per access; as_global: 125 -> 65 instructions per chunk and thread around the eight MFMAs, and the loop is
        //  bound by the SIMDs' instruction issue: four workgroups per CU x ~500 issue cycles per chunk of 2 400)
        gfloat* const Ag = as_global(Abase);
        gfloat* const Bg = as_global(Bbase);
        gfloat* const betag = as_global(G.A);                // virt: d score per row; otherwise any valid address
        // One instantiation of the steady loop per KIND of job -- virtual A operand or not, live-row list or plain rows, a
        // per-sample B operand (row % B) or not -- because the loop is bound by the SIMDs' instruction issue (round-6 counters at
        // config 3, 512 samples: 42 % of the wave cycles stalled on issue, 20 % issuing, MFMA pipes 23 % busy): what a job does
        // not use must not be in its loop.  The commonest kind (live rows, plain operands) drops the score load, the ReLU mask
        // arithmetic and two magic modulos: ~65 -> ~40 instructions per chunk and thread.
        auto run = [&](auto virt_tag, auto cmp_tag, auto bmod_tag) {
            constexpr bool VIRT = decltype(virt_tag)::value, CMPJ = decltype(cmp_tag)::value, BMOD = decltype(bmod_tag)::value;
            float4 ra[DEPTH], rb0[DEPTH], rb1[DEPTH];
            float beta[DEPTH];
            // (32-bit element offsets from the job's base pointers: every operand array is far below 2^32 floats)
            const uint32_t acol = (uint32_t)(n0 + lac), bcol = (uint32_t)(k0 + lbc), ulda = (uint32_t)lda, uldb = (uint32_t)ldb;
            auto issue = [&](int rc_, int c0_, int c1_, int u) {                         // (row indices already mapped)
                ra[u] = ldg4(Ag + ((uint32_t)rc_ * ulda + acol));
                if (VIRT) beta[u] = betag[(uint32_t)rc_];
                rb0[u] = ldg4(Bg + ((uint32_t)(BMOD ? rowmod(c0_) : c0_) * uldb + bcol));
                rb1[u] = ldg4(Bg + ((uint32_t)(BMOD ? rowmod(c1_) : c1_) * uldb + bcol));
            };
            auto fetch = [&](int c, int u) {
                const int r_ = c * CH + la, r0_ = c * CH + lb0, r1_ = r0_ + 32;
                int rc_ = min(r_, rlast), c0_ = min(r0_, rlast), c1_ = min(r1_, rlast);
                if (CMPJ) { rc_ = s_map[rc_]; c0_ = s_map[c0_]; c1_ = s_map[c1_]; }     // (LDS reads: the global loads stay branch-free)
                issue(rc_, c0_, c1_, u);
            };
            // the prefetch ring's first loads: ahead of the list where they do not need it (see `early` above)
            if (!CMPJ) {
#pragma unroll
                for (int u = 0; u < DEPTH - 1; ++u) fetch(cbeg + u, u);
            } else if (ident0) issue(la, lb0, lb0 + 32, 0);
            if (use_map) __syncthreads();                      // s_map is complete
#ifdef MMG_TIMING
            if (threadIdx.x == 0 && tile < 2000) dbg2[8192 + 4 * tile + 0] = (long long)wall_clock64();
#endif
            if (CMPJ) {
                set_rows(nact);
#pragma unroll
                for (int u = 0; u < DEPTH - 1; ++u) if (u > 0 || !ident0) fetch(cbeg + u, u);
            }
            const int nouter = (nchunks + DEPTH - 1) / DEPTH;
            for (int o = 0; o < nouter; ++o) {
#pragma unroll
                for (int u = 0; u < DEPTH; ++u) {
                    const int c = cbeg + o * DEPTH + u;
                    const int buf = (o * DEPTH + u) & 1;
                    const bool va = (c * CH + la) < rows_end;
                    float4 av = ra[u];
                    if (VIRT) {
                        av.x = av.x > 0.f ? beta[u] * vw.x : 0.f; av.y = av.y > 0.f ? beta[u] * vw.y : 0.f;
                        av.z = av.z > 0.f ? beta[u] * vw.z : 0.f; av.w = av.w > 0.f ? beta[u] * vw.w : 0.f;
                    }
                    // rows beyond the slice: the A slice is zeroed; the B slice may stay what the clamped load fetched -- a LIVE row's
                    // finite values times zero (eight multiplies per chunk and thread less)
                    const float ma = va ? 1.f : 0.f;
                    *reinterpret_cast<float4*>(&s_a[buf][la][lac]) = make_float4(av.x * ma, av.y * ma, av.z * ma, av.w * ma);
                    *reinterpret_cast<float4*>(&s_b[buf][lb0][lbc]) = rb0[u];
                    *reinterpret_cast<float4*>(&s_b[buf][lb0 + 32][lbc]) = rb1[u];
                    __syncthreads();
                    fetch(c + DEPTH - 1, (u + DEPTH - 1) % DEPTH);        // unconditional: clamped beyond the last row
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int r = wave * 16 + v * 4 + q;
                        const float a = s_a[buf][r][i];
                        acc0 = mfma16(a, s_b[buf][r][i], acc0);
                        acc1 = mfma16(a, s_b[buf][r][16 + i], acc1);
                    }
                }
            }
        };
        auto run_b = [&](auto virt_tag, auto cmp_tag) {
            if (bmod != 0) run(virt_tag, cmp_tag, std::true_type{}); else run(virt_tag, cmp_tag, std::false_type{});
        };
        auto run_c = [&](auto virt_tag) { if (cmp) run_b(virt_tag, std::true_type{}); else run_b(virt_tag, std::false_type{}); };
        if (interior) {
            if (virt) run_c(std::true_type{}); else run_c(std::false_type{});
        } else {
            // edge tiles (N or K tail inside the tile, unaligned rows): guarded loads, one chunk at a time
            if (use_map) __syncthreads();                      // s_map is complete
            if (cmp) set_rows(nact);
            for (int c = cbeg; c < cend; ++c) {
                const int buf = c & 1;
                const int rl = c * CH + la;
                const bool rv = rl < rows_end;
                const int r = cmp ? (int)s_map[min(rl, rlast)] : rl;
                float4 av = load4_guard(Abase, (size_t)(rv ? r : 0) * lda, n0 + lac, N, rv, veca);
                if (virt) {
                    const float be = rv ? betag[r] : 0.f;
                    av.x = av.x > 0.f ? be * vw.x : 0.f; av.y = av.y > 0.f ? be * vw.y : 0.f;
                    av.z = av.z > 0.f ? be * vw.z : 0.f; av.w = av.w > 0.f ? be * vw.w : 0.f;
                }
                const int r0l = c * CH + lb0, r1l = r0l + 32;
                const bool v0 = r0l < rows_end, v1 = r1l < rows_end;
                const int r0 = cmp ? (int)s_map[min(r0l, rlast)] : r0l, r1 = cmp ? (int)s_map[min(r1l, rlast)] : r1l;
                const int m0 = bmod ? (r0 % bmod) : r0, m1 = bmod ? (r1 % bmod) : r1;
                const float4 b0v = load4_guard(Bbase, (size_t)(v0 ? m0 : 0) * ldb, k0 + lbc, K, v0, vecb);
                const float4 b1v = load4_guard(Bbase, (size_t)(v1 ? m1 : 0) * ldb, k0 + lbc, K, v1, vecb);
                *reinterpret_cast<float4*>(&s_a[buf][la][lac]) = av;
                *reinterpret_cast<float4*>(&s_b[buf][lb0][lbc]) = b0v;
                *reinterpret_cast<float4*>(&s_b[buf][lb0 + 32][lbc]) = b1v;
                __syncthreads();
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int rr2 = wave * 16 + v * 4 + q;
                    const float a = s_a[buf][rr2][i];
                    acc0 = mfma16(a, s_b[buf][rr2][i], acc0);
                    acc1 = mfma16(a, s_b[buf][rr2][16 + i], acc1);
                }
            }
        }
        __syncthreads();                                   // every wave is done reading the staging buffers
#ifdef MMG_TIMING
        if (threadIdx.x == 0 && tile < 2000) dbg2[8192 + 4 * tile + 1] = (long long)wall_clock64();
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) { s_acc[wave][q * 4 + r][i] = acc0[r]; s_acc[wave][q * 4 + r][16 + i] = acc1[r]; }
        __syncthreads();
        float sq = 0.f;
        float gv2[2]; int64_t gi2[2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
            const int rr = threadIdx.x >> 4, cc = (threadIdx.x & 15) + 16 * h2;
            const int n = n0 + rr, k = k0 + cc;
            const float v = (s_acc[0][rr][cc] + s_acc[1][rr][cc]) + (s_acc[2][rr][cc] + s_acc[3][rr][cc]);
            gv2[h2] = v; gi2[h2] = -1;
            if (ns > 1) st_wt(&wpart[(size_t)tile * 512 + rr * 32 + cc], v);       // raw partial tile, written through
            else if (n < N && k < K) {
                as_global_w(G.C)[(size_t)n * G.ldc + k] = v;
                sq = fmaf(v, v, sq);
                gi2[h2] = (G.C - wo.grads) + (int64_t)n * G.ldc + k;
            }
        }
        bool adds_up = false;
        if (ns > 1) {
            // Row slices of one output tile (thousands of (step, sample) rows): the slice that ARRIVES LAST adds the ns raw partial
            // tiles -- in slice order whoever it is, so the sum does not depend on the arrival order -- and stores the gradient tile
            // and its sum of squares.  (Through round 5 a second launch, k_wreduce, did that: ~5 us per minibatch of configs 3 / 5
            // at full batch and of config 5's shard.)  Nobody waits: the count is a returning atomic behind the workgroup's own
            // write-through stores, the partial tiles are read with agent-scope loads, and the last slice zeroes the count.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int tile0 = tile - sp;
            if (threadIdx.x == 0) s_arrived = (int)__hip_atomic_fetch_add(wcnt + tile0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            adds_up = __builtin_amdgcn_readfirstlane(s_arrived) == ns - 1;
            if (adds_up) {
                if (threadIdx.x == 0) __hip_atomic_store(wcnt + tile0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int rr = threadIdx.x >> 4, cc = (threadIdx.x & 15) + 16 * h2;
                    float v = 0.f;
                    for (int s0 = 0; s0 < ns; s0 += 8) {
                        float pv[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) pv[u] = ld_cc(&wpart[(size_t)(tile0 + min(s0 + u, ns - 1)) * 512 + rr * 32 + cc]);
#pragma unroll
                        for (int u = 0; u < 8; ++u) v += (s0 + u < ns) ? pv[u] : 0.f;
                    }
                    const int n = n0 + rr, k = k0 + cc;
                    if (n < N && k < K) { as_global_w(G.C)[(size_t)n * G.ldc + k] = v; sq = fmaf(v, v, sq); }
                }
            }
        }
        sq = block_sum(sq, s_red);
        if (threadIdx.x == 0) {
            if (ns == 1) part[tile] = sq;
            else {                                         // (the tile's sum of squares sits in its FIRST slice's entry, the others are zero)
                if (sp != 0) part[tile] = 0.f;
                if (adds_up) part[tile - sp] = sq;
            }
        }
        if (OPT) {
            if (threadIdx.x == 0) st_ll(wo.gnll, tile, sq, oepoch);
            // parameter / state elements of this thread: requested now, they arrive while the norm role adds up
            float w[2], s1[2], s2[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int64_t ic = gi2[h2] >= 0 ? gi2[h2] : 0;
                w[h2] = wo.params[ic]; s1[h2] = wo.state[ic];
                s2[h2] = (wo.oa.optim_type == MMG_OPT_ADAM) ? wo.state[wo.oa.total + ic] : 0.f;
            }
#ifdef MMG_TIMING
            if (threadIdx.x == 0 && tile < 2000) dbg2[8192 + 4 * tile + 2] = (long long)wall_clock64();
#endif
            const float coef = opt_wait_coef(wo, jt->wblock_agent[tile], oepoch, sync);
#ifdef MMG_TIMING
            if (threadIdx.x == 0 && tile < 2000) dbg2[8192 + 4 * tile + 3] = (long long)wall_clock64();
#endif
            if (coef >= 0.f) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
                    if (gi2[h2] >= 0) opt_update_one(wo.oa, wo.params, wo.state, gi2[h2], gv2[h2], coef, w[h2], s1[h2], s2[h2], ostep);
            }
        }
        __syncthreads();                                   // (the staging buffers / s_acc are the next tile's)
      }
        MMG_WG_END();
        return;
    }
    // ---- column sums: 16 columns x 16 row groups per block, 4 independent loads in flight per thread
    const int cb = bid - hd.gemm_tiles;
    const int j = __popcll(__ballot(jt->c_begin[lane] <= cb)) - 1;
    const ColJob& C = jt->c[j];
    const bool ccmp = use_map && C.compact;
    if (use_map) __syncthreads();                          // s_map is complete
    const int c0 = (cb - C.blk_begin) * 16;
    const int rows = ccmp ? nact : C.rows, ld = C.ld;
    const bool virt = C.vbeta != nullptr;
    const bool wsum = C.wrow != nullptr;
    gfloat* const rowv = as_global(virt ? C.vbeta : wsum ? C.wrow : C.src);      // (neither: any valid address, the value is unused)
    // Sixteen whole columns of 16-byte-aligned rows (every job but a ragged last block): FOUR columns per thread, 64 row groups,
    // eight rows in flight per thread -- 512 rows per memory trip and 16-byte loads; the scalar form below (one column per
    // thread, 256 rows per trip, 64-byte segments) kept config 3 at 512 samples waiting for its 141 column blocks: 5 632 rows,
    // 22 trips, 88 us while the GEMM tiles were through after 77 (round-6 timeline).  Loads are unconditional (rows clamped,
    // contributions masked): the compiler keeps them all in flight.
    const bool vec4 = (c0 + 16 <= C.cols) && !(ld & 3) && !(((uintptr_t)(C.src + c0)) & 15) && rows > 0;
    int ngroups = 16;
    if (vec4) {
        ngroups = 64;
        const int cq = (threadIdx.x & 3) * 4, g = threadIdx.x >> 2;
        gfloat* const sp = as_global(C.src) + (c0 + cq);
        float4 vw = make_float4(0.f, 0.f, 0.f, 0.f);
        if (virt) vw = make_float4(C.vw2[c0 + cq], C.vw2[c0 + cq + 1], C.vw2[c0 + cq + 2], C.vw2[c0 + cq + 3]);
        float4 A0 = make_float4(0.f, 0.f, 0.f, 0.f), A1 = A0;
        constexpr int UN = 8;
        const int rlast = rows - 1;
        for (int r0 = g; r0 < rows; r0 += 64 * UN) {
            float4 hv[UN]; float bv[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int rl = min(r0 + 64 * u, rlast);
                const int r = ccmp ? (int)s_map[rl] : rl;
                hv[u] = ldg4(sp + (size_t)r * ld);
                bv[u] = rowv[(virt || wsum) ? r : 0];
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const bool ok = (r0 + 64 * u) < rows;
                float4& A = (u & 1) ? A1 : A0;
                if (wsum) {
                    const float w = ok ? bv[u] : 0.f;
                    A.x = fmaf(hv[u].x, w, A.x); A.y = fmaf(hv[u].y, w, A.y); A.z = fmaf(hv[u].z, w, A.z); A.w = fmaf(hv[u].w, w, A.w);
                } else if (virt) {
                    const float w = ok ? bv[u] : 0.f;
                    A.x += hv[u].x > 0.f ? w * vw.x : 0.f; A.y += hv[u].y > 0.f ? w * vw.y : 0.f;
                    A.z += hv[u].z > 0.f ? w * vw.z : 0.f; A.w += hv[u].w > 0.f ? w * vw.w : 0.f;
                } else {
                    A.x += ok ? hv[u].x : 0.f; A.y += ok ? hv[u].y : 0.f; A.z += ok ? hv[u].z : 0.f; A.w += ok ? hv[u].w : 0.f;
                }
            }
        }
        *reinterpret_cast<float4*>(&s_part[g * 16 + cq]) = make_float4(A0.x + A1.x, A0.y + A1.y, A0.z + A1.z, A0.w + A1.w);
    } else {
        // a ragged last block or a narrow job (the N = 1 products: ONE column over all (step, sample) rows): the columns on the
        // low lane bits, every other thread of the workgroup a row group of its own -- 256 row groups for one column instead of 16
        const int ncol = min(16, C.cols - c0);
        const int lp = ncol > 8 ? 4 : ncol > 4 ? 3 : ncol > 2 ? 2 : ncol > 1 ? 1 : 0, P = 1 << lp, NG = MMG_BLOCK >> lp;
        const int cc = threadIdx.x & (P - 1), g = threadIdx.x >> lp;
        const bool cv = cc < ncol;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (cv) {
            const float* sp = C.src + c0 + cc;
            constexpr int UN = 16;                         // rows in flight per thread
            const float vw = virt ? C.vw2[c0 + cc] : 0.f;
            for (int r0 = g; r0 < rows; r0 += NG * UN) {
                float hv[UN], bv[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int rl = r0 + NG * u;
                    const bool ok = rl < rows;
                    const int r = (ccmp && ok) ? (int)s_map[rl] : rl;
                    hv[u] = ok ? sp[(size_t)r * ld] : 0.f;
                    bv[u] = (ok && (virt || wsum)) ? rowv[r] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < UN; u += 4) {
                    if (wsum) {
                        a0 = fmaf(hv[u], bv[u], a0); a1 = fmaf(hv[u + 1], bv[u + 1], a1);
                        a2 = fmaf(hv[u + 2], bv[u + 2], a2); a3 = fmaf(hv[u + 3], bv[u + 3], a3);
                    } else if (virt) {
                        a0 += hv[u] > 0.f ? bv[u] * vw : 0.f; a1 += hv[u + 1] > 0.f ? bv[u + 1] * vw : 0.f;
                        a2 += hv[u + 2] > 0.f ? bv[u + 2] * vw : 0.f; a3 += hv[u + 3] > 0.f ? bv[u + 3] * vw : 0.f;
                    } else {
                        a0 += hv[u]; a1 += hv[u + 1]; a2 += hv[u + 2]; a3 += hv[u + 3];
                    }
                }
            }
        }
        // across the row groups: the lanes P apart within a wave (fixed butterfly), then the four waves below
        float av = (a0 + a1) + (a2 + a3);
        for (int off = 32; off >= P; off >>= 1) av += __shfl_xor(av, off);
        ngroups = 4;
        if ((threadIdx.x & 63) < P) s_part[(threadIdx.x >> 6) * 16 + cc] = av;
    }
    __syncthreads();
    float v = 0.f;
    if (threadIdx.x < 16 && (c0 + (int)threadIdx.x) < C.cols) {
        for (int k = 0; k < ngroups; ++k) v += s_part[k * 16 + threadIdx.x];
        if (C.scale) v *= C.scale[c0 + threadIdx.x];
        C.dst[c0 + threadIdx.x] = v;
    }
    const float sq = block_sum(v * v, s_red);
    if (threadIdx.x == 0) part[bid] = sq;
    if (OPT) {
        if (threadIdx.x == 0) st_ll(wo.gnll, bid, sq, oepoch);
        const bool mine = threadIdx.x < 16 && (c0 + (int)threadIdx.x) < C.cols;
        const int64_t idx = mine ? (C.dst - wo.grads) + c0 + threadIdx.x : 0;
        const float w = wo.params[idx], s1 = wo.state[idx], s2 = (wo.oa.optim_type == MMG_OPT_ADAM) ? wo.state[wo.oa.total + idx] : 0.f;
        const float coef = opt_wait_coef(wo, jt->wblock_agent[bid], oepoch, sync);
        if (coef >= 0.f && mine) opt_update_one(wo.oa, wo.params, wo.state, idx, v, coef, w, s1, s2, ostep);
    }
    MMG_WG_END();
}

// ---------------------------------------------------------------------------------------------
// Per-agent gradient norm (clip_grad_norm, model.py:1310) and the optimizer update.
// ---------------------------------------------------------------------------------------------
// fix_bg > 0 (continuous messages, data parallel): the tail quad behind the all-reduced gradients holds the GLOBAL sum of
// rewards and hit count (k_wgrad's spare block wrote this rank's share): the logged NLL / hits / running hit total, which
// that block derived from the local sums, are rewritten here.
__global__ __launch_bounds__(MMG_BLOCK) void k_gradnorm(const JobTable* __restrict__ jt, const float* __restrict__ grads,
                                                        float* __restrict__ part, uint32_t* __restrict__ counter,
                                                        const float* __restrict__ grad_tail, float* __restrict__ losses,
                                                        double* __restrict__ totals, int fix_bg) {
    __shared__ float s_red[8];
    const int blk = blockIdx.x;
    const int64_t b0 = jt->np.begin[blk], b1 = jt->np.end[blk];      // multiples of 4 floats
    float acc = 0.f;
    for (int64_t i = b0 + (int64_t)threadIdx.x * 4; i < b1; i += (int64_t)blockDim.x * 4) {
        const float4 g = *reinterpret_cast<const float4*>(grads + i);
        acc = fmaf(g.x, g.x, acc); acc = fmaf(g.y, g.y, acc); acc = fmaf(g.z, g.z, acc); acc = fmaf(g.w, g.w, acc);
    }
    acc = block_sum(acc, s_red);
    if (threadIdx.x == 0) {
        part[blk] = acc;
        if (blk == 0) counter[1] += 1u;                               // optimizer step count (Adam bias correction)
        if (blk == 0 && fix_bg > 0) {
            const float hits = grad_tail[2];
            totals[1] += (double)hits - (double)losses[7];
            losses[0] = -grad_tail[1] / (float)fix_bg;                // NLL of the global minibatch (model.py:1271)
            losses[7] = hits;
        }
    }
}


__global__ __launch_bounds__(MMG_BLOCK) void k_opt(const JobTable* __restrict__ jt, OptArgs oa, float* __restrict__ params,
                                                   const float* __restrict__ grads, float* __restrict__ state,
                                                   const float* __restrict__ part, const uint32_t* __restrict__ counter,
                                                   const uint32_t* __restrict__ sync, uint32_t* __restrict__ err_host,
                                                   const float* __restrict__ grad_tail, float* __restrict__ losses) {
    // An in-launch dependency wait of this minibatch timed out (device_utils.h: role_wait sets sync[MMG_SYNC_ERR]) -- on this
    // rank, or (grad_tail: the flag quad that travelled through the gradient all-reduce) on ANY rank of a data-parallel job:
    // the gradients may be built from stale data -- leave parameters and optimizer state untouched, on every rank alike.
    uint32_t err = __hip_atomic_load(sync + MMG_SYNC_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (err == 0u && grad_tail && grad_tail[0] != 0.f) err = MMG_SYNC_ERR_REMOTE;
    // fused step of kernels_game.h: no role of its launch may see the minibatch counter (the Philox stream) or the launch epoch of
    // the (value, epoch) pairs move, so the LAST launch of the step commits them (nothing in this kernel reads either)
    if (oa.bump_mb && blockIdx.x == 0 && threadIdx.x == 0) { const_cast<uint32_t*>(counter)[0] += 1u; const_cast<uint32_t*>(counter)[3] += 1u; }
    // the word goes to a pinned HOST word (device-mapped) with a posted store: the host reads it before the next minibatch
    // without any stream operation or synchronisation; it is sticky, every later training call fails (mmg.hip: sticky_error)
    if (err_host && blockIdx.x == 0 && threadIdx.x == 0 && (err != 0u || *err_host == 0u)) *err_host = err;
    if (err != 0u) return;
    // oa.from_wgrad: the squared-norm partials are the ones k_wgrad left per block (single GPU);
    // otherwise the MMG_GN_BLOCKS partials of k_gradnorm over the all-reduced gradient (data parallel).
    __shared__ float s_coef[4];
    __shared__ float s_ss[4][4];
    // this thread's element quad: issue its loads first so they share one memory round trip with the partials
    float* st1 = state; float* st2 = state + oa.total;
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const bool have = i0 < oa.total;
    const int64_t il = have ? i0 : 0;
    const float4 g0 = *reinterpret_cast<const float4*>(grads + il);
    const float4 w0 = *reinterpret_cast<const float4*>(params + il);
    const float4 s0 = *reinterpret_cast<const float4*>(st1 + il);
    const float4 v0 = (oa.optim_type == MMG_OPT_ADAM) ? *reinterpret_cast<const float4*>(st2 + il) : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint32_t step = counter[1] + (oa.bump_step ? 1u : 0u);      // k_gradnorm bumps it in the DP path
    {
        const int n = oa.from_wgrad ? jt->n_wblocks : MMG_GN_BLOCKS;
        float ss[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = threadIdx.x; k0 < n; k0 += 8 * MMG_BLOCK) {      // 8 partials (16 loads) in flight per thread
            float v[8]; int a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * MMG_BLOCK;
                const bool ok = k < n;
                v[u] = ok ? part[k] : 0.f;
                a[u] = ok ? (oa.from_wgrad ? (int)jt->wblock_agent[k] : jt->np.agent[k]) : -1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ss[0] += (a[u] == 0) ? v[u] : 0.f; ss[1] += (a[u] == 1) ? v[u] : 0.f;
                ss[2] += (a[u] == 2) ? v[u] : 0.f; ss[3] += (a[u] == 3) ? v[u] : 0.f;
            }
        }
        // fixed reduction tree (DPP inside a wave, then the four waves in order): deterministic
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float wsum = dpp_wave_sum(ss[a]);
            if ((threadIdx.x & 63) == 0) s_ss[a][threadIdx.x >> 6] = wsum;
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            const float tot = (s_ss[threadIdx.x][0] + s_ss[threadIdx.x][1]) + (s_ss[threadIdx.x][2] + s_ss[threadIdx.x][3]);
            const float norm = sqrtf(tot);
            const float coef = 1.0f / (norm + 1e-6f);                // max_norm = 1 (model.py:1310)
            s_coef[threadIdx.x] = coef < 1.f ? coef : 1.f;
            // non-finite guard (k_wgrad<OPT>'s norm role): a non-finite gradient norm of any agent makes the logged NLL NaN
            if (losses && blockIdx.x == 0 && (!(tot == tot) || tot > 3.0e38f)) losses[0] = __builtin_nanf("");
        }
    }
    __syncthreads();
    const float b1 = 0.9f, b2 = 0.999f;
    float bc1 = 1.f, bc2s = 1.f;
    if (oa.optim_type == MMG_OPT_ADAM) {
        bc1 = 1.f - powf(b1, (float)step);
        bc2s = sqrtf(1.f - powf(b2, (float)step));
    }
    for (int64_t i = i0; i < oa.total; i += (int64_t)gridDim.x * blockDim.x * 4) {
        int a = 0;
        if (i >= oa.agent_begin[1]) a = 1;
        if (i >= oa.agent_begin[2]) a = 2;
        if (i >= oa.agent_begin[3]) a = 3;
        if (oa.only_receiver && a != 0) continue;                     // model.py:1313
        const float coef = s_coef[a];
        const bool first = (i == i0);
        float4 g = first ? g0 : *reinterpret_cast<const float4*>(grads + i);
        float4 w = first ? w0 : *reinterpret_cast<float4*>(params + i);
        float gv[4] = {g.x * coef, g.y * coef, g.z * coef, g.w * coef};
        float wv[4] = {w.x, w.y, w.z, w.w};
        if (oa.optim_type == MMG_OPT_RMSPROP) {                       // torch.optim.RMSprop defaults (model.py:1128)
            float4 sq = first ? s0 : *reinterpret_cast<float4*>(st1 + i);
            float sv[4] = {sq.x, sq.y, sq.z, sq.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sv[k] = 0.99f * sv[k] + (1.f - 0.99f) * gv[k] * gv[k];
                wv[k] -= oa.lr * gv[k] / (sqrtf(sv[k]) + 1e-8f);
            }
            *reinterpret_cast<float4*>(st1 + i) = make_float4(sv[0], sv[1], sv[2], sv[3]);
        } else if (oa.optim_type == MMG_OPT_ADAM) {                   // torch.optim.Adam defaults (model.py:1120)
            float4 m = first ? s0 : *reinterpret_cast<float4*>(st1 + i), v = first ? v0 : *reinterpret_cast<float4*>(st2 + i);
            float mv[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mv[k] = mv[k] + (gv[k] - mv[k]) * (1.f - b1);
                vv[k] = b2 * vv[k] + (1.f - b2) * gv[k] * gv[k];
                const float denom = sqrtf(vv[k]) / bc2s + 1e-8f;
                wv[k] -= (oa.lr / bc1) * mv[k] / denom;
            }
            *reinterpret_cast<float4*>(st1 + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
            *reinterpret_cast<float4*>(st2 + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        } else {                                                      // SGD (model.py:1112)
#pragma unroll
            for (int k = 0; k < 4; ++k) wv[k] -= oa.lr * gv[k];
        }
        *reinterpret_cast<float4*>(params + i) = make_float4(wv[0], wv[1], wv[2], wv[3]);
    }
    // every block has read counter[1] above; the bump is published by the kernel boundary
    if (oa.bump_step && blockIdx.x == gridDim.x - 1) {
        __syncthreads();
        if (threadIdx.x == 0) const_cast<uint32_t*>(counter)[2] = step;   // committed to counter[1] by the next k_prep
    }
}


// ---------------------------------------------------------------------------------------------
// k_log_snapshot: everything the log block of a minibatch prints (model.py:1342-1461) gathered into ONE flat f64 vector by ONE
// launch (round 6; the host ran ~16 small torch kernels for it -- softmax / log / sums over [T, B, D], six dtype casts, a cat --
// 0.2 ms of GPU time per log block, 4 us per minibatch at -log_interval 50).  Layout (include/mmg.h: mmg_log_snapshot):
//   with_losses: losses[8] | totals[1] | stats[NSTAT] | ent[T] | argmax[B] | target[B]
//     ent[t] = mean_b sum_d softmax(y_t)[b, d] log(softmax(y_t)[b, d] + 1e-8)   ("Entropy Receiver Predictions", model.py:880-886,
//     over the WHOLE batch: the tape of a run-all minibatch); argmax[b] = first maximum of dist[b, :] ("Predictions")
//   dump = k > 0:  alive[T] | pz[T, k, W] | pw[T, k, W] | z[T, k, W] | w[T, k, W] | ps[T, k] | mask[1:, :k][T, k]
// grid = T + 2 workgroups: one per step's entropy, one for the scalars / predictions, one for the dump.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int64_t log_snapshot_count(int T, int B, int W, int k, bool with_losses) {
    return (with_losses ? 8 + 1 + (int64_t)stat_count(T) + T + 2 * (int64_t)B : 0) + (k > 0 ? (int64_t)T + 4 * (int64_t)T * k * W + 2 * (int64_t)T * k : 0);
}
__global__ __launch_bounds__(MMG_BLOCK) void k_log_snapshot(Dims dm, Tape tp, const int64_t* __restrict__ target, int k, int with_losses, double* __restrict__ out) {
    __shared__ float s_red[8];
    const int T = dm.T, B = dm.B, D = dm.D, W = dm.W, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nst = stat_count(T);
    const int64_t o_ent = 9 + nst, o_arg = o_ent + T, o_tgt = o_arg + B, o_dump = with_losses ? o_tgt + B : 0;
    const int blk = blockIdx.x;
    if (blk < T) {
        if (!with_losses) return;
        // one wave per row, lanes over the classes
        float acc = 0.f;
        for (int b = wave; b < B; b += MMG_BLOCK / 64) {
            const float* yr = tp.y + ((size_t)blk * B + b) * D;
            float m = -3.0e38f;
            for (int d = lane; d < D; d += 64) m = fmaxf(m, yr[d]);
            m = wave_max(m);
            float se = 0.f;
            for (int d = lane; d < D; d += 64) se += __expf(yr[d] - m);
            se = wave_sum(se);
            const float inv = 1.0f / se;
            float e = 0.f;
            for (int d = lane; d < D; d += 64) { const float p = __expf(yr[d] - m) * inv; e += p * __logf(p + 1e-8f); }
            acc += wave_sum(e);
        }
        if (lane == 0) s_red[wave] = acc;
        __syncthreads();
        if (tid == 0) out[o_ent + blk] = (double)(((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / (float)B);
        return;
    }
    if (blk == T) {
        if (!with_losses) return;
        if (tid < 8) out[tid] = (double)tp.losses[tid];
        if (tid == 8) out[8] = tp.totals[1];
        for (int i = tid; i < nst; i += MMG_BLOCK) out[9 + i] = tp.stats[i];
        for (int b = tid; b < B; b += MMG_BLOCK) {
            const float* dr = tp.dist + (size_t)b * D;
            float best = dr[0]; int arg = 0;
            for (int d = 1; d < D; ++d) { const float v = dr[d]; if (v > best) { best = v; arg = d; } }
            out[o_arg + b] = (double)arg;
            out[o_tgt + b] = target ? (double)target[b] : 0.0;
        }
        return;
    }
    if (k <= 0) return;
    // ---- the sample dump of the first k samples (model.py:1411-1461): every step, stopped samples too (run-all tape)
    double* od = out + o_dump;
    for (int t = wave; t < T; t += MMG_BLOCK / 64) {                          // live samples after every step
        float a = 0.f;
        for (int b = lane; b < B; b += 64) a += (float)tp.mask[(size_t)(t + 1) * B + b];
        a = wave_sum(a);
        if (lane == 0) od[t] = (double)a;
    }
    const int64_t nW = (int64_t)T * k * W;
    const float* src[4] = {tp.pz, tp.pw, tp.z, tp.w};
#pragma unroll
    for (int a = 0; a < 4; ++a)
        for (int64_t i = tid; i < nW; i += MMG_BLOCK) {
            const int t = (int)(i / ((int64_t)k * W)), r = (int)(i % ((int64_t)k * W)), b = r / W, j = r % W;
            od[T + a * nW + i] = (double)src[a][((size_t)t * B + b) * W + j];
        }
    for (int i = tid; i < T * k; i += MMG_BLOCK) {
        const int t = i / k, b = i % k;
        od[T + 4 * nW + i] = (double)tp.ps[(size_t)t * B + b];
        od[T + 4 * nW + T * k + i] = (double)tp.mask[(size_t)(t + 1) * B + b];
    }
}

}  // namespace mmg
