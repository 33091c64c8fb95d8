// mmg.hip -- C-ABI (include/mmg.h) of the MI355X-native exchange path.  Host side: layout queries,
// job-table construction, kernel launches on the caller's stream.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/mmg.h"
#include "layout.h"
#include "device_utils.h"
#include "kernels_fwd.h"
#include "kernels_bwd.h"
#include "kernels_fast.h"
#include "kernels_fast3.h"
#include "kernels_game.h"
#include "kernels_tile.h"
#include "kernels_mc.h"
#include "kernels_mc3.h"
#include "kernels_mc3p.h"
#include "kernels_rc.h"
#ifdef MMG_ROLE_DIAG
#include "diag_kernels.h"
#endif

using namespace mmg;

static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return -1;
}
#define HIP_OK(expr)                                                                      \
    do { hipError_t e_ = (expr);                                                          \
         if (e_ != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

struct KernelTimer { std::string name; hipEvent_t t0, t1; };

struct mmg_handle {
    mmg_config cfg;
    Dims dm;
    ParamLayout pl;
    TapeLayout tl;
    Params P, G;
    Tape tp;
    float *params, *grads, *opt_state;
    void* ws;
    JobTable* d_jt;
    JobTable jt;
    int conv_smem, conv_smem_agent, conv_threads, bwd_smem, prep_smem, prep_cpb;
    bool profiling;
    bool scores_in_parts;      // the last forward left baseline scores as partials (k_baselines2)
    bool sw_merge_bas;         // (= merge_roles) the baselines' forward pass rides in the backward / statistics launch; off: its own launch
    bool defer_bas;            // set by mmg_train_step around its forward call: the baselines may ride in the backward launch
    bool bas_deferred;         // ... and this forward pass left them to it (k_bwd_conv_fast: baseline roles)
    bool bas_pending;          // phased step: the forward pass left the baselines to mmg_loss_stats (k_bas_stats: one launch for both)
    bool sw_merge_prep;        // k_prep's blocks as roles of k_conversation_fast3's launch (MMG_NO_MERGE_PREP=1: a launch of their own)
    bool game_ok;              // fused step of the small Adaptive agents: conversation + statistics + baselines + backward in ONE launch (kernels_game.h); MMG_NO_GAME=1: off
    int game_bas_ub;           // ... 64-unit blocks of a baseline per role: 2 when the block count is even
    int game_nbas;             // ... its baseline roles (a multiple of 2 * ceil(K / 64), sized by the co-residency budget)
    bool game_step;            // set by mmg_train_step around clip_step_impl: k_opt commits the minibatch counter / launch epoch
    int wgrad_stride;          // > 0: k_wgrad's GEMM tiles are walked by this many resident workgroups (more tiles than slots); 0: one workgroup per tile
    bool wgrad_opt_ok;         // the clip + optimizer step can run inside k_wgrad's launch (k_wgrad<true>: every block co-resident, no row splits); MMG_NO_WGRAD_OPT=1: off
    bool wgrad_opt;            // set by mmg_train_step: this step's k_wgrad carries the optimizer (no k_opt launch)
    bool use_fast;             // debugging switches, read once at mmg_create: MMG_NO_FAST=1 forces the generic kernels,
    bool basehx_ready;         // this forward pass formed tape.basehx inside the conversation launch
    bool merge_roles;          // MMG_NO_MERGE=1 keeps k_stats / k_dC / basehx as separate launches / in-kernel work
    // sample-tile MFMA path (kernels_tile.h): every shape the register-resident kernels do not cover
    bool tile_ok;              // its LDS plan fits (MMG_NO_TILE=1: never use it)
    bool rc_bwd;               // ... and the reverse-time loop of its backward as co-resident roles over 16-unit slices (k_rc_bwd); MMG_NO_RC_BWD=1: k_bwd_tile's loop
    bool rc_persist;           // ... as ONE launch of co-resident roles (k_rc_persist) when they all fit on the device; MMG_NO_RC_PERSIST=1: per-step launches
    int rc_budget;
    bool rc_fwd;               // wide receiver (kernels_rc.h): the tile's receiver step as three chip-wide launches over 16-unit slices -- the
                               // one-workgroup-per-tile forward does not fit its LDS plan (R > 128 with a 256-bit message); MMG_NO_RC=1: off
    bool tile_force;           // MMG_TILE=1: use it even where the register-resident kernels apply (cross-checks)
    bool tile_ext;             // the sender MLP of a step runs as its own chip-wide launches (k_send_s1 / k_send_s2)
    int tile_nt, tile_smem;    // threads per tile workgroup, dynamic LDS bytes
    int tile_bwd_smem, send_bwd_smem;
    bool tile_persist;         // the whole conversation as one launch of co-resident roles (k_conv_persist)
    bool tile_split;           // many classes: idle CUs as class helpers of the sample tiles (k_conv_split)
    int split_nh, split_per, split_smem;
    int persist_ns1, persist_ns2, persist_smem;
    std::vector<KernelTimer> timers;
    size_t timers_used;
    uint32_t* h_err;           // pinned host copy of sync[MMG_SYNC_ERR], written by k_opt of every step (posted store to mapped host memory)
    uint32_t* d_err;           // its device-side address
    // debugging switches of the launch paths (environment, read ONCE at mmg_create -- never on the per-minibatch path)
    bool sw_rsample, sw_rmsg, sw_fused_s, rs_capable;
    bool persist_ll;           // k_conv_persist's fused sender roles hand over (value, epoch) pairs in per-step slots (tape.pll_*); MMG_NO_PERSIST_LL=1: counters
    bool mc_ok;                // many-class register-resident conversation (kernels_mc.h); MMG_NO_MC=1: off
    bool mc3p_ok;              // ... for batches of several rounds of workgroups: two sample tiles per workgroup, pipelined (kernels_mc3p.h); MMG_NO_MC3P=1: off
    bool mc3_ok;               // continuous messages: the one-wave-per-SIMD many-class kernel (kernels_mc3.h); binary messages: k_conversation_mc
    bool any_split;            // some k_wgrad job splits its rows over workgroups (the last slice to arrive adds the partial tiles)
    bool wgrad_small_split;    // jobs with few output tiles split their (step, sample) rows further (layout.h: wgrad_job_nsplit)
    int mc_per, mc_xcd;        // classes per member of a tile; mc_xcd: a tile's 16 workgroups on one XCD
    bool xcd_rule_ok;          // probed at mmg_create (k_xcc_probe): workgroup i of a launch runs on XCD i % 8 -- a hand-off between workgroups of one XCD may stay in its L2
    // workgroups of 512 threads that are guaranteed to be resident together on this device (occupancy query at mmg_create,
    // minus a margin): the role launches (k_conv_persist / k_conv_split / k_conversation_mc) spin on each other, so a launch
    // may never hold more roles than this
    int resident_budget, split_budget, n_cu;
    // fail-soft (round 6): no_roles = only launches without in-launch waits are selected (select_paths).  Set at mmg_create by
    // MMG_NO_ROLES=1 / a CU mask in the environment, or by recover() after a timed-out dependency (degraded)
    bool no_roles, degraded;
    int recoveries;            // recover() calls so far (bounded: a wait that keeps timing out without roles is a real fault)
    uint32_t last_code;        // the dependency word of the last recovery
    // data-parallel step inside the library (mmg_dp_set_allreduce): RCCL's ncclAllReduce by address + the caller's communicator
    void* ar_fn; void* ar_comm;
    mmg_handle() : params(nullptr), grads(nullptr), opt_state(nullptr), ws(nullptr), d_jt(nullptr), h_err(nullptr), d_err(nullptr),
                   no_roles(false), degraded(false), recoveries(0), last_code(0u), ar_fn(nullptr), ar_comm(nullptr) {}
    ~mmg_handle() {
        for (auto& t : timers) { hipEventDestroy(t.t0); hipEventDestroy(t.t1); }
        if (h_err) hipHostFree(h_err);
    }
};

// ---------------------------------------------------------------------------------------------
// Fail-soft (round 6).  An in-launch dependency wait that hits its spin bound (fewer compute units than the launch's roles
// need: a shared or CU-masked GPU) sets sync[MMG_SYNC_ERR] on the device; k_opt / the norm role of THAT minibatch leave
// parameters and optimizer state untouched and post the word to a pinned host word (no synchronisation).  The reference has
// no such failure mode (model.py:1218-1330 simply keeps training), so the library recovers instead of failing the run:
// the next call that STARTS a minibatch (mmg_train_step[s], mmg_exchange_forward(train), mmg_dp_train_step) drains the
// stream once, clears both words, re-selects the kernels WITHOUT in-launch waits (select_paths with no_roles: per-step /
// per-phase launches, what MMG_NO_ROLES=1 selects up front), uploads the job table of that path and continues.  The call
// returns 1 (ok, with a warning in mmg_last_error()).  Entry points in the MIDDLE of a phased minibatch do nothing: that
// minibatch's update is skipped on the device anyway and the next minibatch start recovers.  Data parallel: the flag travels
// in the all-reduced gradient tail, every rank skips the same update and every rank posts a word (its own code or 1001).
// ---------------------------------------------------------------------------------------------
static int select_paths(mmg_handle* h);
#define MMG_MAX_RECOVERIES 8
static int clear_error_words(mmg_handle* h, hipStream_t st) {
    HIP_OK(hipStreamSynchronize(st));                    // rare path: nothing of this handle is in flight afterwards
    const uint32_t zero = 0u;
    HIP_OK(hipMemcpy(h->tp.sync + MMG_SYNC_ERR, &zero, sizeof(zero), hipMemcpyHostToDevice));
    if (h->h_err) *(volatile uint32_t*)h->h_err = 0u;
    return 0;
}
// step_start: this call begins a minibatch.  0 = nothing to report, 1 = recovered (warning text in g_err), < 0 = error
static int error_gate(mmg_handle* h, hipStream_t st, bool step_start) {
    if (!h->h_err) return 0;
    const uint32_t code = *(volatile uint32_t*)h->h_err;
    if (code == 0u || !step_start) return 0;
    if (h->recoveries >= MMG_MAX_RECOVERIES)
        return fail("in-launch dependency %u timed out on the device again after %d recoveries (the launches in use hold no in-launch "
                    "waits: device fault?)", code - 1u, h->recoveries);
    if (clear_error_words(h, st)) return -1;
    const bool was_roles = !h->no_roles;
    if (was_roles) {
        h->no_roles = true;
        if (select_paths(h)) return -1;
        HIP_OK(hipMemcpy(h->d_jt, &h->jt, sizeof(JobTable), hipMemcpyHostToDevice));
        h->degraded = true;
    }
    ++h->recoveries; h->last_code = code;
    if (code == MMG_SYNC_ERR_REMOTE)
        fail("warning: another rank of the data-parallel job reported a timed-out in-launch dependency; every rank skipped that "
             "optimizer update%s", was_roles ? " and continues on the launches without in-launch waits" : "");
    else
        fail("warning: in-launch dependency %u timed out on the device (fewer compute units available than the launch's workgroup "
             "roles need?); the optimizer update of that minibatch was skipped%s", code - 1u,
             was_roles ? " and training continues on the launches without in-launch waits (MMG_NO_ROLES=1 selects them up front)" : "");
    return 1;
}

extern "C" int mmg_clear_error(mmg_handle* h, void* stream) {
    if (!h) return fail("NULL handle");
    return clear_error_words(h, (hipStream_t)stream);
}
extern "C" int mmg_degraded(const mmg_handle* h) { return (h && h->no_roles) ? (h->degraded ? 2 : 1) : 0; }

extern "C" const char* mmg_last_error(void) { return g_err; }
extern "C" int mmg_version(void) { return MMG_VERSION; }

static int validate(const mmg_config* c) {
    if (!c) return fail("config is NULL");
    if (c->batch <= 0 || c->n_classes <= 0 || c->feat_dim <= 0 || c->h_dim <= 0 || c->w_dim <= 0 ||
        c->rec_hidden <= 0 || c->wv_dim <= 0 || c->bas_hidden <= 0 || c->max_exchange <= 0)
        return fail("all dimensions must be positive");
    if (c->max_exchange > 64) return fail("max_exchange must be <= 64");
    if (c->w_dim > MMG_BLOCK || c->rec_hidden > MMG_BLOCK)
        return fail("w_dim and rec_hidden must be <= %d (got %d, %d)", MMG_BLOCK, c->w_dim, c->rec_hidden);
    if (c->wv_dim > MMG_BLOCK) return fail("wv_dim must be <= %d", MMG_BLOCK);
    if (c->optim_type < 0 || c->optim_type > 2) return fail("unknown optim_type %d", c->optim_type);
    if (c->global_batch > 0 && c->global_batch < c->batch) return fail("global_batch < batch");
    return 0;
}

extern "C" int64_t mmg_param_count(const mmg_config* cfg) {
    if (validate(cfg)) return -1;
    return param_layout(*cfg).total;
}

extern "C" int mmg_param_table(const mmg_config* cfg, mmg_param_entry* out, int max_entries) {
    if (validate(cfg)) return -1;
    ParamLayout L = param_layout(*cfg);
    if (out) {
        if (max_entries < P_COUNT) return fail("need room for %d entries", (int)P_COUNT);
        for (int i = 0; i < P_COUNT; ++i) {
            memset(&out[i], 0, sizeof(out[i]));
            strncpy(out[i].name, L.name[i], sizeof(out[i].name) - 1);
            out[i].agent = L.agent[i]; out[i].rows = L.rows[i]; out[i].cols = L.cols[i]; out[i].offset = L.off[i];
        }
    }
    return P_COUNT;
}

extern "C" int64_t mmg_grad_floats(const mmg_config* cfg) {
    if (validate(cfg)) return -1;
    return param_layout(*cfg).total + MMG_GRAD_TAIL;
}

extern "C" int64_t mmg_workspace_bytes(const mmg_config* cfg) {
    if (validate(cfg)) return -1;
    return tape_layout(*cfg).total;
}

extern "C" int mmg_tape_table(const mmg_config* cfg, mmg_tape_entry* out, int max_entries) {
    if (validate(cfg)) return -1;
    TapeLayout L = tape_layout(*cfg);
    if (out) {
        if (max_entries < L.n) return fail("need room for %d entries", L.n);
        memcpy(out, L.e, sizeof(mmg_tape_entry) * L.n);
    }
    return L.n;
}

// ---------------------------------------------------------------------------------------------
// job table: every parameter tensor's gradient is produced by exactly one GEMM / column-sum job
// (two for the matrices whose input is a concatenation: y1, both baselines' linear1).
// ---------------------------------------------------------------------------------------------
static bool fast_shape(const mmg_handle* h);
static bool tile_path(const mmg_handle* h);
static bool mc_path(const mmg_handle* h);

static int build_jobs(mmg_handle* h) {
    JobTable& jt = h->jt;
    memset(&jt, 0, sizeof(jt));
    const Dims& d = h->dm;
    const Tape& tp = h->tp;
    const Params& G = h->G;
    const int B = d.B, T = d.T, H = d.H, W = d.W, R = d.R, V = d.V, K = d.K, D = d.D, F = d.F;
    const int TB = T * B;
    int tiles = 0, ng = 0;
    auto gemm = [&](const float* A, int lda, const float* Bm, int ldb, int bmod, int bsrc, float* C, int ldc,
                    int rows, int N, int Kk) {
        GemmJob& g = jt.g[ng++];
        g.A = A; g.Bm = Bm; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.rows = rows; g.N = N; g.K = Kk;
        g.bmod = bmod; g.bsrc = bsrc; g.tile_begin = tiles; g.tiles_k = (Kk + 31) / 32;     // 16 x 32 outputs per block
        g.vhid = nullptr; g.vw2 = nullptr; g.compact = (rows == TB) ? 1 : 0;
        g.nsplit = (rows != TB) ? 1 : h->wgrad_small_split ? wgrad_job_nsplit(TB, h->pl.total, ((N + 15) / 16) * ((Kk + 31) / 32))
                                                           : wgrad_nsplit(TB, h->pl.total);
        tiles += ((N + 15) / 16) * g.tiles_k * g.nsplit;
    };
    // dW = (dbeta * w2 * relu'(hid))^T . input  with the first factor formed on the fly
    auto gemm_virt = [&](const float* dbeta, const float* hid, const float* w2, const float* Bm, int ldb, int bmod,
                         float* C, int ldc, int rows, int N, int Kk) {
        gemm(dbeta, N, Bm, ldb, bmod, SRC_STATIC, C, ldc, rows, N, Kk);
        jt.g[ng - 1].vhid = hid; jt.g[ng - 1].vw2 = w2;
    };
    int cblocks = 0, nc = 0;
    // bias gradient = column sums of a (step, sample)-row tape.  With thousands of rows the 16-column blocks of a column job
    // are a handful of latency-bound workgroups: run it through the row-split GEMM pipeline instead, as delta^T . ones (K = 1)
    const bool bias_as_gemm = wgrad_nsplit(TB, h->pl.total) > 1 || (h->wgrad_small_split && TB > 2048);
    auto col = [&](const float* src, int ld, int rows, int cols, float* dst, const float* scale) {
        ColJob& c = jt.c[nc++];
        c.src = src; c.dst = dst; c.scale = scale; c.ld = ld; c.rows = rows; c.cols = cols; c.blk_begin = cblocks;
        c.vbeta = nullptr; c.vw2 = nullptr; c.wrow = nullptr; c.compact = (rows == TB) ? 1 : 0; c.special = 0;
        cblocks += (cols + 15) / 16;
    };
    auto bias = [&](const float* src, int ld, int cols, float* dst) {        // plain column sums over the (step, sample) rows
        if (bias_as_gemm) gemm(src, ld, tp.ones, 0, 0, SRC_STATIC, dst, 1, TB, cols, 1);
        else col(src, ld, TB, cols, dst, nullptr);
    };
    const Params& P = h->P;
    const bool bin = d.use_binary;
    // ---- receiver ----
    gemm(tp.dgi, 3 * R, tp.z, W, 0, SRC_STATIC, G.p[R_WIH], W, TB, 3 * R, W);          // rnn.weight_ih
    gemm(tp.dgh, 3 * R, tp.h, R, 0, SRC_STATIC, G.p[R_WHH], R, TB, 3 * R, R);          // rnn.weight_hh (h before the step)
    bias(tp.dgi, 3 * R, 3 * R, G.p[R_BIH]);
    bias(tp.dgh, 3 * R, 3 * R, G.p[R_BHH]);
    gemm(tp.dA, R, tp.hstar, R, 0, SRC_STATIC, G.p[R_Y1_W], R + V, B, R, R);           // y1.weight[:, :R]
    gemm(tp.dC, R, tp.descc, V, 0, SRC_STATIC, G.p[R_Y1_W] + R, R + V, D, R, V);      // y1.weight[:, R:]
    col(tp.dC, R, D, R, G.p[R_Y1_B], nullptr);
    col(tp.Py2, R, D, R, G.p[R_Y2_W], nullptr);
    col(tp.dysum, 1, B, 1, G.p[R_Y2_B], nullptr);
    if (bin) {
        gemm(tp.dgpre, R, tp.h + (size_t)B * R, R, 0, SRC_STATIC, G.p[R_WH_W], R, TB, R, R);   // w_h (h after the step)
        bias(tp.dgpre, R, R, G.p[R_WH_B]);
        gemm(tp.dgpre, R, tp.dbar, V, 0, SRC_STATIC, G.p[R_WD_W], V, TB, R, V);        // w_d
        gemm(tp.dlw, W, tp.g, R, 0, SRC_STATIC, G.p[R_W_W], R, TB, W, R);              // w
        bias(tp.dlw, W, W, G.p[R_W_B]);
        col(tp.h + (size_t)B * R, R, TB, R, G.p[R_S_W], nullptr);                      // s.weight = dls^T . h_after
        jt.c[nc - 1].wrow = tp.dls;
        col(tp.dls, 1, TB, 1, G.p[R_S_B], nullptr);
        // ---- sender ----
        gemm(tp.dhx, H, nullptr, F, 0, SRC_X, G.p[S_IMG_W], F, B, H, F);               // image_layer (sum over steps first)
        col(tp.dhx, H, B, H, G.p[S_IMG_B], nullptr);
        gemm(tp.dpre, H, tp.c, W, 0, SRC_STATIC, G.p[S_CODE_W], W, TB, H, W);          // code_layer
        bias(tp.dpre, H, H, G.p[S_CODE_B]);
        if (tile_path(h)) {
            // code_bias: dsig[j] * sum_h code_layer.weight[h, j] * u0[h], u0 = sum_b dpre[t = 0, b, :] (k_dhx): a row-weighted
            // column sum over the weight matrix itself
            col(P.p[S_CODE_W], W, H, W, G.p[S_CODE_BIAS], tp.dsig);
            jt.c[nc - 1].wrow = tp.u0; jt.c[nc - 1].compact = 0;
        } else if (fast_shape(h)) {
            // code_bias: dsig[j] * sum_h code_layer.weight[h, j] * (sum_b dpre[t = 0, b, h]) -- one workgroup of k_wgrad;
            // the register-resident backward kernel then needs no per-sample W_c^T dpre_0 product at its tail
            col(tp.dpre, H, B, 1, G.p[S_CODE_BIAS], tp.dsig);
            jt.c[nc - 1].special = 1; jt.c[nc - 1].wrow = P.p[S_CODE_W]; jt.c[nc - 1].compact = 0; jt.c[nc - 1].cols = W;
        } else {
            col(tp.dc0, W, B, W, G.p[S_CODE_BIAS], tp.dsig);                           // code_bias
        }
        gemm(tp.dlz, W, tp.a, H, 0, SRC_STATIC, G.p[S_BIN_W], H, TB, W, H);            // binary_layer
        bias(tp.dlz, W, W, G.p[S_BIN_B]);
        // ---- baseline_rec: input [z || h_after] ----
        gemm_virt(tp.dbr, tp.hid_r, P.p[BR_L2_W], tp.z, W, 0, G.p[BR_L1_W], W + R, TB, K, W);
        gemm_virt(tp.dbr, tp.hid_r, P.p[BR_L2_W], tp.h + (size_t)B * R, R, 0, G.p[BR_L1_W] + W, W + R, TB, K, R);
        col(tp.hid_r, K, TB, K, G.p[BR_L1_B], nullptr);
        jt.c[nc - 1].vbeta = tp.dbr; jt.c[nc - 1].vw2 = P.p[BR_L2_W];
        col(tp.hid_r, K, TB, K, G.p[BR_L2_W], nullptr);
        jt.c[nc - 1].wrow = tp.dbr;
        col(tp.dbr, 1, TB, 1, G.p[BR_L2_B], nullptr);
        // ---- baseline_sen: input [h_x || z_r] ----
        gemm_virt(tp.dbs, tp.hid_s, P.p[BS_L2_W], tp.hx, H, B, G.p[BS_L1_W], H + W, TB, K, H);
        gemm_virt(tp.dbs, tp.hid_s, P.p[BS_L2_W], tp.zr, W, 0, G.p[BS_L1_W] + H, H + W, TB, K, W);
        col(tp.hid_s, K, TB, K, G.p[BS_L1_B], nullptr);
        jt.c[nc - 1].vbeta = tp.dbs; jt.c[nc - 1].vw2 = P.p[BS_L2_W];
        col(tp.hid_s, K, TB, K, G.p[BS_L2_W], nullptr);
        jt.c[nc - 1].wrow = tp.dbs;
        col(tp.dbs, 1, TB, 1, G.p[BS_L2_B], nullptr);
    }
    if (ng > MMG_MAX_GEMM || nc > MMG_MAX_COL) return fail("job table overflow");
    h->any_split = false;
    for (int g = 0; g < ng; ++g) h->any_split = h->any_split || jt.g[g].nsplit > 1;
    jt.n_gemm = ng; jt.n_col = nc; jt.gemm_tiles = tiles; jt.gemm_blocks = tiles; jt.col_blocks = cblocks;
    jt.n_wblocks = tiles + cblocks;
    jt.special_block = -1; jt.special_job = -1;
    for (int c = 0; c < nc; ++c) if (jt.c[c].special) { jt.special_job = c; jt.special_block = tiles + jt.c[c].blk_begin; }
    for (int k = 0; k < 64; ++k) {
        jt.g_begin[k] = k < ng ? jt.g[k].tile_begin : 0x7fffffff;
        jt.c_begin[k] = k < nc ? jt.c[k].blk_begin : 0x7fffffff;
    }
    if (jt.n_wblocks > MMG_MAX_WBLOCKS && h->wgrad_small_split) { h->wgrad_small_split = false; return build_jobs(h); }   // (k_wgrad addresses 16384 workgroups)
    if (jt.n_wblocks > MMG_MAX_WBLOCKS) return fail("too many weight-gradient tiles (%d)", jt.n_wblocks);
    {
        auto agent_of = [&](const float* dst) {
            const int64_t off = dst - h->grads;
            int a = 0;
            for (int k = 1; k < 4; ++k) if (off >= h->pl.agent_begin[k]) a = k;
            return (signed char)a;
        };
        for (int g = 0; g < ng; ++g) {
            const int end = (g + 1 < ng) ? jt.g[g + 1].tile_begin : tiles;
            for (int t = jt.g[g].tile_begin; t < end; ++t) jt.wblock_agent[t] = agent_of(jt.g[g].C);
        }
        for (int c = 0; c < nc; ++c) {
            const int end = (c + 1 < nc) ? jt.c[c + 1].blk_begin : cblocks;
            for (int bk = jt.c[c].blk_begin; bk < end; ++bk) jt.wblock_agent[tiles + bk] = agent_of(jt.c[c].dst);
        }
    }
    // ---- gradient-norm plan: MMG_GN_BLOCKS chunks, each inside one agent ----
    const ParamLayout& pl = h->pl;
    int nb[4];
    int left = MMG_GN_BLOCKS - 4;
    for (int a = 0; a < 4; ++a) {
        const double frac = (double)(pl.agent_begin[a + 1] - pl.agent_begin[a]) / (double)pl.total;
        nb[a] = 1 + (int)(frac * left);
    }
    int blk = 0;
    for (int a = 0; a < 4; ++a) {
        const int64_t b0 = pl.agent_begin[a], b1 = pl.agent_begin[a + 1];
        const int64_t quads = (b1 - b0) / 4;
        for (int k = 0; k < nb[a]; ++k) {
            jt.np.begin[blk] = b0 + 4 * (quads * k / nb[a]);
            jt.np.end[blk] = b0 + 4 * (quads * (k + 1) / nb[a]);
            jt.np.agent[blk] = a;
            ++blk;
        }
    }
    for (; blk < MMG_GN_BLOCKS; ++blk) { jt.np.begin[blk] = jt.np.end[blk] = 0; jt.np.agent[blk] = -1; }
    if (sizeof(JobTable) > 98304) return fail("job table does not fit its tape slot");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Path selection: which kernels serve this handle's shape on this device.  Runs at mmg_create and again when the library
// falls back to launches WITHOUT in-launch waits (h->no_roles: after a timed-out dependency, for a CU budget / CU mask that
// cannot hold the role launches, or MMG_NO_ROLES=1).  Environment switches are read here only -- never on the per-minibatch path.
// ---------------------------------------------------------------------------------------------
static int select_paths(mmg_handle* h) {
    const mmg_config& cfg = h->cfg;
    const bool no_roles = h->no_roles;
    h->use_fast = !getenv("MMG_NO_FAST"); h->merge_roles = !getenv("MMG_NO_MERGE") && !no_roles;
    h->sw_merge_prep = !getenv("MMG_NO_MERGE_PREP") && !no_roles;
    h->sw_merge_bas = h->merge_roles; h->defer_bas = false; h->bas_deferred = false;
    h->persist_ll = !getenv("MMG_NO_PERSIST_LL") && persist_ll_shape(cfg.batch, cfg.h_dim, cfg.w_dim, cfg.rec_hidden, cfg.wv_dim, cfg.n_classes, cfg.max_exchange);
    h->sw_rsample = !getenv("MMG_NO_RSAMPLE"); h->sw_rmsg = !getenv("MMG_NO_RMSG"); h->sw_fused_s = !getenv("MMG_NO_FUSED_S");
    h->mc_ok = h->use_fast && mc_shape(h->dm.H, h->dm.W, h->dm.R, h->dm.V, h->dm.D, h->dm.T) && !getenv("MMG_NO_MC") && !no_roles;
    h->mc_per = (((h->dm.D + 15) / 16) + 3) & ~3;
    h->mc_xcd = 1;                                  // a tile's 16 workgroups on one XCD (measured at config 5, 256 samples: 192 us per minibatch against 201); cleared below on a device without room for it
    h->mc3_ok = false; h->mc3p_ok = false;
    h->wgrad_small_split = true;
    int n_cu = 0;
    {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) return fail("cannot query the device (multiProcessorCount)");
        // caller-supplied budget (mmg_config.cu_budget): a process that shares the GPU, or runs under a CU mask, states how many
        // compute units it can count on -- every co-residency budget below is sized from it
        if (cfg.cu_budget > 0 && cfg.cu_budget < n_cu) n_cu = cfg.cu_budget;
    }
    // co-resident workgroups a role launch may hold: occupancy of the kernel at its LDS size x compute units, minus a margin
    // of 1/16 of the chip (256 CUs -> 240, the value the role launches were tuned with).  A partitioned device (CPX), a
    // smaller SKU or a masked process simply gets a smaller budget and, where the roles do not fit, the per-step / generic launches.
    auto budget_of = [&](const void* fn, int threads, int smem) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, (size_t)smem) != hipSuccess || nb < 1) return 0;
        const int total = nb * n_cu;
        return total - (total + 15) / 16;
    };
    h->resident_budget = 0; h->split_budget = 0; h->n_cu = n_cu;
    // few samples and large sender matrices or class tables: 512-thread variant of the generic conversation kernel
    h->conv_threads = (h->dm.B <= 256 && ((int64_t)h->dm.H * h->dm.W >= 65536 || (int64_t)h->dm.D * (h->dm.R + h->dm.V) >= 65536)) ? 512 : 256;
    h->conv_smem = conv_smem_floats(h->dm, h->conv_threads) * 4;
    h->conv_smem_agent = conv_smem_floats(h->dm, MMG_BLOCK) * 4;
    h->bwd_smem = bwd_smem_floats(h->dm) * 4;
    h->prep_smem = ((h->dm.V > h->dm.W ? h->dm.V : h->dm.W) + 16) * 4;
    // hundreds of classes: 8 per class block of k_prep (weight rows in registers across them); few classes: one per block (latency)
    h->prep_cpb = (h->dm.D >= 256 && h->dm.R <= 64 && h->dm.V <= 128 && !(h->dm.V & 3) && 2 * h->dm.R <= MMG_BLOCK) ? 2 : 1;
    if (h->prep_cpb > 1 && (int)(h->prep_cpb * (h->dm.V + h->dm.R) * 4) > h->prep_smem) h->prep_smem = h->prep_cpb * (h->dm.V + h->dm.R) * 4;
    if (h->conv_smem > 160 * 1024 || h->bwd_smem > 160 * 1024) return fail("dimensions need more than 160 KB of LDS per sample");
    hipError_t e = hipSuccess;
    {
        const Dims& d = h->dm;
        const int tiles = (d.B + MMG_TM - 1) / MMG_TM;
        // few tiles and a large sender MLP: one step's sender products as chip-wide launches of their own
        h->tile_ext = tiles < 64 && (int64_t)d.H * d.W >= 65536;
        // one tile per CU up to 256 tiles: 16 waves hide the LDS / L2 latency of the tile's phases; beyond that several
        // smaller workgroups share a CU.  Fewer waves also mean smaller split-K staging areas.
        const int nts[2] = {512, 256};                            // (a 1024-thread variant spilled at 128 registers per lane: deleted)
        for (int k = (tiles <= 512 ? 0 : 1); k < 2; ++k) {
            h->tile_nt = nts[k];
            h->tile_smem = tile_lds(d, h->tile_nt / 64, !h->tile_ext).total * 4;
            if (h->tile_smem <= 160 * 1024) break;
        }
        if (!h->tile_ext && d.H > h->tile_nt) {                      // the in-kernel sender keeps the tile's h_x in 16 registers per thread
            h->tile_ext = true;
            h->tile_smem = tile_lds(d, h->tile_nt / 64, false).total * 4;
        }
        // 16-byte aligned weight rows (float4 fragments): every BASELINE shape; odd dimensions take the per-sample kernels
        const bool aligned = !(d.H & 3) && !(d.W & 3) && !(d.R & 3) && !(d.V & 3);
        h->rc_fwd = aligned && h->tile_ext && h->tile_smem > 160 * 1024 && rc_shape(d.B, d.H, d.W, d.R, d.V, d.D) && !getenv("MMG_NO_RC");
        h->tile_ok = aligned && (h->tile_smem <= 160 * 1024 || h->rc_fwd) && !getenv("MMG_NO_TILE");
        h->tile_force = getenv("MMG_TILE") != nullptr;
        // many classes, small agents, fewer than 64 tiles: a workgroup per SAMPLE fills the chip (256 samples = 256 CUs) and
        // beats 16 tiles + class helpers (measured at D = 1000, B = 256: 557 us against 1 010 us per minibatch; B = 2048:
        // 2 091 against 1 189) -- the tile kernels take over from 1024 samples (MMG_TILE=1: always)
        if (h->tile_ok && !h->tile_force && !h->tile_ext && d.D * MMG_TM > 8 * 512 && d.B < 1024) h->tile_ok = false;
        // many classes and fewer sample tiles than CUs: class helpers (k_conv_split)
        h->split_nh = split_helpers(d.B);
        h->split_per = (((d.D + h->split_nh) / (h->split_nh + 1)) + 3) & ~3;
        h->tile_split = h->tile_ok && !h->tile_ext && d.D * MMG_TM > 8 * 512 && h->split_nh >= 1 && tiles * (1 + h->split_nh) <= 224 &&
                        !getenv("MMG_NO_SPLIT") && !no_roles;
        if (h->tile_split) {
            const int a = tile_lds(d, 512 / 64, true, h->split_per).total * 4, b = helper_lds(d, 512 / 64, h->split_per).total * 4;
            h->split_smem = a > b ? a : b;
            if (h->split_smem > 160 * 1024) h->tile_split = false;
            else if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_conv_split<512>, hipFuncAttributeMaxDynamicSharedMemorySize, h->split_smem);
            if (h->tile_split && e == hipSuccess) {
                h->split_budget = budget_of((const void*)k_conv_split<512>, 512, h->split_smem);
                if (tiles * (1 + h->split_nh) > h->split_budget) h->tile_split = false;     // not all co-resident here: k_conv_tile instead
            }
        }
        // per-step sender products as ROLES of one persistent launch when all of them fit on the chip together
        h->persist_ns1 = d.H / 64; h->persist_ns2 = d.W / 32;
        // (receiver shape of the register-resident kernels: per-sample receiver roles, and batches too large for one launch of
        //  co-resident roles run as consecutive launches over sample ranges)
        const bool rs_capable = d.R == 64 && d.V == 100 && d.D <= 32 && d.T <= 16 && h->sw_rsample;
        h->tile_persist = h->tile_ok && h->tile_ext && !(d.H % 64) && !(d.W % 32) && tiles <= 64 &&
                          MMG_TM * d.W <= 8 * 512 && !getenv("MMG_NO_PERSIST") && !no_roles;
        h->rs_capable = rs_capable;
        if (h->tile_persist) {
            const int a = tile_lds(d, 512 / 64, false).total * 4, b = srole_lds(d, 512 / 64).total * 4;
            h->persist_smem = a > b ? a : b;
            if (h->persist_smem > 160 * 1024) h->tile_persist = false;
            else if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_conv_persist<512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->persist_smem);
            if (h->tile_persist && e == hipSuccess) e = hipFuncSetAttribute((const void*)k_conv_persist<512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, h->persist_smem);
            if (h->tile_persist && e == hipSuccess) e = hipFuncSetAttribute((const void*)k_conv_persist<512, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->persist_smem);
            if (h->tile_persist && e == hipSuccess) {
                h->resident_budget = rs_capable ? budget_of((const void*)k_conv_persist<512, true>, 512, h->persist_smem)
                                               : budget_of((const void*)k_conv_persist<512, false>, 512, h->persist_smem);
                // tile roles: every tile's roles in one launch; per-sample receiver roles: at least ONE whole tile per launch
                const bool fits = rs_capable ? (MMG_TM + d.H / 64 + d.W / 16 <= h->resident_budget || MMG_TM + h->persist_ns1 + h->persist_ns2 <= h->resident_budget)
                                             : tiles * (1 + h->persist_ns1 + h->persist_ns2) <= h->resident_budget;
                if (!fits) h->tile_persist = false;                                       // per-step launches instead (no co-residency needed)
            }
        }
        h->tile_bwd_smem = bwd_tile_lds(d, 512 / 64).total * 4;
        h->send_bwd_smem = (MMG_TM * ld16(d.W) + 7 * 64 + 16 + tile_raw_floats_nn(64, MMG_BLOCK / 64)) * 4;
        if (h->tile_bwd_smem > 160 * 1024 || d.W > 256 || d.R > 256) h->tile_ok = false;     // (k_bwd_tile keeps a step's GRU tape in 4 registers per thread per 32 hidden units)
        if (h->tile_ok && e == hipSuccess && h->tile_bwd_smem > 48 * 1024)
        {
            e = hipFuncSetAttribute((const void*)k_bwd_tile<512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, h->tile_bwd_smem);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bwd_tile<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, h->tile_bwd_smem);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bwd_tile<512, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, h->tile_bwd_smem);
        }
        if (h->tile_ok && e == hipSuccess && bwd_pre_lds_floats(d) * 4 > 48 * 1024)
        {
            e = hipFuncSetAttribute((const void*)k_bwd_pre<8>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_pre_lds_floats(d) * 4);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bwd_pre<16>, hipFuncAttributeMaxDynamicSharedMemorySize, bwd_pre_lds_floats(d) * 4);
        }
        if (h->tile_ok && e == hipSuccess && h->send_bwd_smem > 48 * 1024)
            e = hipFuncSetAttribute((const void*)k_send_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, h->send_bwd_smem);
        if (h->tile_ok && e == hipSuccess) {
            const int smem = bwd_pre_lds_floats(d) * 4 > h->send_bwd_smem ? bwd_pre_lds_floats(d) * 4 : h->send_bwd_smem;
            if (smem > 48 * 1024) e = hipFuncSetAttribute((const void*)k_bwd_pre_send<8>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (smem > 48 * 1024 && e == hipSuccess) e = hipFuncSetAttribute((const void*)k_bwd_pre_send<16>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        }
        if (!h->tile_ok) h->rc_fwd = false;
        h->rc_persist = false; h->rc_budget = 0; h->rc_bwd = false;
        if (h->rc_fwd && e == hipSuccess)
            h->rc_bwd = tiles <= 64 && tiles * (d.R / 16) <= budget_of((const void*)k_rc_bwd, 256, 0) && !getenv("MMG_NO_RC_BWD") && !no_roles;
        if (h->rc_fwd && e == hipSuccess) {
            const int nj = d.R / 16, njw = d.W / 16, per_tile = (nj > njw ? nj : njw) + njw + (d.H + 63) / 64 + 1;
            h->rc_budget = budget_of((const void*)k_rc_persist, 256, 0);
            // (up to two consecutive launches over tile ranges; beyond that the per-step launches over the whole batch win:
            //  profiles/r04_rc_batch_sweep.log)
            const int ct = h->rc_budget / per_tile;
            h->rc_persist = !(d.H & 15) && d.H <= 1024 && tiles <= 15 && ct >= 1 && (tiles + ct - 1) / ct <= 2 && !getenv("MMG_NO_RC_PERSIST") && !no_roles;
        }
        if (h->tile_ok && h->tile_smem > 48 * 1024 && !h->rc_fwd) {
            e = hipFuncSetAttribute((const void*)k_conv_tile<256>, hipFuncAttributeMaxDynamicSharedMemorySize, h->tile_smem);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_conv_tile<512>, hipFuncAttributeMaxDynamicSharedMemorySize, h->tile_smem);
        }
    }
    if (h->mc_ok) {
        // k_conversation_mc's 16 workgroups per tile spin on each other: with the per-XCD mapping a tile's members are 16 of 128
        // consecutive ids, so in-order dispatch needs 128 of them resident (16 with consecutive ids); below that the tile /
        // generic kernels run instead -- never a timed-out wait on a partitioned or masked device
        const int mc_budget = budget_of((const void*)(k_conversation_mc<256, 32, 64, 100, 64>), 512, 0);
        if (mc_budget < 128) h->mc_xcd = 0;
        if (mc_budget < 16) h->mc_ok = false;
        h->mc3_ok = h->mc_ok && !h->dm.use_binary;
        if (h->mc3_ok) {
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)(k_conversation_mc3<256, 32, 64, 100, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, mc3_lds_bytes());
            const int b3 = budget_of((const void*)(k_conversation_mc3<256, 32, 64, 100, 64>), 256, mc3_lds_bytes());
            if (b3 < (h->mc_xcd ? 128 : 16)) h->mc3_ok = false;
            // two tiles per workgroup (kernels_mc3p.h): from 512 samples on, where the one-tile kernel needs several rounds of workgroups
            if (h->mc3_ok && h->mc_xcd && mc3p_shape(h->dm.B, h->dm.T, h->dm.D) && !getenv("MMG_NO_MC3P")) {
                if (e == hipSuccess) e = hipFuncSetAttribute((const void*)(k_conversation_mc3p<256, 32, 64, 100, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, mc3p_lds_bytes(h->dm.T));
                h->mc3p_ok = e == hipSuccess && budget_of((const void*)(k_conversation_mc3p<256, 32, 64, 100, 64>), 256, mc3p_lds_bytes(h->dm.T)) >= 128;
            }
        }
    }
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)(k_conversation_fast3<256, 32, 64, 100, false>), hipFuncAttributeMaxDynamicSharedMemorySize, fast3_lds_bytes());
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)(k_conversation_fast3<256, 32, 64, 100, true>), hipFuncAttributeMaxDynamicSharedMemorySize, fast3_lds_bytes());
    h->game_ok = false; h->game_nbas = 0; h->game_step = false; h->game_bas_ub = 1;
    {
        const Dims& d = h->dm;
        const bool shape = h->use_fast && h->merge_roles && h->sw_merge_prep && h->sw_merge_bas && d.H == 256 && d.W == 32 && d.R == 64 && d.V == 100 &&
                           d.D <= 32 && d.T <= 15 && d.B <= 64 && d.use_binary && !d.fixed && (d.K + 63) / 64 <= 8 && d.K <= 512 &&
                           !(h->tile_ok && h->tile_force) && h->prep_cpb == 1 && h->prep_smem <= game_lds_bytes() && !getenv("MMG_NO_GAME") && !no_roles;
        if (shape && e == hipSuccess) {
            const void* fn = d.D == 30 ? (const void*)k_game_fast<30> : (const void*)k_game_fast<32>;
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, game_lds_bytes());
            if (e == hipSuccess) {
                // every spinning role must be resident together with the sample roles (the sample roles wait for the statistics roles,
                // those for the baseline roles): B + n_stats + n_bas + D workgroups inside the co-residency budget of this device
                const int budget = budget_of(fn, 256, game_lds_bytes());
                const int npb_ = (d.K + 63) / 64;
                h->game_bas_ub = !(npb_ & 1) ? 2 : 1;
                const int n_stats = (5 * d.T + 2 + 3) / 4, per = 2 * npb_ / h->game_bas_ub;
                int nb = ((budget - d.B - n_stats - d.D) / per) * per;
                const int want = ((d.T * d.B + 15) / 16) * per;
                if (nb > want) nb = want;
                if (nb >= per && prep_blocks(d, h->prep_cpb, true) + d.B <= n_cu) { h->game_ok = true; h->game_nbas = nb; }
            }
        }
    }
#ifdef MMG_DEBUG_CREATE                                  // (compile with -DMMG_DEBUG_CREATE: what select_paths decided)
    if (true)
        fprintf(stderr, "mmg_create: game_ok %d game_nbas %d\n", (int)h->game_ok, h->game_nbas);
    if (true)
        fprintf(stderr, "mmg_create: tile_ok %d tile_nt %d tile_smem %d tile_ext %d tile_persist %d persist_smem %d resident_budget %d tile_bwd_smem %d bwd_pre %d send_bwd %d split %d mc %d fast %d rc %d rc_persist %d rc_budget %d rc_bwd %d\n",
                (int)h->tile_ok, h->tile_nt, h->tile_smem, (int)h->tile_ext, (int)h->tile_persist, h->persist_smem, h->resident_budget, h->tile_bwd_smem,
                bwd_pre_lds_floats(h->dm) * 4, h->send_bwd_smem, (int)h->tile_split, (int)h->mc_ok, (int)h->use_fast, (int)h->rc_fwd, (int)h->rc_persist, h->rc_budget, (int)h->rc_bwd);
#endif
    if (h->conv_smem > 48 * 1024) {
        e = hipFuncSetAttribute((const void*)k_conversation<256>, hipFuncAttributeMaxDynamicSharedMemorySize, h->conv_smem);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_conversation<512>, hipFuncAttributeMaxDynamicSharedMemorySize, h->conv_smem);
    }
    if (e == hipSuccess && h->bwd_smem > 48 * 1024) e = hipFuncSetAttribute((const void*)k_bwd_conv<false>, hipFuncAttributeMaxDynamicSharedMemorySize, h->bwd_smem);
    if (e == hipSuccess && h->bwd_smem > 48 * 1024) e = hipFuncSetAttribute((const void*)k_bwd_conv<true>, hipFuncAttributeMaxDynamicSharedMemorySize, h->bwd_smem);
    if (e != hipSuccess) return fail("device init failed: %s", hipGetErrorString(e));
    // jobs with few output tiles split their rows further only when the whole table leaves the chip idle otherwise (continuous
    // mode: the receiver's dozen small matrices; measured at config 5, 256 samples: k_wgrad 32 -> 22 us.  With a full table --
    // config 3 at 512 samples, 1 660 tiles -- the extra tiles made it slower: 104 -> 205 us)
    {
        const bool want = h->wgrad_small_split;
        h->wgrad_small_split = false;
        if (build_jobs(h)) return -1;
        if (want && h->jt.gemm_tiles <= 256 && h->dm.T * h->dm.B > 2048) {
            h->wgrad_small_split = true;
            if (build_jobs(h)) return -1;
        }
    }
    {
        // the optimizer inside k_wgrad: its blocks spin on the norm role of the same launch, so ALL of them must be resident together
        int nb = 0;
        h->wgrad_opt = false;
        h->wgrad_opt_ok = !h->any_split && h->dm.use_binary && h->d_err != nullptr && !getenv("MMG_NO_WGRAD_OPT") && !no_roles &&
                          hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_wgrad<true>, MMG_BLOCK, 0) == hipSuccess &&
                          h->jt.n_wblocks + 5 <= nb * n_cu - 8;
        {
            int nb2 = 0;
            h->wgrad_stride = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb2, (const void*)k_wgrad<false>, MMG_BLOCK, 0) == hipSuccess) {
                const int others = h->jt.n_wblocks + 1 - h->jt.gemm_tiles;
                const int slots = ((nb2 * n_cu - others) / 8) * 8;
                // (measured, round 5: 1 336 tiles on 856 slots 289 -> 281 us per minibatch, 3 848 on 672 219 -> 217; 5 120 on 552 388 -> 396 --
                //  beyond ~6 tiles per workgroup the static split loses more to its ragged last round than the walk saves;
                //  a balanced stride (tiles / rounds) gave the gain away again: as many workgroups as are resident)
                if (h->jt.gemm_tiles > slots && slots >= 64 && h->jt.gemm_tiles <= 6 * slots) h->wgrad_stride = slots;
            }
        }
#ifdef MMG_DEBUG_CREATE
        fprintf(stderr, "mmg_create: wgrad_stride %d (gemm tiles %d)\n", h->wgrad_stride, h->jt.gemm_tiles);
        fprintf(stderr, "mmg_create: wgrad_opt_ok %d (blocks %d, resident %d x %d)\n", (int)h->wgrad_opt_ok, h->jt.n_wblocks + 5, nb, n_cu);
#endif
    }
    if (no_roles) {
        // nothing that spins on another workgroup of its own launch: per-step / per-phase launches only
        //   (MMG_NO_MERGE + MMG_NO_MERGE_PREP + MMG_NO_GAME + MMG_NO_WGRAD_OPT + MMG_NO_PERSIST + MMG_NO_SPLIT + MMG_NO_MC + MMG_NO_RC_PERSIST + MMG_NO_RC_BWD)
        if (h->game_ok || h->wgrad_opt_ok || h->tile_persist || h->tile_split || h->mc_ok || h->rc_persist || h->rc_bwd || h->merge_roles || h->sw_merge_prep)
            return fail("internal: a role launch survived the no-roles selection");
    }
    return 0;
}

extern "C" mmg_handle* mmg_create(const mmg_config* cfg, void* d_workspace, int64_t workspace_bytes,
                                  float* d_params, float* d_grads, float* d_opt_state) {
    if (validate(cfg)) return nullptr;
    if (!d_workspace || !d_params || !d_grads || !d_opt_state) { fail("NULL device buffer"); return nullptr; }
    mmg_handle* h = new mmg_handle();
    h->cfg = *cfg;
    if (h->cfg.global_batch <= 0) h->cfg.global_batch = h->cfg.batch;
    h->dm = make_dims(h->cfg);
    h->pl = param_layout(h->cfg);
    h->tl = tape_layout(h->cfg);
    if (workspace_bytes < h->tl.total) { fail("workspace too small: %lld < %lld", (long long)workspace_bytes, (long long)h->tl.total); delete h; return nullptr; }
    h->ws = d_workspace; h->params = d_params; h->grads = d_grads; h->opt_state = d_opt_state;
    h->P = resolve_params(h->pl, d_params);
    h->G = resolve_params(h->pl, d_grads);
    h->tp = resolve_tape(h->tl, d_workspace);
    h->d_jt = reinterpret_cast<JobTable*>(h->tp.tables);
    h->profiling = false; h->timers_used = 0; h->scores_in_parts = false;
    h->h_err = nullptr;
    h->d_err = nullptr;
    if (hipHostMalloc((void**)&h->h_err, sizeof(uint32_t), hipHostMallocMapped) == hipSuccess) {
        *h->h_err = 0u;
        if (hipHostGetDevicePointer((void**)&h->d_err, h->h_err, 0) != hipSuccess) h->d_err = nullptr;
    } else h->h_err = nullptr;
    h->no_roles = getenv("MMG_NO_ROLES") != nullptr;
    // a process whose compute units are masked (HSA_CU_MASK / ROC_GLOBAL_CU_MASK) still sees the whole chip in
    // hipGetDeviceProperties: the occupancy budgets below would promise co-residency the dispatcher cannot deliver
    if (getenv("HSA_CU_MASK") || getenv("ROC_GLOBAL_CU_MASK")) h->no_roles = true;
    if (select_paths(h)) { delete h; return nullptr; }
    {
        // does this device place workgroup i of a launch on XCD i % 8 (8 distinct XCDs)?  The XCD-aware launches only PREFER that
        // placement; the many-class conversation may additionally keep its hand-off pairs inside the XCD's L2 when it holds.
        h->xcd_rule_ok = false;
        if (h->mc_ok && h->mc_xcd) {
            const int n = 2048;
            uint32_t* d_probe = reinterpret_cast<uint32_t*>(h->tp.tables);      // (the job-table slot: uploaded below)
            if (hipMemset(d_probe, 0xff, n * sizeof(uint32_t)) == hipSuccess) {
                hipLaunchKernelGGL(k_xcc_probe, dim3(n), dim3(64), 0, 0, d_probe);
                std::vector<uint32_t> got(n);
                if (hipMemcpy(got.data(), d_probe, n * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess) {
                    bool ok = true;
                    for (int i = 0; i < n && ok; ++i) ok = got[i] < 8u && got[i] == got[i & 7];
                    for (int a = 0; a < 8 && ok; ++a) for (int b = a + 1; b < 8 && ok; ++b) ok = got[a] != got[b];
                    h->xcd_rule_ok = ok;
                }
            }
        }
        hipError_t e = hipMemset(d_workspace, 0, h->tl.total);
        if (e == hipSuccess) e = hipMemset(d_grads, 0, sizeof(float) * (h->pl.total + MMG_GRAD_TAIL));
        if (e != hipSuccess) { fail("device init failed: %s", hipGetErrorString(e)); delete h; return nullptr; }
    }
    hipError_t e = hipSuccess;
    {
        std::vector<float> one(256, 1.0f);
        e = hipMemcpy(h->tp.ones, one.data(), sizeof(float) * one.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { fail("constant upload failed: %s", hipGetErrorString(e)); delete h; return nullptr; }
    }
    e = hipMemcpy(h->d_jt, &h->jt, sizeof(JobTable), hipMemcpyHostToDevice);
    if (e != hipSuccess) { fail("job table upload failed: %s", hipGetErrorString(e)); delete h; return nullptr; }
    return h;
}

extern "C" void mmg_destroy(mmg_handle* h) { delete h; }      // (~mmg_handle releases the events and the pinned error word)

// ---------------------------------------------------------------------------------------------
// launch helper with optional HIP-event timing on the launch stream
// ---------------------------------------------------------------------------------------------
struct Scope {
    mmg_handle* h; hipStream_t st; KernelTimer* kt;
    Scope(mmg_handle* h_, hipStream_t st_, const char* name) : h(h_), st(st_), kt(nullptr) {
        if (!h->profiling) return;
        if (h->timers_used == h->timers.size()) {
            KernelTimer t; hipEventCreate(&t.t0); hipEventCreate(&t.t1); h->timers.push_back(t);
        }
        kt = &h->timers[h->timers_used++];
        kt->name = name;
        hipEventRecord(kt->t0, st);
    }
    ~Scope() { if (kt) hipEventRecord(kt->t1, st); }
};

extern "C" int mmg_set_profiling(mmg_handle* h, int enabled) {
    if (!h) return fail("NULL handle");
    h->profiling = enabled != 0; h->timers_used = 0;
    return 0;
}

extern "C" int mmg_get_kernel_times(mmg_handle* h, char* names, int names_bytes, float* ms, int max_kernels) {
    if (!h) return fail("NULL handle");
    int n = 0; std::string all;
    for (size_t i = 0; i < h->timers_used && n < max_kernels; ++i, ++n) {
        hipEventSynchronize(h->timers[i].t1);
        float t = 0.f; hipEventElapsedTime(&t, h->timers[i].t0, h->timers[i].t1);
        ms[n] = t;
        all += h->timers[i].name; all += ";";
    }
    if (names && names_bytes > 0) { strncpy(names, all.c_str(), names_bytes - 1); names[names_bytes - 1] = 0; }
    h->timers_used = 0;
    return n;
}

static int launch_check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

extern "C" int64_t mmg_log_snapshot_count(const mmg_config* cfg, int dump, int with_losses) {
    if (validate(cfg)) return -1;
    const int k = dump < cfg->batch ? (dump > 0 ? dump : 0) : cfg->batch;
    return log_snapshot_count(cfg->max_exchange, cfg->batch, cfg->w_dim, k, with_losses != 0);
}

extern "C" int mmg_log_snapshot(mmg_handle* h, const int64_t* d_target, int dump, int with_losses, double* d_out, void* stream) {
    if (!h) return fail("NULL handle");
    if (!d_out) return fail("out must not be NULL");
    const int k = dump < h->dm.B ? (dump > 0 ? dump : 0) : h->dm.B;
    if (!with_losses && k == 0) return 0;
    Scope sc(h, (hipStream_t)stream, "k_log_snapshot");
    hipLaunchKernelGGL(k_log_snapshot, dim3(h->dm.T + 2), dim3(MMG_BLOCK), 0, (hipStream_t)stream, h->dm, h->tp, d_target, k, with_losses ? 1 : 0, d_out);
    return launch_check("k_log_snapshot");
}

static int launch_gemm_nt(mmg_handle* h, hipStream_t st, const char* name, const float* X, int ldx, const float* Wm, int ldw,
                          const float* bias, float* out, int ldo, int M, int N, int K) {
    Scope sc(h, st, name);
    const int tiles = ((M + 15) / 16) * ((N + 15) / 16);
    hipLaunchKernelGGL(k_gemm_nt, dim3(tiles), dim3(MMG_BLOCK), 0, st, X, ldx, Wm, ldw, bias, out, ldo, M, N, K);
    return launch_check(name);
}

// x != NULL: also computes h_x = image_layer(x) in the same launch
static int launch_prep(mmg_handle* h, hipStream_t st, const float* desc, const float* x, int bump_mb) {
    Scope sc(h, st, x ? "k_prep+h_x" : "k_prep");
    const Dims& d = h->dm;
    const int hx_tiles = x ? ((d.B + 15) / 16) * ((d.H + 15) / 16) : 0;
    const int cpb = h->prep_cpb, nC = (d.D + cpb - 1) / cpb;
    (void)nC; (void)hx_tiles;
    hipLaunchKernelGGL(k_prep, dim3(prep_blocks(d, cpb, x != nullptr)), dim3(MMG_BLOCK), h->prep_smem, st, h->dm, h->P, h->tp, desc, x, cpb, bump_mb);
    return launch_check("k_prep");
}

static int launch_baselines_fused(mmg_handle* h, hipStream_t st) {
    const Dims& d = h->dm; const Tape& tp = h->tp; const Params& P = h->P;
    const int rows = d.T * d.B;
    BasArgs rec, sen;
    memset(&rec, 0, sizeof(rec)); memset(&sen, 0, sizeof(sen));
    rec.rows = rows; rec.x1 = tp.z; rec.ld1 = d.W; rec.k1 = d.W;
    rec.x2 = tp.h + (size_t)d.B * d.R; rec.ld2 = d.R; rec.k2 = d.R;
    rec.W1 = P.p[BR_L1_W]; rec.ldw = d.W + d.R; rec.col0 = 0; rec.b1 = P.p[BR_L1_B];
    rec.W2 = P.p[BR_L2_W]; rec.b2 = P.p[BR_L2_B]; rec.hid = tp.hid_r; rec.score = tp.br;
    sen.rows = rows; sen.x1 = tp.hx; sen.ld1 = d.H; sen.k1 = d.H; sen.mod1 = d.B;      // h_x is per sample, not per step
    sen.x2 = tp.zr; sen.ld2 = d.W; sen.k2 = d.W;
    sen.W1 = P.p[BS_L1_W]; sen.ldw = d.H + d.W; sen.col0 = 0; sen.b1 = P.p[BS_L1_B];
    sen.W2 = P.p[BS_L2_W]; sen.b2 = P.p[BS_L2_B]; sen.hid = tp.hid_s; sen.score = tp.bs;
    Scope sc(h, st, "k_baselines");
    hipLaunchKernelGGL(k_baselines, dim3((rows + 15) / 16, 2), dim3(MMG_BLOCK), 0, st, d.K, rec, sen);
    return launch_check("k_baselines");
}

// register-resident kernels exist for the agent shape of BASELINE configs 1-3
static bool fast_shape(const mmg_handle* h) {
    const Dims& d = h->dm;
    if (h->tile_ok && h->tile_force) return false;
    return h->use_fast && d.H == 256 && d.W == 32 && d.R == 64 && d.V == 100 && d.D <= 32 && d.T <= 16;   // (D = 30: own instantiation, other D <= 32: capacity 32)
}
// every other shape: sample tiles on the matrix cores (kernels_tile.h); the per-sample generic kernels remain for
// dimensions whose tile does not fit the LDS and for the agent-level entry points
// the small agents with many classes (32 < D <= 1024): register-resident conversation with class slices (kernels_mc.h) up to 2 048
// samples per GPU (measured at D = 1000: 2 048 samples 1 064 us per minibatch against 1 113 on the sample tiles, 4 096 samples
// 2 090 against 1 242 -- from 256 tiles on, the tiles fill the chip and a workgroup per sample is 16 waves of it;
// MMG_TILE=1 forces the tiles)
static bool mc_path(const mmg_handle* h) { return h->mc_ok && !(h->tile_ok && h->tile_force) && (h->dm.B <= 2048 || !h->tile_ok); }
static bool tile_path(const mmg_handle* h) { return h->tile_ok && !fast_shape(h) && !mc_path(h); }

static int launch_conv_tile(mmg_handle* h, hipStream_t st, ConvArgs ar) {
    const Dims& d = h->dm;
    const int tiles = (d.B + MMG_TM - 1) / MMG_TM;
    auto conv = [&](const ConvArgs& a) {
        if (h->tile_nt == 512) hipLaunchKernelGGL(k_conv_tile<512>, dim3(tiles), dim3(512), h->tile_smem, st, h->dm, h->P, h->tp, a);
        else hipLaunchKernelGGL(k_conv_tile<256>, dim3(tiles), dim3(256), h->tile_smem, st, h->dm, h->P, h->tp, a);
    };
    if (h->tile_split) {
        Scope sc(h, st, "k_conv_split");
        ar.phases = 3; ar.t_begin = 0; ar.t_end = d.T; ar.nhelp = h->split_nh; ar.per = h->split_per;
        hipLaunchKernelGGL(k_conv_split<512>, dim3(tiles * (1 + ar.nhelp)), dim3(512), h->split_smem, st, h->dm, h->P, h->tp, ar, tiles);
        return launch_check("k_conv_split");
    }
    if (!h->tile_ext) {
        Scope sc(h, st, "k_conv_tile");
        ar.phases = 3; ar.t_begin = 0; ar.t_end = d.T;
        conv(ar);
        return launch_check("k_conv_tile");
    }
    if (h->tile_persist) {
        ar.phases = 2; ar.t_begin = 0; ar.t_end = d.T; ar.persist = 1; ar.ns1 = h->persist_ns1; ar.ns2 = h->persist_ns2;
        // receiver shape of the register-resident kernels: one receiver role per SAMPLE (rs_role) beside the tiles' sender roles
        ar.rsample = (h->rs_capable) ? 1 : 0;
        if (ar.rsample && d.W == 256 && h->sw_rmsg) ar.rsample = 2;      // ... which also form the receiver's message
        if (ar.rsample == 2 && d.H % 64 == 0 && d.H / 64 <= 16 && h->sw_fused_s) {
            ar.rsample = 3;                                 // fused sender roles (sa_role / sb_role)
            ar.ns1 = d.H / 64; ar.ns2 = d.W / 16;
        }
        // all roles of a launch must be co-resident (resident_budget workgroups, occupancy query at mmg_create): as many whole
        // tiles per launch as fit, the batch in consecutive launches (the conversations of different samples are independent).
        // The path is chosen BEFORE the timing scope opens (a fall-back to the per-step launches leaves no empty timer).
        const int per_tile = MMG_TM + ar.ns1 + ar.ns2;
        int ct = h->resident_budget / per_tile;
        if (ct < 1) ct = 1;
        const int nchunk = (tiles + ct - 1) / ct;
        ct = (tiles + nchunk - 1) / nchunk;
        // (measured with config 4's agents: 256 samples in 4 launches 471 us against 858 us as per-step launches; 1024 samples
        //  in 13 launches 1 723 against 1 544 -- beyond six launches the per-step GEMM launches over the whole batch win)
        const bool sample_roles = ar.rsample && nchunk <= 6 && per_tile <= h->resident_budget;
        const bool tile_roles = !ar.rsample && tiles * (1 + ar.ns1 + ar.ns2) <= h->resident_budget;
        if (sample_roles) {
            Scope sc(h, st, "k_conv_persist");
            // basehx tiles for k_baselines4 ride along as trailing workgroups (training minibatches of <= 64 samples)
            const bool want_base = nchunk == 1 && ar.train && d.use_binary && !ar.run_all && d.B <= 64 && !(d.H & 3) && h->merge_roles;
            const int bt = want_base ? ((d.B + 15) / 16) * ((d.K + 15) / 16) : 0;
            for (int c = 0; c < nchunk; ++c) {
                ar.b_begin = c * ct * MMG_TM;
                ar.b_count = (d.B - ar.b_begin < ct * MMG_TM) ? d.B - ar.b_begin : ct * MMG_TM;
                if (ar.b_count <= 0) break;
                const int ctiles = (ar.b_count + MMG_TM - 1) / MMG_TM;
                if (ar.rsample == 3 && h->persist_ll)
                    hipLaunchKernelGGL((k_conv_persist<512, true, true>), dim3(ar.b_count + ctiles * (ar.ns1 + ar.ns2) + bt), dim3(512), h->persist_smem, st, h->dm, h->P, h->tp, ar, tiles);
                else
                hipLaunchKernelGGL((k_conv_persist<512, true>), dim3(ar.b_count + ctiles * (ar.ns1 + ar.ns2) + bt), dim3(512), h->persist_smem, st, h->dm, h->P, h->tp, ar, tiles);
            }
            h->basehx_ready = want_base;
            return launch_check("k_conv_persist");
        }
        if (tile_roles) {
            Scope sc(h, st, "k_conv_persist");
            const int roles = 1 + ar.ns1 + ar.ns2;
            hipLaunchKernelGGL((k_conv_persist<512, false>), dim3(tiles * roles), dim3(512), h->persist_smem, st, h->dm, h->P, h->tp, ar, tiles);
            return launch_check("k_conv_persist");
        }
    }
    // per-step launches: no co-residency needed (any device, any batch)
    ar.persist = 0; ar.rsample = 0;
    const int skip = (!ar.run_all && !d.fixed && ar.train) ? 1 : 0;
    if (h->rc_fwd && h->rc_persist) {
        // wide receiver, all roles co-resident: one launch for the whole conversation (kernels_rc.h: k_rc_persist)
        Scope sc(h, st, "k_conv_rc");
        const int nj = d.R / 16, njw = d.W / 16, per_tile = (nj > njw ? nj : njw) + njw + (d.H + 63) / 64 + 1;
        ar.phases = 2;
        // basehx tiles for k_baselines4 ride along as trailing workgroups (training minibatches of <= 64 samples)
        int ct = h->rc_budget / per_tile;
        const int nchunk = (tiles + ct - 1) / ct;
        ct = (tiles + nchunk - 1) / nchunk;
        const bool want_base = nchunk == 1 && ar.train && d.use_binary && !ar.run_all && d.B <= 64 && !(d.H & 3) && h->merge_roles;
        const int bt = want_base ? ((d.B + 15) / 16) * ((d.K + 15) / 16) : 0;
        for (int c = 0; c < nchunk; ++c) {
            const int t0 = c * ct, nt = (tiles - t0 < ct) ? tiles - t0 : ct;
            if (nt <= 0) break;
            hipLaunchKernelGGL(k_rc_persist, dim3(nt * per_tile + bt), dim3(256), 0, st, h->dm, h->P, h->tp, ar, nt, t0);
        }
        h->basehx_ready = want_base;
        return launch_check("k_rc_persist");
    }
    if (h->rc_fwd) {
        // wide receiver: the receiver step of a tile as three launches over 16-unit / 16-bit slices (kernels_rc.h)
        Scope sc(h, st, "k_conv_rc");
        const int nj = d.R / 16, njw = d.W / 16;
        ar.phases = 2;
        for (int t = 0; t < d.T; ++t) {
            hipLaunchKernelGGL(k_send_s1, dim3(tiles * ((d.H + 15) / 16)), dim3(MMG_BLOCK), 0, st, h->dm, h->P, h->tp, t, skip);
            hipLaunchKernelGGL(k_send_s2, dim3(tiles * ((d.W + 15) / 16)), dim3(MMG_BLOCK), 0, st, h->dm, h->P, h->tp, ar, t, skip);
            hipLaunchKernelGGL(k_rc_gru, dim3(tiles * nj), dim3(256), 0, st, h->dm, h->P, h->tp, ar, t, skip);
            hipLaunchKernelGGL(k_rc_heads, dim3(tiles * nj), dim3(256), 0, st, h->dm, h->P, h->tp, ar, t, skip);
            hipLaunchKernelGGL(k_rc_query, dim3(tiles * njw), dim3(256), 0, st, h->dm, h->P, h->tp, ar, t, skip);
        }
        hipLaunchKernelGGL(k_rc_tail, dim3(tiles), dim3(256), 0, st, h->dm, h->P, h->tp, ar);
        return launch_check("k_conv_rc");
    }
    for (int t = 0; t < d.T; ++t) {
        {
            Scope sc(h, st, "k_send_s1");
            hipLaunchKernelGGL(k_send_s1, dim3(tiles * ((d.H + 15) / 16)), dim3(MMG_BLOCK), 0, st, h->dm, h->P, h->tp, t, skip);
        }
        {
            Scope sc(h, st, "k_send_s2");
            hipLaunchKernelGGL(k_send_s2, dim3(tiles * ((d.W + 15) / 16)), dim3(MMG_BLOCK), 0, st, h->dm, h->P, h->tp, ar, t, skip);
        }
        {
            Scope sc(h, st, "k_conv_tile");
            ar.phases = 2; ar.t_begin = t; ar.t_end = t + 1;
            conv(ar);
        }
        if (launch_check("k_conv_tile (step)")) return -1;
    }
    return 0;
}

static int exchange_forward_impl(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc,
                                 const float* d_u_z, const float* d_u_s, const float* d_u_w, uint64_t seed,
                                 int train, int run_all_steps, void* stream) {
    if (!d_x || !d_desc) return fail("x / desc must not be NULL");
    hipStream_t st = (hipStream_t)stream;
    const Dims& d = h->dm;
    // register-resident forward (k_conversation_fast3): k_prep's blocks run as leading roles of the conversation's launch -- when
    // every prep and sample role has a CU of its own (the launch holds ONE workgroup per CU: with 512 samples the 531 prep roles would be two
    // more rounds of workgroups ahead of the conversations: 318 us per minibatch against 306 with k_prep as its own launch)
    const bool merge_prep = h->sw_merge_prep && !tile_path(h) && !mc_path(h) && fast_shape(h) &&
                            h->prep_smem <= fast3_lds_bytes() &&
                            prep_blocks(d, h->prep_cpb, true) + d.B <= h->n_cu;
    if (!merge_prep && launch_prep(h, st, d_desc, d_x, train ? 1 : 0)) return -1;
    const bool bas = train && d.use_binary;
    ConvArgs ar;
    memset(&ar, 0, sizeof(ar));
    ar.x = d_x; ar.target = d_target; ar.desc = d_desc; ar.u_z = d_u_z; ar.u_s = d_u_s; ar.u_w = d_u_w; ar.seed = seed;
    // run_all_steps == 2: training-minimal (as 0; the class logits y[t] of the steps before the output step are not kept)
    const int y_last_only = (run_all_steps == 2 && d.fixed) ? 1 : 0;
    const int lean = (run_all_steps == 2 && !d.use_binary) ? 1 : 0;
    if (run_all_steps == 2) run_all_steps = 0;
    // run_all_steps == 3 (the minibatches whose log block reads the whole tape): the CONVERSATION runs every sample through all
    // steps, everything else is the training step's -- the baselines over the live rows only, in the statistics / backward launch
    // (the log block prints no baseline score; the generic all-rows baselines launch of mode 1 costs 55 us at config 2).
    // Register-resident agents of a training pass only: other paths take it as mode 1.
    const bool tape_all = run_all_steps == 3 && train && d.use_binary && fast_shape(h) && !tile_path(h) && !mc_path(h);
    if (run_all_steps == 3) run_all_steps = tape_all ? 0 : 1;
    ar.y_last_only = y_last_only; ar.lean = lean;
    ar.train = train; ar.run_all = (run_all_steps || tape_all) ? 1 : 0; ar.t_begin = 0; ar.t_end = d.T; ar.phases = 3; ar.sprod_first = 1;
    bool base_ready = false;
    h->basehx_ready = false;
    h->bas_deferred = false;
    h->bas_pending = false;
    if (tile_path(h)) {
        if (launch_conv_tile(h, st, ar)) return -1;
    } else if (mc_path(h)) {
        Scope sc(h, st, "k_conversation_mc");
        const int ntile = (d.B + 15) / 16;
        ar.per = h->mc_per;
        ar.l2_handoff = (h->mc_xcd && h->xcd_rule_ok) ? 1 : 0;
        const int grid = h->mc_xcd ? ((ntile + 7) / 8) * 128 : ntile * 16;     // (mc_xcd assumes the 8 XCDs of an unpartitioned MI355X; mmg_create clears it otherwise)
        // ... when it needs fewer rounds: a round of 16 pairs takes ~1.55x a round of 16 single tiles (measured, scripts/mc3p_ab.py:
        // 768 samples = 24 pairs = two rounds lose to three rounds of single tiles, every other multiple of 256 from 512 on wins)
        const int mc3p_rounds = (((ntile + 1) / 2) + h->n_cu / 16 - 1) / (h->n_cu / 16 > 0 ? h->n_cu / 16 : 1), mc3_rounds = (ntile * 16 + h->n_cu - 1) / h->n_cu;
        if (h->mc3_ok && h->mc3p_ok && lean && 31 * mc3p_rounds < 20 * mc3_rounds) {
            // two sample tiles per workgroup, half a step apart (kernels_mc3p.h): 128 consecutive workgroups = 8 pairs of tiles x 16 members
            const int npair = (ntile + 1) / 2;
            int nblk = (npair + 7) / 8;                     // blocks of 128 workgroups = 8 pairs x 16 members; one workgroup per CU: the launch is persistent
            if (nblk > h->n_cu / 128) nblk = h->n_cu / 128 > 0 ? h->n_cu / 128 : 1;
            hipLaunchKernelGGL((k_conversation_mc3p<256, 32, 64, 100, 64>), dim3(nblk * 128), dim3(256), mc3p_lds_bytes(d.T), st, h->dm, h->P, h->tp, ar, ntile, y_last_only);
        } else if (h->mc3_ok)
            hipLaunchKernelGGL((k_conversation_mc3<256, 32, 64, 100, 64>), dim3(grid), dim3(256), mc3_lds_bytes(), st, h->dm, h->P, h->tp, ar, ntile, h->mc_xcd, y_last_only);
        else
            hipLaunchKernelGGL((k_conversation_mc<256, 32, 64, 100, 64>), dim3(grid), dim3(512), 0, st, h->dm, h->P, h->tp, ar, ntile, h->mc_xcd, y_last_only);
        if (launch_check("k_conversation_mc")) return -1;
    } else {
        Scope sc(h, st, "k_conversation");
        const bool fast = fast_shape(h);
        base_ready = fast && bas && !run_all_steps && h->merge_roles;
        const int base_tiles = base_ready ? ((d.B + 15) / 16) * ((d.K + 15) / 16) : 0;
        if (fast) {
            ar.nprep = merge_prep ? prep_blocks(d, h->prep_cpb, true) : 0; ar.prep_cpb = h->prep_cpb; ar.nbase = base_tiles;
            if (merge_prep) hipLaunchKernelGGL((k_conversation_fast3<256, 32, 64, 100, true>), dim3(ar.nprep + d.B + base_tiles + 1), dim3(256), fast3_lds_bytes(), st, h->dm, h->P, h->tp, ar);
            else hipLaunchKernelGGL((k_conversation_fast3<256, 32, 64, 100, false>), dim3(d.B + base_tiles), dim3(256), fast3_lds_bytes(), st, h->dm, h->P, h->tp, ar);
        }
        else
            if (h->conv_threads == 512)
                hipLaunchKernelGGL(k_conversation<512>, dim3(d.B), dim3(512), h->conv_smem, st, h->dm, h->P, h->tp, ar);
            else
                hipLaunchKernelGGL(k_conversation<256>, dim3(d.B), dim3(MMG_BLOCK), h->conv_smem, st, h->dm, h->P, h->tp, ar);
        if (launch_check("k_conversation")) return -1;
    }
    h->scores_in_parts = false;
    if (bas) {
        if (run_all_steps) {                         // exchange(): every row, scores materialised directly
            if (launch_baselines_fused(h, st)) return -1;
        } else if (h->defer_bas && base_ready && d.B <= 64 && (d.K + 63) / 64 <= 8 && h->sw_merge_bas && !d.fixed &&
                   h->n_cu >= 2 * (d.B + (5 * d.T + 2 + 3) / 4)) {
            // fused step, register-resident kernels: the baselines' live-row pass (k_baselines3's body) runs as workgroup roles of
            // the backward launch, beside the sample roles' statistics-independent prologue (kernels_fast.h) -- one launch less.
            // (The sample and statistics roles of that launch sit ahead of these producers and spin: only with CUs to spare.
            //  Adaptive conversations only: with early stopping ~140 of the 640 (step, sample) rows are live = ~144 baseline roles;
            //  Fixed mode keeps all 640 rows live and the backward kernel holds ONE workgroup per CU -- measured at config 3:
            //  101.3 us per minibatch with the roles against 94.6 with k_baselines3 as its own launch; config 2: 66.0 against 72.1.)
            h->bas_deferred = true;
            h->scores_in_parts = true;
        } else if (!h->defer_bas && base_ready && d.B <= 64 && (d.K + 63) / 64 <= 8 && h->sw_merge_bas &&
                   h->n_cu >= 2 * ((5 * d.T + 2 + 3) / 4)) {
            // phased (data-parallel) step, register-resident kernels: the baselines run in mmg_loss_stats' launch, as roles beside
            // the statistics roles that consume their scores (k_bas_stats) -- one launch instead of two before the statistics all-reduce
            h->bas_pending = true;
            h->scores_in_parts = true;
        } else {
            Scope sc(h, st, "k_baselines");
            const bool live_rows = base_ready && d.B <= 64;      // k_baselines3: live (step, sample) rows only
            if (tile_path(h) && d.B <= 64 && !(d.H & 3)) {
                // any message / state width: basehx as a GEMM launch, then one MFMA pass over the live rows (kernels_tile.h)
                const int bt = ((d.B + 15) / 16) * ((d.K + 15) / 16);
                if (!h->basehx_ready)
                    hipLaunchKernelGGL(k_gemm_nt, dim3(bt), dim3(MMG_BLOCK), 0, st, (const float*)h->tp.hx, d.H, (const float*)h->P.p[BS_L1_W], d.H + d.W,
                                       (const float*)nullptr, h->tp.basehx, d.K, d.B, d.K, d.H);
                hipLaunchKernelGGL(k_baselines4, dim3((d.T * d.B + 15) / 16, (d.K + 63) / 64, 2), dim3(MMG_BLOCK), 0, st, h->dm, h->P, h->tp);
            } else if (live_rows) {
                hipLaunchKernelGGL(k_baselines3, dim3((d.T * d.B + 15) / 16, (d.K + 63) / 64, 2), dim3(MMG_BLOCK), 0, st, h->dm, h->P, h->tp);
            } else {
                // grid.z = 2 baselines x 2 step ranges: 128 workgroups at config 1 instead of 64
                hipLaunchKernelGGL(k_baselines2, dim3((d.B + 15) / 16, (d.K + 63) / 64, 2 * (d.T >= 4 ? 2 : 1)), dim3(MMG_BLOCK), 0, st,
                                   h->dm, h->P, h->tp, 1, base_ready ? 1 : 0);
            }
            if (launch_check("k_baselines")) return -1;
            h->scores_in_parts = true;
        }
    }
    return 0;
}

extern "C" int mmg_exchange_forward(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc,
                                    const float* d_u_z, const float* d_u_s, const float* d_u_w, uint64_t seed,
                                    int train, int run_all_steps, void* stream) {
    if (!h) return fail("NULL handle");
    const int warn = train ? error_gate(h, (hipStream_t)stream, true) : 0;     // a training forward pass starts a minibatch
    if (warn < 0) return -1;
    const int rc = exchange_forward_impl(h, d_x, d_target, d_desc, d_u_z, d_u_s, d_u_w, seed, train, run_all_steps, stream);
    return rc ? rc : warn;
}

extern "C" int mmg_loss_stats(mmg_handle* h, void* stream) {
    if (!h) return fail("NULL handle");
    hipStream_t st = (hipStream_t)stream;
    if (h->bas_pending) {
        Scope sc(h, st, "k_bas_stats");
        const Dims& d = h->dm;
        const int n_stats = (5 * d.T + 2 + 3) / 4, n_bas = ((d.T * d.B + 15) / 16) * 2 * ((d.K + 63) / 64);
        hipLaunchKernelGGL(k_bas_stats, dim3(n_stats + n_bas), dim3(MMG_BLOCK), 0, st, h->dm, h->P, h->tp, n_stats);
        h->bas_pending = false;
        return launch_check("k_bas_stats");
    }
    Scope sc(h, st, "k_stats");
    hipLaunchKernelGGL(k_stats, dim3(5 * h->dm.T + 2), dim3(64), 0, st, h->dm, h->P, h->tp, h->scores_in_parts ? 1 : 0);
    return launch_check("k_stats");
}

// single-GPU minibatch: the statistics run as extra roles of the backward launch (no all-reduce in between)
// continuous many-class path: the two-launch backward of kernels_mc.h
static bool mc_bwd(const mmg_handle* h) { return mc_path(h) && !h->dm.use_binary; }
static bool merge_stats(const mmg_handle* h) {
    if (mc_bwd(h)) return h->merge_roles;            // (sum of rewards / hits only: one extra workgroup of k_bwd_mc2)
    return fast_shape(h) && h->dm.use_binary && h->scores_in_parts && h->merge_roles;
}

static int backward_impl(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc, hipStream_t st, bool with_stats, bool conv_done = false) {
    const Dims& d = h->dm;
    bool row_map = false;
    if (conv_done) {
        row_map = true;                                  // k_game_fast ran the reverse pass and its first class role listed the live rows
    } else if (tile_path(h)) {
        row_map = d.T * d.B <= 2048;                     // k_wgrad keeps the live-row list in LDS (2048 entries)
        const int zero_dead = (!row_map && !d.fixed) ? 1 : 0;
        const int tiles = (d.B + MMG_TM - 1) / MMG_TM;
        // the sender's backward rides in the same launch as k_bwd_pre (independent latency chains side by side) while the row
        // blocks are few: it then walks all T * B rows instead of the live-row list (MMG_NO_MERGE=1: separate launches)
        const bool merged_send = d.use_binary && h->merge_roles && d.T * d.B <= 2048;
        bool dhx_done = false;
        {
            Scope sc(h, st, "k_bwd_tile");
            // the dh-independent part of the receiver's BPTT (seeds, dgpre, dhin) for all (step, sample) rows, then the recurrence
            // wide receiver whose reverse-time loop runs as roles (k_rc_bwd adds the partials): four column bands per (step, tile)
            const int pre_bands = (h->rc_fwd && h->rc_bwd && d.R == 256) ? 4 : 1;
            if (d.use_binary && merged_send) {
                const int nbands = (d.H + 63) / 64, nrb = (d.T * d.B + MMG_TM - 1) / MMG_TM;
                const int smem = bwd_pre_lds_floats(d) * 4 > h->send_bwd_smem ? bwd_pre_lds_floats(d) * 4 : h->send_bwd_smem;
                if (d.R <= 128) hipLaunchKernelGGL(k_bwd_pre_send<8>, dim3(d.T * tiles + nrb * nbands), dim3(MMG_BLOCK), smem, st, h->dm, h->P, h->tp, zero_dead, d.T * tiles, nbands, 1);
                else hipLaunchKernelGGL(k_bwd_pre_send<16>, dim3(d.T * tiles * pre_bands + nrb * nbands), dim3(MMG_BLOCK), smem, st, h->dm, h->P, h->tp, zero_dead, d.T * tiles * pre_bands, nbands, pre_bands);
            } else if (d.use_binary) {
                if (d.R <= 128) hipLaunchKernelGGL(k_bwd_pre<8>, dim3(d.T * tiles), dim3(MMG_BLOCK), bwd_pre_lds_floats(d) * 4, st, h->dm, h->P, h->tp, zero_dead);
                else hipLaunchKernelGGL(k_bwd_pre<16>, dim3(d.T * tiles), dim3(MMG_BLOCK), bwd_pre_lds_floats(d) * 4, st, h->dm, h->P, h->tp, zero_dead);
            }
            if (h->rs_capable) {
                // receiver shape of the register-resident kernels: one workgroup per sample (+ one for the live-row list)
                // (+ k_dhx's blocks when the sender's backward already ran: its dpre is complete)
                const int nblk = (d.B * (d.H / 4) + MMG_BLOCK - 1) / MMG_BLOCK, ndhx = merged_send ? nblk + (d.H / 4 + 63) / 64 : 0;
                dhx_done = merged_send;
                if (d.D == 30) hipLaunchKernelGGL((k_bwd_sample<64, 100, 30>), dim3(d.B + 1 + ndhx), dim3(256), 0, st, h->dm, h->P, h->tp, d_target, zero_dead, row_map ? 1 : 0, nblk);
                else hipLaunchKernelGGL((k_bwd_sample<64, 100, 32>), dim3(d.B + 1 + ndhx), dim3(256), 0, st, h->dm, h->P, h->tp, d_target, zero_dead, row_map ? 1 : 0, nblk);
            } else if (d.R <= 64)
                hipLaunchKernelGGL((k_bwd_tile<512, 2>), dim3(tiles), dim3(512), h->tile_bwd_smem, st, h->dm, h->P, h->tp, d_target, zero_dead, row_map ? 1 : 0);
            else if (d.R <= 128)
                hipLaunchKernelGGL((k_bwd_tile<512, 4>), dim3(tiles), dim3(512), h->tile_bwd_smem, st, h->dm, h->P, h->tp, d_target, zero_dead, row_map ? 1 : 0);
            else if (h->rc_fwd && h->rc_bwd) {
                // wide receiver: the output-step prelude and the reverse-time loop as roles over 16-unit slices (kernels_rc.h)
                if (!d.use_binary) hipMemsetAsync(h->tp.rcflags, 0, 64 * 64 * sizeof(uint32_t), st);   // (binary mode: zeroed by k_bwd_pre)
                hipLaunchKernelGGL(k_rc_bwd, dim3(tiles * (d.R / 16)), dim3(256), 0, st, h->dm, h->P, h->tp, d_target, zero_dead,
                                   (d.use_binary && merged_send) ? pre_bands : 1, row_map ? 3 : 1);
            } else
                hipLaunchKernelGGL((k_bwd_tile<512, 8>), dim3(tiles), dim3(512), h->tile_bwd_smem, st, h->dm, h->P, h->tp, d_target, zero_dead, row_map ? 1 : 0);
            if (launch_check("k_bwd_tile")) return -1;
        }
        if (d.use_binary) {
            Scope sc(h, st, "k_send_bwd");
            if (!merged_send)
                hipLaunchKernelGGL(k_send_bwd, dim3((d.T * d.B + MMG_TM - 1) / MMG_TM, (d.H + 63) / 64), dim3(MMG_BLOCK), h->send_bwd_smem, st,
                                   h->dm, h->P, h->tp, (const int*)(row_map ? h->tp.rmap : nullptr), (const int*)(row_map ? h->tp.rcount : nullptr));
            const int nblk = (d.B * (d.H / 4) + MMG_BLOCK - 1) / MMG_BLOCK;
            if (!dhx_done) hipLaunchKernelGGL(k_dhx, dim3(nblk + (d.H / 4 + 63) / 64), dim3(MMG_BLOCK), 0, st, h->dm, h->tp, nblk);
            if (launch_check("k_send_bwd")) return -1;
        }
    } else if (mc_bwd(h)) {
        Scope sc(h, st, "k_bwd_mc");
        const int ntile = (d.B + 15) / 16;
        int ngroup = ntile;                              // one sample tile per workgroup up to 16 groups (measured at 256 samples: 4 groups 26 us, 8: 15, 16: 10)
        ngroup = ngroup < 1 ? 1 : (ngroup > 16 ? 16 : ngroup);
        hipLaunchKernelGGL((k_bwd_mc1<64, 64>), dim3(16 * ngroup), dim3(512), 0, st, h->dm, h->P, h->tp, d_target, h->mc_per, ntile, ngroup);
        const int nred = (2 * d.D * d.R / 4 + MMG_BLOCK - 1) / MMG_BLOCK;
        hipLaunchKernelGGL((k_bwd_mc2<64, 100>), dim3(d.B + nred + (with_stats ? 1 : 0)), dim3(MMG_BLOCK), 0, st, h->dm, h->P, h->tp, ngroup, nred, with_stats ? 1 : 0);
        if (launch_check("k_bwd_mc")) return -1;
    } else {
        Scope sc(h, st, "k_bwd_conv");
        const bool fast = fast_shape(h);
        const bool merge_dc = fast && h->merge_roles;
        row_map = merge_dc && d.T * d.B <= 2048;         // class role 0 lists the live (step, sample) rows for k_wgrad
        // k_conversation_fast3 stores softmax rows, not dbar = softmax(y) . desc: trailing workgroups form it (16 rows each)
        const int n_dbar = (fast && d.use_binary) ? (d.T * d.B + 15) / 16 : 0;
        if (fast && with_stats) {
            const int n_stats = (5 * d.T + 2 + 3) / 4;       // statistics roles: one (stream, step) pair per wave
            const int n_bas = h->bas_deferred ? ((d.T * d.B + 15) / 16) * 2 * ((d.K + 63) / 64) : 0;     // baseline roles: 16 live rows x 64 hidden units each
            if (d.D == 30) hipLaunchKernelGGL((k_bwd_conv_fast<256, 32, 64, 100, 30, true, true>), dim3(n_stats + d.B + d.D + n_dbar + n_bas), dim3(256), 0, st, h->dm, h->P, h->tp, d_target, n_stats, row_map ? 0 : 1, n_dbar, n_bas);
            else hipLaunchKernelGGL((k_bwd_conv_fast<256, 32, 64, 100, 32, true, true>), dim3(n_stats + d.B + d.D + n_dbar + n_bas), dim3(256), 0, st, h->dm, h->P, h->tp, d_target, n_stats, row_map ? 0 : 1, n_dbar, n_bas);
        } else if (merge_dc)
            if (d.D == 30) hipLaunchKernelGGL((k_bwd_conv_fast<256, 32, 64, 100, 30, false, true>), dim3(d.B + d.D + n_dbar), dim3(256), 0, st, h->dm, h->P, h->tp, d_target, 0, row_map ? 0 : 1, n_dbar, 0);
            else hipLaunchKernelGGL((k_bwd_conv_fast<256, 32, 64, 100, 32, false, true>), dim3(d.B + d.D + n_dbar), dim3(256), 0, st, h->dm, h->P, h->tp, d_target, 0, row_map ? 0 : 1, n_dbar, 0);
        else if (fast)      // (a 512-thread variant of this kernel measured slower: 31.8 vs 28.8 us -- it is not issue-bound)
            if (d.D == 30) hipLaunchKernelGGL((k_bwd_conv_fast<256, 32, 64, 100, 30, false, false>), dim3(d.B + n_dbar), dim3(256), 0, st, h->dm, h->P, h->tp, d_target, 0, 1, n_dbar, 0);
            else hipLaunchKernelGGL((k_bwd_conv_fast<256, 32, 64, 100, 32, false, false>), dim3(d.B + n_dbar), dim3(256), 0, st, h->dm, h->P, h->tp, d_target, 0, 1, n_dbar, 0);
        else
            if (d.B > 512)
                hipLaunchKernelGGL(k_bwd_conv<true>, dim3(d.B), dim3(MMG_BLOCK), h->bwd_smem, st, h->dm, h->P, h->tp, d_target);
            else
                hipLaunchKernelGGL(k_bwd_conv<false>, dim3(d.B), dim3(MMG_BLOCK), h->bwd_smem, st, h->dm, h->P, h->tp, d_target);
        if (launch_check("k_bwd_conv")) return -1;
    }
    if (conv_done) {
    } else if (tile_path(h)) {
        Scope sc(h, st, "k_dC");
        const int RL = d.R < MMG_BLOCK ? d.R : MMG_BLOCK, CPB = MMG_BLOCK / RL, nsb = dc_slices(d.B);
        hipLaunchKernelGGL(k_dC_tile, dim3((d.D + CPB - 1) / CPB, nsb), dim3(MMG_BLOCK), 0, st, h->dm, h->P, h->tp, nsb, 0);
        if (nsb > 1) hipLaunchKernelGGL(k_dC_tile, dim3((d.D + CPB - 1) / CPB, 1), dim3(MMG_BLOCK), 0, st, h->dm, h->P, h->tp, nsb, 1);
        if (launch_check("k_dC_tile")) return -1;
    } else if (!(fast_shape(h) && h->merge_roles) && !mc_bwd(h)) {
        Scope sc(h, st, "k_dC");
        hipLaunchKernelGGL(k_dC, dim3(d.D), dim3(MMG_BLOCK), 0, st, h->dm, h->P, h->tp);
        if (launch_check("k_dC")) return -1;
    }
    {
        Scope sc(h, st, "k_wgrad");
        WgOpt wo;
        memset(&wo, 0, sizeof(wo));
        WgHead hd;
        hd.gemm_tiles = h->jt.gemm_tiles; hd.n_wblocks = h->jt.n_wblocks; hd.special_block = h->jt.special_block; hd.special_job = h->jt.special_job;
        if (h->wgrad_opt) {
            wo.oa.optim_type = h->cfg.optim_type; wo.oa.only_receiver = 0; wo.oa.lr = h->cfg.learning_rate;
            wo.oa.from_wgrad = 1; wo.oa.bump_step = 1; wo.oa.bump_mb = h->game_step ? 1 : 0;
            for (int a = 0; a < 5; ++a) wo.oa.agent_begin[a] = h->pl.agent_begin[a];
            wo.oa.total = h->pl.total;
            wo.params = h->params; wo.state = h->opt_state; wo.grads = h->grads; wo.gnll = h->tp.gnll; wo.coefll = h->tp.coefll;
            wo.counter = h->tp.counter; wo.err_host = h->d_err;
            hipLaunchKernelGGL(k_wgrad<true>, dim3(h->jt.n_wblocks + 1 + 4), dim3(MMG_BLOCK), 0, st,     // tiles + column blocks | spare / closing block | four norm roles
                               (const JobTable*)h->d_jt, d_x, d_desc, h->tp.gnpart, h->dm, (const double*)h->tp.stats,
                               h->tp.losses, h->tp.totals, (const int*)(row_map ? h->tp.rmap : nullptr),
                               (const int*)(row_map ? h->tp.rcount : nullptr), h->tp.wpart, reinterpret_cast<uint32_t*>(h->tp.wcnt), (const uint32_t*)h->tp.sync, h->grads + h->pl.total, wo, 0, hd
#ifdef MMG_TIMING
                               , h->tp.dbg2
#endif
                               );
        } else
        hipLaunchKernelGGL(k_wgrad<false>, dim3(h->wgrad_stride > 0 ? h->wgrad_stride + h->jt.n_wblocks + 1 - h->jt.gemm_tiles : h->jt.n_wblocks + 1), dim3(MMG_BLOCK), 0, st,
                           (const JobTable*)h->d_jt, d_x, d_desc, h->tp.gnpart, h->dm, (const double*)h->tp.stats,
                           h->tp.losses, h->tp.totals, (const int*)(row_map ? h->tp.rmap : nullptr),
                           (const int*)(row_map ? h->tp.rcount : nullptr), h->tp.wpart, reinterpret_cast<uint32_t*>(h->tp.wcnt), (const uint32_t*)h->tp.sync, h->grads + h->pl.total, wo, h->wgrad_stride, hd
#ifdef MMG_TIMING
                           , h->tp.dbg2
#endif
                           );
        if (launch_check("k_wgrad")) return -1;
    }
    return 0;
}

extern "C" int mmg_backward(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc, void* stream) {
    if (!h) return fail("NULL handle");
    if (!d_x || !d_target || !d_desc) return fail("x / target / desc must not be NULL");
    if (h->bas_pending) return fail("mmg_loss_stats must run between mmg_exchange_forward(train) and mmg_backward (it carries the baselines' forward pass)");
    // continuous messages: the statistics are this rank's sum of rewards and hit count only, nothing a gradient depends on
    // (model.py:1297-1305) -- the call forms them itself (as a workgroup of the backward launch where the path has one, else as
    // k_stats) and no mmg_loss_stats / statistics all-reduce is needed; they reach the other ranks in the gradient tail
    bool own_stats = false;
    if (!h->dm.use_binary) {
        own_stats = mc_bwd(h) && h->merge_roles;
        if (!own_stats && mmg_loss_stats(h, stream)) return -1;
    }
    return backward_impl(h, d_x, d_target, d_desc, (hipStream_t)stream, own_stats);
}

static int clip_step_impl(mmg_handle* h, hipStream_t st, bool from_wgrad) {
    // from_wgrad: use the squared-norm partials k_wgrad left behind (valid only if d_grads has not been
    // modified since mmg_backward, i.e. single GPU); otherwise recompute them from d_grads.
    float* part = h->tp.gnpart + (from_wgrad ? 0 : MMG_MAX_WBLOCKS);
    if (!from_wgrad) {
        Scope sc(h, st, "k_gradnorm");
        hipLaunchKernelGGL(k_gradnorm, dim3(MMG_GN_BLOCKS), dim3(MMG_BLOCK), 0, st, (const JobTable*)h->d_jt,
                           (const float*)h->grads, part, h->tp.counter, (const float*)(h->grads + h->pl.total), h->tp.losses,
                           h->tp.totals, h->dm.use_binary ? 0 : h->dm.Bg);
        if (launch_check("k_gradnorm")) return -1;
    }
    OptArgs oa;
    oa.optim_type = h->cfg.optim_type; oa.only_receiver = h->cfg.use_binary ? 0 : 1; oa.lr = h->cfg.learning_rate;
    oa.from_wgrad = from_wgrad ? 1 : 0; oa.bump_step = from_wgrad ? 1 : 0; oa.bump_mb = h->game_step ? 1 : 0;
    for (int a = 0; a < 5; ++a) oa.agent_begin[a] = h->pl.agent_begin[a];
    oa.total = h->pl.total;
    int blocks = (int)((oa.total / 4 + MMG_BLOCK - 1) / MMG_BLOCK);
    if (blocks > 1024) blocks = 1024;
    {
        Scope sc(h, st, "k_opt");
        hipLaunchKernelGGL(k_opt, dim3(blocks), dim3(MMG_BLOCK), 0, st, (const JobTable*)h->d_jt, oa, h->params,
                           (const float*)h->grads, h->opt_state, (const float*)part, (const uint32_t*)h->tp.counter, (const uint32_t*)h->tp.sync, h->d_err,
                           (const float*)(from_wgrad ? nullptr : h->grads + h->pl.total), h->tp.losses);
        if (launch_check("k_opt")) return -1;
    }
    return 0;
}

extern "C" int mmg_clip_step(mmg_handle* h, void* stream) {
    if (!h) return fail("NULL handle");
    return clip_step_impl(h, (hipStream_t)stream, false);
}

static int train_step_impl(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc,
                           const float* d_u_z, const float* d_u_s, const float* d_u_w, uint64_t seed, void* stream) {
    if (h->game_ok) {
        // the small Adaptive agents: conversation, baselines, statistics and the reverse pass in ONE launch (kernels_game.h), then
        // k_wgrad and k_opt -- three launches per minibatch
        hipStream_t st = (hipStream_t)stream;
        const Dims& d = h->dm;
        if (!d_x || !d_desc) return fail("x / desc must not be NULL");
        if ((d_u_z || d_u_s || d_u_w) && !(d_u_z && d_u_s && d_u_w)) return fail("injected uniforms: all three streams or none");
        ConvArgs ar;
        memset(&ar, 0, sizeof(ar));
        ar.x = d_x; ar.target = d_target; ar.desc = d_desc; ar.u_z = d_u_z; ar.u_s = d_u_s; ar.u_w = d_u_w; ar.seed = seed;
        ar.train = 1; ar.run_all = 0; ar.t_begin = 0; ar.t_end = d.T; ar.phases = 3; ar.sprod_first = 1;
        ar.nprep = prep_blocks(d, h->prep_cpb, true); ar.prep_cpb = h->prep_cpb; ar.nbase = ((d.B + 15) / 16) * ((d.K + 15) / 16);
        GameArgs ga; ga.n_stats = (5 * d.T + 2 + 3) / 4; ga.n_bas = h->game_nbas; ga.bas_ub = h->game_bas_ub;
        h->basehx_ready = true; h->bas_deferred = false; h->bas_pending = false; h->scores_in_parts = true;
        {
            Scope sc(h, st, "k_game");
            const int grid = d.B + ar.nprep + ar.nbase + ga.n_stats + ga.n_bas + d.D;
            if (d.D == 30) hipLaunchKernelGGL(k_game_fast<30>, dim3(grid), dim3(256), game_lds_bytes(), st, h->dm, h->P, h->tp, ar, ga);
            else hipLaunchKernelGGL(k_game_fast<32>, dim3(grid), dim3(256), game_lds_bytes(), st, h->dm, h->P, h->tp, ar, ga);
            if (launch_check("k_game_fast")) return -1;
        }
        h->game_step = true;
        h->wgrad_opt = h->wgrad_opt_ok;
        int rc = backward_impl(h, d_x, d_target, d_desc, st, true, true);
        if (!rc && !h->wgrad_opt) rc = clip_step_impl(h, st, true);
        h->game_step = false; h->wgrad_opt = false;
        return rc;
    }
    h->defer_bas = true;
    const int frc = exchange_forward_impl(h, d_x, d_target, d_desc, d_u_z, d_u_s, d_u_w, seed, 1, 2, stream);
    h->defer_bas = false;
    if (frc) return -1;
    const bool merged = merge_stats(h);
    if (h->bas_deferred && !merged) return fail("internal: deferred baselines without the merged backward launch");
    if (!merged && mmg_loss_stats(h, stream)) return -1;
    h->wgrad_opt = h->wgrad_opt_ok;
    const int brc = backward_impl(h, d_x, d_target, d_desc, (hipStream_t)stream, merged);
    const bool opt_done = h->wgrad_opt;
    h->wgrad_opt = false;
    if (brc) return -1;
    return opt_done ? 0 : clip_step_impl(h, (hipStream_t)stream, true);
}

extern "C" int mmg_train_step(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc,
                              const float* d_u_z, const float* d_u_s, const float* d_u_w, uint64_t seed, void* stream) {
    if (!h) return fail("NULL handle");
    if (h->cfg.global_batch != h->cfg.batch) return fail("mmg_train_step is single-GPU; with several ranks: mmg_dp_train_step, or all-reduce between the phases");
    if (!d_target) return fail("target must not be NULL");
    const int warn = error_gate(h, (hipStream_t)stream, true);
    if (warn < 0) return -1;
    const int rc = train_step_impl(h, d_x, d_target, d_desc, d_u_z, d_u_s, d_u_w, seed, stream);
    return rc ? rc : warn;
}

// n consecutive minibatches of the epoch loop (model.py:1218-1240) enqueued from C: minibatch i reads rows [i * B, (i + 1) * B)
// of d_x [n * B, F] / d_target [n * B] -- the batch-ordered gather of the epoch (misc.py:257-302) the caller laid out once.  The
// sampling streams advance with the device-side minibatch counter exactly as under n mmg_train_step calls.
extern "C" int mmg_train_steps(mmg_handle* h, const float* d_x, const int64_t* d_target, int64_t n, const float* d_desc,
                               uint64_t seed, void* stream) {
    if (!h) return fail("NULL handle");
    if (h->cfg.global_batch != h->cfg.batch) return fail("mmg_train_steps is single-GPU; with several ranks: mmg_dp_train_steps");
    if (!d_x || !d_target || !d_desc || n < 0) return fail("x / target / desc must not be NULL, n >= 0");
    int warn = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int w = error_gate(h, (hipStream_t)stream, true);
        if (w < 0) return -1;
        warn |= w;
        if (train_step_impl(h, d_x + (size_t)i * h->dm.B * h->dm.F, d_target + (size_t)i * h->dm.B, d_desc, nullptr, nullptr, nullptr, seed, stream)) return -1;
    }
    return warn;
}

// ---------------------------------------------------------------------------------------------
// The data-parallel minibatch (SURVEY 8e option A; couplings model.py:912-915, 947-961, 1310) in ONE call: forward + batch
// statistics | all-reduce of the f64 statistics (binary messages only) | backward | ONE all-reduce of the flat gradient buffer
// incl. its tail quad | norm of the REDUCED gradient + optimizer.  The collectives are RCCL's ncclAllReduce, called through the
// address the caller hands over (the library does not link RCCL) on the caller's communicator and on THIS stream; `reduce` = 0
// skips them (one rank).  What python did in four ctypes calls + two per step (50 us of host time against 77 us of device time).
// ---------------------------------------------------------------------------------------------
typedef int (*mmg_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
extern "C" int mmg_dp_set_allreduce(mmg_handle* h, void* nccl_all_reduce, void* comm) {
    if (!h) return fail("NULL handle");
    h->ar_fn = nccl_all_reduce; h->ar_comm = comm;
    return 0;
}
static int dp_step_impl(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc,
                        const float* d_u_z, const float* d_u_s, const float* d_u_w, uint64_t seed, int full_tape, int reduce, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    mmg_allreduce_fn ar = (mmg_allreduce_fn)h->ar_fn;
    if (reduce && !ar) return fail("mmg_dp_train_step: no collective set (mmg_dp_set_allreduce)");
    if (exchange_forward_impl(h, d_x, d_target, d_desc, d_u_z, d_u_s, d_u_w, seed, 1, full_tape ? 3 : 2, stream)) return -1;
    if (h->dm.use_binary) {
        if (mmg_loss_stats(h, stream)) return -1;
        if (reduce) {
            const int rc = ar(h->tp.stats, h->tp.stats, (size_t)stat_count(h->dm.T), 8 /* ncclFloat64 */, 0 /* ncclSum */, h->ar_comm, st);
            if (rc) return fail("ncclAllReduce (statistics) failed: %d", rc);
        }
    }
    if (mmg_backward(h, d_x, d_target, d_desc, stream)) return -1;
    if (reduce) {
        const int rc = ar(h->grads, h->grads, (size_t)(h->pl.total + MMG_GRAD_TAIL), 7 /* ncclFloat32 */, 0, h->ar_comm, st);
        if (rc) return fail("ncclAllReduce (gradients) failed: %d", rc);
    }
    return clip_step_impl(h, st, false);
}
extern "C" int mmg_dp_train_step(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc,
                                 const float* d_u_z, const float* d_u_s, const float* d_u_w, uint64_t seed, int full_tape, int reduce, void* stream) {
    if (!h) return fail("NULL handle");
    if (!d_x || !d_target || !d_desc) return fail("x / target / desc must not be NULL");
    const int warn = error_gate(h, (hipStream_t)stream, true);
    if (warn < 0) return -1;
    const int rc = dp_step_impl(h, d_x, d_target, d_desc, d_u_z, d_u_s, d_u_w, seed, full_tape, reduce, stream);
    return rc ? rc : warn;
}
extern "C" int mmg_dp_train_steps(mmg_handle* h, const float* d_x, const int64_t* d_target, int64_t n, const float* d_desc,
                                  uint64_t seed, int reduce, void* stream) {
    if (!h) return fail("NULL handle");
    if (!d_x || !d_target || !d_desc || n < 0) return fail("x / target / desc must not be NULL, n >= 0");
    int warn = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int w = error_gate(h, (hipStream_t)stream, true);
        if (w < 0) return -1;
        warn |= w;
        if (dp_step_impl(h, d_x + (size_t)i * h->dm.B * h->dm.F, d_target + (size_t)i * h->dm.B, d_desc, nullptr, nullptr, nullptr, seed, 0, reduce, stream)) return -1;
    }
    return warn;
}

// ---------------------------------------------------------------------------------------------
// Host helper of the epoch loop (misc.py:270-271 `random.seed(11 + epoch); random.shuffle(order)`): Fisher-Yates exactly as
// CPython's random.shuffle draws it -- j = _randbelow(i + 1) for i = n-1 .. 1 with _randbelow(m) = rejection sampling of
// getrandbits(bit_length(m)) = genrand_uint32() >> (32 - k) -- continuing the Mersenne-Twister state the caller read with
// random.getstate().  The Python loop costs ~0.4 us per sample (1.2 ms per 3000-sample epoch, a third of the epoch's device
// time at config 2); this is ~10 ns per sample.  The caller verifies it against random.shuffle once per process.
// ---------------------------------------------------------------------------------------------
extern "C" int mmg_host_shuffle(const uint32_t* mt_state, int pos, int64_t n, int64_t* perm) {
    if (!mt_state || !perm || n < 0 || pos < 0 || pos > 624) return fail("mmg_host_shuffle: bad arguments");
    if (n > 0x7fffffffLL) return fail("mmg_host_shuffle: more than 2^31 - 1 elements");
    uint32_t mt[624];
    memcpy(mt, mt_state, sizeof(mt));
    int mti = pos;
    auto next = [&]() -> uint32_t {
        if (mti >= 624) {
            const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
            int kk = 0;
            for (; kk < 624 - 397; ++kk) { const uint32_t y = (mt[kk] & UP) | (mt[kk + 1] & LO); mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
            for (; kk < 623; ++kk) { const uint32_t y = (mt[kk] & UP) | (mt[kk + 1] & LO); mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u); }
            const uint32_t y = (mt[623] & UP) | (mt[0] & LO);
            mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
            mti = 0;
        }
        uint32_t y = mt[mti++];
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
        return y;
    };
    for (int64_t i = n - 1; i >= 1; --i) {
        const uint32_t m = (uint32_t)(i + 1);
        const int k = 32 - __builtin_clz(m);                           // bit_length(i + 1)
        uint32_t r = next() >> (32 - k);
        while (r >= m) r = next() >> (32 - k);
        const int64_t t = perm[i]; perm[i] = perm[r]; perm[r] = t;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// agent-level entry points (one exchange step, forward only)
// ---------------------------------------------------------------------------------------------
extern "C" int mmg_sender_forward(mmg_handle* h, const float* d_x, const float* d_w, int t, int train,
                                  const float* d_u_z, uint64_t seed, float* d_message, float* d_probs, float* d_h_x,
                                  void* stream) {
    if (!h) return fail("NULL handle");
    if (!d_x || !d_message) return fail("x / message must not be NULL");
    if (t < 0 || t >= h->dm.T) return fail("t out of range");
    if (t > 0 && !d_w) return fail("w must not be NULL for t > 0");
    hipStream_t st = (hipStream_t)stream;
    const Dims& d = h->dm;
    // k_prep needs a description matrix only for Cd (unused here); reuse the tape's zero-initialised Cd as a
    // dummy source so that hw0 / dsig are refreshed from the current parameters.
    {
        Scope sc(h, st, "k_prep(sender)");
        Dims d1 = h->dm; d1.D = 0;
        hipLaunchKernelGGL(k_prep, dim3((d1.H + 63) / 64), dim3(MMG_BLOCK), h->prep_smem, st, d1, h->P, h->tp, (const float*)nullptr,
                           (const float*)nullptr, 1, train ? 1 : 0);
        if (launch_check("k_prep")) return -1;
    }
    if (launch_gemm_nt(h, st, "k_gemm_nt(h_x)", d_x, d.F, h->P.p[S_IMG_W], d.F, h->P.p[S_IMG_B], h->tp.hx, d.H, d.B, d.H, d.F)) return -1;
    ConvArgs ar;
    memset(&ar, 0, sizeof(ar));
    ar.x = d_x; ar.u_z = d_u_z ? d_u_z - (size_t)t * d.B * d.W : nullptr; ar.seed = seed; ar.train = train; ar.run_all = 1;
    ar.t_begin = t; ar.t_end = t + 1; ar.phases = 1; ar.w_in = d_w;
    hipLaunchKernelGGL(k_conversation<256>, dim3(d.B), dim3(MMG_BLOCK), h->conv_smem_agent, st, h->dm, h->P, h->tp, ar);
    if (launch_check("k_conversation(sender)")) return -1;
    const size_t off = (size_t)t * d.B * d.W;
    HIP_OK(hipMemcpyAsync(d_message, h->tp.z + off, sizeof(float) * d.B * d.W, hipMemcpyDeviceToDevice, st));
    if (d_probs && d.use_binary) HIP_OK(hipMemcpyAsync(d_probs, h->tp.pz + off, sizeof(float) * d.B * d.W, hipMemcpyDeviceToDevice, st));
    if (d_h_x) HIP_OK(hipMemcpyAsync(d_h_x, h->tp.hx, sizeof(float) * d.B * d.H, hipMemcpyDeviceToDevice, st));
    return 0;
}

extern "C" int mmg_receiver_forward(mmg_handle* h, const float* d_z, const float* d_desc, float* d_h_z,
                                    float* d_s_prob_prod, int first, int t, int train,
                                    const float* d_u_s, const float* d_u_w, uint64_t seed,
                                    float* d_s, float* d_s_prob, float* d_w, float* d_w_probs, float* d_y,
                                    float* d_h_w, void* stream) {
    if (!h) return fail("NULL handle");
    if (!d_z || !d_desc || !d_h_z) return fail("z / desc / h_z must not be NULL");
    if (t < 0 || t >= h->dm.T) return fail("t out of range");
    hipStream_t st = (hipStream_t)stream;
    const Dims& d = h->dm;
    const size_t offW = (size_t)t * d.B * d.W, offB = (size_t)t * d.B;
    HIP_OK(hipMemcpyAsync(h->tp.z + offW, d_z, sizeof(float) * d.B * d.W, hipMemcpyDeviceToDevice, st));
    if (launch_prep(h, st, d_desc, nullptr, train ? 1 : 0)) return -1;
    ConvArgs ar;
    memset(&ar, 0, sizeof(ar));
    ar.desc = d_desc; ar.seed = seed; ar.train = train; ar.run_all = 1;
    ar.u_s = d_u_s ? d_u_s - offB : nullptr; ar.u_w = d_u_w ? d_u_w - offW : nullptr;
    ar.t_begin = t; ar.t_end = t + 1; ar.phases = 2; ar.h_state = d_h_z; ar.sprod_state = d_s_prob_prod; ar.sprod_first = first;
    hipLaunchKernelGGL(k_conversation<256>, dim3(d.B), dim3(MMG_BLOCK), h->conv_smem_agent, st, h->dm, h->P, h->tp, ar);
    if (launch_check("k_conversation(receiver)")) return -1;
    if (d_s) HIP_OK(hipMemcpyAsync(d_s, h->tp.s + offB, sizeof(float) * d.B, hipMemcpyDeviceToDevice, st));
    if (d_s_prob) HIP_OK(hipMemcpyAsync(d_s_prob, h->tp.ps + offB, sizeof(float) * d.B, hipMemcpyDeviceToDevice, st));
    if (d_w) HIP_OK(hipMemcpyAsync(d_w, h->tp.w + offW, sizeof(float) * d.B * d.W, hipMemcpyDeviceToDevice, st));
    if (d_w_probs && d.use_binary) HIP_OK(hipMemcpyAsync(d_w_probs, h->tp.pw + offW, sizeof(float) * d.B * d.W, hipMemcpyDeviceToDevice, st));
    if (d_y) HIP_OK(hipMemcpyAsync(d_y, h->tp.y + (size_t)t * d.B * d.D, sizeof(float) * d.B * d.D, hipMemcpyDeviceToDevice, st));
    if (d_h_w) HIP_OK(hipMemcpyAsync(d_h_w, h->tp.g + (size_t)t * d.B * d.R, sizeof(float) * d.B * d.R, hipMemcpyDeviceToDevice, st));
    return 0;
}

extern "C" int mmg_baseline_forward(mmg_handle* h, int which, const float* d_x, const float* d_binary,
                                    const float* d_inp, int rows, float* d_score, void* stream) {
    if (!h) return fail("NULL handle");
    if (!d_binary || !d_score || rows <= 0) return fail("binary / score must not be NULL, rows > 0");
    hipStream_t st = (hipStream_t)stream;
    const Dims& d = h->dm; const Params& P = h->P;
    BasArgs rec, sen;
    memset(&rec, 0, sizeof(rec)); memset(&sen, 0, sizeof(sen));
    if (which == MMG_AGENT_BASELINE_REC) {
        if (!d_inp) return fail("baseline_rec needs inp (receiver hidden state)");
        rec.rows = rows; rec.x1 = d_binary; rec.ld1 = d.W; rec.k1 = d.W; rec.x2 = d_inp; rec.ld2 = d.R; rec.k2 = d.R;
        rec.W1 = P.p[BR_L1_W]; rec.ldw = d.W + d.R; rec.b1 = P.p[BR_L1_B]; rec.W2 = P.p[BR_L2_W]; rec.b2 = P.p[BR_L2_B];
        rec.score = d_score;
    } else if (which == MMG_AGENT_BASELINE_SEN) {
        if (!d_x) return fail("baseline_sen needs x (sender.h_x)");
        sen.rows = rows; sen.x1 = d_x; sen.ld1 = d.H; sen.k1 = d.H; sen.x2 = d_binary; sen.ld2 = d.W; sen.k2 = d.W;
        sen.W1 = P.p[BS_L1_W]; sen.ldw = d.H + d.W; sen.b1 = P.p[BS_L1_B]; sen.W2 = P.p[BS_L2_W]; sen.b2 = P.p[BS_L2_B];
        sen.score = d_score;
    } else {
        return fail("which must be MMG_AGENT_BASELINE_REC or MMG_AGENT_BASELINE_SEN");
    }
    hipLaunchKernelGGL(k_baselines, dim3((rows + 15) / 16, 2), dim3(MMG_BLOCK), 0, st, d.K, rec, sen);
    return launch_check("k_baselines(agent)");
}
