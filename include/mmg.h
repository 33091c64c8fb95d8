/*
 * mmg.h -- C-ABI of the MI355X-native REINFORCE exchange path (libmmg.so).
 *
 * The reference (nyu-dl/MultimodalGame) has no FFI: its hot path is reached through the
 * Python objects of model.py.  This header is the boundary a maintainer would bind (ctypes,
 * see INTEGRATION.md) to replace exactly these reference call sites:
 *
 *   mmg_exchange_forward   <- exchange()                    model.py:725-876  (called at 1240, 640)
 *                             + output selection / NLL      model.py:879-904, 1264-1275
 *   mmg_loss_stats         <- mask derivation + the batch statistics of
 *                             multistep_loss_binary/_bas    model.py:1248-1262, 907-988
 *   mmg_backward           <- the four loss.backward()      model.py:1309, 1316, 1322, 1328
 *   mmg_clip_step          <- clip_grad_norm + optimizer    model.py:1310-1311, 1317-1318, 1323-1324, 1329-1330
 *   mmg_train_step         <- the whole per-minibatch block model.py:1240-1339 (single GPU)
 *   mmg_train_steps        <- n iterations of the epoch loop's body, model.py:1218-1240 (batches of misc.py:257-302)
 *   mmg_dp_train_step[s]   <- the same block on one rank of a data-parallel job (statistics + gradient
 *                             all-reduce; couplings model.py:912-915, 947-961, 1310)
 *   mmg_sender_forward     <- Sender.forward                model.py:144-238 (non-attention, sender_mix=sum)
 *   mmg_receiver_forward   <- Receiver.forward              model.py:303-477 (non-desc_attn)
 *   mmg_baseline_forward   <- Baseline.forward              model.py:496-516
 *
 * Conventions
 *   - plain pointers and sizes; every pointer named d_* is DEVICE memory owned by the caller
 *     (torch tensors' data_ptr()).  The library never allocates device memory: the caller hands
 *     it one workspace of mmg_workspace_bytes() bytes at mmg_create().
 *   - all work is enqueued on the hipStream_t passed as `stream` (void*, 0 = null stream) and is
 *     asynchronous w.r.t. the host; no host synchronisation happens inside any call.
 *   - return value: 0 = ok, negative = error (message via mmg_last_error()), 1 = ok with a warning in
 *     mmg_last_error() (training entry points only: the library recovered from a timed-out in-launch
 *     dependency, see mmg_clear_error); nothing throws across the ABI.  One handle per device per process; not thread-safe (the reference is
 *     single-threaded).
 *   - all floating-point data is fp32, row-major, PyTorch [out,in] weight layout; masks are
 *     uint8; targets int64 (as the reference's LongTensor).
 */
#ifndef MMG_H_
#define MMG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMG_VERSION 3
#define MMG_GRAD_TAIL 4       /* floats behind the gradients in d_grads, owned by the library (see mmg_grad_floats) */

enum { MMG_OPT_RMSPROP = 0, MMG_OPT_ADAM = 1, MMG_OPT_SGD = 2 };          /* model.py:1725 */
enum { MMG_AGENT_RECEIVER = 0, MMG_AGENT_SENDER = 1, MMG_AGENT_BASELINE_REC = 2, MMG_AGENT_BASELINE_SEN = 3 };

/* Frozen copy of the gflags the hot path reads (model.py:1639-1741). */
typedef struct mmg_config {
    int32_t batch;            /* B: samples handled by THIS process (-batch_size / world_size)      */
    int32_t global_batch;     /* samples of the whole minibatch over all ranks (== batch if 1 GPU)   */
    int32_t batch_offset;     /* index of this rank's first sample inside the global minibatch       */
    int32_t n_classes;        /* D: rows of the description matrix                                   */
    int32_t feat_dim;         /* F: -img_feat_dim                                                    */
    int32_t h_dim;            /* H: -img_h_dim                                                       */
    int32_t w_dim;            /* W: -rec_w_dim == -sender_out_dim (model.py:1756)                    */
    int32_t rec_hidden;       /* R: -rec_hidden                                                      */
    int32_t wv_dim;           /* V: -wv_dim                                                          */
    int32_t bas_hidden;       /* K: -baseline_hid_dim                                                */
    int32_t max_exchange;     /* T: -max_exchange                                                    */
    int32_t use_binary;       /* -use_binary                                                         */
    int32_t fixed_exchange;   /* -fixed_exchange                                                     */
    int32_t s_prob_prod;      /* -s_prob_prod (eval-mode stop bit, model.py:423-427)                 */
    int32_t has_entropy_s, has_entropy_sen, has_entropy_rec;   /* flag is not None (model.py:925)    */
    float   entropy_s, entropy_sen, entropy_rec;               /* model.py:1730-1732                 */
    float   first_rec;        /* -first_rec (model.py:786)                                           */
    int32_t optim_type;       /* MMG_OPT_*                                                           */
    float   learning_rate;    /* -learning_rate                                                      */
    int32_t top_k;            /* -top_k_train                                                        */
    int32_t cu_budget;        /* compute units this process can count on (0 = the whole device).  The persistent
                                 "role" launches need their workgroups co-resident; their budgets are sized from this
                                 number, and a budget they do not fit selects launches without in-launch waits up front.
                                 Set it when the GPU is shared or the process runs under a CU mask (no reference
                                 counterpart: additive)                                                              */
} mmg_config;

/* One parameter tensor inside the flat parameter / gradient / optimizer-state buffers. */
typedef struct mmg_param_entry {
    char    name[48];         /* state_dict key, e.g. "rnn.weight_ih" (SURVEY.md §8 b)               */
    int32_t agent;            /* MMG_AGENT_*                                                         */
    int32_t rows, cols;       /* cols == 0 for 1-D tensors                                           */
    int64_t offset;           /* in floats from the start of the flat buffer (16-byte aligned)       */
} mmg_param_entry;

/* One named array inside the workspace ("tape"): everything exchange() returns plus what the
 * backward pass re-reads. */
typedef struct mmg_tape_entry {
    char    name[32];
    int32_t dtype;            /* 0 = f32, 1 = u8, 2 = i32, 3 = f64                                   */
    int32_t ndim;
    int64_t dims[4];
    int64_t offset;           /* bytes from the start of the workspace                               */
} mmg_tape_entry;

typedef struct mmg_handle mmg_handle;

const char* mmg_last_error(void);
int     mmg_version(void);

/* Layout queries (host only, no GPU needed). */
int64_t mmg_param_count(const mmg_config* cfg);                                   /* floats incl. padding */
/* Floats the caller allocates for d_grads: mmg_param_count() + MMG_GRAD_TAIL.  The tail quad is written by mmg_backward:
 * [0] = 1.0 if an in-launch dependency wait of this minibatch timed out on this rank (stale gradients), else 0.0.  A
 * data-parallel caller all-reduces (sum) ALL mmg_grad_floats() floats, so the flag reaches every rank with the gradients
 * and mmg_clip_step skips the update on all ranks alike (the reference has no counterpart: model.py:1307-1330 is
 * single-process); [1], [2] = this rank's sum of rewards (log-likelihood of the target, model.py:1274) and top-k hit count,
 * from which mmg_clip_step rewrites the logged NLL / hits of the GLOBAL minibatch in continuous mode (use_binary == 0), where
 * the shards couple through nothing else and no statistics all-reduce is needed; [3] = 0. */
int64_t mmg_grad_floats(const mmg_config* cfg);
int     mmg_param_table(const mmg_config* cfg, mmg_param_entry* out, int max_entries);   /* returns count */
int64_t mmg_workspace_bytes(const mmg_config* cfg);
int     mmg_tape_table(const mmg_config* cfg, mmg_tape_entry* out, int max_entries);     /* returns count */

/* d_params: flat fp32 buffer of mmg_param_count() floats; d_grads: mmg_grad_floats() floats.  d_opt_state: 2x mmg_param_count()
 * (RMSprop square_avg | unused; Adam exp_avg | exp_avg_sq; SGD unused), zero-initialised by the
 * caller.  d_workspace: mmg_workspace_bytes() bytes, 256-byte aligned. */
mmg_handle* mmg_create(const mmg_config* cfg, void* d_workspace, int64_t workspace_bytes,
                       float* d_params, float* d_grads, float* d_opt_state);
void    mmg_destroy(mmg_handle* h);

/* Runs the T-step conversation of one minibatch (model.py:725-876) and the output selection /
 * log-softmax / NLL reward (model.py:879-904, 1264-1275).
 *   d_x [B,F] f32, d_target [B] i64 (may be NULL when train==0 and no accuracy is wanted),
 *   d_desc [D,V] f32.
 *   Sampling (train==1): if d_u_z/d_u_s/d_u_w ([T,B,W],[T,B],[T,B,W] f32 uniforms, the reference's
 *   np.random.rand draws in its call order) are non-NULL they are consumed; otherwise the in-kernel
 *   Philox4x32-10 generator keyed by (seed, the handle's device-side minibatch counter) is used.
 *   train==0: messages are round(p), stop bit round(prod p_s) (model.py:229, 423-427, 462).
 *   run_all_steps==1: every sample runs all T steps (what exchange() returns to Python);
 *   ==0: a sample stops computing once its own conversation has ended (its later steps are masked
 *   out of every loss, so training results are identical); per-(step, sample) arrays -- messages,
 *   baseline scores and hiddens, gradient tapes -- are then defined on the LIVE rows only
 *   (t <= the sample's own last step); the others keep whatever an earlier call left there.
 *   ==2: as 0, and only what the training step itself reads is guaranteed to be stored: in Fixed mode the class logits "y"
 *   are kept for the output step (T-1) only -- the losses read nothing else of them (model.py:885-904, 1264-1275) -- and in
 *   continuous mode (only the receiver is trained, model.py:1313) the sender-side and message arrays (a, c, zr, dbar, g, w)
 *   may be left unwritten.  This is what mmg_train_step runs.
 *   ==3 (train == 1): every sample runs all T steps, as with 1 -- messages, probabilities, stop bits, masks and class logits
 *   of every (step, sample) are on the tape -- but the baselines are the training step's: their scores "bs" / "br" are defined
 *   on the live rows only.  For the minibatches whose log block prints the whole conversation (model.py:1342-1518), which
 *   prints no baseline score.
 * Results land in the workspace arrays listed by mmg_tape_table(). */
int mmg_exchange_forward(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc,
                         const float* d_u_z, const float* d_u_s, const float* d_u_w, uint64_t seed,
                         int train, int run_all_steps, void* stream);

/* Per-rank partial sums of every batch statistic the losses need (counts, sums and squared sums
 * of reward-minus-baseline per stream and step, ...) into the f64 tape array "stats".  With more
 * than one rank the caller all-reduces (sum) that array between this call and mmg_backward
 * (model.py:912-915, 947-961: the REINFORCE weights are normalised by statistics of the WHOLE minibatch).
 * Continuous mode (use_binary == 0; model.py:1297-1305: loss = NLL mean) has no such coupling: the call and the
 * all-reduce may be skipped, mmg_backward forms the two logged sums itself (see mmg_grad_floats).
 * Binary mode: the call is REQUIRED between mmg_exchange_forward(train, run_all_steps != 1) and mmg_backward -- on the
 * register-resident path it also carries the baselines' forward pass over the live rows (model.py:835-843; one launch for the
 * baselines and the statistics that consume their scores); mmg_backward fails if it was skipped. */
int mmg_loss_stats(mmg_handle* h, void* stream);

/* Gradient of the four losses (model.py:1296-1305) w.r.t. all parameters into d_grads (the whole
 * flat buffer is overwritten).  Also writes the six loss scalars into the tape array "losses". */
int mmg_backward(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc, void* stream);

/* Per-agent clip_grad_norm(max_norm=1) + optimizer update on the flat buffers.  In continuous mode
 * (use_binary==0) only the receiver is updated (model.py:1313).  Skips the update when the dependency-error flag of this
 * rank or -- through the tail quad of d_grads -- of any rank is set; the next call that starts a minibatch recovers
 * (mmg_clear_error).
 * Non-finite guard: when the gradient norm of ANY agent is not finite, tape "losses"[0] (the NLL) is set to NaN in the same
 * step.  (The class-logit ReLU is v_max_f32, which reads a NaN pre-activation as "unit off" where torch's relu propagates
 * it: without the guard a NaN in receiver.y1.weight[:, :R] or in the GRU leaves a plausible NLL = log D while every loss of
 * the reference is NaN.  The reference's NLL is non-finite => this one is, in the same step; it is finite whenever all six
 * losses of the reference are.) */
int mmg_clip_step(mmg_handle* h, void* stream);

/* forward(train, run_all_steps = 2) + stats + backward + clip_step for a single-GPU minibatch; nothing returns to the
 * host.  Equivalent to the four calls above in sequence.  Launches per minibatch depend on the shape: 2 for the agents of
 * BASELINE configs 1-2 (k_game_fast: conversation, k_prep's blocks, baselines, statistics and the reverse pass as roles of
 * one launch; k_wgrad<OPT>: weight gradients + clip + optimizer), 4 for config 3's shard (k_conversation_fast3, k_baselines3,
 * k_bwd_conv_fast, k_wgrad<OPT>; + k_prep with more samples than CUs, + k_opt with row splits), 6 for config 5's
 * shard (k_prep, k_conversation_mc, k_bwd_mc1, k_bwd_mc2, k_wgrad, k_opt), 9 for config 4 (k_prep,
 * k_conv_persist, k_baselines4, k_stats, k_bwd_pre_send, k_bwd_sample, k_dC_tile, k_wgrad, k_opt), 10 for
 * config 4 with rec_hidden 256 (k_prep, k_rc_persist, k_gemm_nt, k_baselines4, k_stats, k_bwd_pre_send, k_rc_bwd, k_dC_tile,
 * k_wgrad, k_opt). */
int mmg_train_step(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc,
                   const float* d_u_z, const float* d_u_s, const float* d_u_w, uint64_t seed, void* stream);

/* n consecutive minibatches of the epoch loop (model.py:1218-1240) enqueued by ONE call: minibatch i reads rows
 * [i * B, (i + 1) * B) of d_x [n * B, F] / d_target [n * B] -- the epoch's samples laid out in the reference's batch order
 * (misc.py:257-302: random.seed(11 + epoch) shuffle, consecutive slices, indices sorted inside a batch), which the caller
 * gathers once per epoch.  Philox sampling only (the device-side minibatch counter advances as under n mmg_train_step
 * calls: bit-identical results).  The caller runs the minibatches that write a log block / evaluate / checkpoint itself. */
int mmg_train_steps(mmg_handle* h, const float* d_x, const int64_t* d_target, int64_t n, const float* d_desc,
                    uint64_t seed, void* stream);

/* Data-parallel minibatch in ONE call (one rank; batch < global_batch): forward + mmg_loss_stats | all-reduce(sum) of the
 * f64 "stats" array (binary messages only) | mmg_backward | ONE all-reduce of all mmg_grad_floats() floats of d_grads |
 * mmg_clip_step on the reduced gradient.  The collectives are RCCL's
 *     ncclAllReduce(sendbuff, recvbuff, count, datatype, op, comm, stream)
 * called through the ADDRESS the caller registers (libmmg does not link RCCL) on the caller's communicator, in place, on
 * `stream`.  reduce == 0 skips both (a one-rank job).  full_tape != 0: every sample runs all steps (run_all_steps = 1: the
 * minibatches whose log block reads the whole tape), same update.  mmg_dp_train_steps: n minibatches laid out as for
 * mmg_train_steps (this rank's B rows of each). */
int mmg_dp_set_allreduce(mmg_handle* h, void* nccl_all_reduce, void* comm);
int mmg_dp_train_step(mmg_handle* h, const float* d_x, const int64_t* d_target, const float* d_desc,
                      const float* d_u_z, const float* d_u_s, const float* d_u_w, uint64_t seed,
                      int full_tape, int reduce, void* stream);
int mmg_dp_train_steps(mmg_handle* h, const float* d_x, const int64_t* d_target, int64_t n, const float* d_desc,
                       uint64_t seed, int reduce, void* stream);

/* Fail-soft.  The persistent launches hold workgroup "roles" that wait on each other inside one launch; every wait is
 * bounded.  A wait that expires (fewer compute units than mmg_create assumed: shared or CU-masked GPU) makes that minibatch's
 * optimizer update a no-op on every rank, and the NEXT call that starts a minibatch (mmg_train_step[s],
 * mmg_exchange_forward(train), mmg_dp_train_step[s]) recovers by itself: it drains `stream`, clears the error words,
 * re-selects the kernels WITHOUT in-launch waits for the rest of the handle's life (what MMG_NO_ROLES=1, a CU mask in the
 * environment or a too-small cu_budget select up front), and returns 1 with the warning in mmg_last_error().  The reference
 * has no such failure mode (model.py:1218-1330 keeps training) -- neither has the caller of this library.
 *   mmg_clear_error: the same clearing on request, without changing the selected kernels (drains `stream`).
 *   mmg_degraded:    0 = role launches in use, 1 = launches without in-launch waits selected at mmg_create,
 *                    2 = ... selected by a recovery. */
int mmg_clear_error(mmg_handle* h, void* stream);
int mmg_degraded(const mmg_handle* h);

/* Agent-level entry points (forward only; one exchange step), mirroring the reference modules.
 *   sender:   x[B,F], w[B,W] (ignored when t==0), t -> message[B,W], probs[B,W] (NULL if continuous),
 *             h_x[B,H]                                                   model.py:193-238
 *   receiver: z[B,W], desc[D,V], h_z[B,R] in/out (zeros for a fresh conversation), s_prob_prod[B]
 *             in/out (eval; `first` != 0 restarts the product) -> s[B], s_prob[B], w[B,W],
 *             w_probs[B,W], y[B,D], h_w[B,R]                             model.py:333-342, 411-477
 *   baseline: which = MMG_AGENT_BASELINE_REC/SEN; x[B,x_dim] or NULL, binary[B,W], inp[B,R] or NULL
 *             -> score[B]                                                model.py:496-516
 * Uniform pointers as in mmg_exchange_forward (NULL + train==1 -> Philox with `seed`, step t). */
int mmg_sender_forward(mmg_handle* h, const float* d_x, const float* d_w, int t, int train,
                       const float* d_u_z, uint64_t seed,
                       float* d_message, float* d_probs, float* d_h_x, void* stream);
int mmg_receiver_forward(mmg_handle* h, const float* d_z, const float* d_desc, float* d_h_z,
                         float* d_s_prob_prod, int first, int t, int train,
                         const float* d_u_s, const float* d_u_w, uint64_t seed,
                         float* d_s, float* d_s_prob, float* d_w, float* d_w_probs, float* d_y,
                         float* d_h_w, void* stream);
int mmg_baseline_forward(mmg_handle* h, int which, const float* d_x, const float* d_binary,
                         const float* d_inp, int rows, float* d_score, void* stream);

/* The log block of a minibatch (model.py:1342-1461) gathered on the device: ONE launch writes one flat float64 vector the
 * caller copies to the host (asynchronously) and formats.  Layout (mmg_log_snapshot_count() doubles):
 *   with_losses != 0:  the tape's losses[8] | running top-k hit count totals[1] | the batch statistics "stats" |
 *                      ent[T]: mean over the WHOLE batch of sum_d softmax(y_t) log(softmax(y_t) + 1e-8) at every step ("Entropy
 *                      Receiver Predictions", model.py:880-886 -- meaningful on the tape of a run_all_steps = 1 minibatch) |
 *                      argmax[B] of the selected log-probabilities | target[B]                    ("Predictions", model.py:1379)
 *   dump = k > 0:      what the sample dump of the first k samples prints at every step (model.py:1411-1461): live samples
 *                      after each step [T] | sender probs, receiver probs, sender bits, receiver bits [T, k, W] each | stop
 *                      probabilities [T, k] | stop masks after each step [T, k]
 * Call it after the minibatch whose block is to be logged, on the same stream. */
int64_t mmg_log_snapshot_count(const mmg_config* cfg, int dump, int with_losses);
int mmg_log_snapshot(mmg_handle* h, const int64_t* d_target, int dump, int with_losses, double* d_out, void* stream);

/* Host-only helper of the epoch loop (no GPU work): shuffles perm[0..n) in place exactly as CPython's
 * `random.shuffle` would from the Mersenne-Twister state (624 words + position, `random.getstate()[1]`) -- the batch order of
 * misc.py:270-271 (`random.seed(11 + epoch); random.shuffle(order)`) without the per-sample interpreter loop. */
int mmg_host_shuffle(const uint32_t* mt_state, int pos, int64_t n, int64_t* perm);

/* Test/bench hooks: per-kernel timing of the most recent mmg_train_step measured with HIP events on
 * the launch stream (enable before the step; read after a stream sync). */
int mmg_set_profiling(mmg_handle* h, int enabled);
int mmg_get_kernel_times(mmg_handle* h, char* names, int names_bytes, float* ms, int max_kernels);

#ifdef __cplusplus
}
#endif
#endif /* MMG_H_ */
