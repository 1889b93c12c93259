/*
 * twotower.h -- C ABI of the MI355X-native two-tower training step (libtwotower_hip.so).
 *
 * The reference (zoongahn/jodalroB-twoTower) has no FFI: its hot path sits behind plain PyTorch
 * modules.  This header is the boundary a binding for that path would use; every entry point names
 * the reference code it replaces (paths relative to the reference root).  See INTEGRATION.md for
 * the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - every data pointer is a DEVICE pointer borrowed from the caller (never freed / resized here);
 *     descriptor structs themselves live in HOST memory and are read during the call only.
 *   - scratch comes from the caller: <op>_workspace_bytes() + (workspace, workspace_bytes).
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t); no hidden synchronisation.
 *   - returns TT_OK (0) or a negative TT_ERR_*; never throws, never exits;
 *     tt_last_error_string() gives the thread's last message.
 *   - row-major everywhere; `ld*` = leading dimension in ELEMENTS.
 */
#ifndef TWOTOWER_H_
#define TWOTOWER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TT_ABI_VERSION 2

#define TT_OK 0
#define TT_ERR_INVALID_ARG (-1)
#define TT_ERR_HIP (-2)
#define TT_ERR_WORKSPACE (-3)
#define TT_ERR_UNSUPPORTED (-4)
#define TT_ERR_DEVICE (-5) /* tt_ctx_check_device_errors: a kernel raised the context's sticky device error word */

#define TT_F32 0
#define TT_BF16 1

#define TT_MAX_SIDES 4   /* towers fused into one lookup / gradient launch */
#define TT_MAX_HIDDEN 8  /* [Linear,ReLU,BatchNorm1d,Dropout] blocks per tower */

typedef struct tt_ctx tt_ctx; /* opaque: device id, cached device properties */
typedef void* tt_stream;      /* hipStream_t */

int tt_abi_version(void);
int tt_ctx_create(int device, tt_ctx** out);
int tt_ctx_destroy(tt_ctx* ctx);
const char* tt_last_error_string(void);
/* number of compute units of the context's device (used by callers to size synthetic work) */
int tt_ctx_num_cus(const tt_ctx* ctx);
/* kernels this library has launched (or recorded into a stream capture) so far in the process: the difference across one
 * captured step is the step's launch count (collectives and the caller's own kernels are not in it); a measurement aid */
uint64_t tt_launch_count(void);

/* Options of a context.  TT_OPT_DEFER_SLAB_REDUCE (default 0): tt_towers_mlp_bwd's one-launch first-block backward leaves the
 * split-K slab reduction of its weight gradients queued in the context instead of launching it; the next tt_embed_grad_bwd
 * with TT_GRAD_PLANNED runs it in the first workgroups of its own launch (the two are independent: one launch fewer in the
 * step's dependent chain); tt_flush_deferred launches it on its own, and the tt_adam_* entries do that themselves before
 * they read a gradient.  Nothing else may write the towers' workspace while tt_deferred_pending() is 1. */
#define TT_OPT_DEFER_SLAB_REDUCE 1
/* TT_OPT_KEYED_PARTS (default 0 = chosen from the batch size): workgroups per key of tt_dedup_plan_keyed* -- every value gives
 * the same plan bit for bit (tests vary it).  TT_OPT_SCORE_BWD_ROWS_MIN (default 32768): number of rows from which
 * tt_score_bwd_bf16 takes the workgroup-staged form instead of the b-split form (same results up to summation order; tests
 * force either form). */
#define TT_OPT_KEYED_PARTS 2
#define TT_OPT_SCORE_BWD_ROWS_MIN 3
/* TT_OPT_DEFER_RIDERS (default 0; 1 or 3 = both riders, 2 = only the loss reduction -- a caller that reads the plan right after
 * building it, as the sharded exchange does): tt_dedup_plan_keyed* leaves the plan's compaction, and tt_score_fwd_sym_* its last reduction
 * (loss_out / out8), QUEUED in the context instead of launching them: tt_towers_mlp_fwd / tt_towers_mlp_bwd run them in an extra
 * grid row of their fused tail kernels (two launches fewer in the step's dependent chain; same bodies: bit-identical), and
 * tt_flush_deferred, tt_embed_grad_bwd, tt_embed_grad_finish and the tt_adam_* entries launch whatever is still queued.  Until
 * then the plan's unique_rows / seg_offsets / n_unique and the forward's loss_out / out8 are NOT written: for callers that run
 * the whole step back to back (GraphedTrainStep); tt_deferred_pending() covers the queue. */
#define TT_OPT_DEFER_RIDERS 4
/* TT_OPT_FP8_GRAD (default 1): tt_score_bwd_fp8 forms the gradient products dA = W B from e4m3 operands as well (softmax weights
 * block-scaled per row and 32 consecutive b rows, the diagonal weight kept apart in f32: tt_score_bwd_fp8 below); 0 = bf16 weights and bf16
 * B rows for those products (round 2's arithmetic, 1.5 x the matrix-pipe time). */
#define TT_OPT_FP8_GRAD 5
/* TT_OPT_CHAINED (default 1): the segment-head pass of tt_dedup_plan / tt_dedup_plan_runs runs as ONE launch, its cross-workgroup
 * prefix sum chained through a small context-owned buffer (a tile publishes its head count with a ready bit and sums its
 * predecessors'; the last one through clears the buffer); 0 = count and write as separate launches.  Same results bit for bit
 * (tests compare the two). */
#define TT_OPT_CHAINED 6
/* TT_OPT_LOOKUP_NT (default 0): tt_batch_ingest_lookup stores its bf16 rows with non-temporal stores (same bits; an A/B switch:
 * they leave L2 during the launch instead of at its end). */
#define TT_OPT_LOOKUP_NT 7
/* TT_OPT_CHAIN_SPIN (default 2^22): polls after which a tile of a chained launch stops waiting for a predecessor (it then raises
 * the device error word instead of hanging the GPU); tests lower it to provoke the error path. */
#define TT_OPT_CHAIN_SPIN 8
int tt_ctx_set_option(tt_ctx* ctx, int32_t option, int32_t value);
/* Device-side errors.  A kernel that cannot complete its contract without hanging or faulting the GPU (a tile of a chained
 * segment-head launch whose bounded wait for a predecessor expired; a lookup whose decoded row lies outside the table it was given
 * -- key offsets / vocabularies of another table: the launch reads the table's last row instead) raises a STICKY word in device memory owned by the context
 * (TT_DEVERR_* bits) and finishes; nothing reaches the host by itself.  tt_ctx_check_device_errors reads the word (synchronises
 * `stream`), and if it is set: clears it, zeroes the context's chain buffers, sets the error string and returns TT_ERR_DEVICE --
 * every plan built since the previous check must then be considered corrupt.  Call it where the host synchronises anyway (end of an
 * epoch, before a checkpoint, when a captured step is closed). */
#define TT_DEVERR_CHAIN_TIMEOUT 1u
#define TT_DEVERR_ROW_RANGE 2u
int tt_ctx_check_device_errors(tt_ctx* ctx, tt_stream stream);
/* The hand-over launch as a node of a captured graph.  A tt_batch_ingest* call on a stream that is being captured becomes a kernel
 * node; tt_handover_captured_node hands that node out (NULL if the last call was not captured) and forgets it.  Later, between
 * tt_handover_retarget(ctx, graph_exec, node) and tt_handover_retarget(ctx, NULL, NULL), every tt_batch_ingest* call launches
 * NOTHING: it re-points `node` of the executable graph `graph_exec` (hipGraphExec_t) at the call's own kernel, grid and arguments
 * (hipGraphExecKernelNodeSetParams), so that the next launch of the graph hands over THAT batch -- launches already enqueued keep
 * theirs.  What it is for: a graph that holds SEVERAL training steps, hand-overs included (graph.GraphedTrainStep(unroll=U)): one graph
 * launch costs ~8 us on top of its nodes on this runtime (tools/probe/graph_setparams.hip), U steps per launch pay it once.
 * Reference counterpart: none (torch.compile(mode="reduce-overhead") replays one step per launch, scripts/train.py:223-225). */
int tt_handover_retarget(tt_ctx* ctx, void* graph_exec, void* node);
int tt_handover_captured_node(tt_ctx* ctx, void** node);
int tt_flush_deferred(tt_ctx* ctx, tt_stream stream);
/* only the queued slab reduction (the one thing that lives in the caller's shared scratch buffer) */
int tt_flush_deferred_slabs(tt_ctx* ctx, tt_stream stream);
/* bit 0: a slab reduction is queued; bit 1: riders (TT_OPT_DEFER_RIDERS) are queued */
int tt_deferred_pending(const tt_ctx* ctx);

/* ------------------------------------------------------------------------------------------------
 * Categorical embedding lookup  -- replaces CategoricalEmbedder._kjt_to_dict + .forward
 * (src/towers/cat_embed.py:88-123 id unpack + clamp, :157-178 per-key nn.Embedding gather + cat)
 * and the torch.cat of src/towers/tower/base_tower.py:139 (rows are written straight into the MLP
 * input buffer at a column offset).
 *
 * All per-key tables of all towers live in ONE fused row space `table[table_rows, E]` (f32).
 * For side s, sample b, key k:   id  = ids[b*K + k]                       (sample-major KJT values)
 *                                row = key_row_offset[k] + clamp(id, 0, key_vocab[k]-1)
 *                                out[b*ld_out + k*E .. +E) = table[row*E .. +E)
 * table == NULL (with rows_out set): rows-only mode -- nothing is gathered, only rows_out is written
 * (used to route ids to the rank that owns the row when the table is sharded).
 * rows_out (optional, may be NULL): fused row of every slot, slot = side_slot_base + b*K + k with
 * side_slot_base = sum of B*K of the earlier sides -- the input of tt_dedup_plan.
 * ---------------------------------------------------------------------------------------------- */
typedef struct tt_embed_side {
  const int64_t* ids;            /* [B*K] */
  const int64_t* key_row_offset; /* [K] first fused row of key k */
  const int64_t* key_vocab;      /* [K] rows of key k (reference: metadata count + 10) */
  void* out;                     /* first output element of sample 0 / key 0 */
  int64_t ld_out;
  int32_t K;
  int32_t out_dtype; /* TT_F32 (bit-exact row copy) or TT_BF16 (round-to-nearest-even) */
} tt_embed_side;

/* Measurement hook: per-launch device-clock stamps of the lookup kernel, usable inside a captured graph (where
 * HIP events cannot bracket one kernel), without host synchronisation and WITHOUT an extra launch.
 * `ring_dev` = 8192 + n_slots * 8192 * 2 uint64 words of device memory, zero-initialised by the caller:
 *   [b], b < 8192: launch counter of workgroup b (each workgroup keeps its own: [0] = launches so far);
 *   then n_slots blocks of 8192 pairs {start, end}: block (n mod n_slots), pair b = workgroup b in launch n --
 *   start = its first instruction, end = all its stores have completed (0, 0 = workgroup not in the grid).
 *   Kernel duration of launch n = (max end - min start over the block) * 10 ns (100 MHz clock); the caller reduces.
 *   ring_dev == NULL switches the hook off.  The fused hand-over + lookup launch (tt_batch_ingest_lookup) writes the same ring:
 *   workgroup b = its linear index, the tiles (gather phase) first, the copy roles behind them. */
int tt_embed_lookup_set_profile(tt_ctx* ctx, uint64_t* ring_dev, int32_t n_slots);
int tt_embed_lookup_fwd(tt_ctx* ctx, const float* table, int64_t table_rows, int32_t E,
                        const tt_embed_side* sides, int32_t n_sides, int64_t B, int32_t* rows_out,
                        tt_stream stream);
/* The same gather + concat (cat_embed.py:157-178, base_tower.py:139) from PRECOMPUTED fused rows: rows[slot], slot = side_base +
 * b*K + k (side_base = sum of B*K of the earlier sides), already clamped -- what tt_batch_ingest / tt_batch_ingest_store leave in
 * `rows_sm` when a captured step's batch is handed over (they decode and clamp every id anyway).  sides[i] gives K / out / ld_out /
 * out_dtype (ids, key_row_offset, key_vocab are ignored and may be NULL).  Same bits in the outputs as tt_embed_lookup_fwd on the
 * ids the rows came from (test); the kernel reads a 4-byte row instead of an 8-byte id + its key's offset and vocabulary.
 * The rows are trusted to lie in [0, table_rows): the hand-over that formed them has checked them (its table_rows argument);
 * tt_embed_lookup_fwd checks the rows it decodes itself (TT_DEVERR_ROW_RANGE).
 * E = 4 x a power of two, outputs 4-element aligned (TT_ERR_UNSUPPORTED otherwise). */
int tt_embed_lookup_rows_fwd(tt_ctx* ctx, const float* table, int64_t table_rows, int32_t E,
                             const tt_embed_side* sides, int32_t n_sides, int64_t B, const int32_t* rows,
                             tt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Duplicate-row plan for the sparse gradient (a16): stable LSD radix sort of the M slot rows, then
 * segment boundaries of equal rows.  Depends on ids only, so callers run it during the forward.
 *   sorted_src[i]   slot index of the i-th smallest row (ties in ascending slot order)
 *   unique_rows[u]  u-th distinct row, ascending;   seg_offsets[u]..seg_offsets[u+1] its span
 *   n_unique[0]     number of distinct rows U (stays on the device; consumers read it there)
 * ---------------------------------------------------------------------------------------------- */
size_t tt_dedup_workspace_bytes(int64_t M);
int tt_dedup_plan(tt_ctx* ctx, const int32_t* rows, int64_t M, int64_t table_rows,
                  int32_t* sorted_src, int32_t* unique_rows, int32_t* seg_offsets, int32_t* n_unique,
                  void* workspace, size_t workspace_bytes, tt_stream stream);

/* Same plan for slot rows that come straight from tt_embed_lookup_fwd (slot = side_base + b*K + k): key k's
 * rows lie in key k's own row range, so the sort decomposes into sum(K) independent sorts of B ids.  One
 * workgroup per key sorts its B ids entirely in LDS (stable LSD radix, ballot ranking) and a second small
 * launch compacts the per-key unique lists: 2 launches instead of 3 per radix pass + 2.  Needs B <= 8192 and
 * key row ranges ascending in (side, k) order (true for the fused store); same outputs as tt_dedup_plan. */
size_t tt_dedup_keyed_workspace_bytes(int64_t M, int32_t n_keys);
int tt_dedup_plan_keyed(tt_ctx* ctx, const int32_t* rows, const int32_t* side_K /* host [n_sides] */,
                        int32_t n_sides, int64_t B, int32_t* sorted_src, int32_t* unique_rows,
                        int32_t* seg_offsets, int32_t* n_unique, void* workspace, size_t workspace_bytes,
                        tt_stream stream);

/* The same plan from rows in KEY-MAJOR order, rows_km[side_base + k*B + b] (written by tt_batch_ingest): a key's B rows are
 * one contiguous run instead of one word per K*4 bytes -- the sort's own strided load is 6.5 of a workgroup's 18 us. */
int tt_dedup_plan_keyed_km(tt_ctx* ctx, const int32_t* rows_km, const int32_t* side_K /* host [n_sides] */,
                           int32_t n_sides, int64_t B, int32_t* sorted_src, int32_t* unique_rows,
                           int32_t* seg_offsets, int32_t* n_unique, void* workspace, size_t workspace_bytes,
                           tt_stream stream);

/* The per-key plan that ALSO prepares the gradient reduction: rows with more than 64 slots (the two-row keys: thousands each)
 * are summed chunk by chunk, and the list of those rows / chunks depends on the plan only.  Built here -- into the workspace
 * tt_embed_grad_bwd will be called with (size tt_embed_grad_workspace_bytes(M, E), flag TT_GRAD_PLANNED; nobody else may touch
 * it in between) -- the reduction's row pass and chunk pass run as ONE launch instead of one after the other.
 * rows_key_major: 0 = the lookup's slot-major rows, 1 = tt_batch_ingest's [key][sample] rows. */
int tt_dedup_plan_keyed_long(tt_ctx* ctx, const int32_t* rows, int32_t rows_key_major, const int32_t* side_K /* host [n_sides] */,
                             int32_t n_sides, int64_t B, int32_t E, int32_t* sorted_src, int32_t* unique_rows,
                             int32_t* seg_offsets, int32_t* n_unique, void* grad_workspace, size_t grad_workspace_bytes,
                             void* workspace, size_t workspace_bytes, tt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Embedding gradient -- replaces autograd's nn.Embedding backward (dense index_add; reference
 * builds its tables with sparse=False, src/towers/cat_embed.py:42-45; scripts/train.py:326).
 * Slot s of source i reads d_out_i[b*ld + k*E .. +E).  Atomic-free: one wave-group per distinct
 * row sums its contributions in ascending slot order; rows with more than 64 contributions are
 * summed in 64-slot chunks whose partial sums are then added in chunk order (bitwise reproducible).
 *   TT_GRAD_SPARSE     out[u*E .. +E)               = sum   (u < U; out has room for M rows)
 *   TT_GRAD_DENSE_SET  out[unique_rows[u]*E .. +E)  = sum   (caller zeroed the dense buffer)
 *   TT_GRAD_DENSE_ACC  out[unique_rows[u]*E .. +E) += sum
 *   | TT_GRAD_SHORT_SEGMENTS (flag): every row is summed by one lane group whatever its length, and the chunk passes
 *     are not launched -- for callers that know no row has many contributions (the owner side of the row exchange:
 *     at most one per rank).  Results do not depend on the flag, only the time does.
 * ---------------------------------------------------------------------------------------------- */
#define TT_GRAD_SPARSE 0
#define TT_GRAD_DENSE_SET 1
#define TT_GRAD_DENSE_ACC 2
#define TT_GRAD_SHORT_SEGMENTS 0x100
#define TT_GRAD_PLANNED 0x200 /* the workspace was handed to tt_dedup_plan_keyed_long, which left the long-row list in it */
/* with TT_GRAD_PLANNED | TT_GRAD_SPARSE: leave the long rows' chunk sums unadded (their rows of `out` are not written) -- the
 * caller completes `out` on the same workspace with tt_adam_fused_step_finish (inside the optimiser's launch: one launch
 * fewer in the step's dependent chain) or tt_embed_grad_finish, before anything else reads those rows */
#define TT_GRAD_DEFER_FINISH 0x400

typedef struct tt_grad_src {
  const void* d_out; /* gradient w.r.t. the lookup output of this side */
  int64_t ld;
  int32_t K;
  int32_t dtype; /* TT_F32 / TT_BF16 */
} tt_grad_src;

size_t tt_embed_grad_workspace_bytes(int64_t M, int32_t E);
/* counters: NULL, or 3 int32 words of device memory that are ZERO on entry and that the caller keeps (per stream) between
 * calls: the last kernel of the reduction leaves them zero again, so no zeroing launch precedes the reduction. */
int tt_embed_grad_bwd(tt_ctx* ctx, const tt_grad_src* srcs, int32_t n_srcs, int64_t B, int32_t E,
                      const int32_t* sorted_src, const int32_t* seg_offsets,
                      const int32_t* unique_rows, const int32_t* n_unique, int64_t M, int32_t mode,
                      float* out, int32_t* counters, void* workspace, size_t workspace_bytes, tt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Adam -- replaces torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay coupled)
 * (scripts/train.py:231, :327).  `step` is the 1-based step count used for bias correction.
 *   tt_adam_dense_step : exact dense Adam over n contiguous elements (tower weights / small tables)
 *   tt_sparse_adam_step: the same update applied only to the U looked-up rows; entries of unique_rows that
 *   are >= table_rows are skipped (pads of the multi-GPU fixed-capacity routing) (rows not in the batch
 *                        keep weight, m and v untouched -- the documented difference to the
 *                        reference's dense update; DESIGN.md "optimiser semantics")
 * ---------------------------------------------------------------------------------------------- */
/* Graph capture: when `hparams_dev` is non-NULL the kernels read the per-step scalars from DEVICE memory
 * instead of the host arguments, so a captured launch can be replayed with a new lr / step:
 *   hparams_dev[0] = lr / (1 - beta1^step)   hparams_dev[1] = 1 / sqrt(1 - beta2^step)
 *   hparams_dev[2] = beta1  [3] = beta2  [4] = eps  [5] = weight_decay        (tt_adam_hparams fills them) */
void tt_adam_hparams(int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                     float out6[6]);
int tt_adam_dense_step(tt_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t n,
                       int64_t step, float lr, float beta1, float beta2, float eps,
                       float weight_decay, const float* hparams_dev, tt_stream stream);
/* the same dense update over n_tensors separate tensors in one launch per 32 tensors */
typedef struct tt_adam_tensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} tt_adam_tensor;
int tt_adam_multi_step(tt_ctx* ctx, const tt_adam_tensor* tensors /* host array */, int32_t n_tensors,
                       int64_t step, float lr, float beta1, float beta2, float eps,
                       float weight_decay, const float* hparams_dev, tt_stream stream);
int tt_sparse_adam_step(tt_ctx* ctx, float* table, float* m, float* v, int64_t table_rows, int32_t E,
                        const int32_t* unique_rows, const float* grad_rows, const int32_t* n_unique,
                        int64_t M, int64_t step, float lr, float beta1, float beta2, float eps,
                        float weight_decay, const float* hparams_dev, tt_stream stream);

/* tt_adam_multi_step (<= 32 tensors) and tt_sparse_adam_step with the same hyper-parameters in ONE launch: the
 * tower weights and the looked-up table rows of a step (optim.FusedAdam uses it when both share a parameter group
 * and step number). */
int tt_adam_fused_step(tt_ctx* ctx, const tt_adam_tensor* tensors, int32_t n_tensors, float* table, float* m, float* v,
                       int64_t table_rows, int32_t E, const int32_t* unique_rows, const float* grad_rows, const int32_t* n_unique, int64_t M,
                       int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                       const float* hparams_dev, tt_stream stream);

/* tt_adam_fused_step for a gradient whose long-row finish was deferred (TT_GRAD_DEFER_FINISH): extra workgroups add each long
 * row's chunk partials (same order as the reduction's own finish: bit-identical), store the sum into grad_rows and update that
 * table row; grad_rows is complete when the launch has run. */
int tt_adam_fused_step_finish(tt_ctx* ctx, const tt_adam_tensor* tensors, int32_t n_tensors, float* table, float* m, float* v,
                              int64_t table_rows, int32_t E, const int32_t* unique_rows, float* grad_rows, const int32_t* n_unique,
                              int64_t M, const int32_t* seg_offsets, void* grad_workspace, size_t grad_workspace_bytes,
                              int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay,
                              const float* hparams_dev, tt_stream stream);
/* The deferred finish on its own (sparse mode: out = grad_rows [M, E]) -- for a consumer other than the fused optimiser. */
int tt_embed_grad_finish(tt_ctx* ctx, int32_t E, const int32_t* seg_offsets, int64_t M, float* out, void* workspace,
                         size_t workspace_bytes, tt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Tower MLP -- replaces BaseTower.forward after the lookup (src/towers/tower/base_tower.py:133-145)
 * and its autograd backward:
 *   x[:, 0:h0]      = dense W_proj^T + b_proj                 (:133; x[:, h0:] holds the lookup rows)
 *   per hidden i    : a = relu(in W_i^T + b_i); BatchNorm1d(a) (train: batch stats, biased var,
 *                     eps 1e-5, running stats momentum 0.1 with unbiased var; eval: running stats);
 *                     Dropout(p) in train mode (counter-based mask from `seed`)        (:88-93)
 *   y = h W_out^T + b_out ;  emb = y / max(||y||_2, 1e-12)                             (:97, :145)
 * ---------------------------------------------------------------------------------------------- */
typedef struct tt_tower_params {
  int32_t din, h0, kcat_e, n_hidden, d_out;
  int32_t hidden[TT_MAX_HIDDEN]; /* output width of hidden block i */
  const float* w_proj;           /* [h0, din] */
  const float* b_proj;           /* [h0] */
  const float* w[TT_MAX_HIDDEN]; /* [hidden[i], in_i], in_0 = h0 + kcat_e, in_i = hidden[i-1] */
  const float* b[TT_MAX_HIDDEN];
  const float* bn_w[TT_MAX_HIDDEN];
  const float* bn_b[TT_MAX_HIDDEN];
  float* bn_rm[TT_MAX_HIDDEN]; /* running_mean, updated by the train-mode forward */
  float* bn_rv[TT_MAX_HIDDEN]; /* running_var */
  int64_t* bn_nbt[TT_MAX_HIDDEN]; /* num_batches_tracked (+1 per train-mode forward); may be NULL */
  const float* w_out;          /* [d_out, in_last] */
  const float* b_out;
  int32_t compute_dtype; /* TT_F32: exact-f32 MFMA (parity); TT_BF16: GEMM operands rounded to bf16 (RNE) on the way
                            into LDS, f32 accumulate on v_mfma_f32_32x32x16_bf16 */
  int32_t x_dtype;       /* element type of tt_tower_acts.x  (TT_BF16 only with compute_dtype TT_BF16: the GEMMs that read x
                            round it to bf16 anyway, so results are bit-identical and x costs half the HBM bytes) */
  int32_t dx_dtype;      /* element type of tt_tower_grads.d_x (TT_BF16 only with compute_dtype TT_BF16; the per-slot row
                            gradients are then rounded to bf16 before tt_embed_grad_bwd sums them in f32) */
  int32_t flags;         /* TT_TOWER_*: 0 by default */
  /* Data-parallel BatchNorm over several ranks (SyncBN): the pass is cut at the batch-wide reduction so that the caller
   * can exchange the statistics in between (the library makes no collective calls).
   *   sync_phase 0: the whole pass (default).
   *   sync_phase 1: forward up to the local BN statistics -> acts.bn_sync_local [3][H] = (n, mean, M2) per column;
   *                 backward up to the local column sums   -> grads.s_sync_local [2][H] = (S1, S2).
   *   sync_phase 2: the rest, with the statistics of ALL ranks as acts.bn_sync_all [sync_ranks][3][H] (merged in rank
   *                 order with Chan's formula) / grads.s_sync_all [sync_ranks][2][H] (added in rank order); BN normalises
   *                 over sync_ranks * B rows, and the BN weight / bias gradients -- identical on every rank -- are stored
   *                 divided by sync_ranks so that the caller's SUM over ranks of the dense gradients leaves them right.
   * Phases 1 / 2 need a training pass over towers with exactly ONE hidden block (any width, either compute_dtype: the
   * reference's [128, 64] -> 64 on the fused tail kernels, scripts/train.py's [512, 256] -> 128 on the separate ones);
   * anything else returns TT_ERR_UNSUPPORTED. */
  int32_t sync_phase;
  int32_t sync_ranks;
  int64_t rng_row_offset; /* first global row of this rank's batch: dropout masks are drawn per GLOBAL (row, column), so a
                             split batch drops the same elements as the whole one */
  /* Optional bf16 SHADOWS of w_proj / w[i] (same shapes, RNE of the f32 values; NULL = none): the one-launch front reads them
   * instead of the f32 weights -- every 64-row workgroup re-reads its tower's weights, and the f32 copies are two thirds of the bytes
   * it moves.  The caller keeps them equal to bf16(w) (a captured step: refreshed by the hand-over launch, tt_cvt_list); results
   * are bit-identical to the f32 path, which rounds the same values on the way into LDS. */
  const void* w_proj_bf16;
  const void* w_bf16[TT_MAX_HIDDEN];
} tt_tower_params;
/* run the tail of a training pass (BN of the last block, output Linear, L2 normalise) as the separate kernels even when
   the fused form applies (last hidden width and d_out <= 64, compute_dtype TT_BF16): for A/B comparison */
#define TT_TOWER_UNFUSED_TAIL 1
#define TT_TOWER_UNFUSED_FRONT 2 /* keep projection GEMM, block GEMM and slab/statistics pass as separate launches */
#define TT_TOWER_UNFUSED_BACK 4  /* first-block weight / data gradients and the projection's weight gradient as separate launches */

typedef struct tt_tower_acts { /* caller-allocated; kept between forward and backward */
  const float* dense;          /* [B, din] */
  void* x;                     /* [B, h0 + kcat_e] of params.x_dtype */
  float* pre[TT_MAX_HIDDEN];   /* [B, hidden[i]] Linear output before ReLU */
  float* act[TT_MAX_HIDDEN];   /* [B, hidden[i]] block output (after BN and dropout) */
  float* mean[TT_MAX_HIDDEN];  /* [hidden[i]] statistics used by BN in this pass */
  float* rstd[TT_MAX_HIDDEN];
  float* y;   /* [B, d_out] before normalisation */
  float* emb; /* [B, d_out] unit rows */
  void* emb_packed; /* optional, tt_score_pack_bytes(B, d_out) bytes, 16-byte aligned: on return also holds the score kernels'
                       bf16 operand images of emb (what tt_score_pack_bf16 would produce) -- written by the fused tail kernel
                       itself where that applies, by the pack kernel otherwise */
  float emb_pack_scale; /* the images hold bf16(emb_pack_scale * emb) (0 = 1): tt_score_pack_bf16's `scale` */
  float* bn_sync_local;     /* sync_phase 1 out: [3][hidden[0]] */
  const float* bn_sync_all; /* sync_phase 2 in:  [sync_ranks][3][hidden[0]] */
  int64_t bn_sync_stride;   /* floats between two ranks' triples in bn_sync_all (0 = 3 * hidden[0]): lets one all-gather
                               carry every tower's statistics */
} tt_tower_acts;

typedef struct tt_tower_grads { /* every buffer is overwritten, not accumulated */
  float* w_proj;
  float* b_proj;
  float* w[TT_MAX_HIDDEN];
  float* b[TT_MAX_HIDDEN];
  float* bn_w[TT_MAX_HIDDEN];
  float* bn_b[TT_MAX_HIDDEN];
  float* w_out;
  float* b_out;
  void* d_x;                     /* [B, h0 + kcat_e] of params.dx_dtype; columns [h0, ..) feed tt_embed_grad_bwd.  Columns [0, h0)
                                    (the projection output's gradient) are scratch: the bf16 path with edge-free shapes does not write them */
  float* scratch[TT_MAX_HIDDEN]; /* [B, hidden[i]] */
  float* d_y;                    /* [B, d_out] */
  float* s_sync_local;     /* sync_phase 1 out: [2][hidden[0]] */
  const float* s_sync_all; /* sync_phase 2 in:  [sync_ranks][2][hidden[0]] */
  int64_t s_sync_stride;   /* floats between two ranks' pairs in s_sync_all (0 = 2 * hidden[0]) */
} tt_tower_grads;

/* scratch for either pass (split-K slabs of the weight gradients, column-reduction partials) */
size_t tt_tower_workspace_bytes(const tt_tower_params* p, int64_t B);
/* seed_dev (may be NULL): when set, the dropout seed is read from device memory (seed + *seed_dev), so a
 * captured launch draws a new mask on every replay once the caller updates that word. */
int tt_tower_mlp_fwd(tt_ctx* ctx, const tt_tower_params* p, const tt_tower_acts* a, int64_t B,
                     int32_t train, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                     void* workspace, size_t workspace_bytes, tt_stream stream);
int tt_tower_mlp_bwd(tt_ctx* ctx, const tt_tower_params* p, const tt_tower_acts* a,
                     const float* d_emb, const tt_tower_grads* g, int64_t B, int32_t train,
                     float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* workspace,
                     size_t workspace_bytes, tt_stream stream);

/* The same two passes for n towers (1..TT_MAX_SIDES) with ONE launch per layer step: every kernel takes the
 * towers' argument blocks side by side and blockIdx selects the tower ("horizontal fusion" -- each tower alone
 * fills at most half of the 256 CUs).  All towers share B and the number of hidden blocks; widths may differ.
 * Arrays are HOST arrays of n pointers; workspaces[t] >= tt_tower_workspace_bytes(p[t], B). */
int tt_towers_mlp_fwd(tt_ctx* ctx, int32_t n, const tt_tower_params* const* p, const tt_tower_acts* const* a,
                      int64_t B, int32_t train, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                      void* const* workspaces, const size_t* workspace_bytes, tt_stream stream);
int tt_towers_mlp_bwd(tt_ctx* ctx, int32_t n, const tt_tower_params* const* p, const tt_tower_acts* const* a,
                      const float* const* d_emb, const tt_tower_grads* const* g, int64_t B, int32_t train,
                      float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* const* workspaces,
                      const size_t* workspace_bytes, tt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * In-batch-negative score + symmetric softmax cross-entropy, never materialising the score matrix
 * -- replaces TwoTowerTrainTask._compute_similarity_matrix / _compute_loss / _compute_metrics
 * (src/towers/two_tower_train_task.py:99-134, :162-179), the argmax checks of
 * _verify_positive_pair_alignment (:253-276) and, through the diagonal ranks, the evaluator's
 * Recall@K / MRR (src/evaluation/evaluator.py:20-71).
 *
 * One direction:  for every row a of A[Ra, D] against all rows of Bm[Rb, D], s = <a, b> * inv_t,
 * positive column = a + diag_offset:
 *   sumexp[a] = sum_b exp(s_ab - shift)        (shift >= max s; unit rows => shift = inv_t)
 *   diag[a]   = s at the positive column
 *   rank[a]   = #{b: s_ab > diag} + #{b < positive: s_ab == diag}   (0 <=> torch.argmax hits)
 * The loss needs direction (N,C) and direction (C,N); tt_score_loss_finish combines them:
 *   out[0]=loss  out[1]=accuracy  out[2]=pos mean  out[3]=neg mean  out[4]=gap
 *   out[5]=column-direction top-1 rate  out[6]=sum of all scores;  loss_out[0] = loss (own tensor for autograd)
 * ---------------------------------------------------------------------------------------------- */
int tt_score_dir_fwd(tt_ctx* ctx, const float* A, const float* Bm, int64_t Ra, int64_t Rb, int32_t D,
                     float inv_t, float shift, int64_t diag_offset, float* sumexp, float* diag,
                     int32_t* rank, float* sumscore /* [Ra] sum_b s_ab, may be NULL */,
                     tt_stream stream);
int tt_score_loss_finish(tt_ctx* ctx, int64_t B, float shift, const float* rowsum,
                         const float* colsum, const float* diag, const int32_t* row_rank,
                         const int32_t* col_rank, const float* sumscore, float* out8,
                         float* loss_out /* [1], may be NULL */, tt_stream stream);
/* gradient of one direction's operand:
 *   dA[a] = d_loss[0] * scale * sum_b (e_ab/sumexp_a[a] + e_ab/sumexp_b[b] - 2[b == a+diag_offset]) Bm[b]
 * with e_ab = exp(s_ab - shift) and scale = inv_t / (2 * batch); call once for (N,C) -> dN and once
 * for (C,N) -> dC.  D <= 256. */
int tt_score_dir_bwd(tt_ctx* ctx, const float* A, const float* Bm, int64_t Ra, int64_t Rb, int32_t D,
                     float inv_t, float shift, int64_t diag_offset, const float* sumexp_a,
                     const float* sumexp_b, const float* d_loss, float scale, float* dA,
                     tt_stream stream);
/* ---- bf16 fast path of the same three steps: operands rounded to bf16 (round-to-nearest-even), f32
 * accumulation on v_mfma_f32_32x32x16_bf16, exp/log in f32.  tt_score_pack_bf16 writes the two bf16
 * operand images of X[R, D] (row-major, and fragment-ordered for the gradient product) into `packed`
 * (tt_score_pack_bytes(R, D) bytes, 16-byte aligned); the fwd / bwd calls take packed operands and
 * process one or two directions in ONE launch (the loss needs (N,C) and (C,N)).  D <= 256. */
typedef struct tt_score_fwd_dir {
  const void* A_packed;
  const void* B_packed;
  int64_t Ra, Rb, diag_offset;
  float* sumexp; /* [Ra] */
  float* diag;     /* [Ra] or NULL */
  int32_t* rank;   /* [Ra] or NULL */
  float* sumscore; /* [Ra] sum_b s_ab, or NULL */
  int32_t rank_mode; /* with rank != NULL: 0 or 2 = full rank; 1 = top-1 flag only (rank = 0 if the positive is the
                        row's argmax, first index winning ties, else 1) -- what the training metrics need, at a
                        quarter of the per-element work */
  float ab_scale;    /* product of the scales the two operand images were packed with (0 = 1).  When it equals
                        tt_score_unit_scale(inv_t) in EVERY direction of a call, the kernels run their "unit" form: the softmax
                        term is exp2(acc) -- one VALU op less per score -- and the fixed shift's factor goes onto the finished
                        sums.  Results (sums, diagonal, ranks, gradients) are scale-free either way. */
  float* inv_sumexp; /* [Ra] or NULL: the reciprocal tt_score_bwd_bf16 multiplies by (its inv_a / inv_b): with it the backward
                        kernel takes no reciprocals in its tile loop.  Valid for a backward call with the same ab_scale. */
} tt_score_fwd_dir;
typedef struct tt_score_bwd_dir {
  const void* A_packed;
  const void* B_packed;
  int64_t Ra, Rb, diag_offset;
  const float* sumexp_a; /* [Ra] */
  const float* sumexp_b; /* [Rb], 16-byte aligned */
  float* dA;             /* [Ra, D] f32 */
  float ab_scale;        /* as in tt_score_fwd_dir */
  float b_scale;         /* scale of the B image alone (0 = 1): dA is divided by it */
  /* sumexp_b and inv_b are read a whole 32-row tile at a time: [round_up(Rb, 32)] readable floats, the entries past Rb
     finite (and non-zero in sumexp_b); 16-byte aligned */
  const float* inv_a;    /* [Ra] / [Rb] or NULL: tt_score_fwd_dir.inv_sumexp of the A rows' direction and of the B rows' */
  const float* inv_b;    /* direction (16-byte aligned); NULL = reciprocals of sumexp_a / sumexp_b are taken in the kernel */
} tt_score_bwd_dir;
size_t tt_score_pack_bytes(int64_t R, int32_t D);
/* `scale` (0 = 1): the image holds bf16(scale * X).  Packing ONE of the two towers with tt_score_unit_scale(inv_t) =
 * inv_t * log2(e) folds the softmax's exponent scale into the MFMA (see tt_score_fwd_dir.ab_scale). */
float tt_score_unit_scale(float inv_t);
int tt_score_pack_bf16(tt_ctx* ctx, const float* X, int64_t R, int32_t D, float scale, void* packed, tt_stream stream);
/* two operands in one launch (the notice and company embeddings of a step) */
int tt_score_pack2_bf16(tt_ctx* ctx, const float* X0, int64_t R0, void* packed0, const float* X1, int64_t R1,
                        void* packed1, int32_t D, float scale0, float scale1, tt_stream stream);
int tt_score_fwd_bf16(tt_ctx* ctx, const tt_score_fwd_dir* dirs, int32_t n_dirs, int32_t D, float inv_t,
                      float shift, tt_stream stream);
int tt_score_bwd_bf16(tt_ctx* ctx, const tt_score_bwd_dir* dirs, int32_t n_dirs, int32_t D, float inv_t,
                      float shift, const float* d_loss, float scale, tt_stream stream);
/* Single-pass symmetric forward of the SQUARE training problem (B notice rows against the B company rows of the same
 * pairs; replaces two_tower_train_task.py:99-179's mm + two cross-entropies + metrics): every 32 x 32 tile of S is computed
 * once and feeds both softmax directions.  Inputs: the two packed operand images (tt_score_pack_bf16 / the tower pass);
 * ab_scale as tt_score_fwd_dir.ab_scale.  Outputs: rowsum / colsum / inv_row / inv_col are [round_up(B, 64)] (the entries past B
 * are written too -- 1 and 0 -- so that they can go straight into tt_score_bwd_bf16), diag and row_rank [B]: rowsum / colsum (shifted exp-sums, as tt_score_fwd_bf16's
 * sumexp of the two directions), inv_row / inv_col (tt_score_fwd_dir.inv_sumexp of the two directions: what
 * tt_score_bwd_bf16 takes as inv_a / inv_b), diag (s_ii / T), row_rank (want_rank != 0: 0 where the positive is the row's
 * first maximum, else 1); out8 / loss_out as tt_score_loss_finish (out8[5], the column-direction top-1 rate, is 0 here).
 * Three launches (sweep, per-row finish, scalar finish), every sum in a fixed order: bitwise reproducible.
 * workspace: tt_score_fwd_sym_workspace_bytes(B, D) bytes. */
size_t tt_score_fwd_sym_workspace_bytes(int64_t B, int32_t D);
int tt_score_fwd_sym_bf16(tt_ctx* ctx, const void* N_packed, const void* C_packed, int64_t B, int32_t D, float inv_t,
                          float shift, float ab_scale, int32_t want_rank, float* rowsum, float* colsum, float* inv_row,
                          float* inv_col, float* diag, int32_t* row_rank, float* out8, float* loss_out, void* workspace,
                          size_t workspace_bytes, tt_stream stream);
/* fp8 (OCP e4m3) form of the score kernels -- BASELINE.json configs[4] (final_embedding_dim 256, batch 65536): the S products
 * run on v_mfma_scale_f32_32x32x64_f8f6f4 (twice the bf16 MFMA rate), f32 accumulate; softmax and loss as on the bf16 path.
 * Per-tensor scale: the images hold fp8(64 * scale * x), the factor 2^6 leaves again through the instruction's block scales,
 * so ab_scale / b_scale mean what they mean in the bf16 calls.  tt_score_pack2_fp8 writes, per operand, the fp8 rows image,
 * the bf16 fragment image and the fp8 fragment image (tt_score_pack_fp8_bytes(R, D) bytes, 16-byte aligned).
 * tt_score_fwd_sym_fp8 = tt_score_fwd_sym_bf16 on such operands (same workspace).  tt_score_bwd_fp8 = tt_score_bwd_bf16 on such
 * operands, always in the workgroup-staged form, with the gradient products dA = W B
 *   TT_OPT_FP8_GRAD 1 (default): on the fp8 MFMA as well, K = 64 b rows per instruction.  The softmax weights
 *     w_ab = e_ab (1/rowsum_a + 1/colsum_b) of row a and of one BLOCK of 32 consecutive b rows (32 k .. 32 k + 31) are divided
 *     by 2^(floor(log2 m) - 7), m the block's largest weight, and rounded to e4m3 (nearest even); the power of two goes into
 *     the instruction as the block's scale.  The diagonal's weight w_aa - 2 stays out of the block (f32) and multiplies the
 *     bf16 image's row: dA[a] = sum_b q(w_ab) B8[b] + (w_aa - 2) B16[pos_a].
 *   TT_OPT_FP8_GRAD 0: bf16 weights x bf16 operand images, as tt_score_bwd_bf16.
 * Tolerance: tests/test_gpu_parity.py::test_score_fp8_vs_rounded_oracle (the f64 oracle with the same roundings). */
size_t tt_score_pack_fp8_bytes(int64_t R, int32_t D);
int tt_score_pack2_fp8(tt_ctx* ctx, const float* X0, int64_t R0, void* packed0, const float* X1, int64_t R1,
                       void* packed1, int32_t D, float scale0, float scale1, tt_stream stream);
int tt_score_fwd_sym_fp8(tt_ctx* ctx, const void* N_packed, const void* C_packed, int64_t B, int32_t D, float inv_t,
                         float shift, float ab_scale, int32_t want_rank, float* rowsum, float* colsum, float* inv_row,
                         float* inv_col, float* diag, int32_t* row_rank, float* out8, float* loss_out, void* workspace,
                         size_t workspace_bytes, tt_stream stream);
int tt_score_bwd_fp8(tt_ctx* ctx, const tt_score_bwd_dir* dirs, int32_t n_dirs, int32_t D, float inv_t,
                     float shift, const float* d_loss, float scale, tt_stream stream);
/* Dense loss path: the loss variants the fused kernels do not cover, on the MATERIALISED score matrix --
 * label-smoothed cross-entropy (loss_type 0; two_tower_train_task.py:114-133, F.cross_entropy(label_smoothing=e) in both
 * directions) and cosine-embedding loss (loss_type 1; :135-158: F.cosine_embedding_loss of [s] against [1], positives on the
 * diagonal, negatives off it, mean over all B^2 entries).  f32 throughout, exact-f32 GEMMs, O(B^2) memory: S [B, B] (holds
 * d loss / d (N C^T) after the backward call), stats [6 B] floats and hit [2 B] int32 are caller memory kept between the calls.
 * out8 = {loss, row top-1 accuracy, positive mean, negative mean, gap, column top-1 accuracy, sum of S, 0} as
 * tt_score_loss_finish. */
size_t tt_score_dense_workspace_bytes(int64_t B, int32_t D);
int tt_score_dense_fwd(tt_ctx* ctx, const float* N, const float* Cm, int64_t B, int32_t D, float inv_t, int32_t loss_type,
                       float label_smoothing, float* S, float* stats, int32_t* hit, float* out8, float* loss_out,
                       tt_stream stream);
int tt_score_dense_bwd(tt_ctx* ctx, const float* N, const float* Cm, int64_t B, int32_t D, float inv_t, int32_t loss_type,
                       float label_smoothing, float* S, const float* stats, const float* d_loss, float* dN, float* dC,
                       void* workspace, size_t workspace_bytes, tt_stream stream);
/* dense score matrix S[Ra, Rb] = A Bm^T * inv_t (result["similarity_matrix"], predict_batch
 * "all_similarities": two_tower_train_task.py:94, :206) */
int tt_score_matrix(tt_ctx* ctx, const float* A, const float* Bm, int64_t Ra, int64_t Rb, int32_t D,
                    float inv_t, float* S, int64_t lds, tt_stream stream);
/* rank of the positive column of every row of a dense matrix (evaluator MRR / Recall@K on a given
 * similarity matrix: src/evaluation/evaluator.py:45-71):
 *   rank[r] = #{c: S[r,c] > S[r,r+off]} + #{c < r+off: S[r,c] == S[r,r+off]} */
int tt_diag_rank_rows(tt_ctx* ctx, const float* S, int64_t R, int64_t Ccols, int64_t lds,
                      int64_t diag_offset, int32_t* rank, tt_stream stream);
/* per-row top-k of a dense matrix, descending, ties -> lower column first (torch.topk use in
 * predict_batch :195 and evaluator.py:35); k <= 64 */
int tt_topk_rows(tt_ctx* ctx, const float* S, int64_t R, int64_t Ccols, int64_t lds, int32_t k,
                 float* vals, int64_t* idx, tt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Generic Linear used by the one-off feature projection -- replaces FeatureProjector.forward
 * (src/torchrec_preprocess/feature_projector.py:20-28):  Y = act(X W^T + b)
 * ---------------------------------------------------------------------------------------------- */
int tt_linear_fwd(tt_ctx* ctx, const float* X, int64_t ldx, const float* W, const float* bias,
                  float* Y, int64_t ldy, int64_t M, int32_t N, int32_t K, int32_t relu,
                  tt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Batch assembly on the device -- replaces the numpy fancy-index gather of collate_fn_gpu_optimized
 * (src/towers/pairs/unified_bid_data_loader.py:630-684) and _build_batch_kjt (:827-841):
 *   dense_out[b, :] = dense_store[entity[b], :] ;  ids_out[b*K + k] = cat_store[entity[b]*K + k]
 * ---------------------------------------------------------------------------------------------- */
int tt_batch_gather(tt_ctx* ctx, const int64_t* entity, int64_t B, const float* dense_store,
                    int32_t dense_dim, const int64_t* cat_store, int32_t K, float* dense_out,
                    int64_t* ids_out, tt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * Row-wise sharded tables (multi-GPU; the reference is single-process): owner(row) = row % G, local
 * index at the owner = row / G.  The U distinct rows of a duplicate-row plan are routed to their owners through
 * FIXED-CAPACITY buckets, so no size ever travels to the host and the whole exchange can sit inside a captured graph:
 *   send_ids [G, C] int32  local row index at owner g, in plan order; unused entries = pad_id[g]
 *   send_u   [G, C] int32  plan index u of each entry; unused entries = pad_u (caller: index of an all-zero row)
 *   pos_u    [M]    int32  g*C + position of plan row u; G*C for rows that did not fit -- the caller keeps row G*C of the
 *                          buffer the exchanged rows are placed from all-zero, so an overflowing step (which it must
 *                          reject: overflow[0]) never feeds a tower another row's embedding
 *   counts   [G]    int32  entries wanted per owner; overflow[0] is set to 1 when any count exceeds C (sticky: never
 *                          cleared here, the caller owns the flag)
 * tt_route_expand: idx_slot[slot] = pos_u[u] for every slot of plan row u (int64: ids of the placing lookup).
 * tt_route_bucket_expand: both in one call.
 * ---------------------------------------------------------------------------------------------- */
#define TT_MAX_RANKS 64
size_t tt_route_workspace_bytes(int64_t M, int32_t G);
int tt_route_bucket(tt_ctx* ctx, const int32_t* unique_rows, const int32_t* n_unique, int64_t M, int32_t G, int32_t C,
                    const int32_t* pad_id, int32_t pad_u, int32_t* send_ids, int32_t* send_u, int32_t* pos_u,
                    int32_t* counts, int32_t* overflow, void* workspace, size_t workspace_bytes, tt_stream stream);
int tt_route_expand(tt_ctx* ctx, const int32_t* sorted_src, const int32_t* seg_offsets, const int32_t* n_unique,
                    const int32_t* pos_u, int64_t M, int64_t* idx_slot, tt_stream stream);
int tt_route_bucket_expand(tt_ctx* ctx, const int32_t* unique_rows, const int32_t* n_unique, int64_t M, int32_t G, int32_t C,
                           const int32_t* pad_id, int32_t pad_u, int32_t* send_ids, int32_t* send_u, int32_t* pos_u,
                           int32_t* counts, int32_t* overflow, void* workspace, size_t workspace_bytes,
                           const int32_t* sorted_src, const int32_t* seg_offsets, int64_t* idx_slot, tt_stream stream);
/* Duplicate-row plan of G ASCENDING runs of C row ids each (what an owner receives: every source sends its distinct rows in
 * ascending order, pads -- the largest value -- at the end): a stable merge by binary searches instead of radix passes.
 * Same outputs as tt_dedup_plan over the concatenated runs; workspace tt_dedup_workspace_bytes(G * C).
 * row_limit > 0: ids >= row_limit are pads -- they sort to the end as ONE last group that is left out of n_unique (the
 * gradient reduction and the optimiser then never touch the thousands of pad entries); row_limit <= 0: every id counts. */
int tt_dedup_plan_runs(tt_ctx* ctx, const int32_t* rows, int32_t G, int64_t C, int64_t row_limit, int32_t* sorted_src,
                       int32_t* unique_rows, int32_t* seg_offsets, int32_t* n_unique, void* workspace,
                       size_t workspace_bytes, tt_stream stream);
/* out[i, :] = rows[i] < 0 ? 0 : table[min(rows[i], table_rows - 1), :] -- the owner's gather of requested rows and the
 * hand-over of per-row gradients into the send buckets, whose unused entries carry -1 (E a multiple of 4, 16-byte aligned
 * f32 table).  out_dtype TT_BF16: rows leave rounded to bf16 (RNE) -- bit-identical to rounding them where a bf16 tower
 * input is filled, at half the bytes on the wire. */
int tt_gather_rows(tt_ctx* ctx, const float* table, int64_t table_rows, int32_t E, const int32_t* rows, int64_t n,
                   void* out, int32_t out_dtype, tt_stream stream);

/* ------------------------------------------------------------------------------------------------
 * n (<= 8) device-to-device copies in ONE launch -- the per-step refresh of a captured step's static input
 * buffers (dense features and ids of both towers); sizes in bytes, all pointers 16-byte aligned.
 * ---------------------------------------------------------------------------------------------- */
#define TT_MAX_COPIES 8
/* f32 -> bf16 (round to nearest even) conversions that ride in a hand-over launch: dst[i][0 .. count[i]) = bf16(src[i][...]).  A captured
 * step refreshes the towers' bf16 weight shadows (tt_tower_params.w_proj_bf16 / w_bf16) this way at every hand-over, so whoever changed
 * the weights since the last step -- the captured Adam, an eager step, a checkpoint load -- the shadows are current when the replay
 * reads them.  Sources 16-byte, destinations 8-byte aligned; NULL list or n = 0: none. */
#define TT_MAX_CVT 8
typedef struct tt_cvt_list {
  int32_t n;
  int32_t reserved;
  const float* src[TT_MAX_CVT];
  void* dst[TT_MAX_CVT];
  int64_t count[TT_MAX_CVT];
} tt_cvt_list;
int tt_copy_multi(tt_ctx* ctx, int32_t n, void* const* dst, const void* const* src, const int64_t* bytes,
                  tt_stream stream);

/* Batch hand-over of a graph-replayed step in ONE launch: the n copy segments of tt_copy_multi (the batch's dense features and
 * ids into the graph's static buffers, the step scalars) and, for every side, the fused rows of the batch's ids in key-major
 * order for tt_dedup_plan_keyed_km:  rows_km[side_base + k*B + b] = key_row_offset[k] + clamp(ids[b*K + k], 0, key_vocab[k] - 1)
 * (the lookup's own id -> row rule, src/towers/cat_embed.py:114-117; side_base = sum of B*K of the earlier sides).
 * sides[i].ids is the SOURCE of the hand-over (the incoming batch); out / ld_out / out_dtype are ignored.  1 <= K <= 64 per side. */
int tt_batch_ingest(tt_ctx* ctx, int32_t n, void* const* dst, const void* const* src, const int64_t* bytes,
                    const tt_embed_side* sides, int32_t n_sides, int64_t B, int32_t* rows_km,
                    int32_t* rows_sm /* or NULL: the same rows in slot order, rows_sm[side_base + b*K + k], for tt_embed_lookup_rows_fwd */,
                    int64_t table_rows /* rows of the table the fused rows index: a row outside [0, table_rows) -- key offsets /
                                        * vocabularies of another table -- is stored as table_rows - 1 and raises TT_DEVERR_ROW_RANGE,
                                        * so no consumer of rows_km / rows_sm reads or writes out of bounds; 0: unchecked */,
                    const tt_cvt_list* cvt /* or NULL */, tt_stream stream);

/* The same hand-over STRAIGHT FROM THE DEVICE-RESIDENT FEATURE STORES (two-level gather: pair -> entity row -> dense features
 * and ids -> fused table rows) -- replaces UnifiedBidDataset.__getitem__ + collate_fn_gpu_optimized + _build_batch_kjt
 * (src/towers/pairs/unified_bid_data_loader.py:461-504, :630-684, :827-841) and the host-to-device copy of the batch
 * (scripts/train.py:261-273) with ONE launch and no intermediate batch tensors.  For side i, sample b:
 *   o = order ? order[b] : b ;  e = stores[i].entity[o * stores[i].entity_stride]      (entity row of the pair's side)
 *   dense_out[b, :]   = dense_store[e, :]                                              (f32 copy)
 *   ids_out[b*K + k]  = cat_store[e*K + k]                                             (sample-major: the KJT values())
 *   rows_km[side_base + k*B + b] = key_row_offset[k] + clamp(ids_out[b*K + k], 0, key_vocab[k] - 1)   (as tt_batch_ingest)
 * plus the n copy segments (the step scalars).  sides[i] gives K / key_row_offset / key_vocab (ids, out, ld_out, out_dtype
 * ignored); rows_km may be NULL (no key-major rows wanted).  Entity indices are expected to lie inside the stores (the
 * loader validates the pair list once: KeyError as unified_bid_data_loader.py:495-498); with stores[i].n_rows set, one that
 * does not is clamped into the store rather than read out of bounds.  Bit-identical to tt_batch_gather per
 * side followed by tt_batch_ingest (test). */
typedef struct tt_store_side {
  const int64_t* entity;     /* entity index per pair, read at [o * entity_stride] (an interleaved [P, 2] pair list: stride 2) */
  int64_t entity_stride;
  const float* dense_store;  /* [N, dense_dim] */
  const int64_t* cat_store;  /* [N, K] */
  float* dense_out;          /* [B, dense_dim] */
  int64_t* ids_out;          /* [B * K] */
  int32_t dense_dim;
  int32_t n_rows;            /* entity rows N of the store; > 0: an index outside [0, N) is clamped into it instead of read out of
                              * bounds (the loader has validated the pair list; this guards the device), 0: unchecked */
} tt_store_side;
int tt_batch_ingest_store(tt_ctx* ctx, int32_t n, void* const* dst, const void* const* src, const int64_t* bytes,
                          const tt_embed_side* sides, const tt_store_side* stores, int32_t n_sides, int64_t B,
                          const int64_t* order /* [B] or NULL */, int32_t* rows_km /* or NULL */,
                          int32_t* rows_sm /* or NULL: as tt_batch_ingest */, int64_t table_rows /* as tt_batch_ingest */,
                          const tt_cvt_list* cvt /* or NULL */, tt_stream stream);


/* Hand-over AND lookup in ONE launch: tt_batch_ingest / tt_batch_ingest_store whose tile workgroups -- they hold the batch's
 * clamped fused rows in LDS anyway -- also gather the table rows and write them into the towers' input, i.e. the hand-over
 * followed by tt_embed_lookup_fwd (src/towers/cat_embed.py:103-121, :157-178; tower/base_tower.py:133-139) without the second
 * launch, its re-read of the ids and its boundary; the dense features are copied beside the gathers.  sides[i].out / ld_out /
 * out_dtype say where side i's rows go (as in tt_embed_lookup_fwd: out = first element of sample 0 / key 0); rows_km may be
 * NULL.  E must be 8, 16, 32 or 64 (TT_ERR_UNSUPPORTED otherwise: use the two separate calls).  Results are bit-identical to
 * the two separate calls (test).  The tile workgroups write tt_embed_lookup_set_profile's stamps (gather phase). */
typedef struct tt_ingest_lookup {
  const float* table;  /* fused [table_rows, E] f32 table */
  int64_t table_rows;
  int32_t E;
  int32_t reserved;
} tt_ingest_lookup;
int tt_batch_ingest_lookup(tt_ctx* ctx, int32_t n, void* const* dst, const void* const* src, const int64_t* bytes,
                           const tt_embed_side* sides, int32_t n_sides, int64_t B, int32_t* rows_km /* or NULL */,
                           const tt_ingest_lookup* lookup, const tt_cvt_list* cvt /* or NULL */, tt_stream stream);
int tt_batch_ingest_store_lookup(tt_ctx* ctx, int32_t n, void* const* dst, const void* const* src, const int64_t* bytes,
                                 const tt_embed_side* sides, const tt_store_side* stores, int32_t n_sides, int64_t B,
                                 const int64_t* order /* [B] or NULL */, int32_t* rows_km /* or NULL */,
                                 const tt_ingest_lookup* lookup, const tt_cvt_list* cvt /* or NULL */, tt_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* TWOTOWER_H_ */
