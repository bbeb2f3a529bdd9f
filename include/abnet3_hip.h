/* abnet3_hip.h -- C ABI of libabnet3_hip.so, the MI355X (gfx950) implementation
 * of bootphon/abnet3's Siamese training hot path.
 *
 * The reference has no FFI, plugin registry or native code: its boundary for
 * this path is Python duck-typing (abnet3/gridsearch.py:145-202 builds the
 * objects by class name).  This header is therefore NEW: it is what a
 * maintainer of the reference would bind with ctypes to replace the torch op
 * sequences cited per function below (citations are file:line relative to the
 * reference checkout).  INTEGRATION.md shows the reference-side stubs.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch / C++ types.
 *  - Every pointer named in a signature is a DEVICE pointer unless the comment
 *    says "host".  The caller owns all buffers; the library never allocates device
 *    memory, frees or synchronises (graph-capturable).  Process state it keeps: a
 *    thread-local error string, per-kernel "attribute already set" flags (dynamic LDS
 *    opt-in, set once per device), and the A/B switches it reads from the environment ONCE,
 *    when the library is loaded (ABN_PLANES, ABN_FUSED, ABN_FUSED_MIN_ROWS, ABN_BN_PLANES,
 *    ABN_WGRAD_XCD, ABN_BF16X3_PLANES, ABN_GEMM_TILE, ABN_BWD_PAIR, ABN_DTW_F40, ABN_DTW_PC:
 *    kernel choice only, never results beyond fp32 summation order; no call reads the
 *    environment, and none of them makes a call do less than its contract).  No streams, no events.
 *  - Row-major contiguous fp32 tensors; sizes are int64_t; `stream` is a
 *    hipStream_t passed as void* (NULL = the null stream).
 *  - Return value: 0 = ok, negative = error (ABN_E_*); abn_last_error() gives
 *    the message.  Nothing throws across the ABI.
 */
#ifndef ABNET3_HIP_H
#define ABNET3_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ABN_ABI_VERSION 19
#define ABN_MAX_LAYERS 16

enum { ABN_OK = 0, ABN_E_ARG = -1, ABN_E_LAUNCH = -2, ABN_E_WORKSPACE = -3,
       ABN_E_UNSUPPORTED = -4 };

/* activation_functions table, abnet3/model.py:19-23.  'softmax' (last_non_linearity only,
 * model.py:161-166) is not an epilogue: the tower ends with ABN_ACT_NONE and abn_softmax_rows follows. */
enum { ABN_ACT_NONE = 0, ABN_ACT_SIGMOID = 1, ABN_ACT_RELU = 2, ABN_ACT_TANH = 3 };

/* abnet3/loss.py:37 (coscos2), :70 (cosmargin) */
enum { ABN_LOSS_COSCOS2 = 0, ABN_LOSS_COSMARGIN = 1 };

/* label element types accepted by the pair loss: the reference compares with
 * torch.eq(y, 1) / torch.eq(y, -1) on whatever dtype arrives (int64 in
 * test/test_loss.py:28, float64 from abnet3/dataloader.py:206,231) */
enum { ABN_Y_I8 = 0, ABN_Y_I32 = 1, ABN_Y_I64 = 2, ABN_Y_F32 = 3, ABN_Y_F64 = 4 };

/* optimizer_type, abnet3/trainer.py:68-87 (torch.optim defaults otherwise) */
enum { ABN_OPT_SGD = 0, ABN_OPT_ADADELTA = 1, ABN_OPT_ADAM = 2,
       ABN_OPT_ADAGRAD = 3, ABN_OPT_RMSPROP = 4 };

/* abn_tower_desc.precision: arithmetic of the tower GEMMs (see the field's comment) */
enum { ABN_PREC_F32 = 0, ABN_PREC_BF16 = 1, ABN_PREC_BF16X3 = 2, ABN_PREC_F16X2 = 3 };

/* abn_tower_desc.bn_sync_fn: SUM-all-reduces n float64 values at a DEVICE pointer in place over the replicas, in the
 * stream's order (torch.distributed.all_reduce / ncclAllReduce on that stream); returns 0 on success. */
typedef int (*abn_allreduce_fn)(void* ctx, void* device_doubles, int64_t n, void* stream);

int abn_abi_version(void);
const char* abn_last_error(void);          /* host string, thread-local */

/* One SiameseNetwork tower (abnet3/model.py:110-170): n_layers Linear layers
 * dims[0] -> dims[1] -> ... -> dims[n_layers], each followed by Dropout (caller-
 * drawn masks, see drop_mask), optional BatchNorm1d, and an activation (`act`, or `last_act` for
 * the output layer, model.py:161-166).  W[l] is [dims[l+1], dims[l]] row-major
 * exactly as nn.Linear stores it; gradients are written to dW/db/dbn_* (same
 * shapes).  All pointers are device pointers held in this HOST struct. */
typedef struct abn_tower_desc {
    int32_t n_layers;
    int32_t act;
    int32_t last_act;
    int32_t batch_norm;
    int64_t dims[ABN_MAX_LAYERS + 1];
    const float* W[ABN_MAX_LAYERS];
    const float* b[ABN_MAX_LAYERS];
    const float* bn_w[ABN_MAX_LAYERS];     /* gamma; NULL when !batch_norm */
    const float* bn_b[ABN_MAX_LAYERS];     /* beta */
    float* bn_rm[ABN_MAX_LAYERS];          /* running_mean (updated in train) */
    float* bn_rv[ABN_MAX_LAYERS];          /* running_var */
    float* dW[ABN_MAX_LAYERS];             /* backward outputs; may be NULL   */
    float* db[ABN_MAX_LAYERS];             /* when only forward is called     */
    float* dbn_w[ABN_MAX_LAYERS];
    float* dbn_b[ABN_MAX_LAYERS];
    /* nn.Dropout(p) between Linear and BatchNorm/activation (model.py:137,148,157):
     * per layer a [rows, dims[l+1]] multiplier (0 or 1/(1-p)) drawn by the caller;
     * NULL = identity (p = 0 or eval mode).  The same masks must be passed to
     * the backward call. */
    const float* drop_mask[ABN_MAX_LAYERS];
    /* 0 = fp32 on the exact-fp32 MFMA (the parity path, default); 1 = throughput mode:
     * matrix operands rounded to bf16 at fragment time, fp32 accumulation, everything
     * else (storage, BatchNorm, loss, optimizer) still fp32.  NOT within the 1e-5 bar.
     * 2 = bf16 x 3: every operand split into three bf16 terms, six bf16 MFMAs per product
     * block: fp32-grade results (as close to a float64 evaluation as mode 0 is) that are not
     * bit-identical to mode 0.
     * 3 = fp16 x 2 (the Python classes' default): every operand scaled by a power of two (per operand row /
     * 32-row weight block) and split into two fp16 terms (22 significant bits), three fp16 MFMAs per product
     * block, the scales taken out again in the epilogue (exactly): the same grade at half the MFMAs and two
     * thirds of the operand stream.  On the operand-plane kernels only; where a call falls back to the GEMM
     * kernels it runs as mode 2. */
    int32_t precision;
    /* backward only: d_out already IS d loss / d z of the output layer (abn_pair_loss_dz), so
     * the activation derivative / dropout step in front of the last layer's GEMMs is skipped.
     * Not with batch_norm (its backward needs d loss / d a). */
    int32_t d_out_is_dz;
    /* backward only: leave the weight gradients as unreduced split-K slabs in `scratch`; the
     * caller finishes with abn_tower_reduce_step (reduction + optimizer step in one launch;
     * BatchNorm's gamma / beta gradients are final either way). */
    int32_t defer_reduce;
    /* Optional persistent image of the weights as MFMA operand fragments (the split arithmetics, precision 1-3):
     * wpack = abn_tower_wpack_floats() floats owned by the caller, zero before the first use, or
     * NULL (the forward then builds the image inside its workspace every call).  wpack_valid != 0:
     * the caller vouches that the image matches W as it stands -- it does after a forward that
     * was given the buffer with wpack_valid = 0 (which rebuilds it: ~6 us) until anything writes
     * to W (abn_optimizer_step and abn_tower_reduce_step included).  What it buys: repeated
     * forwards with unchanged weights (embedding extraction) skip the rebuild.  (Keeping the image
     * in step inside abn_tower_reduce_step was measured: the transposed half is 2-byte scattered
     * stores, +10 us on that launch against the 6 us saved.) */
    int32_t wpack_valid;
    /* forward only: no backward will follow this forward (inference): it may skip whatever it
     * would store for one (the default arithmetic writes more than half of its bytes for the
     * backward).  A backward after such a forward reads garbage. */
    int32_t forward_only;
    /* backward only, data-parallel overlap (operand-plane launches without BatchNorm; ABN_E_UNSUPPORTED elsewhere):
     * the backward in two calls on the same workspace and scratch so that the caller can start the all-reduce of the
     * upper layers' gradients while the lower layers' are still being computed.  0 = everything (default);
     * 1 = the data-gradient launches and the weight gradients (with their slab reduction) of layers >= wgrad_split;
     * 2 = the weight gradients (and reduction) of layers < wgrad_split -- after a part-1 call (d_out is not read and may be
     * NULL), no defer_reduce. */
    int32_t wgrad_part;
    void* wpack;
    /* Dropout drawn inside the kernels instead of read from drop_mask (the split arithmetics only;
     * where drop_mask[l] is given it wins): drop_seed = device pointer to one uint64 the caller
     * draws per forward (NULL: off), drop_p = nn.Dropout's p.  The multiplier of element
     * (layer, row, feature) is a hash of (seed, layer, row, feature): 0 with probability p
     * (to 2^-16), else 1 / (1 - p) -- the backward, given the same descriptor fields, regenerates
     * it.  A forward / backward that cannot run on the operand-plane kernels
     * (abn_tower_uses_planes) returns ABN_E_UNSUPPORTED when only a seed is given. */
    const void* drop_seed;
    float drop_p;
    int32_t reserved2_;
    /* Cross-replica BatchNorm statistics (data-parallel training; SURVEY.md 8e's exact mode): bn_sync_fn != NULL with bn_sync_world >= 1
     * (a group of ONE replica is a group: its reduction is the identity, the launches are the group's)
     * makes a TRAINING forward / backward of a BatchNorm tower sum its per-call statistics -- [sum z, sum z^2] per
     * layer in the forward, [sum dy, sum dy xhat] in the backward, float64 -- over the replicas through bn_sync_fn
     * (called on the host between two launches, once per layer and direction) and normalise with the replicas' row
     * count, which travels with the forward's sums (n_calls more values: the replicas' batches may differ in size):
     * R replicas on B_r rows each then step like one process on sum B_r rows.  EVERY replica of the group must make the
     * same calls (the exchange is a collective: the caller agrees beforehand which steps take this path).
     * bn_sync_fn NULL or bn_sync_world 0: per-replica statistics.  Operand-plane launches only (ABN_E_UNSUPPORTED otherwise). */
    int32_t bn_sync_world;
    int32_t wgrad_split;                   /* see wgrad_part */
    abn_allreduce_fn bn_sync_fn;
    void* bn_sync_ctx;
    /* A PADDED batch through a BatchNorm tower in training (device int32, or NULL: every row is real): only the first
     * *n_valid rows of EVERY forward_once call are real (abn_gather_pairs writes such batches: zero rows behind the real
     * ones, tower 2 starting at row rows / n_calls).  The batch statistics, the running statistics' update and the backward
     * then span the real rows only and the padded rows get no gradient and give none, so that one captured step serves
     * every batch size of a bucket (see abn_tower_backward_loss's n_valid: the same pointer).  The forward and the
     * backward of a step must be given the same value.  Only on the BatchNorm layer launches (abn_tower_path =
     * ABN_PATH_BN_LAYERS) with per-replica statistics: ABN_E_UNSUPPORTED elsewhere.  Towers without BatchNorm and
     * inference forwards ignore it (their rows do not see each other). */
    const int32_t* n_valid;
    /* BatchNorm1d.num_batches_tracked of every layer (device int64 each, or NULL): a TRAINING forward of a batch_norm tower
     * adds n_calls to each -- torch's BatchNorm1d counts its training calls, and the reference's forward runs forward_once
     * twice (abnet3/model.py:194-195) -- inside the forward's own launches where it can (ABN_PATH_BN_TOWER), else in one
     * small launch behind them.  Nothing else reads them (momentum is the fixed 0.1). */
    void* bn_nbt[ABN_MAX_LAYERS];
    /* Optional sync buffer of the resident BatchNorm tower (ABN_PATH_BN_TOWER): abn_tower_sync_ws_bytes() bytes, 16-byte
     * aligned, owned by the caller, ZERO before its first use and never written by the caller afterwards; one per tower
     * and stream (the forward and the backward of a step share it; two streams driving one tower need two).  It holds the
     * launch counter the kernels' hand-over tags derive from and the hand-over granules themselves.  NULL: BatchNorm
     * training runs one launch per layer (ABN_PATH_BN_LAYERS).  Should a launch ever give up on a hand-over (the grid was
     * not resident: the outputs then read NaN) the buffer's failure word stays set: zero the buffer again.  While it is
     * set abn_tower_reduce_step drops its step (parameters, state and gradients untouched), and a backward WITHOUT
     * defer_reduce (the data-parallel step: the caller all-reduces the gradient next) writes a ZERO gradient. */
    void* sync_ws;
    /* Optional, backward with defer_reduce only (ABI v19): the workspace of the forward whose gradients are pending (the
     * `ws` both calls were given) and that forward's n_calls, lent to abn_tower_reduce_step.  Small batches on the
     * layer-per-launch kernels (ABN_PATH_WIDE, fp16 x 2) then skip the split-K weight-gradient launch in the backward:
     * abn_tower_reduce_step computes every layer's weight gradient over ALL rows and applies the optimizer's rule in ONE
     * launch (csrc/tower_wgrad_step.h: a workgroup per 64 x 64 tile of [dW | db], no slabs).  The workspace must stay
     * untouched until that call.  NULL: weight gradients as slabs in the backward, their sum in abn_tower_reduce_step. */
    const float* fwd_ws;
    int64_t fwd_calls;
    /* Optional (ABI v19): the step reads its batch from a PLAN of the whole pass instead of from x1 / x2 / y -- see
     * abn_step_source below.  Layer-per-launch kernels (ABN_PATH_WIDE) in training, two forward_once calls, the pair loss
     * inside the backward (abn_tower_backward_loss); ABN_E_UNSUPPORTED elsewhere. */
    const struct abn_step_source* source;
} abn_tower_desc;

/* A pass's batches as the trainer's batch plan holds them (abnet3/dataloader.py:166-261: every batch = the frame pairs of
 * some word pairs, in the order the reference's iterator yields them): pair p of the plan aligns row idx1[p] of `table`
 * (tower 1) with row idx2[p] (tower 2) under label labels[p]; step s of the pass takes pairs [steps[2 s], steps[2 s] +
 * steps[2 s + 1]).  With abn_tower_desc.source set, a training step needs NO gather launch and no per-step argument: the
 * first layer's launch stages its 32 rows straight from the table (rows behind the step's last pair: zero rows, as
 * abn_gather_pairs pads them), the pair loss reads its labels and its real-pair count from here, and the step's last launch
 * (abn_tower_reduce_step) advances *step_ctr -- a captured hipGraph replays unchanged for every batch of its size.
 * x1 / x2 still name the (unused) input buffers of the padded size: rows = 2 x the padded pair count as before.
 * All arrays on the device; the struct itself is host memory read during the calls. */
typedef struct abn_step_source {
    const float* table;            /* [table_rows, dims[0]] */
    int64_t table_rows;
    const int64_t* idx1;
    const int64_t* idx2;
    const void* labels;            /* dtype: abn_tower_backward_loss's y_dtype (its y argument is ignored) */
    const int64_t* steps;          /* [n_steps][2]: first pair, pairs */
    int32_t* step_ctr;             /* the step being run; += 1 by abn_tower_reduce_step */
} abn_step_source;

/* A ready-made abn_allreduce_fn for abn_tower_desc.bn_sync_fn over RCCL, so that no host language stands between
 * two launches of a data-parallel BatchNorm step: ctx = an abn_rccl_ctx the caller fills once -- `comm` its ncclComm_t,
 * `all_reduce` the address of RCCL's ncclAllReduce (the library does not link RCCL: the caller already has it loaded,
 * e.g. torch's librccl.so) -- and the function issues ncclAllReduce(buf, buf, n, ncclFloat64, ncclSum, comm, stream)
 * in place on the launch stream.  Returns 0, or 1 when RCCL reports an error. */
typedef struct abn_rccl_ctx {
    void* comm;
    void* all_reduce;       /* ncclResult_t (*)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) */
    int64_t calls;          /* counts the all-reduces issued through this context (the caller's to read and reset) */
} abn_rccl_ctx;
int abn_rccl_allreduce_f64(void* ctx, void* device_doubles, int64_t n, void* stream);

/* Workspace of one forward call (what the backward needs: the saved activations, for the
 * default arithmetic also the weights as MFMA operand fragments; its layout is the library's
 * own and depends on the descriptor), in floats, and the offset inside it of the
 * [rows, dims[n_layers]] output embedding -- the only public position in it. */
int64_t abn_tower_ws_floats(const abn_tower_desc* t, int64_t rows, int64_t n_calls);
int64_t abn_tower_out_offset(const abn_tower_desc* t, int64_t rows, int64_t n_calls);
/* Scratch of one backward call (split-K slabs + the dZ of every layer), in floats. */
int64_t abn_tower_bwd_scratch_floats(const abn_tower_desc* t, int64_t rows);
/* Size of the optional abn_tower_desc.sync_ws buffer, in bytes (host function, no device call). */
int64_t abn_tower_sync_ws_bytes(void);
/* Size of the optional abn_tower_desc.wpack buffer, in floats (0: this tower has no such image). */
int64_t abn_tower_wpack_floats(const abn_tower_desc* t);
/* 1 when abn_tower_forward(train) / the backward after it with these arguments run on the
 * operand-plane kernels (the ones that read, and with wpack_valid = 0 rebuild, wpack), 0 when on
 * the per-layer GEMMs, < 0 on a bad descriptor.  Depends on the descriptor, the row count, pointer
 * alignment and the library's environment switches.  train = 0 asks about the inference forward:
 * a batch_norm tower takes the operand-plane kernel there (forward_only descriptors: running
 * statistics folded into the epilogue), and only there. */
int abn_tower_uses_planes(const abn_tower_desc* t, int64_t rows, const float* x1, const float* x2,
                          const float* ws, int train);

/* Which kernel family a call takes, and the arithmetic its GEMMs compute in -- a PURE function of
 * the arguments and the environment switches (the library keeps no record of past calls):
 * backward = 0 asks about abn_tower_forward(t, x1, x2, rows, n_calls, train, ws), backward = 1 about
 * the abn_tower_backward / abn_tower_backward_loss after it.  Returns one of ABN_PATH_* (< 0: bad
 * descriptor); *precision_out (may be NULL) receives the abn_tower_desc.precision code actually
 * used: a 'f16x2' tower that falls to the GEMM kernels (widths > 512 or not multiples of 4,
 * BatchNorm on < 256 rows, ABN_PLANES=0) computes in bf16x3 there, and this is where a caller
 * learns it. */
enum {
    ABN_PATH_PER_LAYER = 0,        /* one GEMM launch per layer (gemm_f32.h) */
    ABN_PATH_FUSED_F32 = 1,        /* the fp32 tower in one launch (tower_fused.h) */
    ABN_PATH_PLANES = 2,           /* operand-plane chain, everything a backward needs is kept */
    ABN_PATH_PLANES_INFER = 3,     /* operand-plane chain, inference (forward_only) */
    ABN_PATH_PLANES_INFER_BN = 4,  /* ... with BatchNorm's running statistics in the epilogue */
    ABN_PATH_BN_LAYERS = 5,        /* BatchNorm training: one operand-plane launch per layer */
    ABN_PATH_WIDE = 6,             /* small batches: one launch per layer over up to 8 workgroups per row block */
    ABN_PATH_BN_TOWER = 7          /* BatchNorm training: the whole tower in ONE resident launch per direction, grid barriers
                                      between the layers (csrc/tower_bn_persist.h): batches of 256 .. 32 x (CUs of the device)
                                      tower rows with per-replica statistics and a sync buffer (abn_tower_desc.sync_ws);
                                      ABN_BN_PERSIST=0 keeps ABN_PATH_BN_LAYERS.  The answer therefore also depends on the
                                      current device's CU count. */
};
int abn_tower_path(const abn_tower_desc* t, const float* x1, const float* x2, int64_t rows,
                   int64_t n_calls, int train, const float* ws, int backward,
                   int32_t* precision_out);

/* Float offset of one of the library's operand images, for callers that decode them (the tests'
 * decoders, tests/planes_decode.py) -- which = 0 packed W_l, 1 packed W_l^T (relative to
 * abn_tower_desc.wpack when given, else to the forward workspace), 2 the transposed image of
 * [input of layer l | 1], 4 a BatchNorm layer's z_l, 5 the row-major output of layer l (forward
 * workspace), 3 the transposed image of dZ_l (backward scratch); -1 when there is no such image.
 * The formats are csrc/tower_planes.h's; they change with ABN_ABI_VERSION. */
int64_t abn_tower_image_offset(const abn_tower_desc* t, int64_t rows, int64_t n_calls, int which,
                               int l);

/* The library reads its A/B switches (ABN_PLANES, ABN_WIDE, ABN_DTW_PC, ... : kernel choice only)
 * from the environment once, when it is loaded; this reads them again (tests and A/B tools that
 * change one inside a process). */
void abn_reload_switches(void);

/* SiameseNetwork.forward_once / forward, abnet3/model.py:179-196.
 * `rows` input rows in total, made of `n_calls` forward_once calls of
 * rows/n_calls rows each (1 = embed, 2 = Siamese pair): call c reads rows
 * [c*rows/n_calls, ...) from x1 (c == 0) or x2 (c == 1; x2 may be NULL when
 * x1 already holds all rows contiguously).  BatchNorm statistics and running
 * stat updates are per call, as in the reference (two updates per Siamese
 * forward).  train != 0: batch statistics, activations saved in ws;
 * train == 0: running statistics.  Output: ws + abn_tower_out_offset().
 * A batch_norm tower in the default arithmetic runs one operand-plane launch per layer in
 * training (its backward likewise), and the single-launch forward with the running statistics folded in
 * when train == 0 and forward_only != 0; otherwise the per-layer kernels.  (ABN_PATH_BN_TOWER: where the whole grid is
 * resident at once -- at most one workgroup of 32 rows per CU -- training runs as ONE launch per direction with grid
 * hand-overs between the layers; a hand-over that cannot complete gives up after a bounded spin and the launch leaves NaN
 * in the embeddings / the loss instead of hanging.)  Results agree to
 * rounding; forward and backward of one pass must see the same environment switches. */
int abn_tower_forward(const abn_tower_desc* t, const float* x1, const float* x2,
                      int64_t rows, int64_t n_calls, int train, float* ws,
                      void* stream);

/* Autograd of the above (what loss.backward() runs, abnet3/trainer.py:239).
 * d_out: [rows, dims[n_layers]] gradient w.r.t. the output embeddings.
 * Writes dW/db (+dbn_w/dbn_b) summed over all rows (both towers), and dx
 * ([rows, dims[0]], may be NULL: the reference never needs it).  t, x1, x2, rows, n_calls and
 * ws must be the forward call's (ws unchanged since). */
int abn_tower_backward(const abn_tower_desc* t, const float* x1, const float* x2,
                       const float* d_out, int64_t rows, int64_t n_calls,
                       const float* ws, float* scratch, int64_t scratch_floats,
                       float* dx, void* stream);

/* ONE of the two launches of an ABN_PATH_PLANES backward, for per-launch measurements (bench.py's
 * roofline legs): part = 1 the data-gradient chain, 2 the weight gradients of every layer -- after
 * a complete abn_tower_backward with the same arguments has left the other launch's output in
 * place.  The split-K slabs are never reduced (dW / db are not written).  ABN_E_UNSUPPORTED on
 * every other path. */
int abn_tower_backward_launch(const abn_tower_desc* t, const float* x1, const float* x2,
                              const float* d_out, int64_t rows, int64_t n_calls, const float* ws,
                              float* scratch, int64_t scratch_floats, int part, void* stream);

/* abn_pair_loss_dz + abn_tower_backward in the backward's own launches (abnet3/trainer.py:238-239:
 * loss = self.loss(emb1, emb2, y); loss.backward()): rows = 2 B tower rows, [tower 1: pairs 0..B-1 |
 * tower 2: pairs 0..B-1], whose embeddings the forward left in ws; the first phase of the data
 * gradient chain computes the loss and d loss / d z of the output layer (same arithmetic, fp64 per
 * pair) instead of reading d_out; for a BatchNorm tower on its layer launches (per-replica statistics,
 * wgrad_part 0) the launch that sums the output layer's dy and dy xhat does, and leaves d loss / d a for
 * the top layer's launch.  Only for towers the operand-plane kernels take (default arithmetic, widths <= 512
 * and multiples of 4): ABN_E_UNSUPPORTED otherwise, and the caller uses the two separate calls.  loss_ws: abn_tower_backward_loss_ws_bytes(rows) bytes whose
 * first 8 (a ticket counter) are zero before the first call and are left zero.
 * n_valid (device int32, or NULL): a PADDED batch -- only the first *n_valid pairs of the B = rows / 2
 * are real (abn_gather_pairs writes such batches: zero rows behind the real ones, tower 2 starting at
 * row B); the others contribute no loss term and their d loss / d z is zero, so nothing of them reaches
 * a gradient, and avg divides by *n_valid.  One captured step then serves every batch size of a bucket
 * (the reference's batches of 8 word pairs have a different number of frame pairs every step,
 * abnet3/dataloader.py:248-255).  loss_accum (device double, or NULL): the call's loss is also added to
 * it -- the epoch's running sum the reference keeps on the host (abnet3/trainer.py:242). */
int64_t abn_tower_backward_loss_ws_bytes(int64_t rows);
int abn_tower_backward_loss(const abn_tower_desc* t, const float* x1, const float* x2, const void* y,
                            int y_dtype, int loss_kind, float margin, int avg, int64_t rows,
                            const float* ws, float* scratch, int64_t scratch_floats,
                            float* loss_out, void* loss_ws, const int32_t* n_valid,
                            double* loss_accum, void* stream);

/* Finishes an abn_tower_backward that ran with defer_reduce = 1 (same descriptor, rows and
 * scratch): sums the split-K slabs in their fixed order, writes the gradients to dW / db AND
 * applies abn_optimizer_step's update to the same elements -- one launch for what is otherwise
 * the slab reduction followed by the optimizer step (abnet3/trainer.py:239-240 back to back, no
 * gradient exchange in between: single process).  With batch_norm the BatchNorm tensors (whose gradients
 * the backward wrote directly) are stepped by the same launch.  params / grads / state1 / state2 are the flat
 * buffers (n floats each) that hold every tensor of the descriptor at the same offsets: element j
 * of layer l's weight lives at (dW[l] - grads) + j in all four. */
int abn_tower_reduce_step(const abn_tower_desc* t, int64_t rows, const float* scratch,
                          int64_t scratch_floats, int kind, float* params, float* grads,
                          float* state1, float* state2, int64_t n, float lr, float hp0,
                          float hp1, float eps, int64_t step, float grad_scale, void* stream);

/* The data-parallel step's gradient exchange (the reference has no multi-process path: this sits
 * between loss.backward() and optimizer.step(), abnet3/trainer.py:239 -> :240) as a ONE-SHOT all-reduce over peer-mapped
 * mailboxes instead of a ring: every rank pushes shard s of its bucket into rank s's mailbox, sums the world's contributions
 * to its own shard in RANK ORDER (deterministic: replicas stay bit-identical) and pushes the reduced shard to everybody --
 * two hops over all xGMI links at once, one kernel launch per rank on the caller's stream (graph-capturable, no host call
 * inside).  The caller owns the mailboxes: rank s allocates abn_oneshot_mail_bytes(world, cap_floats) bytes of FINE-GRAINED
 * device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)), zeroes it once, exports it (hipIpcGetMemHandle) and maps
 * every peer's (hipIpcOpenMemHandle) into mail[]; mail[rank] is its own.  Every rank of the group makes the same sequence of
 * calls (same n).  SUM of fp32 in place over buf[0 .. n), n % 4 == 0, n <= cap_floats, buf 16-byte aligned.  A rank whose
 * peer never arrives gives up after a bounded spin, leaves NaN in buf and its mailbox's failure word set (zero the mailboxes
 * again before the next use). */
#define ABN_ONESHOT_MAX_RANKS 8
typedef struct abn_oneshot_ctx {
    int32_t rank, world;
    void* mail[ABN_ONESHOT_MAX_RANKS];     /* mail[s]: rank s's mailbox as mapped in THIS process */
    int64_t cap_floats;                    /* most floats one call reduces (what the mailboxes were sized for) */
} abn_oneshot_ctx;
int64_t abn_oneshot_mail_bytes(int32_t world, int64_t cap_floats);
int abn_allreduce_oneshot(const abn_oneshot_ctx* ctx, float* buf, int64_t n, void* stream);

/* One nn.Linear at a time with the same kernels (what abn_tower_* chains):
 *   forward  y = act(x W^T + b)                       (addmm + activation)
 *   dgrad    dx = (dz W) * act'(a_prev)  [a_prev NULL: plain dz W]
 *   wgrad    dW = dz^T a_in, db = colsum(dz); scratch: split-K slabs
 * x [rows,in], W [out,in], y/dz [rows,out], a_prev/dx/a_in [rows,in]. */
int abn_linear_forward(const float* x, const float* W, const float* b, int64_t rows,
                       int64_t in_dim, int64_t out_dim, int act, float* y, void* stream);
int abn_linear_dgrad(const float* dz, const float* W, int64_t rows, int64_t in_dim,
                     int64_t out_dim, const float* a_prev, int act_prev, float* dx,
                     void* stream);
int64_t abn_linear_wgrad_scratch_floats(int64_t rows, int64_t in_dim, int64_t out_dim);
int abn_linear_wgrad(const float* dz, const float* a_in, int64_t rows, int64_t in_dim,
                     int64_t out_dim, float* dW, float* db, float* scratch,
                     int64_t scratch_floats, void* stream);
/* dgrad + wgrad of one nn.Linear the way abn_tower_backward issues them: both read dz
 * only, so they go out as ONE grid (the dgrad's workgroups take over the CUs as the
 * wgrad's retire), followed by the slab reduction.  a_in [rows,in] is the layer's input =
 * the previous activation (act_prev = 0: no activation derivative).  scratch as for
 * abn_linear_wgrad. */
int abn_linear_backward(const float* dz, const float* W, const float* a_in, int64_t rows,
                        int64_t in_dim, int64_t out_dim, int act_prev, float* dW,
                        float* db, float* dx, float* scratch, int64_t scratch_floats,
                        void* stream);
/* ... in the arithmetic abn_tower_desc.precision names (0 exact fp32 = abn_linear_backward,
 * 1 bf16 operands, 2 bf16 x 3, 3 = run as 2: fp16 x 2 exists on the operand planes only): the grid abn_tower_backward
 * issues for that network.  dW and db
 * both NULL: only that grid runs and the split-K slabs stay unreduced in scratch (bench.py times
 * the grid alone this way). */
int abn_linear_backward_prec(const float* dz, const float* W, const float* a_in, int64_t rows,
                             int64_t in_dim, int64_t out_dim, int act_prev, int precision,
                             float* dW, float* db, float* dx, float* scratch,
                             int64_t scratch_floats, void* stream);

/* coscos2.forward / cosmargin.forward fused with their backward,
 * abnet3/loss.py:46-67 and :85-105 (nn.CosineSimilarity(dim=1, eps=1e-6)).
 * loss_out: device scalar (fp32).  de1/de2: [B, D] gradients of the (already
 * /B-scaled when avg) loss; both may be NULL for a forward-only call.
 * ws: abn_pair_loss_ws_bytes(B) bytes of device scratch whose FIRST 8 bytes (a ticket
 * counter) must be zero before the first call on this buffer; every call leaves them zero
 * (ONE launch: the workgroup that finishes last sums the per-workgroup partial losses in a
 * fixed order).  Calls sharing a ws buffer must be ordered on one stream. */
int64_t abn_pair_loss_ws_bytes(int64_t B);
int abn_pair_loss(const float* e1, const float* e2, const void* y, int y_dtype,
                  int64_t B, int64_t D, int kind, float margin, int avg,
                  float* loss_out, float* de1, float* de2, void* ws,
                  void* stream);
/* abn_pair_loss on a PADDED batch (see abn_gather_pairs / abn_tower_backward_loss): only the first *n_valid
 * (device int32, NULL = all B) pairs are real -- the others add nothing to the loss and get zero gradient
 * rows, a mean loss divides by *n_valid -- and the loss is also added to *loss_accum (device double, may be
 * NULL): the evaluation pass of a trainer whose batches sit in bucket-sized static buffers
 * (abnet3/trainer.py:244-248: the dev loss summed over the batches). */
int abn_pair_loss_padded(const float* e1, const float* e2, const void* y, int y_dtype,
                         int64_t B, int64_t D, int kind, float margin, int avg,
                         const int32_t* n_valid, float* loss_out, double* loss_accum,
                         float* de1, float* de2, void* ws, void* stream);
/* The same with the output layer's activation derivative (and dropout multipliers, mask1 /
 * mask2 [B, D] or NULL) folded in: e1 / e2 are the tower's outputs act(z), and dz1 / dz2
 * receive d loss / d z = d loss / d e * act'(e) [* mask] -- the first step of
 * loss.backward() (abnet3/trainer.py:239) through a tower without BatchNorm.  Hand
 * [dz1; dz2] to abn_tower_backward with abn_tower_desc.d_out_is_dz = 1. */
int abn_pair_loss_dz(const float* e1, const float* e2, const void* y, int y_dtype,
                     int64_t B, int64_t D, int kind, float margin, int avg, int act,
                     const float* mask1, const float* mask2, float* loss_out,
                     float* dz1, float* dz2, void* ws, void* stream);

/* torch.optim.{SGD(momentum),Adadelta,Adam,Adagrad,RMSprop}.step over one flat
 * fp32 parameter buffer (abnet3/trainer.py:68-87, :240).  state1/state2: flat
 * buffers of n floats, zero-initialised by the caller before the first step
 * (momentum_buffer | square_avg, acc_delta | exp_avg, exp_avg_sq | sum | -).
 * `step` counts from 1.  hp0/hp1: momentum | rho=0.9 | beta1,beta2 | - | alpha.
 * grad_scale multiplies the gradient first (1/world_size for avg=True DP). */
int abn_optimizer_step(int kind, float* params, const float* grads, float* state1,
                       float* state2, int64_t n, float lr, float hp0, float hp1,
                       float eps, int64_t step, float grad_scale, void* stream);

/* abnet3/utils.py:40-60 (cosine_distance) + :147-153 (get_dtw_alignment ->
 * third-party dtw.DTW) for a batch of token pairs.  Pair p aligns rows
 * [off1[p], off1[p]+n1[p]) of feats1 ([rows1, D] fp32, device) with rows
 * [off2[p], off2[p]+n2[p]) of feats2.  The per-pair metadata (off*, n*) are
 * HOST arrays -- token boundaries come from the pairs file, host data in the
 * reference too -- which the call stages into `ws` through `host_stage` (a
 * caller-owned host buffer, ideally pinned, that must stay untouched until the
 * stream has passed this call).  path1/path2: [npairs, path_stride] int32 (device);
 * the path of pair p, from (0,0) to (n1-1,n2-1), is RIGHT-ALIGNED in its row: entries
 * [path_stride - path_len[p], path_stride), in forward order (the traceback walks from the
 * end and writes each cell where it belongs; the rest of the row is not touched).
 * path_len[p] = 0 marks a pair the reference would have dropped (NaN distance,
 * abnet3/dataloader.py:188-191) or an empty token.  total_cost (device, [npairs] f64) may be
 * NULL.  Tokens of any length.  path_stride >= max(n1 + n2 - 1).
 * One fused kernel per call (distances on the fp32 matrix cores, the reference's division /
 * acosf / pi per cell, float64 dynamic programme, 2-bit back-pointers) plus a traceback
 * kernel, all on `stream`: no library-owned streams, events or other global state.  path_len and total_cost need no
 * clearing by the caller: every pair's entries are written by one of the call's launches.  The cost
 * matrix is never materialised: the workspace holds ~0.26 B per cell (back-pointers) plus
 * per-workgroup boundary rows.  rows1 / rows2 bound the offsets (checked). */
int64_t abn_dtw_ws_bytes(const int32_t* n1_host, const int32_t* n2_host,
                         int64_t npairs, int64_t rows1, int64_t rows2);
int64_t abn_dtw_host_stage_bytes(const int32_t* n1_host, const int32_t* n2_host,
                                 int64_t npairs);
int abn_dtw_batched(const float* feats1, int64_t rows1, const float* feats2,
                    int64_t rows2, const int64_t* off1_host, const int32_t* n1_host,
                    const int64_t* off2_host, const int32_t* n2_host, int64_t npairs,
                    int64_t D, int32_t* path1, int32_t* path2, int32_t* path_len,
                    int64_t path_stride, double* total_cost, void* ws,
                    int64_t ws_bytes, void* host_stage, int64_t host_stage_bytes,
                    void* stream);
/* The same call with the traceback BESIDE the fill kernel instead of behind it (ABI v19): `side_stream` is a second
 * stream of the caller's (same device; NULL or == stream: exactly abn_dtw_batched).  The fill kernel stores what the
 * traceback reads of a pair write-through and flags the pair when it is complete; a traceback launch on `side_stream`
 * polls the flags (bounded, asleep in between) and walks each pair as it completes, so that the ~0.2 ms a 10 000-pair
 * traceback takes -- one pair's chain of dependent window fetches, a few per cent of the chip -- run under the fill's
 * 2.7 ms; `stream` then waits for it and sweeps up whatever it left (normally nothing).  Same results, bit for bit.
 * The library orders the two streams with two events it keeps per host thread and device (made at the first call, the
 * only state this entry point adds to the process); on return both streams carry work of this call and `stream` alone is
 * behind all of it: the caller
 * synchronises with `stream` as before and need not look at `side_stream` again.  40-value frames (the gang kernel);
 * every other frame width runs as abn_dtw_batched. */
int abn_dtw_batched_overlap(const float* feats1, int64_t rows1, const float* feats2,
                            int64_t rows2, const int64_t* off1_host, const int32_t* n1_host,
                            const int64_t* off2_host, const int32_t* n2_host, int64_t npairs,
                            int64_t D, int32_t* path1, int32_t* path2, int32_t* path_len,
                            int64_t path_stride, double* total_cost, void* ws,
                            int64_t ws_bytes, void* host_stage, int64_t host_stage_bytes,
                            void* stream, void* side_stream);
/* The distance matrix alone (utils.py:40-60), one pair, float64 [N, M] out.  The
 * reference computes in the precision of its inputs (utils.py:41-42: both float32 or
 * both float64): abn_cosine_distance is the float32 arithmetic of the hot path (the
 * same cell function as abn_dtw_batched: numpy's norm summation order, one fma chain
 * per dot product, one division, glibc's acosf, / float32(pi) -- bit-identical to the
 * reference's output on its plain numpy/libm path), abn_cosine_distance_f64 the same
 * statements in double.  *bad_flag (device int32, may be NULL) is set when an entry is
 * NaN or negative -- the reference's `assert np.all(d >= 0)` (utils.py:59). */
int abn_cosine_distance(const float* x, int64_t N, const float* y, int64_t M,
                        int64_t D, double* d, int32_t* bad_flag, void* stream);
int abn_cosine_distance_f64(const double* x, int64_t N, const double* y, int64_t M,
                            int64_t D, double* d, int32_t* bad_flag, void* stream);

/* np.arccos on float32 as the reference's cosine_distance evaluates it on numpy's plain
 * path (utils.py:50,53: scipy.arccos = np.arccos = libm acosf): out[i] = acosf(x[i]),
 * bit-identical to glibc 2.35's acosf for every float32 argument (NaN outside [-1, 1]);
 * over_pi != 0: out[i] = acosf(x[i]) / float32(pi), the correctly rounded float32 quotient
 * (utils.py:53).  The cell function of abn_dtw_batched, exposed for verification.
 * over_pi bit 0: divide by pi; bit 1: arguments with 2^-26 < |x| < 0.5 take the straight-line
 * statements the gang kernel uses when a wavefront's cells all lie in that range (same bits). */
int abn_arccos_f32(const float* x, int64_t n, int over_pi, float* out, void* stream);

/* last_non_linearity='softmax' (abnet3/model.py:161-166: nn.Softmax() after the output
 * layer's Linear/Dropout/BatchNorm = softmax over each row of a [rows, n] matrix), and
 * its autograd: dz = a * (da - sum_c(da * a)).  out may alias z; dz may alias da. */
int abn_softmax_rows(const float* z, int64_t rows, int64_t n, float* out, void* stream);
int abn_softmax_rows_backward(const float* a, const float* da, int64_t rows, int64_t n,
                              float* dz, void* stream);

/* X[path] gathers of abnet3/dataloader.py:204-205, :673-684: out[i] = table[idx[i]] */
int abn_gather_rows(const float* table, const int64_t* idx, int64_t n, int64_t D,
                    float* out, void* stream);

/* One training batch of frame pairs, gathered straight into the layout a (captured) train step reads
 * (abnet3/dataloader.py:204-205,227-233 + the vstack / permutation of :248-255, with the index lists
 * prepared once per dataset): x12 is [2 n_pad, D] --
 *   x12[r]         = table[idx1[first + r]]   (tower 1),   x12[n_pad + r] = table[idx2[first + r]]   (tower 2)
 * for r < n, zero rows for n <= r < n_pad.  labels (device, `label_bytes` per element: 8 = the float64 /
 * int64 labels of the loaders; may be NULL together with y_out): y_out[r] = labels[first + r], zero
 * padding.  n_valid (device int32, may be NULL) receives n: abn_tower_backward_loss's n_valid.  An index outside
 * [0, table_rows) reads as a zero row. */
int abn_gather_pairs(const float* table, int64_t table_rows, int64_t D, const int64_t* idx1, const int64_t* idx2,
                     int64_t first, int64_t n, int64_t n_pad, const void* labels,
                     int32_t label_bytes, float* x12, void* y_out, int32_t* n_valid, void* stream);

/* FeaturesGenerator.stack_fbanks, abnet3/features.py:135-159 */
int abn_stack_frames(const float* feats, int64_t T, int64_t D, int32_t nframes,
                     float* out, void* stream);
/* ... for a batch of utterances laid end to end in one [T, D] table (what the reference's
 * h5features_feats2stackedfeats loop does file by file, features.py:299-320): utt_frame_off =
 * cumulative frame counts, device int64 [n_utts + 1]; the window never crosses an utterance boundary. */
int abn_stack_frames_batched(const float* feats, const int64_t* utt_frame_off, int64_t n_utts,
                             int64_t T, int64_t D, int32_t nframes, float* out, void* stream);

/* FeaturesGenerator.mean_variance_normalisation / mean_var_norm_per_file,
 * abnet3/features.py:205-244, :263-297: mean = np.mean, std = np.std over axis 0
 * (per_channel: [D] outputs) or over everything (whole spectrum: [1] outputs),
 * then out = (x - mean) / (std + eps).  ws: abn_mvn_ws_bytes(T, D) bytes. */
int64_t abn_mvn_ws_bytes(int64_t T, int64_t D);
int abn_mvn_stats(const float* feats, int64_t T, int64_t D, int per_channel,
                  float* mean, float* stdv, void* ws, void* stream);
int abn_mvn_apply(const float* feats, int64_t T, int64_t D, const float* mean,
                  const float* stdv, int per_channel, float eps, float* out,
                  void* stream);

/* FeaturesGenerator.do_fbank, abnet3/features.py:99-114 (-> third-party
 * spectral.Spectral): int16 or fp32 mono samples -> [nframes, nfilt] log mel
 * energies (framing of that package's Sphinx-III lineage: a frame's pre-emphasis starts from the
 * last sample of the previous frame, a tail frame repeats its samples cyclically;
 * oracle/features_np.py).  melbank: [nfft/2+1, nfilt] fp32 weights (host side builds it,
 * abnet3_amd/features.py), window: [wlen] fp32.  band: [nfilt][2] int32 (device), first and
 * last bin with a non-zero weight of every filter; with it (and nfft = 1024, the
 * reference's value, nfilt <= 64) a frame is one wavefront's real-input FFT and a sparse mel
 * projection; NULL selects the general (any power-of-two nfft, dense projection) kernel. */
int abn_fbank(const void* samples, int sample_is_i16, int64_t nsamples,
              int32_t wlen, double fshift, int32_t nfft, int32_t nfilt,
              float alpha, const float* window, const float* melbank, const int32_t* band,
              int64_t nframes, float* out, void* stream);

/* ... for a batch of utterances in ONE launch (the reference's h5features_compute loop calls do_fbank
 * file by file, features.py:160-203): the utterances' samples laid end to end in `samples`,
 * utt_sample_off / utt_frame_off = cumulative sample / frame counts (device int64 [n_utts + 1],
 * frames of utterance u = int(len_u / fshift + 1)); out [nframes = utt_frame_off[n_utts], nfilt].
 * Every utterance is framed on its own, as if abn_fbank had been called on it alone. */
int abn_fbank_batched(const void* samples, int sample_is_i16, const int64_t* utt_sample_off,
                      const int64_t* utt_frame_off, int64_t n_utts, int32_t wlen, double fshift,
                      int32_t nfft, int32_t nfilt, float alpha, const float* window,
                      const float* melbank, const int32_t* band, int64_t nframes, float* out,
                      void* stream);

/* do_deltas / do_deltasdeltas of the same call (abnet3/features.py:110-111 -> spectral):
 * the slope over +-4 frames, out[t] = sum_{n=1..4} n (x[t+n] - x[t-n]) / 60, the sequence
 * padded with copies of frame 1 in front and of frame T-2 behind.  Apply twice for the
 * second-order deltas.  feats, out: [T, D] fp32, distinct buffers. */
int abn_deltas(const float* feats, int64_t T, int64_t D, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ABNET3_HIP_H */
