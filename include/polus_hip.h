/* polus_hip.h — C ABI of libpolus_hip.so, the MI355X (gfx950) kernel library behind the
 * Polus training hot path.
 *
 * The reference (bioinformatics-ua/polus @ 0.2.1) has NO native interface: every numeric
 * instruction of `BaseTrainer.train_step` (polus/training.py:150-193) is executed by
 * third-party wheels (TensorFlow/Keras, HuggingFace TF-BERT, tensorflow-addons, Horovod).
 * Each entry point below therefore cites the reference call site whose arithmetic it
 * replaces.  Conventions:
 *   - every function returns 0 on success; on failure a non-zero code and a thread-local
 *     message from polus_last_error();
 *   - every pointer is a caller-owned DEVICE pointer (HBM) unless the name says host;
 *     nothing is allocated inside a call — scratch is passed as (workspace, bytes) and
 *     sized with the matching *_workspace_bytes();
 *   - every launch goes to the caller's stream (`void* stream` is a hipStream_t);
 *     no call synchronises;
 *   - activations are row-major [rows, features]; Dense weights are [out, in]
 *     (PyTorch layout — the reference loads its BERT weights `from_pt=True`,
 *     polus/models.py:229);
 *   - `dtype` selects the activation/weight element type: POLUS_F32 (exact-f32 MFMA,
 *     the parity path) or POLUS_BF16 (bf16 MFMA inputs, f32 accumulation).  Biases,
 *     LayerNorm parameters, statistics, losses, gradients of parameters and optimizer
 *     state are always f32.
 */
#ifndef POLUS_HIP_H
#define POLUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POLUS_ABI_VERSION 1

enum { POLUS_OK = 0, POLUS_ERR_INVALID = 1, POLUS_ERR_HIP = 2, POLUS_ERR_WORKSPACE = 3 };
enum { POLUS_F32 = 0, POLUS_BF16 = 1 };
/* operand storage for polus_gemm: K_CONTIG = [rows][K] (K fastest), K_STRIDED = [K][rows] */
enum { POLUS_K_CONTIG = 0, POLUS_K_STRIDED = 1 };
enum { POLUS_ACT_NONE = 0, POLUS_ACT_GELU = 1, POLUS_ACT_SWISH = 2, POLUS_ACT_RELU = 3, POLUS_ACT_TANH = 4 };
/* polus_gemm flags */
enum {
    POLUS_GEMM_ACCUM_C = 1,  /* C += result (gradient accumulation)                         */
    POLUS_GEMM_ACT_FWD = 2,  /* aux[m][n] = v (pre-activation, if aux != NULL); C = act(v)   */
    POLUS_GEMM_ACT_BWD = 4,  /* C = v * act'(aux[m][n])                                      */
    POLUS_GEMM_DROPOUT = 8   /* polus_gemm_dropout only: v = keep(seed, m*N+n) ? v/(1-p) : 0   */
};

const char* polus_last_error(void);
int polus_abi_version(void);
/* host out-params; arch is a NUL-terminated gcnArchName prefix (e.g. "gfx950") */
int polus_device_info(int* n_cu, int* lds_bytes_per_cu, char* arch, int arch_len);
/* The POLUS_* tuning switches of the library are read from the environment once, at the first call
 * that needs one; this re-reads them (A/B tools and tests that flip a switch inside one process). */
int polus_reload_env(void);
/* POLUS_GEMM_RESERVE_CUS (CUs the GEMM tile-shape choice and the persistent grids leave to concurrent RCCL channel
 * kernels) applies only while this is on (default on).  The data-parallel trainer switches it on for backward, where the
 * bucketed exchange runs beside the GEMMs, and off for the forward pass. */
int polus_set_reserve_active(int on);
/* Per-step scalars from device memory, for steps replayed from a captured HIP graph (the kernel arguments of a
 * replay are frozen).  `dev_block16` points to 16 bytes in HBM, {uint32 salt; float lr; float lr_t; uint32 0},
 * that the caller rewrites before each replay; NULL unregisters.  While a block is registered
 *   - every dropout site uses seed_eff = mix(seed + salt), mix(x) = (x ^ x >> 15) * 0x2C1B3C6D (uint32), so the
 *     caller passes the step-independent part as `seed` and the step term as `salt` (the eager path passes
 *     mix(step-independent + step term) as `seed`: identical masks either way);
 *   - polus_adam_step takes lr and lr_t from the block instead of its arguments.
 * Process-wide (one process drives one GPU). */
int polus_set_dynamic_params(const void* dev_block16);

/* ---- GEMM (HF Dense layers + their gradients; tape.gradient at polus/training.py:185)
 * C[M,N] = epilogue(alpha * A_op[M,K] . B_op[K,N]).
 *   a_layout: K_CONTIG  -> A stored [M][K] (lda = row stride), K_STRIDED -> stored [K][M]
 *   b_layout: K_CONTIG  -> B stored [N][K] (ldb),              K_STRIDED -> stored [K][N]
 *   forward  Y = X W^T        : A=X (K_CONTIG), B=W[out,in] (K_CONTIG)
 *   dX = dY W                 : A=dY (K_CONTIG), B=W[out,in] (K_STRIDED)
 *   dW = dY^T X               : A=dY (K_STRIDED), B=X (K_STRIDED), c_dtype = POLUS_F32
 * epilogue order: v = alpha*acc; v += bias[n]; ACT_FWD: aux=v, v=act(v); ACT_BWD: v*=act'(aux);
 *                 v += resid[m][n]; ACCUM_C: v += C[m][n]; C = v.
 * c_dtype is the element type of C / resid / aux... C only: resid and aux use `dtype`.
 * split_k > 1 writes f32 partial slabs to `workspace` and reduces them in a second,
 * order-fixed kernel (bitwise reproducible).  With bf16 K-contiguous operands and a bf16 C the reduce
 * applies the whole epilogue (residual, activation forward / backward, dropout); otherwise only
 * bias / ACCUM_C epilogues are allowed with split_k > 1.
 * polus_gemm_auto_split: the number of K slices the library recommends for a bf16 Dense GEMM of this size
 * (1 = none; a tuning query with no counterpart in the reference, whose Dense layers are Keras / HF calls):
 * > 1 only for about two thousand tokens and fewer, where even its 128 x 128 tile leaves most of the chip idle. */
size_t polus_gemm_workspace_bytes(int M, int N, int split_k);
int polus_gemm_auto_split(int M, int N, int K);
int polus_gemm(int dtype, int a_layout, int b_layout, int c_dtype,
               const void* A, long lda, const void* B, long ldb, void* C, long ldc,
               int M, int N, int K, float alpha,
               const float* bias, const void* resid, long ldr, void* aux, long ldaux,
               int act, int flags, int split_k, void* workspace, size_t workspace_bytes,
               void* stream);

/* polus_gemm with inverted dropout in the epilogue, applied after bias/activation and before the
 * residual add (HF TFBertSelfOutput / TFBertOutput: dropout(dense(x)) + residual).  The mask is a
 * pure function of (seed, m*N + n); polus_dropout_mask reproduces it. */
int polus_gemm_dropout(int dtype, int a_layout, int b_layout, int c_dtype,
                       const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                       int M, int N, int K, float alpha,
                       const float* bias, const void* resid, long ldr, void* aux, long ldaux,
                       int act, int flags, int split_k, void* workspace, size_t workspace_bytes,
                       float drop_p, uint32_t seed, void* stream);
/* y[i] = keep(seed, i) ? x[i]/(1-p) : 0 (tf.keras.layers.Dropout; apply the same call to dy for backward) */
int polus_dropout(int dtype, const void* x, void* y, int64_t n, float drop_p, uint32_t seed, void* stream);
/* mask[i] = 1 if element i is kept (i = idx0 .. idx0+n-1): the reference for every dropout site */
int polus_dropout_mask(uint32_t seed, float drop_p, uint32_t idx0, int64_t n, uint8_t* mask, void* stream);

/* ---- Dense backward for the parameters (the dW/db part of tape.gradient, polus/training.py:185):
 * dW[n_out, n_in] (+)= dY[T, n_out]^T . X[T, n_in]  (f32) and, if db != NULL, db[n_out] (+)= column
 * sums of dY — one pass over dY (on the bf16 ring kernel the column sums ride on the matrix pipe).
 * Deterministic split-K over T. */
size_t polus_dense_bwd_params_workspace_bytes(int T, int n_out, int n_in, int split_k);
int polus_dense_bwd_params(int dtype, const void* dY, long lddy, const void* X, long ldx,
                           float* dW, long lddw, float* db, int T, int n_out, int n_in,
                           int accumulate, int split_k, void* workspace, size_t workspace_bytes,
                           void* stream);

/* The same for up to POLUS_MAX_GROUP (8) Dense layers in ONE launch -- the four weight gradients of
 * an encoder layer (the per-variable MatMul grads tape.gradient emits for one TFBertLayer,
 * polus/training.py:185 through polus/models.py:205-213): the concatenated tile lists fill the chip
 * with 2-3 K-splits instead of 7-28 per matrix.  All problems share T and `accumulate`; split_k > 0
 * applies to every problem, split_k <= 0 lets the library choose per problem so that the launch is
 * one full round of workgroup slots; db may be NULL per problem.  Falls back to one call per problem when a shape does not fit the
 * grouped kernel. */
typedef struct polus_dw_problem {
    const void* dY; long lddy;     /* [T, n_out] */
    const void* X;  long ldx;      /* [T, n_in]  */
    float* dW; long lddw;          /* [n_out, n_in] f32 */
    float* db;                     /* [n_out] f32 or NULL */
    int n_out, n_in;
} polus_dw_problem;
size_t polus_dense_bwd_params_grouped_workspace_bytes(int n, const polus_dw_problem* problems, int T, int split_k);
int polus_dense_bwd_params_grouped(int dtype, int n, const polus_dw_problem* problems, int T, int accumulate,
                                   int split_k, void* workspace, size_t workspace_bytes, void* stream);

/* ---- Dense layers with at most 8 output units (a token-classification head, polus/ner/models.py:26-44; the last layer of
 * tutorials/classifier_example.py:44-48): y = x W^T + b, W [C][H] row-major in `dtype`.  HBM-bound, one wave per row, no matrix
 * pipe: a 128-wide MFMA tile would compute 97 % padding.  polus_dense_thin_supported: 1 when (dtype, H, C) fits (C <= 8,
 * H a multiple of 16 bytes of elements, H <= 1024).  forward: y [rows][C] in y_dtype (f32 or dtype).  backward: dy [rows][C] in
 * dy_dtype (f32 or dtype); dx [rows][H] in dtype or NULL; dW [C][H] and db [C] (or NULL) f32, (+)= when accumulate; sums in a fixed
 * order (bitwise reproducible); workspace from polus_dense_thin_bwd_workspace_bytes. */
int polus_dense_thin_supported(int dtype, int H, int C);
int polus_dense_thin_fwd(int dtype, const void* x, long ldx, const void* W, long ldw, const float* bias,
                         int y_dtype, void* y, long ldy, int rows, int H, int C, void* stream);
size_t polus_dense_thin_bwd_workspace_bytes(int dtype, int rows, int H, int C);
int polus_dense_thin_bwd(int dtype, const void* x, long ldx, int dy_dtype, const void* dy, long lddy,
                         const void* W, long ldw, void* dx, long lddx, float* dW, long lddw, float* db,
                         int rows, int H, int C, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused scaled-dot-product attention (HF TFBertSelfAttention as driven by
 * TFBertSplited.call, polus/models.py:201-216, with the additive key mask
 * (1-m)*-10000 of polus/models.py:175-195).
 * qkv   [B*S, 3H] fused projections, row blocks Q | K | V, head h at columns h*64..h*64+63
 * mask  [B, S] int32 {0,1} (NULL = all ones)
 * ctx   [B*S, H]; lse [B, A, S] f32 = log-sum-exp of the masked, scaled scores
 * head_dim must be 64.  Backward recomputes the probabilities from lse; dqkv [B*S, 3H].
 * drop_p > 0: inverted dropout of the probabilities after the softmax (HF attention_probs_dropout),
 * mask = keep(seed, ((b*A+h)*S+q)*S+key), regenerated in backward.
 * workspace for bwd: B*A*S floats (row dot products dO.O). */
int polus_attention_fwd(int dtype, const void* qkv, const int32_t* mask, void* ctx, float* lse,
                        int B, int S, int n_heads, int head_dim, float drop_p, uint32_t seed, void* stream);
size_t polus_attention_bwd_workspace_bytes(int B, int S, int n_heads);
int polus_attention_bwd(int dtype, const void* qkv, const int32_t* mask, const void* ctx,
                        const void* dctx, const float* lse, void* dqkv,
                        int B, int S, int n_heads, int head_dim, float drop_p, uint32_t seed,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---- LayerNorm over the feature axis, eps inside the sqrt, biased variance
 * (HF TFBertSelfOutput/TFBertOutput/TFBertEmbeddings LayerNorm, eps 1e-12).
 * fwd: y = (x-mean)*rstd*gamma+beta; mean/rstd [rows] f32 are saved for backward.
 * bwd: dx; dgamma/dbeta [H] f32 (+= when accumulate); if dbias != NULL also
 *      dbias[H] (+)= column sums of dx (the bias gradient of the Dense that produced x).
 *      When x = dropout(dense) + residual (drop_p > 0, mask index row*H+col as in polus_gemm_dropout):
 *      dx_masked = dx * mask/(1-p) is the gradient the Dense sees, and dbias sums dx_masked.
 *      With dgamma = dbeta = NULL the call stops after the main kernel and leaves the per-workgroup partial sums in
 *      `workspace` (dbias non-NULL still requests the bias sums); polus_layernorm_bwd_finalize then reduces them -- on any
 *      stream ordered behind the first call (the training step queues it behind the layer's weight-gradient launch on the
 *      side stream, off the critical path).  Same kernels in the same order: same bits. */
size_t polus_layernorm_bwd_workspace_bytes(int rows, int H);
int polus_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta,
                        void* y, float* mean, float* rstd, int rows, int H, float eps, void* stream);
int polus_layernorm_bwd(int dtype, const void* dy, const void* x, const float* gamma,
                        const float* mean, const float* rstd, void* dx,
                        float* dgamma, float* dbeta, float* dbias, int accumulate,
                        int rows, int H, void* dx_masked, float drop_p, uint32_t seed,
                        void* workspace, size_t workspace_bytes, void* stream);
int polus_layernorm_bwd_finalize(void* workspace, size_t workspace_bytes, int rows, int H, float* dgamma, float* dbeta,
                                 float* dbias, int accumulate, void* stream);

/* ---- embeddings: word[ids] + pos[s] + type[tt] -> LayerNorm (HF TFBertEmbeddings; TF gather
 * has no padding_idx, so row 0 receives its gradient).  Tables and their gradients are f32.
 * bwd recomputes the pre-LN sum; gword rows are accumulated with f32 atomics unless
 * `deterministic`, in which case duplicates are summed in index order by one owner wave. */
size_t polus_embed_bwd_workspace_bytes(int B, int S, int H);
int polus_embed_ln_fwd(int dtype, const int32_t* ids, const int32_t* type_ids,
                       const float* word, const float* pos, const float* type,
                       const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                       int B, int S, int H, int vocab, int max_pos, int type_vocab, float eps,
                       float drop_p, uint32_t seed, void* stream);
int polus_embed_ln_bwd(int dtype, const void* dy, const int32_t* ids, const int32_t* type_ids,
                       const float* word, const float* pos, const float* type, const float* gamma,
                       const float* mean, const float* rstd,
                       float* gword, float* gpos, float* gtype, float* ggamma, float* gbeta,
                       int accumulate, int deterministic,
                       int B, int S, int H, int vocab, int max_pos, int type_vocab,
                       float drop_p, uint32_t seed,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- column sums out[c] (+)= sum_r x[r][c]  (bias gradients) */
size_t polus_colsum_workspace_bytes(int rows, int cols);
int polus_colsum(int dtype, const void* x, long ldx, int rows, int cols, float* out, int accumulate,
                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- losses.  logits are f32 [rows, C] (ld = ldl); dlogits is written in `dtype` (ld = lddl)
 * already divided by `rows` (tf.reduce_mean over every leading dim); loss is one f32.
 * softmax_xent: Keras SparseCategoricalCrossentropy(from_logits=True)
 *   (tutorials/classifier_example.py:55); class_weights != NULL gives polus/losses.py:5-18
 *   with one-hot targets (weight = class_weights[label]).
 * sigmoid_xent: polus/losses.py:21-41, y_true f32 multi-hot [rows, C]. */
size_t polus_loss_workspace_bytes(int rows);
int polus_softmax_xent(int dtype, const float* logits, long ldl, const int32_t* labels,
                       const float* class_weights, float* loss, void* dlogits, long lddl,
                       int rows, int C, void* workspace, size_t workspace_bytes, void* stream);
int polus_sigmoid_xent(int dtype, const float* logits, long ldl, const float* y_true, long ldy,
                       const float* class_weights, float negative_weight, float* loss,
                       void* dlogits, long lddl, int rows, int C,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- linear-chain CRF (polus/layers.py:58-126; tensorflow-addons crf_log_likelihood /
 * crf_decode restated).  potentials f32 [B,S,C]; tags int32 [B,S]; lengths int32 [B];
 * trans f32 [C,C] (already masked by the caller, polus/layers.py:58-63);
 * sample_w f32 [B] or NULL.  loss = mean_b(-ll_b * w_b).  dpot in `dtype`, dtrans f32 [C,C].
 * C <= 16. */
size_t polus_crf_workspace_bytes(int B, int S, int C);
int polus_crf_nll(int dtype, const float* potentials, const int32_t* tags, const int32_t* lengths,
                  const float* trans, const float* sample_w, float* loss, void* dpot,
                  float* dtrans, int accumulate, int B, int S, int C,
                  void* workspace, size_t workspace_bytes, void* stream);
int polus_crf_viterbi(const float* potentials, const int32_t* lengths, const float* trans,
                      int32_t* out_tags, int B, int S, int C,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- argmax over the last axis (PolusClassifier.inference, polus/models.py:148-150) */
int polus_argmax(const float* x, long ldx, int32_t* out, int rows, int C, void* stream);

/* ---- confusion matrix of the validation path (polus/metrics.py:51-66, tf.math.confusion_matrix):
 * cm[row_idx[i]][col_idx[i]] += 1 for i < n; cm is int32 [C, C] on the device and is ACCUMULATED into
 * (zero it to start); C <= 128.  Integer atomics: exact.  A pair with an index outside [0, C) adds nothing to cm
 * and 1 to *rejected (device int32, accumulated; may be null): tf.math.confusion_matrix raises on such input, the
 * caller decides (polus_amd/metrics.py raises in evaluate()). */
int polus_confusion_matrix(const int32_t* row_idx, const int32_t* col_idx, int64_t n, int C,
                           int32_t* cm, int32_t* rejected, void* stream);

/* ---- optimizer (optimizer.apply_gradients, polus/training.py:191): Keras Adam /
 * HF AdamWeightDecay over a flat f32 arena.  `seg` is a device table of int64 triples
 * (begin, end, flags) covering [0,n) in chunks; flags bit0 = apply weight decay,
 * bit1 = write a bf16 copy of the updated value to shadow[i] (GEMM weights).
 *   p -= lr*wd*p (decayed tensors) ; m,v update ; p -= lr_t * m / (sqrt(v)+eps)
 * g is multiplied by grad_scale and, if clip_scale != NULL, by *clip_scale (device). */
int polus_adam_step(float* p, const float* g, float* m, float* v, void* shadow_bf16,
                    const int64_t* seg, int n_seg, int64_t n,
                    float lr, float lr_t, float beta1, float beta2, float eps, float weight_decay,
                    float grad_scale, const float* clip_scale, void* stream);
/* sum of squares of g[0..n) -> *out (deterministic two-stage); polus_sqnorm_segments sums only the
 * windows [seg[3k], seg[3k+1]) of g (the polus_adam_step segment table: the variables actually being
 * updated; workspace >= 4096 bytes); then polus_clip_scale writes
 * min(1, clip_norm / sqrt(sum_k sqnorm[k] * grad_scale^2)) -- tf.clip_by_global_norm over n_terms partial
 * sums (one per parameter arena). */
size_t polus_sqnorm_workspace_bytes(int64_t n);
int polus_sqnorm(const float* g, int64_t n, float* out, void* workspace, size_t workspace_bytes, void* stream);
int polus_sqnorm_segments(const float* g, const int64_t* seg, int n_seg, float* out,
                          void* workspace, size_t workspace_bytes, void* stream);
int polus_clip_scale(const float* sqnorm, int n_terms, float grad_scale, float clip_norm, float* out_scale, void* stream);
/* f32 -> bf16 copy (shadow weights refresh after load / broadcast) and bf16/f32 casts */
int polus_cast(int src_dtype, const void* src, int dst_dtype, void* dst, int64_t n, void* stream);
/* dst[cols][rows] = src[rows][cols]^T for bf16 (transposed weight shadow read by dX = dY . W) */
int polus_transpose_bf16(const void* src, void* dst, int rows, int cols, void* stream);
/* Many matrices at once: matrix s lives at element offset segs[4s] of BOTH src_base and dst_base
 * (rows segs[4s+1], cols segs[4s+2]); segs[4s+3] = index of its first 64x64 tile in the launch
 * (ascending), total_tiles = sum of tiles.  `segs_dev` is a device array of 4*nseg int64.  Used
 * once per optimizer step to refresh the transposed bf16 weight shadows that dX = dY.W reads
 * (replaces the implicit transpose inside tape.gradient's MatMul grad, polus/training.py:185). */
int polus_transpose_bf16_batched(const void* src_base, void* dst_base, const void* segs_dev, int nseg,
                                 int total_tiles, void* stream);
/* du = dy * act'(u) elementwise (activation gradient of a Dense whose dY is not produced by
 * a polus_gemm epilogue) */
int polus_act_bwd(int dtype, const void* dy, const void* u, void* du, int64_t n, int act, void* stream);
/* y = a*x elementwise, f32 (gradient averaging when the comm backend lacks AVG) */
int polus_scale(float* x, float a, int64_t n, void* stream);

/* ---- data-parallel collectives over RCCL / xGMI: the device side of the six Horovod touch points of the
 * reference -- hvd.init (polus/__init__.py:109-122), hvd.DistributedGradientTape's gradient averaging
 * (polus/training.py:182-185) and hvd.broadcast_variables (polus/training.py:210-211).
 * One communicator per process (one process per GPU).  Rank 0 draws a 128-byte id with polus_comm_unique_id and
 * hands it to the other ranks over any host channel (the Python side uses the torchrun TCP store); every rank
 * then calls polus_comm_init(rank, world, id).  Collectives are queued on `stream` and return immediately; buffers
 * are device pointers; `dtype` is POLUS_F32 or POLUS_BF16 (bf16 = half the bytes on the links, sums rounded to bf16).
 *   allreduce_sum      buf[count] := sum over ranks (in place); the 1/world factor is folded into polus_adam_step
 *   reduce_scatter_sum recv[recv_count] := rank's slice of the sum of send[world * recv_count]
 *   all_gather         recv[world * send_count] := concatenation of every rank's send[send_count]
 *   broadcast          buf[bytes] := root's buf
 * group_start / group_end bracket several collectives into one RCCL launch. */
#define POLUS_COMM_ID_BYTES 128
int polus_comm_unique_id(void* out_id128);
int polus_comm_init(void** comm, int rank, int world, const void* unique_id128);
int polus_comm_destroy(void* comm);
/* what RCCL itself says about the communicator: ranks it spans, this rank, the HIP device it is bound to
 * (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) -- bench.py prints n_ranks as `rccl_ranks` */
int polus_comm_info(void* comm, int* n_ranks, int* rank, int* device);
int polus_comm_broadcast(void* comm, void* buf, size_t bytes, int root, void* stream);
int polus_comm_allreduce_sum(void* comm, void* buf, size_t count, int dtype, void* stream);
int polus_comm_reduce_scatter_sum(void* comm, const void* send, void* recv, size_t recv_count, int dtype, void* stream);
int polus_comm_all_gather(void* comm, const void* send, void* recv, size_t send_count, int dtype, void* stream);
int polus_comm_group_start(void);
int polus_comm_group_end(void);

#ifdef __cplusplus
}
#endif
#endif /* POLUS_HIP_H */
