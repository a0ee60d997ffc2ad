/*
 * quantool_amd.h -- C ABI of the MI355X (gfx950) backend for quantool's GPTQ / AWQ /
 * SmoothQuant per-linear calibration hot path.
 *
 * The reference (langtech-bsc/quantool) has no FFI for this path: its plugins hand the whole
 * job to llmcompressor.oneshot (src/quantool/methods/llm_compressor/base.py:161).  Each entry
 * point below therefore cites the upstream step it replaces by SURVEY.md section 8(a) row
 * (a7..a14) and the reference line through which that step is reached.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says host;
 *   - the caller owns every buffer; nothing persistent is allocated here.  Scratch comes in
 *     through (workspace, workspace_bytes); sizes from the matching *_workspace_bytes();
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); calls are
 *     asynchronous and re-entrant per stream; the current HIP device is used;
 *   - return value: QT_OK (0) or a negative qt_status; qt_last_error() gives a message for
 *     the calling thread.  Nothing throws across this boundary;
 *   - matrices are row-major; "ld" arguments are row strides in ELEMENTS.
 */
#ifndef QUANTOOL_AMD_H
#define QUANTOOL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* qt_stream_t; /* hipStream_t */

enum qt_status {
    QT_OK = 0,
    QT_ERR_INVALID = -1,     /* bad argument / unsupported shape */
    QT_ERR_NOT_PD = -2,      /* Hessian not positive definite (host-side check helper) */
    QT_ERR_WORKSPACE = -3,   /* workspace too small */
    QT_ERR_HIP = -4,         /* a HIP runtime call failed */
    QT_ERR_UNSUPPORTED = -5
};

enum qt_dtype { QT_F32 = 0, QT_BF16 = 1, QT_F16 = 2 };

int qt_version(void);
const char* qt_last_error(void);

/* ---- a7  accumulate_hessian (GPTQ hook under base.py:161) ----------------------------------
 * G[K,K] (fp32, lower triangle incl. diagonal tiles) += X^T X for X[n_tokens,K] in the model's
 * own 16-bit dtype (x_dtype QT_BF16 or QT_F16: the reference injects no dtype, base.py:222-241, and
 * upstream accumulates inp.float() -- both products are exact in the fp32 accumulator).
 * The caller keeps the raw Gram sum G and the sample count n; upstream's running
 * "H = H*n/(n+1) + (2/(n+1)) X^T X" equals (2/n)*G and is applied in qt_hessian_prepare.
 * Requires K % 8 == 0, ldx % 8 == 0, X 16-byte aligned.  Deterministic: a tile is either summed
 * by one workgroup over all tokens and added to G, or split over token chunks whose fp32 slabs are
 * reduced in fixed order -- no atomics. */
size_t qt_xtx_workspace_bytes(int64_t n_tokens, int K);
int qt_xtx_accumulate(const void* X, int x_dtype, int64_t n_tokens, int K, int64_t ldx, float* G,
                      void* workspace, size_t workspace_bytes, qt_stream_t stream);

/* The same accumulation for fp32 activations X [n_tokens, K] (an fp32 checkpoint: upstream's `inp.float()` is
 * then an fp32 Gram product).  fp32-accurate on the bf16 MFMA: the tokens are split into three
 * bf16 planes (residual <= 2^-27 |x|) and the six plane products of weight >= 2^-16 are summed in one fp32
 * accumulator per lower-triangular tile, added into G.  K % 4 == 0, ldx % 4 == 0, X 16-byte aligned.
 * Deterministic; 6x the MFMA work of the 16-bit path. */
size_t qt_xtx_accumulate_f32_workspace_bytes(int64_t n_tokens, int K);
int qt_xtx_accumulate_f32(const float* X, int64_t n_tokens, int K, int64_t ldx, float* G, void* workspace,
                          size_t workspace_bytes, qt_stream_t stream);

/* ---- a12/a13  activation statistics (AWQ / SmoothQuant hooks under base.py:161) ------------
 * abs_sum[K] += sum_t |x[t,k]|;  cmin[k] = min(cmin[k], min_t x);  cmax likewise.  Any of the
 * three outputs may be NULL.  The caller initialises abs_sum = 0, cmin = +inf, cmax = -inf.
 * X [n_tokens, K] bf16 or fp16 (x_dtype), K % 8 == 0.  Deterministic (ordered chunk reduction). */
size_t qt_act_stats_workspace_bytes(int64_t n_tokens, int K);
int qt_act_stats_accumulate(const void* X, int x_dtype, int64_t n_tokens, int K, int64_t ldx, float* abs_sum,
                            float* cmin, float* cmax, void* workspace, size_t workspace_bytes,
                            qt_stream_t stream);

/* ---- a8/a9  dead columns, damping, activation ordering (quantize_weight, gptq.py:86) -------
 * From the Gram sum G (lower triangle) and sample count n builds (one pass; from K = 2048 two coalesced passes
 * through a symmetric copy of G in the workspace, K*K*4 bytes -- the result is the same to the bit)
 *   Hd = P^T (2/n * G) P  with  dead = diag==0 -> 1,  Hd += percdamp*mean(diag) * I
 * and writes A = flat-reversed Hd (A[i][j] = Hd[K-1-i][K-1-j]), upper triangle valid, which is
 * what qt_cholesky_inverse_upper consumes.  perm (int32[K], sweep position -> original column)
 * may be NULL (identity).  dead[K] (uint8, indexed by sweep position) and diag_out[K]
 * (fp32 diag of 2/n*G in ORIGINAL order, before dead/damp; may be NULL) are outputs. */
size_t qt_hessian_prepare_workspace_bytes(int K);
int qt_hessian_prepare(const float* G, int K, int64_t n_samples, float percdamp,
                       const int32_t* perm, float* A, uint8_t* dead, float* diag_out,
                       void* workspace, size_t workspace_bytes, qt_stream_t stream);
/* diag(2/n * G) only (input to the activation-ordering argsort). */
int qt_hessian_diag(const float* G, int K, int64_t n_samples, float* diag_out, qt_stream_t stream);
/* a9: perm = argsort(values, descending), stable (equal values keep ascending index; NaN orders as
 * the largest value, as torch.argsort does, so perm is a permutation for any input); inv (may be
 * NULL) receives the inverse permutation.  Rank counting, deterministic. */
int qt_argsort_desc(const float* values, int K, int32_t* perm, int32_t* inv, qt_stream_t stream);

/* ---- a8  cholesky -> cholesky_inverse -> cholesky(upper) ------------------------------------
 * Given A = flat-reversed damped Hessian (upper triangle read, destroyed), writes
 * U = chol(Hd^-1, upper) [K,K] row-major (strict lower triangle zero-filled).
 * Uses A = R^T R, U = flat-reverse(R^-T) (DESIGN.md "one factorisation instead of three").
 * info (device int32): 0 ok, else 1-based index of the first non-positive pivot; upstream's
 * LinAlgError fallback (U = I) is then applied on the device, with no host synchronisation.
 * For large K (from about 6 k; K % 8 == 0) the two block-row products that carry the K^3 run as fp32-accurate
 * three-plane bf16 products and the workspace grows by 12 K^2 bytes for the plane copies of
 * R and R^-T; QT_CHOL_G3=0 keeps the f32-MFMA chain.  Deterministic either way. */
size_t qt_cholesky_inverse_upper_workspace_bytes(int K);
int qt_cholesky_inverse_upper(float* A, int K, float* U, int32_t* info, void* workspace,
                              size_t workspace_bytes, qt_stream_t stream);

/* The same factorisation for n_problems (1..16) Hessians of ONE size K in every launch of the chain: the Linear groups
 * of a decoder layer that read different inputs of equal width (Llama: q/k/v, o, gate/up at K = hidden; Mixtral: the
 * eight experts' w2 at K = intermediate) -- upstream factorises them one after another inside one `oneshot` call
 * (base.py:161 -> quantize_weight per Linear).  Problem b: A + b * strideA (destroyed), U + b * strideU (elements;
 * multiples of 4, >= K*K), info[b].  Each problem's factor is bit-identical to a single-problem call (same kernels,
 * tile shapes, split-K decisions and summation orders per problem); the latency-bound panel kernels -- one workgroup
 * per problem -- and the short products serve all problems per launch.  Workspace: n_problems times a single
 * problem's share plus one copy of the item tables. */
size_t qt_cholesky_inverse_upper_batched_workspace_bytes(int K, int n_problems);
int qt_cholesky_inverse_upper_batched(float* A, int64_t strideA, int K, float* U, int64_t strideU, int32_t* info,
                                      int n_problems, void* workspace, size_t workspace_bytes, qt_stream_t stream);

/* ---- a10  minmax observer -> calculate_qparams ----------------------------------------------
 * W[R,K] (fp32, bf16 or fp16 by w_dtype -- every w_dtype / out_dtype argument below takes the three) -> scale, zp [R, K/group_size] fp32.  group_size <= 0:
 * channel-wise.  symmetric: scale = absmax/((qmax-qmin)/2), zp = 0.  scale_t / zp_t (may be
 * NULL) receive the same values group-major [G, R], the layout qt_gptq_sweep reads (one
 * coalesced load per column step: lanes are rows). */
int qt_group_minmax_qparams(const void* W, int w_dtype, int R, int K, int64_t ldw, int group_size,
                            int symmetric, int num_bits, float* scale, float* zp, float* scale_t,
                            float* zp_t, qt_stream_t stream);

/* W_f32[R,K] = float(W[:, perm]) with dead sweep positions zeroed (W = weight.clone().float();
 * W[:, perm]; W[:, dead] = 0).  perm / dead may be NULL. */
int qt_weight_gather_f32(const void* W, int w_dtype, int R, int K, int64_t ldw, const int32_t* perm,
                         const uint8_t* dead, float* W_f32, qt_stream_t stream);

/* Both of the above in ONE pass over W (static / no activation ordering, where the observer sees the ORIGINAL columns
 * and the sweep's working copy is in sweep order): scale, zp [R, G] and W_f32 [R, K] as qt_group_minmax_qparams and
 * qt_weight_gather_f32 give them, to the bit; scale_t / zp_t (may be NULL) at [g * ld_t + r] -- ld_t >= R lets a caller
 * write a Linear's columns of a table that spans the stacked rows of several Linears.  A row must fit the LDS
 * (K <= 81920 16-bit or 40960 fp32 elements). */
int qt_weight_gather_qparams(const void* W, int w_dtype, int R, int K, int64_t ldw, const int32_t* perm,
                             const uint8_t* dead, int group_size, int symmetric, int num_bits, float* W_f32, float* scale,
                             float* zp, float* scale_t, float* zp_t, int64_t ld_t, qt_stream_t stream);

/* ---- a11  the column sweep of quantize_weight -----------------------------------------------
 * W[R,K] fp32 in sweep order (updated in place: error-compensated, then dequantised values),
 * U[K,K] upper factor, scale_t/zp_t [G,R] fp32 (group-major, see qt_group_minmax_qparams),
 * g_idx[K] int32 = group of each sweep position.
 * Outputs Qt[K,R] int8 (integer levels, sweep-position major) and loss[R].
 * blocksize must be 128 (upstream default) in this build.  Bit-exact against
 * oracle/gptq_oracle.c:orc_gptq_sweep for identical inputs -- including when the updates of columns
 * far to the right are applied for several blocks in one pass over W (env QT_SWEEP_BATCH, 1..8 blocks,
 * default 4, 8 from K = 8192: every block's product is still its own ascending-k chain, subtracted in block order). */
size_t qt_gptq_sweep_workspace_bytes(int R, int K, int blocksize);
int qt_gptq_sweep(float* W, int R, int K, const float* U, const float* scale_t, const float* zp_t,
                  int G, const int32_t* g_idx, int blocksize, int num_bits, int8_t* Qt,
                  float* loss, void* workspace, size_t workspace_bytes, qt_stream_t stream);

/* The sweeps of several Linear groups of ONE in_features K as one stacked sweep: W holds the rows of all groups
 * (group g: rows [row_end[g-1], row_end[g]), row_end a HOST array, every boundary but the last a multiple of 128),
 * group g's factor is U + g * strideU (elements) and its column groups g_idx + g * K; scale_t / zp_t / Qt / loss span the
 * stacked rows.  Rows are independent given their factor, and per row the operation sequence is qt_gptq_sweep's, so
 * every group's outputs are bit-identical to its own qt_gptq_sweep call; the block kernel and the update products run
 * once per 128-column block for all groups (a Llama layer's q/k/v + o + gate/up: 38 912 rows instead of three launches
 * of 6144 / 4096 / 28 672). */
int qt_gptq_sweep_grouped(float* W, int R, int K, const float* U, int64_t strideU, int n_groups, const int32_t* row_end,
                          const float* scale_t, const float* zp_t, int G, const int32_t* g_idx, int blocksize,
                          int num_bits, int8_t* Qt, float* loss, void* workspace, size_t workspace_bytes,
                          qt_stream_t stream);

/* ---- a14  pack_to_int32 (save path, base.py:188) --------------------------------------------
 * packed[R, ceil(K/8)] int32: nibble j of word w = level(column 8w+j) + 8.  col_src (int32[K],
 * may be NULL) maps an output column to the sweep position holding it (undoes actorder). */
int qt_pack_int4(const int8_t* Qt, int R, int K, const int32_t* col_src, int32_t* packed,
                 qt_stream_t stream);

/* Dequantised weights in original column order: out[r,c] = (q - zp[r,g(c)]) * scale[r,g(c)],
 * out dtype by out_dtype (fp32 / bf16 / fp16, round to nearest even); g_of_col int32[K] group of each
 * ORIGINAL column. */
int qt_dequantize(const int8_t* Qt, int R, int K, const int32_t* col_src, const float* scale,
                  const float* zp, int G, const int32_t* g_of_col, void* out, int out_dtype,
                  int64_t ldo, qt_stream_t stream);

/* ---- a12  AWQ scale search (AWQModifier under awq.py:81) ----------------------------------------
 * w_sum[K] += sum_rows |w| / (group absmax + 1e-6) (call once per balance layer; w_mean = w_sum /
 * total rows).  group_size: any divisor of K; <= 0 = one group per row (W8A16). */
size_t qt_awq_weight_mean_workspace_bytes(int R, int K);
int qt_awq_weight_mean_accumulate(const void* W, int w_dtype, int R, int K, int64_t ldw, int group_size,
                                  float* w_sum, void* workspace, size_t workspace_bytes,
                                  qt_stream_t stream);
/* scales[g][k], g = 0..n_grid-1 (ratio g/n_grid): s = clamp(x_mean^r / (w_mean^(1-r) + 1e-4), 1e-4),
 * s /= sqrt(max s * min s), inf/nan -> 1.  x_mean = x_abs_sum / n_tokens (qt_act_stats_accumulate). */
int qt_awq_scales(const float* x_abs_sum, int64_t n_tokens, const float* w_sum, int64_t n_rows, int K,
                  int n_grid, int duo_scaling, float* scales, qt_stream_t stream);
/* Mirror the lower triangle of the Gram sum into the upper one (G full symmetric afterwards). */
int qt_symmetrize_lower(float* G, int K, qt_stream_t stream);
/* Search loss of one grid point for one balance Linear:  L = mean((X W^T - X Wq^T)^2) with
 * Wq = pseudo_quant(W*s)/s, evaluated as <G, D^T D>_F / (n_tokens * R), D = W - Wq, G = X^T X (full
 * symmetric, see qt_symmetrize_lower).  exact == 0: D rounded to bf16, D^T D on the bf16 MFMA (moves L
 * by < 5e-4 relative; the form the 20-point search runs).  exact != 0: D and D^T D in fp32 on the f32
 * MFMA (16x the MFMA time; used to re-score grid points whose fast losses are closer than that error).
 * loss_out[0] = (accumulate ? loss_out[0] : 0) + weight * L   -- a mapping's loss is the row-weighted
 * mean over its balance Linears, so the caller passes weight = R / total rows. */
size_t qt_awq_loss_workspace_bytes(int R, int K);
int qt_awq_loss(const void* W, int w_dtype, int R, int K, int64_t ldw, const float* s, int group_size,
                int symmetric, int num_bits, const float* Gfull, int64_t n_tokens, int exact, float weight,
                int accumulate, float* loss_out, void* workspace, size_t workspace_bytes, qt_stream_t stream);
/* index_out[0] = index of the first minimum of values[0..n) (n <= 1024): the grid search's argmin. */
/* All n_grid fast search losses of one balance Linear at once (scales [n_grid][K] row-major): losses[g]
 * (+)= weight * loss_g, the same quantity n_grid qt_awq_loss(exact = 0) calls compute, from three launches. */
size_t qt_awq_losses_workspace_bytes(int R, int K, int n_grid);
int qt_awq_losses(const void* W, int w_dtype, int R, int K, int64_t ldw, const float* scales, int n_grid,
                  int group_size, int symmetric, int num_bits, const float* Gfull, int64_t n_tokens, float weight,
                  int accumulate, float* losses, void* workspace, size_t workspace_bytes, qt_stream_t stream);
int qt_argmin_f32(const float* values, int n, int32_t* index_out, qt_stream_t stream);
/* out[R,K] (W's dtype, leading dimension ldo) = pseudo_quant(W * s) / s: the trial weights of one
 * grid point, for mappings whose search loss is measured on a parent module's output (q/k/v under
 * self_attn, gate/up under mlp) rather than on the balance Linear's own.  out may alias W. */
int qt_awq_pseudo_quantize(const void* W, int w_dtype, int R, int K, int64_t ldw, const float* s, int group_size,
                           int symmetric, int num_bits, void* out, int64_t ldo, qt_stream_t stream);
/* out = W * s[None, :] (divide != 0: W / s), rounded to W's dtype (AWQ / SmoothQuant apply step). */
int qt_scale_columns(const void* W, int w_dtype, int R, int K, int64_t ldw, const float* s, int divide,
                     void* out, int64_t ldo, qt_stream_t stream);
/* Plain round-to-nearest levels Qt[K,R] under the given group parameters (AWQ's final step; also
 * the U = I degenerate case of the sweep). */
int qt_rtn_quantize(const void* W, int w_dtype, int R, int K, int64_t ldw, const float* scale,
                    const float* zp, int G, int group_size, int num_bits, int8_t* Qt, qt_stream_t stream);

/* ---- a13  SmoothQuant (SmoothQuantModifier under smoothquant.py:77) -----------------------------
 * wmax[k] = max(wmax[k], max_r |W[r,k]|);  s = (cmax-cmin)^alpha / wmax^(1-alpha), s = cmax-cmin
 * where wmax == 0. */
size_t qt_col_absmax_workspace_bytes(int R, int K);
int qt_col_absmax_accumulate(const void* W, int w_dtype, int R, int K, int64_t ldw, float* wmax,
                             void* workspace, size_t workspace_bytes, qt_stream_t stream);
int qt_smoothquant_scales(const float* cmin, const float* cmax, const float* wmax, int K, float alpha,
                          float* s, qt_stream_t stream);

/* ---- fp32 "TN" GEMM on the f32 MFMA (building block of a8 and a11; exposed for tests) ----------
 * acc[m][n] = sum_{k ascending} A[k*lda + m] * B[k*ldb + n]   (bit-for-bit an fmaf chain from 0)
 * mode 0: Cout = Cin - acc   1: Cout = acc   2: Cout = -acc.   skip_zero_k: B[k][n] == 0 for k < n.
 * allow_split_k != 0 lets latency-bound shapes split k over workgroups (ordered slab reduction:
 * deterministic, but no longer the single ascending chain -- the sweep never allows it). */
size_t qt_sgemm_tn_f32_workspace_bytes(int M, int N);
int qt_sgemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* Cin,
                    int64_t ldcin, float* Cout, int64_t ldcout, int M, int N, int kdim, int skip_zero_k,
                    int mode, int allow_split_k, void* workspace, size_t workspace_bytes, qt_stream_t stream);

/* Host-only self-check of the Gram kernel's tile table for K (runs without a GPU): 0 if every lower-triangular
 * 256 x 256 tile appears exactly once.  Also returns the distinct 256-channel panels per 32-entry chunk (one XCD's
 * workgroups) and per 256-entry round (the chip), summed over the table -- the locality figures DESIGN.md 4.1 quotes. */
int qt_xtx_tile_table_check(int K, int* n_tiles_out, int* chunk_panels_out, int* round_panels_out);

/* ---- scale * <H, X^T X>_F without storing X^T X (the Gram kernel's epilogue multiplies its tile with H's;
 * building block of a12's search loss <X^T X, D^T D>; exposed for tests) -----------------------------
 * X [n_tokens, K] bf16 / fp16 with ldx == K and n_tokens % 64 == 0; H [K, K] fp32, lower triangle read.
 * *out = (accumulate ? *out : 0) + scale * sum_{i,j} H[i][j] (X^T X)[i][j]; fp64 partial sums in a fixed order. */
size_t qt_xtx_dot_workspace_bytes(int64_t n_tokens, int K);
int qt_xtx_dot(const void* X, int x_dtype, int64_t n_tokens, int K, int64_t ldx, const float* H, double scale,
               float* out, int accumulate, void* workspace, size_t workspace_bytes, qt_stream_t stream);

/* ---- fp32-accurate "TN" product on the bf16 MFMA (three bf16 planes per operand, six plane products;
 * building block of a8's K^3 products; exposed for tests) ---------------------------------------
 * A [k][lda], B [k][ldb] fp32, k a multiple of 128, M / N / lda / ldb multiples of 4.
 * kind 0: C -= A^T B (one workgroup per 256x256 tile);  kind 1: C = A^T B, k split into slabs reduced in
 * fixed order.  Not an ascending-k fmaf chain: never used where the oracle pins the order (a11). */
size_t qt_gemm3_tn_f32_workspace_bytes(int M, int N, int k);
int qt_gemm3_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M,
                    int N, int k, int kind, void* workspace, size_t workspace_bytes, qt_stream_t stream);

/* Host-only self-check of the bf16x3 block-row planner (runs without a GPU): 0 if every k chunk of every tile is
 * covered exactly once and the slab / reduction tables are consistent, else a negative code. */
int qt_gemm3_plan_check(int Tm, int Tn, int c_end, int tri, int* n_items_out, int* n_slabs_out, int* longest_out);
/* ---- measurement aid (bench.py roofline leg; not part of the reference surface) -------------
 * When enabled, HIP events are recorded on the launch stream immediately around the named
 * kernel; qt_profile_read synchronises them, returns the summed device time and the launch
 * count since the last read, and resets the slot. */
enum qt_prof_kernel { QT_PROF_XTX = 0, QT_PROF_SWEEP_BLOCK = 1, QT_PROF_NUM_KERNELS = 2 };
int qt_profile_enable(int on);
int qt_profile_read(int kernel_id, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* QUANTOOL_AMD_H */
