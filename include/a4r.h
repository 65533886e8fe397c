/* a4r.h -- C ABI of liba4r_hip.so: the MI355X-native kernels behind the adapter-tuned
 * TransRec training step (reference: westlake-repl/Adapter4Rec).
 *
 * The reference has no FFI: its hot path is eager PyTorch called from Python
 * (Downstream/Text/run.py:595-600 -> model/model.py:48-70).  Each entry point below replaces
 * the op sequence of the cited reference lines; the host mirror of the reference's Python
 * interface (adapter4rec_amd/model, engine.py) binds them through ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless stated; the caller owns all memory;
 *  - `stream` is a hipStream_t (0 = default stream); calls only enqueue work, they never
 *    synchronise, allocate or free -- so they can be captured into a hipGraph;
 *  - matrices are row-major, leading dimensions in ELEMENTS; row counts of activations are
 *    padded by the caller to a multiple of 128 (padding rows hold finite values);
 *  - dtype: A4R_BF16 (0) = bf16 storage / fp32 accumulate, A4R_F32 (1) = fp32 everywhere;
 *  - return value: 0 ok, -1 invalid argument (shape/alignment/dtype), -2 launch failure.
 *    No entry point aborts.
 */
#ifndef A4R_H
#define A4R_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define A4R_BF16 0
#define A4R_F32 1
#define A4R_FP8 2    /* a4r_gemm_nt in_dtype only: OCP e4m3fn, one byte per element, with per-row scales (a4r_gemm_t.scale_a / scale_b) */
#define A4R_ACT_NONE 0
#define A4R_ACT_RELU 1
#define A4R_ACT_GELU 2       /* exact erf form (HF "gelu", nn.GELU) */
#define A4R_ACT_GELU_TANH 3  /* HF "gelu_new" (compacter, model/modules.py:220) */
#define A4R_ACT_LEAKY 4
#define A4R_DACT_MUL 15       /* dact only: multiply by Pre itself (Pre holds a stored derivative, see c2_mode) */
#define A4R_DACT_MUL_Q8 14    /* dact only, out_dtype bf16: the same, Pre is the uint8 tensor c2_mode 2 wrote (ldpre in bytes) */

/* Return codes of the int-returning entry points (those whose comment names a count, a flag or a previous value return that instead).  Every
 * argument check happens BEFORE anything is enqueued: A4R_EINVAL means nothing was launched and no output was touched. */
#ifndef A4R_OK
#define A4R_OK 0
#define A4R_EINVAL (-1)   /* NULL / misaligned pointer, unsupported shape, dtype or flag combination */
#define A4R_ELAUNCH (-2)  /* hipGetLastError() != hipSuccess after the launch */
#endif

/* ABI version: bumped whenever a signature or struct below changes.  The Python binding (adapter4rec_amd/_lib.py) refuses a
 * library whose a4r_version() differs, so an A/B build made before a signature change cannot be called with shifted arguments. */
#define A4R_ABI_VERSION 409
int a4r_version(void);

/* C[M,N] = epilogue(alpha * A[M,K] . B[N,K]^T): every nn.Linear on the path (HF BertSelfAttention
 * q/k/v, BertSelfOutput.dense, BertIntermediate, BertOutput.dense; AdapterBlock fc_down/fc_up
 * model/modules.py:130-134; SASRec w_Q/w_K/w_V/fc/w_1/w_2 modules.py:23-28,63-74) and its dgrad.
 * epilogue, in order: + bias[N]; C2 = pre-activation (or its act' if c2_mode); act; * act'(Pre) if dact (* Pre if A4R_DACT_MUL);
 * [dropout if drop_first]; + R1 + R2 (residuals); [dropout if !drop_first]; store C.  The dropout mask is a
 * pure function of (seed, site, row * N + col), so backward regenerates it instead of reading it.
 * M % 128 == 0, N % 64 == 0, K % 64 == 0.  A,B have in_dtype; C,C2,R1,R2,Pre have out_dtype. */
typedef struct {
    const void* A; const void* B; void* C;
    const float* bias; void* C2; const void* R1; const void* R2; const void* Pre;
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldc2, ldr1, ldr2, ldpre;
    int32_t in_dtype, out_dtype;
    int32_t act, dact;
    int32_t drop_first;   /* 0: dropout after the residual adds (backward form); 1: before them (forward form) */
    int32_t c2_mode;      /* what C2 receives: 0 = the pre-activation, 1 = act'(pre-activation) (so that backward is one multiply),
                           * 2 (act GELU, out_dtype bf16) = gelu'(pre-activation) as 8-bit fixed point, one byte per element, ldc2 in
                           * bytes: q = round((d + 0.1289) / 0.0049326), d in [-0.1289, 1.1289] (the range of gelu'), |error| <= 0.0025
                           * -- bf16's own rounding error for a derivative near 1 -- at half the bytes */
    float alpha;
    float drop_p; uint32_t drop_site; uint64_t drop_seed;
    int64_t drop_row0;    /* row index of A's first row inside the logical matrix the dropout mask is defined on (0 unless the
                           * caller splits one GEMM into several launches: the mask index is (drop_row0 + row) * N + col) */
    /* in_dtype == A4R_FP8 (OCP e4m3fn operands, one byte per element; M % 256 == 0, N % 256 == 0, K % 128 == 0; out bf16):
     * A [M, K] carries one scale per ROW (token), B [N, K] one per ROW (output channel): the accumulator is multiplied by
     * scale_a[m] * scale_b[n] (fp32) before alpha and the bias.  Null for the other dtypes. */
    const float* scale_a; const float* scale_b;
    /* C stored as OCP e4m3 bytes (in_dtype A4R_FP8 only; C then has ONE byte per element and ldc counts bytes; out_dtype stays
     * A4R_BF16 = the type of R1 / R2 / Pre / C2) -- the next fp8 GEMM's A operand straight from this epilogue:
     *   c_fp8 = 1: C = e4m3(sat(v / c_scale)), one static scale for the whole tensor (the FFN's GELU output: O(1) values, the consumer
     *              passes a constant scale_a = c_scale);
     *   c_fp8 = 2: C = e4m3(sat(v / (scale_a[m] * c_scale))) and c_scale_out[m] = scale_a[m] * c_scale: a row of the result inherits the
     *              scale of the row of A it came from times a per-layer constant (dgrad chains: du = (dy W) * gelu', |du[m]| <~ |dy[m]| |W|). */
    int32_t c_fp8; float c_scale; float* c_scale_out;
    /* q8_tiled != 0: the 8-bit derivative tensor (C2 under c2_mode 2, Pre under A4R_DACT_MUL_Q8) is stored in the TILE-NATIVE order of the
     * 256 x 256 kernel instead of row-major: byte offset of (tile tm, tn; wave w; group g = 2 * row16 + pair; lane l) =
     *   ((tm * (N / 256) + tn) * 8 + w) * 8192 + g * 512 + l * 8,
     * so that a wave's store / load instruction moves 512 contiguous bytes (row-major: 16 segments of 32).  The tensor is only ever
     * written by the FFN-up launch and read by the `* derivative` dgrad launch of the same [M, N]; both must set the flag.  Rows that a
     * launch hands to the 128-tile kernel (small shapes) stay row-major -- a function of (M, N) only, so the two launches agree.
     * Rows the launch covers with SHORT tiles (a4r_gemm_tail_plan below: row panels of h = 32 * kp rows from row p_full * 256 on): the panel
     * at row r0 starts at byte r0 * N and holds, per N-tile tn and wave w, kp KiB in the same (group, lane) order:
     *   r0 * N + (tn * 8 + w) * kp * 1024 + g * 512 + l * 8,   g < 2 * kp.          ldc2 / ldpre must equal N. */
    int32_t q8_tiled;
} a4r_gemm_t;
int a4r_gemm_nt(void* stream, const a4r_gemm_t* g);
/* How the 256 x 256-tile kernel covers an [M, N] output whose tile count is not a multiple of the CU count: the first *p_full row
 * panels of 256 rows are whole rounds of full tiles, the rows behind them are cut into SHORT tiles of 32 * *kp rows, at most one per CU,
 * inside the same launch -- used when kp <= A4R_GEMM_TAIL (environment, default 3, up to 7, 0 = never: a short tile costs 0.6 - 0.7 of a
 * full one at kp = 1 and 0.97 at kp = 7).  Returns 1 when a short-tile tail is used (else *p_full = M / 256, *kp = 0).  Results do not
 * depend on the split (same K order per element). */
int a4r_gemm_tail_plan(int M, int N, int* p_full, int* kp);
/* the largest kp a4r_gemm_tail_plan accepts (0 - 7; k < 0 only queries; initial value: A4R_GEMM_TAIL or 3); returns the previous one.  For
 * tests and A/B runs -- change it between, never inside, a write / read pair of a tile-native 8-bit derivative. */
int a4r_gemm_tail_max(int k);
/* Leading rows (a multiple of 256; 0 .. M) of an [M, N] a4r_gemm_nt output that the 256 x 256-tile kernel computes; the rows behind them go
 * to the 128-tile kernel.  A function of (M, N), the CU count and a4r_gemm_variant only, never of the operand type: a tile-native 8-bit
 * derivative (q8_tiled) is tile-native on exactly these rows and row-major behind them for EVERY launch that writes or reads it -- e4m3
 * launches, which always run on the 256-tile kernel, are cut at the same row. */
int a4r_gemm_rows_256(int M, int N);
/* tuning knob for A/B measurements and tests: 0 = 128x128 tile, register-staged K pipeline; 1 = 128x128 tile, direct-to-LDS
 * (global_load_lds) pipeline; 2 (default) = 256x256 tile with a 2-deep LDS-DMA ring kept in flight across barriers
 * wherever M % 256 == 0, N % 256 == 0 (variant 1 elsewhere), chosen automatically: no more 256-tiles than a quarter of the CUs -> the
 * 128-tile kernel; a partial last round -> short tiles in the same launch (a4r_gemm_tail_plan); 4 = the 256 tile
 * forced (tests).  Results of 2 / 4 agree bit for bit, the others to fp32 summation order; returns the previous setting
 * (-1 for the retired variants 3 and 5, any other v only queries).  6 / 7 leave all of that alone and switch the 256 x 256-tile
 * weight-gradient kernel of a4r_gemm_tn / _tn_bias / _tn_multi off / on (tests: the same step on the 64-tile kernels). */
int a4r_gemm_variant(int v);

/* C[P,Q] (fp32, +=) = X[M,P]^T . Y[M,Q]: weight gradients of the trainable adapter matrices
 * (autograd of AdapterBlock, model/modules.py:130-134).  P % 64 == 0, Q % 64 == 0, M % 64 == 0.
 * C must be zeroed (or hold the running sum) before the call; accumulation uses fp32 atomics. */
int a4r_gemm_tn(void* stream, const void* X, int ldx, const void* Y, int ldy, float* C, int ldc,
                int M, int P, int Q, int dtype);
/* a4r_gemm_tn plus xsum[p] += sum_m X[m, p] (p < P): weight AND bias gradient of a trainable nn.Linear from one pass over dy -- dW = dy^T x,
 * db = colsum(dy) -- i.e. the autograd of every dense product of HF BertLayer / ViTLayer under `--fine_tune_to all` (Downstream/Text/run.py:366-371)
 * and in Pretraining/ (Pretraining/Text/run.py:319-324: nothing frozen).  bf16 with P % 256 == 0, Q % 256 == 0, M >= 4096: one launch of the
 * 256 x 256-tile kernel (csrc/a4r_gemm_tn256.hip), which a4r_gemm_tn uses for those shapes too; otherwise a4r_gemm_tn then a4r_colsum.
 * xsum NULL = a4r_gemm_tn.  Same argument checks and error codes as a4r_gemm_tn. */
int a4r_gemm_tn_bias(void* stream, const void* X, int ldx, const void* Y, int ldy, float* C, int ldc,
                     int M, int P, int Q, int dtype, float* xsum);
/* 1 <= n <= 4 such products over the SAME M token rows in one call: C_i[P_i, Q_i] += X_i^T Y_i, xsum_i (optional) += column sums of X_i -- e.g. the
 * query / key / value weight gradients (X_i = the three column slices of the fused qkv gradient, Y_i = the block input) and the attention output's
 * (HF BertSelfAttention / BertSelfOutput under full fine-tuning).  When every product is a large bf16 one (a4r_gemm_tn_bias above) they run as ONE
 * launch whose workgroups share the token splits and the single atomic flush; otherwise one a4r_gemm_tn_bias per product.  n outside 1..4 or any
 * product failing a4r_gemm_tn's checks: A4R_EINVAL, nothing launched. */
typedef struct a4r_tn_prob_t {
    const void* X; const void* Y; float* C; float* xsum;
    int32_t ldx, ldy, ldc, P, Q, pad_;
} a4r_tn_prob_t;
int a4r_gemm_tn_multi(void* stream, const a4r_tn_prob_t* probs, int n, int M, int dtype);
/* One SASRec transformer block of the user encoder per launch and direction (fp32; E = 64, 2 heads x 32, d_inner 256, T <= 32 rows per
 * user, adapter bottleneck d <= 32): TransformerBlock with the two bottleneck adapters of SASRecAdaptedSelfOutput (model/model.py:341-376;
 * inner_res 1) or SASRecCompacterAdaptedSelfOutput (:666-720; inner_res 0, the PHM matrices materialised):
 *   h = dropout(MHA(x) fc^T); x1 = LN1(x + A1(h)); y = LN2(x1 + A2(dropout(relu(x1 W1^T + b1) W2^T + b2)));  A(h) = act(h Wd^T + bd) Wu^T + bu [+ h]
 * attention: causal + log_mask (additive mask_neg on masked keys, model/encoders.py:24-28), probability dropout drop_attn; the two hidden
 * dropouts drop_hidden; masks = counter hash of (drop_seed, drop_site + {0, 1, 2}, element) -- applied only when train != 0.
 * One workgroup per user holds every activation of its rows in LDS; bwd recomputes the forward from x (nothing else is saved), writes dx
 * and ADDS the adapter gradients into g_* (fp32 atomics; null = not wanted; g_wd* [d, ldg_d >= 64], g_wu* [64, ldg_u >= d]).
 * Dense weights and LayerNorms are treated as frozen.  Wd* [>= 16-multiple of d rows, 64] and bd*, Wu* [64, ldwu] are zero-padded past d. */
typedef struct {
    const float *wqkv, *wfc, *w1, *b1, *w2, *b2;              /* [192, 64] (q | k | v rows), [64, 64], [256, 64], [256], [64, 256], [64] */
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;               /* [64] each */
    const float *wd1, *bd1, *wu1, *bu1, *wd2, *bd2, *wu2, *bu2;
    float *g_wd1, *g_bd1, *g_wu1, *g_bu1, *g_wd2, *g_bd2, *g_wu2, *g_bu2;
    int32_t E, n_heads, F, d, ldwu, ldg_d, ldg_u, act, inner_res;
    float eps, mask_neg, drop_attn, drop_hidden;
    uint32_t drop_site; uint64_t drop_seed;
    /* mode 1 = SASRecPfeifferAdaptedSelfOutput (model/model.py:458-471): no adapter on the attention sub-layer (wd1 .. bu1 unused but non-null),
     * va = h2 + x1; t = LN2(va); y = LN3(adapter2(t) + va) with the NEW LayerNorm ln3 (trainable: g_ln3_g / g_ln3_b, fp32 atomics). */
    int32_t mode; const float *ln3_g, *ln3_b; float *g_ln3_g, *g_ln3_b;
} a4r_sasrec_block_t;
int a4r_sasrec_block_fwd(void* stream, const a4r_sasrec_block_t* b, const float* x, const float* log_mask, float* y, int n_users, int T, int train);
int a4r_sasrec_block_bwd(void* stream, const a4r_sasrec_block_t* b, const float* x, const float* log_mask, const float* dy, float* dx,
                         int n_users, int T, int train);

/* Two such products over the same M rows in ONE launch (an adapter's dW_up = dv^T z and dW_down = dzp^T h; bf16;
 * (P1 / 64) (Q1 / 64) == (P2 / 64) (Q2 / 64)).  xsum1 / xsum2 (either may be null): xsum_k[p] (fp32, +=) = sum over the rows of
 * X_k[:, p] -- the bias gradients that go with the two weight gradients (db_up = colsum(dv), db_down = colsum(dzp)), from the same pass. */
int a4r_gemm_tn2(void* stream, const void* X1, int ldx1, const void* Y1, int ldy1, float* C1, int ldc1, int P1, int Q1,
                 const void* X2, int ldx2, const void* Y2, int ldy2, float* C2, int ldc2, int P2, int Q2, int M, int dtype,
                 float* xsum1, float* xsum2);

/* colsum[N] (fp32, +=) = sum over rows of X[M,N]: bias gradients. N % 8 == 0. */
int a4r_colsum(void* stream, const void* X, int ldx, float* out, int M, int N, int dtype);

/* Fused bottleneck adapter + residual(s) + LayerNorm, ONE launch per direction (bf16; H in {128,256,512,768,1024}; the
 * bottleneck width padded to d = 64; M % 16 == 0).  Replaces BertAdaptedSelfOutput.forward after its dense + dropout
 * (model/model.py:292-297) with AdapterBlock (modules.py:116-134), HyperComplexAdapterBlock (modules.py:209-252, effective
 * PHM matrices, no inner residual) or the adapter half of BertPfeifferAdaptedSelfOutput (model.py:321-329):
 *   forward : zp = A Wd^T + bd ; z = act(zp) ; v = z Wu^T + bu + R1 + R2 ; y = LayerNorm(v) gamma + beta
 *             A must be R1 or R2 (Houlsby: A = R1 = dense output, R2 = sub-layer input; parallel form: A = R2) or R2 must be
 *             null (Compacter: A = dense output, R1 = input; Pfeiffer: A = LN(h + input), R1 = h + input): two tensors streamed.
 *             Wd [64, H], Wu [H, 64] row-major bf16; outputs zp, z [M, 64], v, y [M, H], stats [M, 2] (mean, rstd).
 *   backward: dv = LayerNorm'(dy; v, stats, gamma) [+ dres]; dzp = (dv Wu) * act'(zp); dh = dropout_mask * (dzp Wd [+ dv if
 *             inner_res]); column sums dgamma += sum dy xhat, dbeta += sum dy, dbias += sum dv, dbd [64] += sum dzp (the
 *             down-projection's bias gradient, from the fp32 values before their bf16 store) -- each optional, fp32 atomics.
 *             flags bit 0: dbias sums dv AFTER dres is added (pre-LN towers -- HF ViTLayer under Downstream/CV/model/model.py:182-212:
 *             v is the residual stream itself, dres the gradient that reaches it from the layers above; post-LN BERT: dres is not
 *             part of the adapter's output gradient).
 *             WuT [64, H] = Wu^T, WdT [H, 64] = Wd^T; the dropout mask is the dense output's (index row * H + col).
 * Returns A4R_EINVAL for anything else: the caller then uses the three-launch form (a4r_gemm_nt x 2 + a4r_ln_fwd / a4r_ln_bwd).
 * HBM bytes per launch: (4 H + 128) * 2 * M against ~7 H * 2 * M for the three launches. */
int a4r_adapter_ln_fwd(void* stream, const void* A, int lda, const void* R1, int ldr1, const void* R2, int ldr2,
                       const void* Wd, const float* bd, const void* Wu, const float* bu,
                       const float* gamma, const float* beta, float eps, int act,
                       void* zp, void* z, void* v, int ldv, void* y, int ldy, float* stats, int M, int H, int d, int dtype,
                       void* y8, int ld8, float* ys,    /* y8 / ys (optional; y may then be NULL): y as OCP e4m3 + per-row scale, the
                                                           arithmetic of a4r_ln_fwd_fp8 (the fp8 A operand of the GEMM that follows) */
                       const float* res32, int ldres32, float* y32, int ldy32,
                       int w_frag);   /* bit 0: Wd, Wu are in FRAGMENT order (a4r_pack_matrices layouts 1 and 2 below), see the note under the backward;
                                         bit 1 (ABI 409, --residual_dtype bf24): res32 / y32 are BYTE planes (int8 [M, ld], ld in bytes, 8-byte aligned) -- the 24-bit
                                         residual stream: value = (bits(bf16 tensor) << 16) + (signed byte << 8), i.e. the fp32 value cut to 15 explicit mantissa
                                         bits, stored as its bf16 rounding (the tensor the GEMMs read, unchanged) + the offset from it (a zero byte = no offset).  One more byte
                                         read and written per element instead of four, the accuracy of the fp32 stream to 2^-15 (7 + 8 explicit mantissa bits);
                                         bit 2 (with bit 1; --residual_dtype bf20): the planes hold a signed NIBBLE per element ([M, ld] bytes, ld >= H / 2, 4-byte aligned;
                                         eight consecutive elements share a 32-bit word, element j in bits [4 j, 4 j + 4)): value = (bits(bf16) << 16) + (nibble << 12), the offset
                                         rounded to nearest -- 11 explicit mantissa bits for half a byte read and written per element.  The ORDER of the words within a row
                                         is the kernel's own (a lane's words together: one memory instruction per lane and tile) -- the plane is meaningful only to this entry
                                         point at the same H; move whole rows, never columns (tests/sim_lib.py: lo4_word_index restates the order) */
/* The LayerNorm of the forward runs on the fp32 sum v (its bf16 copy `v`, when asked for, is for the backward only).
 * res32 / y32 (both optional; --residual_dtype fp32): the residual stream between sub-layers kept in fp32, as under the reference's
 * autocast (LayerNorm outputs fp32 there and `hidden_states + input_tensor` promotes to it, HF BertSelfOutput / BertOutput under
 * Downstream/Text/model/model.py:292-297): res32 [M, H] fp32 replaces the residual operand that is NOT A (R2 in the Houlsby form, R1
 * in the Compacter form; that bf16 pointer is then only used to tell the forms apart), y32 [M, H] receives y before its bf16 rounding. */
int a4r_adapter_ln_bwd(void* stream, const void* dy, int lddy, const void* v, int ldv, const float* stats, const float* gamma,
                       const void* dres, int lddres, const void* zp, int act, const void* WuT, const void* WdT, int inner_res,
                       void* dv, int lddv, void* dzp, void* dh, int lddh, float* dgamma, float* dbeta, float* dbias,
                       int M, int H, int d, int dtype, float drop_p, uint32_t drop_site, uint64_t drop_seed, float* dbd, int flags,
                       const float* beta_y);      /* beta_y != NULL: the forward was called with v = NULL and `v` here is its y = LN(v): xhat = (y - beta_y) / gamma
                                                   * (post-LN form with frozen LayerNorm, no dres; min |gamma| is the caller's responsibility) */
/* Fragment-ordered weights (ABI 408; forward: w_frag, backward: flags bit 1): every CU of these persistent kernels reads both matrices into MFMA
 * operand fragments before its first tile.  From row-major copies a wave instruction gathers 16 row pieces of 64 bytes, which the CU's address unit
 * takes twice as long over as 1 KiB contiguous: with the copies that a4r_pack_matrices writes in fragment order (layout 1 for the [64, H] matrices
 * Wd and WuT, layout 2 for the [H, 64] matrices Wu and WdT) the first tile starts ~3 us earlier (forward 46.8 -> 43.2 us at M = 40 448, 27.5 -> 24.3
 * at M = 16 896).  Same values, same arithmetic order: results are bit-identical to the row-major form. */

/* Short-sequence self-attention, one wave per (item, head), S <= 32, dh in {32, 64} (fp32 also 128 and 256: the user tower at
 * --embedding_dim 256 / 512 with two heads, Downstream/Text/parameters.py:27-28; <= 16: scalar kernels).
 * BERT layer: HF BertSelfAttention (called from model/encoders.py:53); SASRec: SelfAttention
 * model/modules.py:31-42 with the mask of model/encoders.py:24-28.
 * qkv [n_items*S, ld]: q at column q_off, k at k_off, v at v_off, head h at +h*dh.
 * score = q.k * scale + (allowed ? 0 : mask_neg), allowed = key_mask[item][key] != 0 and
 * (!causal or key <= query); softmax over the S keys; attention-prob dropout; . V.
 * S <= 32; dh 32 or 64 (MFMA kernels) or dh <= 16 (scalar kernels for the narrow heads inside a K-Adapter,
 * Downstream/Text/model/modules.py:161-206; their dropout lots are indexed differently, fwd and bwd agree). */
typedef struct {
    const void* qkv; int32_t ld; int32_t q_off, k_off, v_off;
    void* out; int32_t ldo;              /* fwd: ctx [n_items*S, ldo], head h at column h*dh */
    const void* dout; void* dqkv;        /* bwd: d ctx (ldo) -> d qkv (ld, same offsets) */
    const float* key_mask;               /* [n_items, S] (0 = masked) or NULL */
    int32_t n_items, S, n_heads, dh, causal, dtype;
    float scale, mask_neg;
    float drop_p; uint32_t drop_site; uint64_t drop_seed;
    const int32_t* offsets;              /* ABI 407, a4r_attn_fwd / _bwd with dh 32 / 64 only (else A4R_EINVAL), NULL = off: PACKED items -- item i owns
                                          * rows [offsets[i], offsets[i + 1]) of qkv / out / dout / dqkv (device int32 [n_items + 1], every length in
                                          * 1 .. S), no pad rows exist and key_mask is ignored (a title's pad tokens never reach its CLS output:
                                          * Downstream/Text/model/encoders.py:48-57); the dropout counter stays (item * heads + head, query, key) */
} a4r_attn_t;
int a4r_attn_fwd(void* stream, const a4r_attn_t* a);
int a4r_attn_bwd(void* stream, const a4r_attn_t* a);

/* The same attention for 32 < S <= 256 tokens per item (any S <= 256 is accepted) and dh == 64 or 32 (fp32 also dh == 128 with S <= 128: the user tower
 * at the parser's default --embedding_dim 256 with two heads and --max_seq_len 33 .. 128), no packed items (offsets NULL, else A4R_EINVAL).  key_mask (ABI 408, optional): HF's attention_mask, fp32 [n_items, S], 1 = attend --
 * the text towers when --num_words_title exceeds 32 (Downstream/Text/parameters.py:44, model/encoders.py:48-57); masked keys get probability 0,
 * an item without any attended key (the PAD item) attends uniformly over its S keys, as HF's softmax over S equal scores does.
 * causal 1 is accepted WITH a key_mask only (else A4R_EINVAL): the user tower at --max_seq_len above 32 (parameters.py:29,
 * model/user_encoders.py:20-27: att_mask = log_mask & tril, added as -1e9): query t attends keys j <= t with key_mask[j] != 0; a query without any
 * such key (the left padding of a short history) attends uniformly over all S keys, which is what fp32 softmax over S scores of -1e9 + s gives.
 * Without a mask: the ViT / ViT-MAE item tower (HF ViTSelfAttention under Downstream/CV/model/encoders.py:21-32;
 * S = 197 / 50, dh 64, drop_p 0) and the TransformerBlocks inside VITKAdaptedCVModel's KAdapterBlocks
 * (Downstream/CV/model/model.py:374-404, modules.py:24-36,148-187; width 384 = 12 heads of 32, all-ones mask, dropout
 * drop_p on the probabilities with the counter (item * heads + head, query, key)).  One workgroup per (item, head);
 * nothing S x S reaches HBM.
 * fwd also writes lse [n_items, n_heads, S] fp32 (row max + log row sum of the scaled scores), which bwd reads;
 * bwd reads a->out = the ctx fwd wrote (ldo), a->dout = d ctx, and needs delta_ws [n_items, n_heads, S] fp32 scratch
 * (dO . O per query, produced by its dq launch, consumed by its dk/dv launch).  fp32 reads the transposed operands with 4-byte gathers from the same row-major LDS images (parity instantiation: any S <= 256). */
int a4r_attn_long_fwd(void* stream, const a4r_attn_t* a, float* lse);
int a4r_attn_long_bwd(void* stream, const a4r_attn_t* a, const float* lse, float* delta_ws);

/* ViT / ViT-MAE input side (HF ViTEmbeddings, ViTMAEEmbeddings under Downstream/CV/model/encoders.py:21-32).
 * a4r_patchify: out[(item*n_keep + j), c*P*P + ky*P + kx] = pixel (c, py*P + ky, px*P + kx) of patch keep_idx[item][j]
 * (keep_idx NULL: all patches in raster order, n_keep ignored) -- the im2col of the stride-P patch convolution in the
 * Conv2d weight's own column order, so the projection is a4r_gemm_nt with conv.weight.view(H, C*P*P).
 * src_kind 0: img fp32 [n, C, Himg, Wimg], already normalised (the tensor Build_Lmdb_Dataset hands over,
 * data_utils/dataset.py:85-113).  src_kind 1: img uint8 [n, Himg, Wimg, C] raw pixels; ToTensor + Normalize(0.5, 0.5)
 * of dataset.py:77-81 is applied here: (x / 255 - 0.5) / 0.5.  patch % 8 == 0.
 * a4r_vit_assemble: out[item*(n_keep+1) + 0] = cls + pos[0]; out[.. + 1 + j] = patches[item*n_keep + j] + pos[1 + idx_j]
 * (cls [H], pos [1 + n_patches, H] fp32).  tokens_out (0 = n_keep + 1): rows per item in `out`; a larger value leaves rows
 * n_keep + 1 .. tokens_out - 1 of every item untouched (soft-prompt tokens appended by the caller, Downstream/CV/model/model.py:512-535). */
int a4r_patchify(void* stream, const void* img, int src_kind, void* out, int ldo, const int32_t* keep_idx, int n_keep,
                 int n_items, int C, int Himg, int Wimg, int patch, int dtype);
int a4r_vit_assemble(void* stream, const void* patches, int ldp, const float* cls, const float* pos, const int32_t* keep_idx,
                     void* out, int ldo, int n_items, int n_keep, int H, int dtype, int tokens_out);

/* ViT-MAE random masking (HF ViTMAEEmbeddings.random_masking under Downstream/CV/model/encoders.py:8-22): keep[item][0 .. n_keep) =
 * argsort(noise[item][0 .. n_patches))[: n_keep] in the STABLE order (ties by index).  noise fp32 [n_items, n_patches], or NULL: uniform
 * [0, 1) noise drawn on the device from the counter hash (seed, site, item * n_patches + patch), 24 bits per draw.  NaNs in an explicit
 * noise tensor are not supported (they sort nowhere).  n_patches <= 4096. */
int a4r_mae_keep_indices(void* stream, const float* noise, int32_t* keep, int n_items, int n_patches, int n_keep, uint64_t seed,
                         uint32_t site);

/* One pass of Pillow's 8-bit separable resampler (third party; what torchvision's Resize((R, R)) executes on the PIL image
 * at Downstream/CV/data_utils/dataset.py:77-81): src uint8 [n_outer, in_len, inner] -> dst uint8 [n_outer, out_len, inner],
 * dst = clip8((2^21 + sum_{x < bounds[2*o+1]} src[bounds[2*o] + x] * kk[o*ksize + x]) >> 22).  Horizontal pass of a batch of
 * [n, H, W, C] images: n_outer = n*H, in_len = W, inner = C; vertical pass: n_outer = n, in_len = H, inner = W*C.
 * bounds / kk (device, int32) come from the caller (adapter4rec_amd/cv/image_io.py: resample_tables). */
int a4r_resample_u8(void* stream, const void* src, void* dst, const int32_t* bounds, const int32_t* kk, int ksize,
                    long n_outer, int in_len, int out_len, long inner);

/* HF BertEmbeddings / RobertaEmbeddings: word[id] + pos[pos_id] + type[0] -> LayerNorm -> dropout.
 * ids [n_items, S] int64 with row stride ld_ids (the reference hands over ids||mask rows of 2*S,
 * model/encoders.py:49-52).  roberta != 0: pos_id = cumsum(id != pad) * (id != pad) + pad.  A negative id -(r + 1) reads word
 * row r and counts as a pad for the position ids (soft prompt, model/model.py:586-630: the first n_tokens word vectors are
 * replaced, the positions still follow the original ids). */
int a4r_embed_ln(void* stream, const int64_t* ids, int ld_ids, const float* word, const float* pos,
                 const float* type0, const float* gamma, const float* beta, float eps,
                 void* out, int ldo, int n_items, int S, int H, int roberta, int pad_id, int dtype,
                 float drop_p, uint32_t drop_site, uint64_t drop_seed, void* pre_out, float* stats_out, float* key_mask_out);
/* key_mask_out (optional, fp32 [n_items, S]): the attention key mask = columns S .. 2S-1 of the ids || mask rows (Bert_Encoder splits
 * them at encoders.py:48-57), converted in the same pass.  pre_out (optional, same dtype / ld as out): the embedding sum before the LayerNorm; stats_out (optional, fp32 [rows, 2]):
 * mean, rstd -- what a4r_ln_bwd needs when the embedding LayerNorm or tables are trained (--finetune_layernorm, --fine_tune_to all).
 * a4r_embed_bwd: the nn.Embedding backward: dword[id] += dpre[row], dpos[pos_id] += dpre[row] (either table may be NULL). */
int a4r_embed_bwd(void* stream, const int64_t* ids, int ld_ids, const void* dpre, int ldd, float* dword, float* dpos,
                  int n_items, int S, int H, int roberta, int pad_id, int dtype);

/* y = LayerNorm(v) * gamma + beta over rows of width H (H % 8 == 0, H <= 1024);
 * stats[2*row] = mean, stats[2*row+1] = rstd (fp32) saved for backward.  add (optional, fp32 [add_rows, H])
 * is added to v first with row index (row % add_rows): SASRec position embedding (modules.py:101-106). */
int a4r_ln_fwd(void* stream, const void* v, int ldv, const float* add, int add_rows,
               const float* gamma, const float* beta, float eps,
               void* y, int ldy, float* stats, int M, int H, int dtype,
               float drop_p, uint32_t drop_site, uint64_t drop_seed);
/* y = LayerNorm(h + residual) with the sum taken in fp32 and normalised UNROUNDED (un-adapted sub-layers under --residual_dtype fp32: HF
 * BertSelfOutput / BertOutput as the reference's autocast runs them).  The residual is res32 (fp32 [M, H]) when given, else res (h's dtype).
 * Optional outputs: sum (h's dtype: the v that a4r_ln_bwd re-reads), sum32 (fp32), y32 (y before its rounding).  stats [M, 2] = (mean, rstd). */
int a4r_ln_fwd_sum(void* stream, const void* h, int ldh, const float* res32, int ldres32, const void* res, int ldres,
                   const float* gamma, const float* beta, float eps, void* y, int ldy, void* sum, int ldsum, float* sum32, int ldsum32,
                   float* y32, int ldy32, float* stats, int M, int H, int dtype);
/* fp8 (OCP e4m3fn) operands of the frozen-backbone forward GEMMs (north_star "fp8 MFMA encoder"; the reference's reduced-precision
 * path is fp16 AMP, Downstream/CV/run_adapter.py:565-593).  One fp32 scale per ROW: q[row] = e4m3(x[row] * 448 / amax(row)),
 * scale[row] = amax(row) / 448.  a4r_ln_fwd_fp8 = a4r_ln_fwd that ALSO (y optional) emits the normalised row in that form straight
 * from its fp32 registers -- the A operand of the qkv / FFN-up GEMM costs no extra pass; a4r_quant_rows_fp8 is the standalone pass
 * (H % 8 == 0, H <= 4096) for operands no fused producer exists for. */
int a4r_ln_fwd_fp8(void* stream, const void* v, int ldv, const float* add, int add_rows,
                   const float* gamma, const float* beta, float eps,
                   void* y, int ldy, void* y8, int ld8, float* yscale, float* stats, int M, int H, int dtype);
int a4r_quant_rows_fp8(void* stream, const void* x, int ldx, void* q, int ldq, float* scale, int M, int H, int dtype);
/* dv = LN backward of dy (through the same dropout mask when drop_p > 0); dgamma/dbeta (+=, fp32,
 * optional: --finetune_layernorm, run.py:496-501); dbias (+= column sums of dv, optional: the bias
 * gradient of the Linear whose output feeds v).  dres (optional) is added to dv before it is stored
 * (a residual branch that by-passes this LayerNorm: Pfeiffer, model/model.py:321-329). */
int a4r_ln_bwd(void* stream, const void* dy, int lddy, const void* v, int ldv, const float* add, int add_rows,
               const float* stats, const float* gamma, const void* dres, int lddres, void* dv, int lddv,
               float* dgamma, float* dbeta, float* dbias, int M, int H, int dtype,
               float drop_p, uint32_t drop_site, uint64_t drop_seed,
               void* dv2, int lddv2, float drop2_p, uint32_t drop2_site, uint64_t drop2_seed);
/* dv2 (optional, same dtype): dv through the dropout mask (drop2_*) -- BertSelfOutput / BertOutput apply dropout to the dense output BEFORE the
 * residual add (HF modeling_bert.py BertSelfOutput.forward, wrapped by model/model.py:292-297): the residual branch gets dv, the dense layer's dgrad
 * GEMM gets mask * dv; written by the same launch instead of a separate a4r_dropout_apply pass over dv. */

/* out[i, :] = in[i * row_stride_rows, :] (CLS gather, model/encoders.py:55) and its scatter-transpose. */
int a4r_gather_rows(void* stream, const void* in, int ldi, void* out, int ldo, int n, int row_step, int H, int dtype);
int a4r_scatter_rows(void* stream, const void* in, int ldi, void* out, int ldo, int n, int row_step, int H, int dtype);
/* Rows by index, any element type (sizes in BYTES, multiples of 16): scatter == 0: out[r] = in[idx[r]]; scatter != 0: out[idx[r]] = in[r] (idx without
 * repeats), r < n.  The item rows of a training batch that the loss really reads -- BuildTrainDataset pads short histories with item 0 in the positive
 * AND the negative slot (Downstream/Text/data_utils/dataset.py:24-49), and neither Model.forward nor ModelCPC.forward (model/model.py:48-70, 113-135) reads
 * those slots -- are gathered in front of the item tower and their embeddings scattered back into the [B, L, 2] slot layout. */
int a4r_rows_idx_copy(void* stream, const void* in, int64_t ldi_bytes, void* out, int64_t ldo_bytes, const int32_t* idx, int n,
                      int64_t row_bytes, int scatter);
/* the scatter that also ZEROES every other row of out[0 .. fill_rows): the gradient of a CLS gather is written in one pass
 * (no separate fill of the [tokens, H] buffer) */
int a4r_scatter_rows_fill(void* stream, const void* in, int ldi, void* out, int ldo, int n, int row_step, int H, int dtype, int fill_rows);

/* y = x * keep_mask(seed, site, index) / (1 - p) elementwise on [M,N] (index = row * N + col): backward of a
 * dropout whose forward ran inside a GEMM epilogue, for the cases no GEMM sits behind it. */
int a4r_dropout_apply(void* stream, const void* x, int ldx, void* y, int ldy, int M, int N, int dtype,
                      float drop_p, uint32_t drop_site, uint64_t drop_seed);

/* y = x * act'(pre) elementwise on [M,N] fp32 (GELU backward of the item head, encoders.py:57). */
int a4r_act_bwd_f32(void* stream, const float* dy, const float* pre, float* dx, int64_t n, int act);

/* Scoring head + loss, model/model.py:53-68 (SASRec) / :118-133 (CPC).
 * emb [B, L, 2, E] fp32 item embeddings (L = max_seq_len + 1), prec [B, L-1, E] user-encoder output,
 * log_mask [B, L-1].  fwd: pos/neg scores [B, L-1]; loss_ws = 4 floats zeroed by the caller:
 * [0] loss (mean over valid positions of softplus(-pos) + softplus(neg)), [1] sum, [2] valid count.
 * bwd: d_prec [B, L-1, E] and d_emb [B, L, 2, E] (the target-side part; the caller adds the input side); the incoming
 * gradient of the loss is loss_scale x (*loss_scale_dev if that device pointer is non-null: autograd's grad_output of
 * loss.backward(), run.py:599, stays on the device). */
int a4r_score_bce_fwd(void* stream, const float* emb, const float* prec, const float* log_mask,
                      float* pos, float* neg, float* loss_ws, int B, int L, int E, int cpc);
int a4r_score_bce_bwd(void* stream, const float* emb, const float* prec, const float* log_mask,
                      const float* pos, const float* neg, const float* loss_ws, float loss_scale, const float* loss_scale_dev,
                      float* d_prec, float* d_emb, int B, int L, int E, int cpc);
/* d_emb[b, l, 0, :] += d_in[b*(L-1) + l, :] for l < L-1 (gradient wrt the user-encoder input, model.py:57). */
int a4r_emb_grad_add_inputs(void* stream, const float* d_in, int ldi, float* d_emb, int B, int L, int E);
/* out[b*(L-1) + l, :] = emb[b, l, 0, :], l < L-1  (model.py:57) */
int a4r_take_inputs(void* stream, const float* emb, float* out, int ldo, int B, int L, int E);

/* torch.optim.Adam as configured at Downstream/Text/run.py:524-529 (betas .9/.999, eps 1e-8, no decay)
 * over one flat fp32 buffer; element i belongs to the first segment with seg_end[seg] > i and uses
 * group_lr[seg_group[seg]].  g is multiplied by grad_scale first (1/world for the DDP average). */
int a4r_adam_step(void* stream, float* p, const float* g, float* m, float* v, int64_t n,
                  const int32_t* seg_end, const int32_t* seg_group, int n_seg,
                  const float* group_lr, int step, float beta1, float beta2, float eps, float grad_scale);

/* Refresh the kernel-side copies of trainable matrices after an optimiser step:
 * dst[rows_pad, cols_pad] (dtype) = src (fp32 [rows, cols] at flat + src_off) or its transpose, zero padded.
 * `transpose`: bit 0 = transpose; bits 1-2 = destination layout (ABI 408): 0 row-major (stride dst_ld), 1 / 2 = the fragment order of the one-launch
 * adapter kernels for a [64, H] / an [H, 64] destination (rows_pad resp. cols_pad must be 64, H in {128, 256, 512, 768, 1024}, dst_ld ignored:
 * the destination is rows_pad x cols_pad contiguous elements; the index formulas are in a4r_head.hip: pack_dst_index). */
typedef struct {
    int64_t src_off; void* dst; int32_t rows, cols, rows_pad, cols_pad, transpose, dst_ld;   /* dst_ld 0 = cols_pad */
} a4r_pack_desc_t;
int a4r_pack_matrices(void* stream, const float* flat, const a4r_pack_desc_t* desc_dev, int n_desc, int max_elems, int dtype);

/* ---- parameter-side kernels (a4r_params.hip): what the reference leaves to eager torch ops on the parameter tensors ----
 * a4r_lora_merge: dst[o, i] = dstT[i, o] = W[o, i] + scaling * sum_r B[o, r] A[r, i] (loralib lora.Linear, run.py:414-428): the
 *   merged weight goes straight into the packed q / v rows of the fused qkv operand and its transpose (compute dtype), every step.
 * a4r_phm_build / a4r_phm_bwd: Compacter (PHMLinear, model/layers.py:25-166): E[out, in] = (sum_k kron(rule[k], W_left[k] W_right[k]))^T
 *   for n_desc PHMLinear instances in one launch, and the gradients of rule / W_left / W_right from dL/dE (+=, fp32 atomics; the
 *   gradient buffer has the parameter buffer's layout).
 * a4r_unpack_add: target[dst_off + r * cols + c] += alpha * src[r * ld + c]: the valid corners of zero-padded gradient scratch
 *   matrices into the flat gradient buffer, one launch for all of them.
 * a4r_memset_zero: hipMemsetAsync on the stream. */
typedef struct {
    int64_t rule_off, wl_off, wr_off;   /* fp32 element offsets in `params` (and in `grads`): rule [n,n,n], W_left [n, in/n], W_right [n, out/n] */
    int64_t out_off;                    /* a4r_phm_build: offset of E [out, in] (row-major, ld = in) in `eff` */
    const float* G; int32_t ldg;        /* a4r_phm_bwd: dL/dE [out, in], row stride ldg */
    int32_t in_f, out_f, n, pad_;
} a4r_phm_desc_t;
typedef struct { const float* src; int64_t dst_off; int32_t rows, cols, ld; float alpha; } a4r_add_desc_t;
int a4r_lora_merge(void* stream, const float* W, const float* A, const float* B, float scaling,
                   void* dst, int ld, void* dstT, int ldT, int out_f, int in_f, int r, int dtype);
/* The same for n_desc projections in ONE launch (every LoRA of the model; device table of descriptors, all destinations of one dtype);
 * max_elems = the largest out_f * in_f. */
typedef struct {
    const float* W; const float* A; const float* B; void* dst; void* dstT;
    float scaling; int32_t ld, ldT, out_f, in_f, r;
} a4r_lora_desc_t;
int a4r_lora_merge_batch(void* stream, const a4r_lora_desc_t* desc_dev, int n_desc, int max_elems, int dtype);
/* The low-rank gradients of a block's TWO small-rank LoRAs (r <= 8 each: loralib's Linear on the query and value projections, Downstream/CV/
 * run_adapter.py:384-395, Downstream/Text/run.py:414-428) in one pass over the rows x [M, H], dqa, dqb [M, H] (row stride lddq: slices of the fused
 * qkv gradient):  t = x [Aa ; Ab]^T,  dt = (dqa BTa^T) scale_a | (dqb BTb^T) scale_b,  dAa | dAb += dt^T x  ([8, H] each, row stride lda),
 * dBa += dqa^T t[:, 0:8], dBb += dqb^T t[:, 8:16]  ([H, 8] each, row stride ldb; WITHOUT the LoRA scaling: the caller applies it),
 * dbias_a / dbias_b (optional, element stride ldbias) += the column sums of dqa / dqb.  Aa, Ab, BTa (= B_a^T), BTb: 8 rank rows of H elements each
 * (row stride ldw; rows past a LoRA's rank must be zero).  rank_rows = 16 is the same for ranks up to 15 (the image tower's default r = 12): 16 rank
 * rows per operand, dAa | dAb [16, H], dBa | dBb [H, 15].  t and dt are rounded to bf16 between the two stages, as the separate launches store them.
 * Two launches: the pass over the rows, then the reduction of the workgroups' column sums from the workspace `ws` into the destinations.
 * bf16, H = 768, M % 16 == 0; everything else: A4R_EINVAL (the caller keeps the separate a4r_gemm_nt / a4r_gemm_tn launches for those). */
int a4r_lora_bwd_fused(void* stream, const void* x, int ldx, const void* dqa, const void* dqb, int lddq,
                       const void* Aa, const void* Ab, const void* BTa, const void* BTb, int ldw, float scale_a, float scale_b,
                       float* dAa, float* dAb, int lda, float* dBa, float* dBb, int ldb, float* dbias_a, float* dbias_b, int ldbias,
                       int M, int H, int dtype, int rank_rows, float* ws, int64_t ws_floats);
/* fp32 elements of the workspace `ws` above (the workgroups' column sums before their reduction; contents need not be kept between calls) */
int a4r_lora_bwd_fused_ws_floats(int H);
int a4r_phm_build(void* stream, const float* params, const a4r_phm_desc_t* desc_dev, int n_desc, float* eff);
int a4r_phm_bwd(void* stream, const float* params, const a4r_phm_desc_t* desc_dev, int n_desc, float* grads);
int a4r_unpack_add(void* stream, float* target, const a4r_add_desc_t* desc_dev, int n_desc, int max_elems);
int a4r_memset_zero(void* stream, void* p, int64_t bytes);

/* Eval (data_utils/metrics.py:82-116): for user u with vector prec[u] (fp32 [U,E]) and item table
 * item_emb (fp32 [N1,E], row 0 = pad item): rank[u] = 1 + #{i in 1..N1-1, i not in hist(u),
 * score_i > score_target(u)} without materialising [U,N1].  hist in CSR form (hist_ptr [U+1], hist_idx); a user's
 * history may hold at most A4R_EVAL_MAX_HISTORY ids (the reference keeps max_seq_len + 2, preprocess.py:51-59) --
 * the caller checks this (the host cannot read hist_ptr without a sync); longer lists are NOT silently truncated by
 * the Python mirror (data_utils/metrics.py raises). */
#define A4R_EVAL_MAX_HISTORY 264
int a4r_eval_rank(void* stream, const float* prec, const float* item_emb, const int32_t* target,
                  const int32_t* hist_ptr, const int32_t* hist_idx, int32_t* rank, int U, int N1, int E);

/* ONE post-LN encoder layer per call (ABI 409; SURVEY 8(b) `encoder_layer_fwd / bwd`): HF BertLayer with the reference's serial Houlsby wrappers on
 * both sub-layers (Downstream/Text/model/model.py:292-297 on attention.output and output, injected at run.py:452-465), frozen backbone.  The call
 * ENQUEUES the launches the per-kernel path issues for such a layer, in the same order with the same arguments -- results are bit-identical to
 * calling the entry points one by one:
 *   fwd: a4r_gemm_nt (qkv) | a4r_attn_fwd | a4r_gemm_nt (attention output, + dropout) | a4r_adapter_ln_fwd | a4r_gemm_nt (FFN up, GELU + derivative)
 *        | a4r_gemm_nt (FFN down, + dropout) | a4r_adapter_ln_fwd                                                                   -- 7 launches
 *   bwd: a4r_adapter_ln_bwd | a4r_gemm_tn2 | a4r_gemm_nt (d FFN-down * derivative) | a4r_gemm_nt (d FFN-up + residual) | a4r_adapter_ln_bwd | a4r_gemm_tn2
 *        | a4r_gemm_nt (d attention output) | a4r_attn_bwd | a4r_gemm_nt (d qkv + residual; skipped when dx_in is NULL)              -- 9 launches
 * Scope: dtype A4R_BF16, S <= 32 (the short attention kernels), H in {128, 256, 512, 768}, adapter bottleneck padded to 64, both adapters present;
 * anything else returns A4R_EINVAL and the caller sequences the kernels itself.  Everything is caller-allocated device memory; M = padded token rows. */
typedef struct {
    const void *wd, *wu, *wdT, *wuT;           /* [64, H], [H, 64] and their transposes (bf16) */
    const void *wd_f, *wu_f, *wdT_f, *wuT_f;   /* the same four in fragment order (a4r_pack_matrices layouts 1 / 2), or all NULL */
    const float *bd, *bu;                       /* [64], [H] */
    float *g_wu, *g_wd, *g_bu, *g_bd;           /* gradient targets (fp32, +=): [H, ldg_wu >= 64], [64, ldg_wd >= H], [H], [64]; all NULL: frozen adapter */
    int32_t ldg_wu, ldg_wd, act, pad_;
} a4r_layer_adapter_t;
typedef struct {
    int32_t M, H, F, n_items, S, n_heads, dh, causal;
    float scale, mask_neg, ln_eps, p_attn, p_hidden;   /* dropout: probabilities of the attention map and of the two dense outputs (0 = eval) */
    uint32_t drop_site; uint64_t drop_seed;             /* sites drop_site (attention), + 1 (attention output), + 2 (FFN output) */
    const float* key_mask; const int32_t* offsets;      /* a4r_attn_t.key_mask / offsets (either may be NULL) */
    const void *wqkv, *wqkvT, *wo, *woT, *wi, *wiT, *wo2, *wo2T;      /* forward operands W [out, in] and dgrad operands W^T (bf16) */
    const float *bqkv, *bo, *bi, *bo2, *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    a4r_layer_adapter_t ad[2];                          /* attention-output half, FFN-output half */
    /* saved for backward / workspace (bf16 unless stated): qkv [M, 3H]; ctx, h1, h2 [M, H]; v1 / v2 [M, H] or NULL (then backward rebuilds xhat from
     * x1 / x_out: a4r_adapter_ln_bwd's beta_y form); st1, st2 [M, 2] fp32; zp1, z1, zp2, z2 [M, 64]; u [M, F]; upre [M, F]: gelu'(pre) as bf16, or
     * one byte per element when upre_q8 (tile-native order when q8_tiled: a4r_gemm_t.q8_tiled) */
    void *qkv, *ctx, *h1, *v1, *zp1, *z1, *u, *upre, *h2, *v2, *zp2, *z2; float *st1, *st2;
    int32_t upre_q8, q8_tiled;
    const void* x_lo; void *x1_lo, *xout_lo;            /* byte (or nibble: lo_nibble) planes of the 24- / 20-bit residual stream (a4r_adapter_ln_fwd w_frag bits 1, 2): each may be NULL */
    /* backward scratch: dv1, dv2, d_h (the gradient of a dense output, both halves in turn), dx1, dctx [M, H]; dzp [M, 64]; du [M, F]; dqkv [M, 3H] (rows >= n_items * S must be zero on entry: the attention
     * backward writes the real token rows only) */
    void *dv1, *dv2, *dzp, *d_h, *du, *dx1, *dctx, *dqkv;
    int32_t lo_nibble;                                   /* the planes x_lo / x1_lo / xout_lo hold 4 bits per element (a4r_adapter_ln_fwd w_frag bit 2: [M, H / 2] bytes) */
} a4r_encoder_layer_t;
/* x [M, H] -> x1 [M, H] (the attention half's output, kept: the FFN half's residual and backward's y1) -> x_out [M, H] */
int a4r_encoder_layer_fwd(void* stream, const a4r_encoder_layer_t* l, const void* x, void* x1, void* x_out);
/* dx_out [M, H] -> dx_in [M, H] (NULL: the first layer of a frozen tower -- the d qkv product is skipped); x1 / x_out: the forward's outputs */
int a4r_encoder_layer_bwd(void* stream, const a4r_encoder_layer_t* l, const void* x1, const void* x_out, const void* dx_out, void* dx_in);

#ifdef __cplusplus
}
#endif
#endif
