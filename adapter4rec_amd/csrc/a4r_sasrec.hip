// a4r_sasrec_block_fwd / _bwd: ONE launch per SASRec transformer block and direction.
//
// Replaces, for one block of the user encoder (reference: Downstream/Text/model/modules.py:45-87 TransformerBlock,
// model/model.py:341-376 SASRecAdaptedSelfOutput, :666-720 SASRecCompacterAdaptedSelfOutput), the launch sequence
//   gemm(qkv) | attn | gemm(fc) + dropout | [adapter down | up] | ln | gemm(w_1, ReLU) | gemm(w_2) + dropout | [adapter] | ln
// (15 launches forward, ~20 backward, each 5 - 10 us on a 640-row problem: pure latency) with one workgroup per USER that keeps the
// user's T <= 32 rows x 64 columns of every activation in LDS:
//   h   = dropout(ctx W_fc^T)                 ctx = softmax(Q K^T / sqrt(dh) + mask) V   (2 heads x 32, causal + log_mask, dropout on P)
//   x1  = LN1(x + A1(h))                      A(h) = act(h Wd^T + bd) Wu^T + bu [+ h]    (Houlsby: + h; Compacter: no inner residual)
//   y   = LN2(x1 + A2(dropout(relu(x1 W1^T + b1) W2^T + b2)))
// fp32 throughout on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32 through the 16-byte chunk primitive of a4r_common.h): A operands
// are rows of an LDS image, B operands rows of the [out, in] weight matrix read straight from global memory (L2-resident: 213 KB per
// block shared by every workgroup) -- the `NT` form of every nn.Linear.  Products that contract over the token index (P V, the
// attention backward, the adapter weight gradients) read the LDS images down a column (gather_chunk).
//
// Backward RECOMPUTES the block's forward from its input (nothing but x is saved per block: the flops are nothing, the launches
// were everything), then walks back; it produces dx and the adapter gradients (dWd, dbd, dWu, dbu: fp32 atomics into the flat
// gradient buffer, one per element per workgroup).  The dense weights and the LayerNorms of the block are frozen here (the
// engine keeps the multi-launch path for --fine_tune_to all / --finetune_layernorm / Pfeiffer / parallel / LoRA / K-Adapter blocks).
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

constexpr int E = 64, NH = 2, DH = 32, F = 256, RP = 32;          // widths of the reference's SASRec (embedding_dim 64, 2 heads, d_inner 4E); rows padded to 32
constexpr int SX = E + 4, SQ = 3 * E + 4, SU = F + 4, SPR = RP + 4;      // LDS row strides in floats (+4: a 16-byte skew between rows)

struct Lay {                      // LDS layout in floats (dpe = adapter width rounded up to 16)
    int x, qkv, p, ctx, h, zp1, z1, v1, x1, u, h2, zp2, z2, v2, st, g1, total, sz;
    // fwd_only (round 6): the forward launch keeps nothing for a backward pass (the backward kernel recomputes the forward in ITS full layout), so images with
    // disjoint lifetimes share storage -- 75 KB instead of 150 KB, TWO workgroups per CU (one user per workgroup is latency-bound: the evaluation's user tower,
    // 8 192 users per launch, ran 32 workgroups deep on every CU).  Lifetimes in block_forward: qkv, p die with ctx (step 2); u is written in step 5 over qkv + p;
    // h (steps 3 - 4; Pfeiffer: again in step 7, when u is dead) sits where p was; x dies with v1 (step 4) and h2 (step 6) takes its place; zp2 / z2 / v2 (step 7)
    // re-use zp1 / z1 / v1 (step 4); the log_mask row needs 32 floats, not an image.
    __host__ __device__ explicit Lay(int dpe, bool fwd_only = false) {
        sz = dpe + 4;
        int o = 0;
        auto take = [&](int n) { const int r = o; o += n; return r; };
        if (fwd_only) {
            x = take(RP * SX); h2 = x;
            qkv = take(RP * SQ); p = take(NH * RP * SPR); u = qkv; h = p;
            static_assert(RP * SU <= RP * SQ + NH * RP * SPR && RP * SX <= NH * RP * SPR, "u over qkv + p, h over p");
            ctx = take(RP * SX);
            zp1 = take(RP * sz); z1 = take(RP * sz); zp2 = zp1; z2 = z1;
            v1 = take(RP * SX); v2 = v1;
            x1 = take(RP * SX);
            st = take(6 * RP); g1 = take(RP);
            total = o;
            return;
        }
        x = take(RP * SX); qkv = take(RP * SQ); p = take(NH * RP * SPR); ctx = take(RP * SX); h = take(RP * SX);
        zp1 = take(RP * sz); z1 = take(RP * sz); v1 = take(RP * SX); x1 = take(RP * SX); u = take(RP * SU); h2 = take(RP * SX);
        zp2 = take(RP * sz); z2 = take(RP * sz); v2 = take(RP * SX); st = take(6 * RP); g1 = take(RP * SX);
        total = o;
    }
};

struct BlockW {                   // device pointers of one block (fp32)
    const float *wqkv, *wfc, *w1, *b1, *w2, *b2;                    // [3E, E], [E, E], [F, E], [F], [E, F], [E]
    const float *ln1g, *ln1b, *ln2g, *ln2b, *ln3g, *ln3b;          // ln3: the Pfeiffer form's new LayerNorm (mode 1)
    const float *wd1, *bd1, *wu1, *bu1, *wd2, *bd2, *wu2, *bu2;     // adapters: Wd [dp, E] (ld E), bd [dp], Wu [E, dp] (ld ldwu), bu [E]
    int ldwu, d, act, inner_res, mode;                              // d = true bottleneck width, rows d.. of Wd / columns d.. of Wu are zero padding
                                                                    // mode 0: an adapter after both sub-layers (Houlsby / Compacter); 1: Pfeiffer (adapter after LN2, + LN3)
    float eps, scale, mask_neg;
    float p_attn, p_hidden;
    uint32_t thr_attn, thr_hidden, site;
    float ks_attn, ks_hidden;
    uint64_t seed;
};

struct BlockG {                   // gradient sinks (fp32, +=), any may be null
    float *wd1, *bd1, *wu1, *bu1, *wd2, *bd2, *wu2, *bu2, *ln3g, *ln3b;
    int ldgd, ldgu;               // leading dimensions of the Wd / Wu gradient matrices ([d, E] -> E or the padded scratch's; [E, d] -> d or dp)
};

// ---- C[RP x 16] tiles: rows from an LDS image (row stride lda floats), B rows from a [n][k] matrix (global or LDS), K a multiple of 16
template <int K>
A4R_DEV void mm_nt(const float* A, int lda, const float* B, int ldb, int n0, int lane, f32x4_t (&acc)[2]) {
    const int i = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < K / 16; ++ks) {
        const uint4 b = *reinterpret_cast<const uint4*>(B + (size_t)(n0 + i) * ldb + ks * 16 + kg * 4);
        const uint4 a0 = *reinterpret_cast<const uint4*>(A + i * lda + ks * 16 + kg * 4);
        const uint4 a1 = *reinterpret_cast<const uint4*>(A + (16 + i) * lda + ks * 16 + kg * 4);
        Mma<float>::mma(a0, b, acc[0]);
        Mma<float>::mma(a1, b, acc[1]);
    }
}
// the same with B addressed [k][n] (contraction index down the rows of an LDS image): B chunk = 4 consecutive k at column n0 + i
template <int K>
A4R_DEV void mm_nn(const float* A, int lda, const float* Bt, int ldbt, int n0, int lane, f32x4_t (&acc)[2]) {
    const int i = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < K / 16; ++ks) {
        uint4 b;
        uint32_t* bo = &b.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) bo[j] = __float_as_uint(Bt[(ks * 16 + kg * 4 + j) * ldbt + n0 + i]);
        const uint4 a0 = *reinterpret_cast<const uint4*>(A + i * lda + ks * 16 + kg * 4);
        const uint4 a1 = *reinterpret_cast<const uint4*>(A + (16 + i) * lda + ks * 16 + kg * 4);
        Mma<float>::mma(a0, b, acc[0]);
        Mma<float>::mma(a1, b, acc[1]);
    }
}
// both operands contracted over the ROW index of LDS images: C[m][n] = sum_r At[r][m0 + m] Bt[r][n0 + n]  (one 16 x 16 tile, r < RP)
A4R_DEV void mm_tn(const float* At, int ldat, int m0, const float* Bt, int ldbt, int n0, int lane, f32x4_t& acc) {
    const int i = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < RP / 16; ++ks) {
        uint4 a, b;
        uint32_t *ao = &a.x, *bo = &b.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ao[j] = __float_as_uint(At[(ks * 16 + kg * 4 + j) * ldat + m0 + i]);
            bo[j] = __float_as_uint(Bt[(ks * 16 + kg * 4 + j) * ldbt + n0 + i]);
        }
        Mma<float>::mma(a, b, acc);
    }
}
// accumulator tile (rows rt * 16 + (lane >> 4) * 4 + r, column n0 + (lane & 15)) -> f(row, col, value)
template <typename Fn> A4R_DEV void tile_each(const f32x4_t (&acc)[2], int n0, int lane, Fn f) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) f(rt * 16 + (lane >> 4) * 4 + r, n0 + (lane & 15), acc[rt][r]);
}
A4R_DEV void zero2(f32x4_t (&acc)[2]) { acc[0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; }

A4R_DEV bool keep_elem(uint64_t seed, uint32_t site, uint64_t e, uint32_t thr) { return thr == 0 || dropout_keep(seed, site, e, thr); }

// LayerNorm of T rows (one wave per row, 64 columns = 64 lanes): y = (v - mean) rstd gamma + beta; stats -> st[2 r], st[2 r + 1]
A4R_DEV void ln_rows(const float* v, float* y, float* st, const float* g, const float* b, float eps, int T, int wave, int lane) {
    for (int r = wave; r < RP; r += 4) {
        if (r < T) {
            const float x = v[r * SX + lane];
            const float mean = wave_sum(x) * (1.f / E);
            const float d = x - mean;
            const float rstd = rsqrtf(wave_sum(d * d) * (1.f / E) + eps);
            y[r * SX + lane] = d * rstd * g[lane] + b[lane];
            if (lane == 0) { st[2 * r] = mean; st[2 * r + 1] = rstd; }
        } else {
            y[r * SX + lane] = 0.f;
        }
    }
}

// ---- the block's forward on LDS images.  lds[L.x] holds the input rows (rows >= T zero).  On return L.v2 / st[2..3] hold the second
// LayerNorm's input and statistics; y (LN2 output) is written to `yout` (an LDS image, stride SX).
template <bool TRAIN>
A4R_DEV void block_forward(float* lds, const Lay& L, const BlockW& w, const float* km, int user, int T, float* yout, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    float* X = lds + L.x; float* QKV = lds + L.qkv; float* P = lds + L.p; float* CTX = lds + L.ctx; float* H = lds + L.h;
    float* ZP1 = lds + L.zp1; float* Z1 = lds + L.z1; float* V1 = lds + L.v1; float* X1 = lds + L.x1; float* U = lds + L.u;
    float* H2 = lds + L.h2; float* ZP2 = lds + L.zp2; float* Z2 = lds + L.z2; float* V2 = lds + L.v2; float* ST = lds + L.st;
    const int dpe = L.sz - 4;
    // 1. qkv = x Wqkv^T  (12 column tiles, 3 per wave)
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {                 // (a compile-time trip count: the 12 B-operand loads of the stage are requested together)
        const int ct = wave + 4 * t3;
        f32x4_t acc[2]; zero2(acc);
        mm_nt<E>(X, SX, w.wqkv, E, ct * 16, lane, acc);
        tile_each(acc, ct * 16, lane, [&](int r, int c, float v) { QKV[r * SQ + c] = v; });
    }
    __syncthreads();
    // 2. attention: wave = (head, row tile); scores of 16 queries x 32 keys in two accumulator tiles
    {
        const int hd = wave >> 1, rt = wave & 1, i = lane & 15, kg = lane >> 4;
        f32x4_t s[2];
        s[0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; s[1] = s[0];
#pragma unroll
        for (int ks = 0; ks < DH / 16; ++ks) {
            const uint4 a = *reinterpret_cast<const uint4*>(QKV + (rt * 16 + i) * SQ + hd * DH + ks * 16 + kg * 4);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const uint4 b = *reinterpret_cast<const uint4*>(QKV + (ct * 16 + i) * SQ + E + hd * DH + ks * 16 + kg * 4);
                Mma<float>::mma(a, b, s[ct]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = rt * 16 + kg * 4 + r;
            float x[2], m = -INFINITY;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int key = ct * 16 + i;
                const float v = s[ct][r] * w.scale;
                const bool allowed = key < T && km[key] != 0.f && key <= q;
                x[ct] = key < T ? (allowed ? v : v + w.mask_neg) : -INFINITY;
                m = fmaxf(m, x[ct]);
            }
            m = group16_max(m);
            float e[2], sum = 0.f;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) { e[ct] = __expf(x[ct] - m); sum += e[ct]; }
            sum = group16_sum(sum);
            const float inv = 1.f / sum;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) P[(hd * RP + q) * SPR + ct * 16 + i] = q < T ? e[ct] * inv : 0.f;      // pre-dropout probabilities
        }
    }
    __syncthreads();
    // ctx_h = dropout(P_h) V_h : wave = (head, column tile of the head's 32 columns)
    {
        const int hd = wave >> 1, ct = wave & 1, i = lane & 15, kg = lane >> 4;
        f32x4_t acc[2]; zero2(acc);
#pragma unroll
        for (int ks = 0; ks < RP / 16; ++ks) {
            uint4 b;
            uint32_t* bo = &b.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) bo[j] = __float_as_uint(QKV[(ks * 16 + kg * 4 + j) * SQ + 2 * E + hd * DH + ct * 16 + i]);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                float pa[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int q = rt * 16 + i, key = ks * 16 + kg * 4 + j;
                    float pv = P[(hd * RP + q) * SPR + key];
                    if (TRAIN && w.thr_attn) pv = keep_elem(w.seed, w.site, (((uint64_t)(user * NH + hd) * 32 + q) << 5) + key, w.thr_attn) ? pv * w.ks_attn : 0.f;
                    pa[j] = pv;
                }
                const uint4 a = make_uint4(__float_as_uint(pa[0]), __float_as_uint(pa[1]), __float_as_uint(pa[2]), __float_as_uint(pa[3]));
                Mma<float>::mma(a, b, acc[rt]);
            }
        }
        tile_each(acc, hd * DH + ct * 16, lane, [&](int r, int c, float v) { CTX[r * SX + c] = v; });
    }
    __syncthreads();
    // 3. h = dropout(ctx Wfc^T)   (4 column tiles, one per wave)
    {
        f32x4_t acc[2]; zero2(acc);
        mm_nt<E>(CTX, SX, w.wfc, E, wave * 16, lane, acc);
        tile_each(acc, wave * 16, lane, [&](int r, int c, float v) {
            if (TRAIN && w.thr_hidden) v = keep_elem(w.seed, w.site + 1, ((uint64_t)user * 32 + r) * E + c, w.thr_hidden) ? v * w.ks_hidden : 0.f;
            H[r * SX + c] = v;
        });
    }
    __syncthreads();
    // 4. [adapter 1] + residual -> v1, LN1 -> x1
    if (w.mode == 1) {                                   // Pfeiffer: the attention sub-layer carries no adapter (model.py:458-471)
        for (int id = tid; id < RP * E; id += 256) { const int r = id >> 6, c = id & 63; V1[r * SX + c] = H[r * SX + c] + X[r * SX + c]; }
    } else {
    for (int ct = wave; ct < dpe / 16; ct += 4) {
        f32x4_t acc[2]; zero2(acc);
        mm_nt<E>(H, SX, w.wd1, E, ct * 16, lane, acc);
        tile_each(acc, ct * 16, lane, [&](int r, int c, float v) {
            v += w.bd1[c];
            ZP1[r * L.sz + c] = v;
            Z1[r * L.sz + c] = c < w.d ? act_fwd(v, w.act) : 0.f;
        });
    }
    __syncthreads();
    {
        f32x4_t acc[2]; zero2(acc);
        const int i = lane & 15, kg = lane >> 4;
        for (int ks = 0; ks < dpe / 16; ++ks) {
            const uint4 b = *reinterpret_cast<const uint4*>(w.wu1 + (size_t)(wave * 16 + i) * w.ldwu + ks * 16 + kg * 4);
            const uint4 a0 = *reinterpret_cast<const uint4*>(Z1 + i * L.sz + ks * 16 + kg * 4);
            const uint4 a1 = *reinterpret_cast<const uint4*>(Z1 + (16 + i) * L.sz + ks * 16 + kg * 4);
            Mma<float>::mma(a0, b, acc[0]);
            Mma<float>::mma(a1, b, acc[1]);
        }
        tile_each(acc, wave * 16, lane, [&](int r, int c, float v) {
            V1[r * SX + c] = v + w.bu1[c] + (w.inner_res ? H[r * SX + c] : 0.f) + X[r * SX + c];
        });
    }
    }
    __syncthreads();
    ln_rows(V1, X1, ST, w.ln1g, w.ln1b, w.eps, T, wave, lane);
    __syncthreads();
    // 5. u = relu(x1 W1^T + b1)   (16 column tiles, 4 per wave)
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
        const int ct = wave + 4 * t4;
        f32x4_t acc[2]; zero2(acc);
        mm_nt<E>(X1, SX, w.w1, E, ct * 16, lane, acc);
        tile_each(acc, ct * 16, lane, [&](int r, int c, float v) { v += w.b1[c]; U[r * SU + c] = v > 0.f ? v : 0.f; });
    }
    __syncthreads();
    // 6. h2 = dropout(u W2^T + b2)
    {
        f32x4_t acc[2]; zero2(acc);
        mm_nt<F>(U, SU, w.w2, F, wave * 16, lane, acc);
        tile_each(acc, wave * 16, lane, [&](int r, int c, float v) {
            v += w.b2[c];
            if (TRAIN && w.thr_hidden) v = keep_elem(w.seed, w.site + 2, ((uint64_t)user * 32 + r) * E + c, w.thr_hidden) ? v * w.ks_hidden : 0.f;
            H2[r * SX + c] = v;
        });
    }
    __syncthreads();
    // 7. mode 0: adapter 2 on h2, + x1 -> v2, LN2 -> y
    //    mode 1 (Pfeiffer, model.py:321-329 / :458-471): va = h2 + x1; t = LN2(va); v3 = adapter(t) + va; y = LN3(v3)
    //    (buffers: va -> V2, t -> H (dead since v1), v3 -> H2 (dead once va exists); statistics of LN2 at ST + 2 RP, of LN3 at ST + 4 RP)
    const float* AIN = H2;                               // the adapter's input rows
    if (w.mode == 1) {
        for (int id = tid; id < RP * E; id += 256) { const int r = id >> 6, c = id & 63; V2[r * SX + c] = H2[r * SX + c] + X1[r * SX + c]; }
        __syncthreads();
        ln_rows(V2, H, ST + 2 * RP, w.ln2g, w.ln2b, w.eps, T, wave, lane);
        __syncthreads();
        AIN = H;
    }
    for (int ct = wave; ct < dpe / 16; ct += 4) {
        f32x4_t acc[2]; zero2(acc);
        mm_nt<E>(AIN, SX, w.wd2, E, ct * 16, lane, acc);
        tile_each(acc, ct * 16, lane, [&](int r, int c, float v) {
            v += w.bd2[c];
            ZP2[r * L.sz + c] = v;
            Z2[r * L.sz + c] = c < w.d ? act_fwd(v, w.act) : 0.f;
        });
    }
    __syncthreads();
    {
        f32x4_t acc[2]; zero2(acc);
        const int i = lane & 15, kg = lane >> 4;
        for (int ks = 0; ks < dpe / 16; ++ks) {
            const uint4 b = *reinterpret_cast<const uint4*>(w.wu2 + (size_t)(wave * 16 + i) * w.ldwu + ks * 16 + kg * 4);
            const uint4 a0 = *reinterpret_cast<const uint4*>(Z2 + i * L.sz + ks * 16 + kg * 4);
            const uint4 a1 = *reinterpret_cast<const uint4*>(Z2 + (16 + i) * L.sz + ks * 16 + kg * 4);
            Mma<float>::mma(a0, b, acc[0]);
            Mma<float>::mma(a1, b, acc[1]);
        }
        if (w.mode == 1) {
            tile_each(acc, wave * 16, lane, [&](int r, int c, float v) { H2[r * SX + c] = v + w.bu2[c] + V2[r * SX + c]; });
        } else {
            tile_each(acc, wave * 16, lane, [&](int r, int c, float v) {
                V2[r * SX + c] = v + w.bu2[c] + (w.inner_res ? H2[r * SX + c] : 0.f) + X1[r * SX + c];
            });
        }
    }
    __syncthreads();
    if (w.mode == 1) ln_rows(H2, yout, ST + 4 * RP, w.ln3g, w.ln3b, w.eps, T, wave, lane);
    else ln_rows(V2, yout, ST + 2 * RP, w.ln2g, w.ln2b, w.eps, T, wave, lane);
    __syncthreads();
}

A4R_DEV void load_rows(float* dst, const float* src, int T, int tid) {     // [T, E] global -> LDS image (stride SX), rows >= T zero
    for (int id = tid; id < RP * (E / 4); id += 256) {
        const int r = id / (E / 4), c4 = id % (E / 4);
        const float4 v = r < T ? *reinterpret_cast<const float4*>(src + (size_t)r * E + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(dst + r * SX + c4 * 4) = v;
    }
}
A4R_DEV void store_rows(float* dst, const float* src, int T, int tid) {
    for (int id = tid; id < T * (E / 4); id += 256) {
        const int r = id / (E / 4), c4 = id % (E / 4);
        *reinterpret_cast<float4*>(dst + (size_t)r * E + c4 * 4) = *reinterpret_cast<const float4*>(src + r * SX + c4 * 4);
    }
}

template <bool TRAIN>
__global__ void __launch_bounds__(256) sasrec_block_fwd_kernel(const float* __restrict__ x, const float* __restrict__ log_mask, float* __restrict__ y,
                                                                BlockW w, int T, int dpe) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Lay L(dpe, true);
    const int user = blockIdx.x, tid = threadIdx.x;
    load_rows(lds + L.x, x + (size_t)user * T * E, T, tid);
    float* km = lds + L.g1;                                       // the user's log_mask row (T floats)
    if (tid < RP) km[tid] = tid < T ? log_mask[(size_t)user * T + tid] : 0.f;
    __syncthreads();
    block_forward<TRAIN>(lds, L, w, km, user, T, lds + L.ctx, tid);      // (CTX is dead after step 3: LN2's output lands there)
    store_rows(y + (size_t)user * T * E, lds + L.ctx, T, tid);
}

// ------------------------------------------------------------------------------------------------ backward
// LayerNorm backward of T rows: dv = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)), dxhat = dy gamma; in place over dy
A4R_DEV void ln_bwd_rows(float* dy, const float* v, const float* st, const float* g, int T, int wave, int lane) {
    for (int r = wave; r < RP; r += 4) {
        if (r < T) {
            const float mean = st[2 * r], rstd = st[2 * r + 1];
            const float xh = (v[r * SX + lane] - mean) * rstd;
            const float dxh = dy[r * SX + lane] * g[lane];
            const float m1 = wave_sum(dxh) * (1.f / E), m2 = wave_sum(dxh * xh) * (1.f / E);
            dy[r * SX + lane] = rstd * (dxh - m1 - xh * m2);
        } else {
            dy[r * SX + lane] = 0.f;
        }
    }
}

// the same, also accumulating dgamma += sum_r dy xhat, dbeta += sum_r dy (a trainable LayerNorm: the Pfeiffer form's LN3)
A4R_DEV void ln_bwd_rows_params(float* dy, const float* v, const float* st, const float* g, float* dg, float* db, int T, int wave, int lane) {
    float ag = 0.f, ab = 0.f;
    for (int r = wave; r < RP; r += 4) {
        if (r < T) {
            const float mean = st[2 * r], rstd = st[2 * r + 1];
            const float xh = (v[r * SX + lane] - mean) * rstd;
            const float d0 = dy[r * SX + lane];
            ag += d0 * xh;
            ab += d0;
            const float dxh = d0 * g[lane];
            const float m1 = wave_sum(dxh) * (1.f / E), m2 = wave_sum(dxh * xh) * (1.f / E);
            dy[r * SX + lane] = rstd * (dxh - m1 - xh * m2);
        } else {
            dy[r * SX + lane] = 0.f;
        }
    }
    if (dg) atomicAdd(dg + lane, ag);
    if (db) atomicAdd(db + lane, ab);
}

// adapter backward on LDS images: dv [RP, E] (gradient at the adapter's output = LN input gradient), h = the adapter's input.
//   dzp = (dv Wu) * act'(zp)   [RP, dpe]      -> DZ (an LDS image, stride L.sz)
//   dh  = dzp Wd [+ dv]        [RP, E]        -> DH
//   dWu += dv^T z, dbu += colsum(dv), dWd += dzp^T h, dbd += colsum(dzp)   (atomics)
A4R_DEV void adapter_backward(const float* DV, const float* Hin, const float* ZP, const float* Z, float* DZ, float* DH, const Lay& L,
                              const float* wd, const float* wu, int ldwu, int d, int act, int inner_res,
                              float* gwd, float* gbd, float* gwu, float* gbu, int ldgd, int ldgu, int T, int tid) {
    const int lane = tid & 63, wave = tid >> 6, dpe = L.sz - 4;
    // dzp = dv Wu: B[n = bottleneck column][k = E] = Wu[k][n] -> contraction down the rows of Wu (global, ld ldwu)
    for (int ct = wave; ct < dpe / 16; ct += 4) {
        f32x4_t acc[2]; zero2(acc);
        mm_nn<E>(DV, SX, wu, ldwu, ct * 16, lane, acc);
        tile_each(acc, ct * 16, lane, [&](int r, int c, float v) { DZ[r * L.sz + c] = (c < d && r < T) ? v * act_bwd(ZP[r * L.sz + c], act) : 0.f; });
    }
    __syncthreads();
    // dh = dzp Wd [+ dv]: B[n = E column][k = bottleneck] = Wd[k][n] (global [dp, E])
    {
        f32x4_t acc[2]; zero2(acc);
        const int i = lane & 15, kg = lane >> 4;
        for (int ks = 0; ks < dpe / 16; ++ks) {
            uint4 b;
            uint32_t* bo = &b.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) bo[j] = __float_as_uint(wd[(size_t)(ks * 16 + kg * 4 + j) * E + wave * 16 + i]);
            const uint4 a0 = *reinterpret_cast<const uint4*>(DZ + i * L.sz + ks * 16 + kg * 4);
            const uint4 a1 = *reinterpret_cast<const uint4*>(DZ + (16 + i) * L.sz + ks * 16 + kg * 4);
            Mma<float>::mma(a0, b, acc[0]);
            Mma<float>::mma(a1, b, acc[1]);
        }
        tile_each(acc, wave * 16, lane, [&](int r, int c, float v) { DH[r * SX + c] = v + (inner_res ? DV[r * SX + c] : 0.f); });
    }
    // weight gradients (rows >= T of DV / DZ are zero).  dWu [E, d] = dv^T z ; dWd [d, E] = dzp^T h
    if (gwu) {
        for (int t = wave; t < (E / 16) * (dpe / 16); t += 4) {
            const int mt = t / (dpe / 16), nt = t % (dpe / 16);
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            mm_tn(DV, SX, mt * 16, Z, L.sz, nt * 16, lane, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + (lane >> 4) * 4 + r, n = nt * 16 + (lane & 15);
                if (n < d) atomicAdd(gwu + (size_t)m * ldgu + n, acc[r]);
            }
        }
    }
    if (gwd) {
        for (int t = wave; t < (dpe / 16) * (E / 16); t += 4) {
            const int mt = t / (E / 16), nt = t % (E / 16);
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            mm_tn(DZ, L.sz, mt * 16, Hin, SX, nt * 16, lane, acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + (lane >> 4) * 4 + r, n = nt * 16 + (lane & 15);
                if (m < d) atomicAdd(gwd + (size_t)m * ldgd + n, acc[r]);
            }
        }
    }
    if (gbu && tid < E) {
        float s = 0.f;
        for (int r = 0; r < T; ++r) s += DV[r * SX + tid];
        atomicAdd(gbu + tid, s);
    }
    if (gbd && tid >= 64 && tid < 64 + d) {
        const int c = tid - 64;
        float s = 0.f;
        for (int r = 0; r < T; ++r) s += DZ[r * L.sz + c];
        atomicAdd(gbd + c, s);
    }
    __syncthreads();
}

// LDS reuse in the backward pass (after the recomputed forward; "dead" = no later reader):
//   G1  <- dY, LN2 backward in place = dV2                       CTX (dead) <- dzp of either adapter, later dCTX
//   V2  (dead after LN2 backward) <- adapter 2's input gradient, through h2's dropout mask = dO2
//   U   <- dU in place (relu' = u > 0), later dP / dS [NH][RP][SPR]
//   H2  (dead after adapter 2's weight gradients) <- dX1 = dU W1 + dV2, LN1 backward in place = dV1 (kept to the end: residual of dX)
//   V1  (dead after LN1 backward) <- adapter 1's input gradient, through h's dropout mask = dO1
//   [H, ZP1, Z1, V1, X1] (contiguous, all dead once dCTX is formed) <- dQKV [RP][SQ]
// Pfeiffer (mode 1): v3 sits in H2, t in H, va in V2; dT -> X1, LN2 backward in place there, + dV3 = dVA (kept in X1 as the residual
// gradient), dO2 -> V2; from dU on the two modes share the code.
template <bool TRAIN>
__global__ void __launch_bounds__(256) sasrec_block_bwd_kernel(const float* __restrict__ x, const float* __restrict__ log_mask, const float* __restrict__ dy,
                                                                float* __restrict__ dx, BlockW w, BlockG g, int T, int dpe) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const Lay L(dpe);
    const int user = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    load_rows(lds + L.x, x + (size_t)user * T * E, T, tid);
    float* km = lds + L.g1;                                        // log_mask row: only the recomputed forward needs it (masked P are exact zeros)
    if (tid < RP) km[tid] = tid < T ? log_mask[(size_t)user * T + tid] : 0.f;
    __syncthreads();
    block_forward<TRAIN>(lds, L, w, km, user, T, lds + L.ctx, tid);      // y itself is not needed: it lands in the dead CTX image
    float* QKV = lds + L.qkv; float* P = lds + L.p; float* CTX = lds + L.ctx; float* H = lds + L.h;
    float* ZP1 = lds + L.zp1; float* Z1 = lds + L.z1; float* V1 = lds + L.v1; float* U = lds + L.u;
    float* H2 = lds + L.h2; float* ZP2 = lds + L.zp2; float* Z2 = lds + L.z2; float* V2 = lds + L.v2; float* ST = lds + L.st; float* G1 = lds + L.g1;
    float* DZ = CTX;                                               // [RP][sz] inside an [RP][SX] image (sz <= SX)
    float* X1b = lds + L.x1;
    load_rows(G1, dy + (size_t)user * T * E, T, tid);
    __syncthreads();
    const float* RES2;                                             // the gradient that reaches x1 along the residual of the FFN sub-layer
    if (w.mode == 1) {
        // Pfeiffer: y = LN3(v3), v3 = adapter(t) + va, t = LN2(va), va = h2 + x1   (v3 in H2, t in H, va in V2)
        ln_bwd_rows_params(G1, H2, ST + 4 * RP, w.ln3g, g.ln3g, g.ln3b, T, wave, lane);      // dV3
        __syncthreads();
        adapter_backward(G1, H, ZP2, Z2, DZ, X1b, L, w.wd2, w.wu2, w.ldwu, w.d, w.act, 0, g.wd2, g.bd2, g.wu2, g.bu2, g.ldgd, g.ldgu, T, tid);    // dT -> X1b
        ln_bwd_rows(X1b, V2, ST + 2 * RP, w.ln2g, T, wave, lane);
        __syncthreads();
        for (int id = tid; id < RP * E; id += 256) {               // dVA = LN2 backward + dV3; dO2 = dVA through h2's dropout mask -> V2
            const int r = id >> 6, c = id & 63;
            const float dva = X1b[r * SX + c] + G1[r * SX + c];
            X1b[r * SX + c] = dva;
            float o = dva;
            if (TRAIN && w.thr_hidden) o = keep_elem(w.seed, w.site + 2, ((uint64_t)user * 32 + r) * E + c, w.thr_hidden) ? dva * w.ks_hidden : 0.f;
            V2[r * SX + c] = o;
        }
        __syncthreads();
        RES2 = X1b;
    } else {
    ln_bwd_rows(G1, V2, ST + 2 * RP, w.ln2g, T, wave, lane);       // dV2
    __syncthreads();
    adapter_backward(G1, H2, ZP2, Z2, DZ, V2, L, w.wd2, w.wu2, w.ldwu, w.d, w.act, w.inner_res, g.wd2, g.bd2, g.wu2, g.bu2, g.ldgd, g.ldgu, T, tid);
    if (TRAIN && w.thr_hidden) {                                   // through the dropout of h2 -> dO2
        for (int id = tid; id < RP * E; id += 256) {
            const int r = id >> 6, c = id & 63;
            V2[r * SX + c] = keep_elem(w.seed, w.site + 2, ((uint64_t)user * 32 + r) * E + c, w.thr_hidden) ? V2[r * SX + c] * w.ks_hidden : 0.f;
        }
        __syncthreads();
    }
        RES2 = G1;
    }
    // dU = (dO2 W2) * relu'(u), in place over U: B[n = F column][k = E] = W2[k][n] (global [E, F])
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
        const int ct = wave + 4 * t4;
        f32x4_t acc[2]; zero2(acc);
        mm_nn<E>(V2, SX, w.w2, F, ct * 16, lane, acc);
        tile_each(acc, ct * 16, lane, [&](int r, int c, float v) { U[r * SU + c] = U[r * SU + c] > 0.f ? v : 0.f; });
    }
    __syncthreads();
    // dX1 = dU W1 + (the residual branch's gradient: dV2 / dVA) -> H2: B[n = E column][k = F] = W1[k][n] (global [F, E])
    {
        f32x4_t acc[2]; zero2(acc);
        mm_nn<F>(U, SU, w.w1, E, wave * 16, lane, acc);
        tile_each(acc, wave * 16, lane, [&](int r, int c, float v) { H2[r * SX + c] = v + RES2[r * SX + c]; });
    }
    __syncthreads();
    ln_bwd_rows(H2, V1, ST, w.ln1g, T, wave, lane);                // dV1
    __syncthreads();
    if (w.mode == 1) {                                             // no adapter on the attention sub-layer: dH = dV1
        for (int id = tid; id < RP * E; id += 256) { const int r = id >> 6, c = id & 63; V1[r * SX + c] = H2[r * SX + c]; }
        __syncthreads();
    } else {
        adapter_backward(H2, H, ZP1, Z1, DZ, V1, L, w.wd1, w.wu1, w.ldwu, w.d, w.act, w.inner_res, g.wd1, g.bd1, g.wu1, g.bu1, g.ldgd, g.ldgu, T, tid);
    }
    if (TRAIN && w.thr_hidden) {                                   // through the dropout of h -> dO1
        for (int id = tid; id < RP * E; id += 256) {
            const int r = id >> 6, c = id & 63;
            V1[r * SX + c] = keep_elem(w.seed, w.site + 1, ((uint64_t)user * 32 + r) * E + c, w.thr_hidden) ? V1[r * SX + c] * w.ks_hidden : 0.f;
        }
        __syncthreads();
    }
    // dCTX = dO1 Wfc -> CTX: B[n][k] = Wfc[k][n]
    {
        f32x4_t acc[2]; zero2(acc);
        mm_nn<E>(V1, SX, w.wfc, E, wave * 16, lane, acc);
        tile_each(acc, wave * 16, lane, [&](int r, int c, float v) { CTX[r * SX + c] = v; });
    }
    __syncthreads();
    // attention backward.  dPd_h = dCTX_h V_h^T (rows of dCTX x rows of V) is the gradient of the DROPPED probabilities; dP = dPd * mask
    // (mask = 0 or 1 / (1 - p)); dS = P (dP - sum_k dP_k P_k) scale
    float* DP = U;                                                 // [NH][RP][SPR]
    float* DQKV = lds + L.h;                                       // [RP][SQ] over H .. X1
    {
        const int hd = wave >> 1, rt = wave & 1, i = lane & 15, kg = lane >> 4;
        f32x4_t s[2];
        s[0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; s[1] = s[0];
#pragma unroll
        for (int ks = 0; ks < DH / 16; ++ks) {
            const uint4 a = *reinterpret_cast<const uint4*>(CTX + (rt * 16 + i) * SX + hd * DH + ks * 16 + kg * 4);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const uint4 b = *reinterpret_cast<const uint4*>(QKV + (ct * 16 + i) * SQ + 2 * E + hd * DH + ks * 16 + kg * 4);
                Mma<float>::mma(a, b, s[ct]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = rt * 16 + kg * 4 + r;
            float dpd[2], pp[2], dot = 0.f;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int key = ct * 16 + i;
                pp[ct] = P[(hd * RP + q) * SPR + key];
                float mk = 1.f;
                if (TRAIN && w.thr_attn) mk = keep_elem(w.seed, w.site, (((uint64_t)(user * NH + hd) * 32 + q) << 5) + key, w.thr_attn) ? w.ks_attn : 0.f;
                dpd[ct] = s[ct][r] * mk;
                dot += dpd[ct] * pp[ct];                           // (dpd is already the gradient w.r.t. the PRE-dropout probability)
            }
            dot = group16_sum(dot);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) DP[(hd * RP + q) * SPR + ct * 16 + i] = (q < T) ? pp[ct] * (dpd[ct] - dot) * w.scale : 0.f;
        }
    }
    __syncthreads();
    if (TRAIN && w.thr_attn) {                                     // P -> Pd in place (P itself is no longer needed)
        for (int id = tid; id < NH * RP * RP; id += 256) {
            const int hd = id / (RP * RP), q = (id / RP) % RP, key = id % RP;
            float* pe = P + (hd * RP + q) * SPR + key;
            *pe = keep_elem(w.seed, w.site, (((uint64_t)(user * NH + hd) * 32 + q) << 5) + key, w.thr_attn) ? *pe * w.ks_attn : 0.f;
        }
        __syncthreads();
    }
    // dQKV: 3 x (2 heads x 2 row tiles x 2 column tiles) = 24 output tiles, 6 per wave
    for (int t = wave; t < 24; t += 4) {
        const int which = t / 8, hd = (t / 4) & 1, mt = (t / 2) & 1, nt = t & 1;
        const int i = lane & 15, kg = lane >> 4;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        if (which == 0) {                // dQ[q, d] = sum_key dS[q, key] K[key, d]: rows of dS, K down its rows
#pragma unroll
            for (int ks = 0; ks < RP / 16; ++ks) {
                const uint4 a = *reinterpret_cast<const uint4*>(DP + (hd * RP + mt * 16 + i) * SPR + ks * 16 + kg * 4);
                uint4 b;
                uint32_t* bo = &b.x;
#pragma unroll
                for (int j = 0; j < 4; ++j) bo[j] = __float_as_uint(QKV[(ks * 16 + kg * 4 + j) * SQ + E + hd * DH + nt * 16 + i]);
                Mma<float>::mma(a, b, acc);
            }
        } else if (which == 1) {         // dK[key, d] = sum_q dS[q, key] Q[q, d]
            mm_tn(DP + hd * RP * SPR, SPR, mt * 16, QKV + hd * DH, SQ, nt * 16, lane, acc);
        } else {                         // dV[key, d] = sum_q Pd[q, key] dCTX[q, d]
            mm_tn(P + hd * RP * SPR, SPR, mt * 16, CTX + hd * DH, SX, nt * 16, lane, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = mt * 16 + kg * 4 + r, col = which * E + hd * DH + nt * 16 + i;
            DQKV[row * SQ + col] = row < T ? acc[r] : 0.f;
        }
    }
    __syncthreads();
    // dX = dQKV Wqkv + dV1 (the residual branch of LN1): B[n = E column][k = 3E] = Wqkv[k][n] (global [3E, E])
    {
        f32x4_t acc[2]; zero2(acc);
        mm_nn<3 * E>(DQKV, SQ, w.wqkv, E, wave * 16, lane, acc);
        tile_each(acc, wave * 16, lane, [&](int r, int c, float v) {
            if (r < T) dx[((size_t)user * T + r) * E + c] = v + H2[r * SX + c];
        });
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
namespace {
constexpr size_t LDS_LIMIT = 160 * 1024;

int fill(const a4r_sasrec_block_t* b, int T, int train, BlockW& w, int& dpe) {
    if (!b || T <= 0 || T > RP) return A4R_EINVAL;
    if (b->E != E || b->n_heads != NH || b->F != F || b->d <= 0 || b->d > 64 || b->ldwu < b->d || b->ldwu % 4) return A4R_EINVAL;
    if (b->mode != 0 && b->mode != 1) return A4R_EINVAL;
    if (b->mode == 1 && (!b->ln3_g || !b->ln3_b || ((reinterpret_cast<uintptr_t>(b->ln3_g) | reinterpret_cast<uintptr_t>(b->ln3_b)) & 3u))) return A4R_EINVAL;
    const void* need[] = {b->wqkv, b->wfc, b->w1, b->b1, b->w2, b->b2, b->ln1_g, b->ln1_b, b->ln2_g, b->ln2_b,
                          b->wd1, b->bd1, b->wu1, b->bu1, b->wd2, b->bd2, b->wu2, b->bu2};
    for (const void* q : need)
        if (!q || (reinterpret_cast<uintptr_t>(q) & 15u)) return A4R_EINVAL;           // 16-byte operand chunks / float4 rows
    if (b->drop_attn < 0.f || b->drop_attn >= 1.f || b->drop_hidden < 0.f || b->drop_hidden >= 1.f) return A4R_EINVAL;
    dpe = (b->d + 15) & ~15;
    if (dpe > b->ldwu) return A4R_EINVAL;                                               // (the zero padding of Wd / bd / Wu must cover the 16-column tiles)
    if ((size_t)Lay(dpe).total * sizeof(float) > LDS_LIMIT) return A4R_EINVAL;          // d > 32: the multi-launch path
    w.wqkv = b->wqkv; w.wfc = b->wfc; w.w1 = b->w1; w.b1 = b->b1; w.w2 = b->w2; w.b2 = b->b2;
    w.ln1g = b->ln1_g; w.ln1b = b->ln1_b; w.ln2g = b->ln2_g; w.ln2b = b->ln2_b; w.ln3g = b->ln3_g; w.ln3b = b->ln3_b;
    w.mode = b->mode;
    w.wd1 = b->wd1; w.bd1 = b->bd1; w.wu1 = b->wu1; w.bu1 = b->bu1; w.wd2 = b->wd2; w.bd2 = b->bd2; w.wu2 = b->wu2; w.bu2 = b->bu2;
    w.ldwu = b->ldwu; w.d = b->d; w.act = b->act; w.inner_res = b->inner_res;
    w.eps = b->eps; w.scale = 1.f / sqrtf((float)DH); w.mask_neg = b->mask_neg;
    w.p_attn = train ? b->drop_attn : 0.f; w.p_hidden = train ? b->drop_hidden : 0.f;
    w.thr_attn = a4r_thr16(w.p_attn); w.thr_hidden = a4r_thr16(w.p_hidden);
    w.ks_attn = a4r_keep_scale(w.p_attn); w.ks_hidden = a4r_keep_scale(w.p_hidden);
    w.site = b->drop_site; w.seed = b->drop_seed;
    return A4R_OK;
}

template <typename K> int set_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess ? A4R_OK : A4R_ELAUNCH;
}
}  // namespace

extern "C" int a4r_sasrec_block_fwd(void* stream, const a4r_sasrec_block_t* b, const float* x, const float* log_mask, float* y,
                                    int n_users, int T, int train) {
    BlockW w;
    int dpe = 0;
    if (!x || !log_mask || !y || n_users <= 0) return A4R_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) return A4R_EINVAL;
    if (int rc = fill(b, T, train, w, dpe)) return rc;
    const size_t lds = (size_t)Lay(dpe, true).total * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool tr = w.thr_attn || w.thr_hidden;
    if (tr) {
        if (int rc = set_lds(sasrec_block_fwd_kernel<true>, lds)) return rc;
        hipLaunchKernelGGL(sasrec_block_fwd_kernel<true>, dim3(n_users), dim3(256), lds, s, x, log_mask, y, w, T, dpe);
    } else {
        if (int rc = set_lds(sasrec_block_fwd_kernel<false>, lds)) return rc;
        hipLaunchKernelGGL(sasrec_block_fwd_kernel<false>, dim3(n_users), dim3(256), lds, s, x, log_mask, y, w, T, dpe);
    }
    return a4r_launch_status();
}

extern "C" int a4r_sasrec_block_bwd(void* stream, const a4r_sasrec_block_t* b, const float* x, const float* log_mask, const float* dy, float* dx,
                                    int n_users, int T, int train) {
    BlockW w;
    BlockG g;
    int dpe = 0;
    if (!x || !log_mask || !dy || !dx || n_users <= 0) return A4R_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15u) return A4R_EINVAL;
    if (int rc = fill(b, T, train, w, dpe)) return rc;
    g.wd1 = b->g_wd1; g.bd1 = b->g_bd1; g.wu1 = b->g_wu1; g.bu1 = b->g_bu1; g.wd2 = b->g_wd2; g.bd2 = b->g_bd2; g.wu2 = b->g_wu2; g.bu2 = b->g_bu2;
    g.ln3g = b->g_ln3_g; g.ln3b = b->g_ln3_b;
    g.ldgd = b->ldg_d; g.ldgu = b->ldg_u;
    if ((g.wd1 || g.wd2) && g.ldgd < E) return A4R_EINVAL;
    if ((g.wu1 || g.wu2) && g.ldgu < b->d) return A4R_EINVAL;
    const size_t lds = (size_t)Lay(dpe).total * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool tr = w.thr_attn || w.thr_hidden;
    if (tr) {
        if (int rc = set_lds(sasrec_block_bwd_kernel<true>, lds)) return rc;
        hipLaunchKernelGGL(sasrec_block_bwd_kernel<true>, dim3(n_users), dim3(256), lds, s, x, log_mask, dy, dx, w, g, T, dpe);
    } else {
        if (int rc = set_lds(sasrec_block_bwd_kernel<false>, lds)) return rc;
        hipLaunchKernelGGL(sasrec_block_bwd_kernel<false>, dim3(n_users), dim3(256), lds, s, x, log_mask, dy, dx, w, g, T, dpe);
    }
    return a4r_launch_status();
}
