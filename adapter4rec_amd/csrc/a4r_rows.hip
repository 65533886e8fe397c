// Row kernels (HBM-bound): embeddings + LayerNorm, LayerNorm forward/backward, CLS gather/scatter,
// small elementwise helpers.  One 64-lane wave owns one row of width H <= 1024 (H % 8 == 0): lane l
// holds the 8-element groups l and l + 64, so every global access is 16 B (bf16) / 2 x 16 B (fp32)
// per lane and the row statistics are two wave reductions.  All arithmetic is fp32.
#include <cstdlib>
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

constexpr int MAXG = 2;   // 8-element groups per lane: H <= 64 * 2 * 8 = 1024

template <typename T>
A4R_DEV void row_load(const T* p, int ng, int lane, float (&v)[MAXG][8]) {
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int gi = lane + 64 * g;
        if (gi < ng) load_vec<T, 8>(p + gi * 8, v[g]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[g][e] = 0.f;
        }
    }
}
template <typename T>
A4R_DEV void row_store(T* p, int ng, int lane, const float (&v)[MAXG][8]) {
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int gi = lane + 64 * g;
        if (gi < ng) store_vec<T, 8>(p + gi * 8, v[g]);
    }
}

// mean / rstd of one row held across the wave (two-pass, values already in registers)
A4R_DEV void row_stats(const float (&v)[MAXG][8], int ng, int lane, int H, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[g][e];
    mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        if (lane + 64 * g < ng) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[g][e] - mean; q += d * d; }
        }
    }
    rstd = rsqrtf(wave_sum(q) / (float)H + eps);
}

A4R_DEV void row_dropout(float (&v)[MAXG][8], int ng, int lane, size_t row, int H, uint64_t seed, uint32_t site, uint32_t thr16, float scale) {
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int gi = lane + 64 * g;
        if (gi < ng) {
            const uint64_t e0 = (uint64_t)row * (uint64_t)H + (uint64_t)gi * 8;
            const uint64_t h0 = a4r_hash64(seed, site, e0 >> 2), h1 = a4r_hash64(seed, site, (e0 >> 2) + 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[g][e] = (((uint32_t)(h0 >> (16 * e)) & 0xffffu) >= thr16) ? v[g][e] * scale : 0.f;
                v[g][e + 4] = (((uint32_t)(h1 >> (16 * e)) & 0xffffu) >= thr16) ? v[g][e + 4] * scale : 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------ embeddings + LN
template <typename T>
__global__ void __launch_bounds__(256) embed_ln_kernel(const int64_t* __restrict__ ids, int ld_ids, const float* __restrict__ word,
                                                       const float* __restrict__ pos, const float* __restrict__ type0,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                       T* __restrict__ out, int ldo, int n_rows, int S, int H, int roberta, int pad_id,
                                                       uint64_t seed, uint32_t site, uint32_t thr16, float scale,
                                                       T* __restrict__ pre_out, float* __restrict__ stats_out, float* __restrict__ kmask_out) {
    const int lane = threadIdx.x & 63;
    const int ng = H / 8;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += gridDim.x * 4) {
        const int item = row / S, s = row % S;
        const int64_t* idr = ids + (size_t)item * ld_ids;
        if (kmask_out && lane == 0) kmask_out[row] = (float)idr[S + s];       // the attention key mask: columns S .. 2S-1 of the ids || mask row
        // a NEGATIVE id -(r + 1) reads word row r but counts as a pad token for the position ids: a soft prompt replaces the word
        // vector of a title's first tokens, RoBERTa's positions still follow the ORIGINAL ids (pads inside a short title's prompt)
        const int64_t idraw = idr[s];
        const int64_t id = idraw < 0 ? -idraw - 1 : idraw;
        int pid = s;
        if (roberta) {        // cumsum(id != pad) * (id != pad) + pad
            int c = 0;
            for (int t = 0; t <= s; ++t) c += (idr[t] != pad_id && idr[t] >= 0);
            pid = (idraw != pad_id && idraw >= 0) ? c + pad_id : pad_id;
        }
        float v[MAXG][8], a[MAXG][8], b[MAXG][8];
        row_load<float>(word + (size_t)id * H, ng, lane, v);
        row_load<float>(pos + (size_t)pid * H, ng, lane, a);
        row_load<float>(type0, ng, lane, b);
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[g][e] += a[g][e] + b[g][e];
        if (pre_out) row_store<T>(pre_out + (size_t)row * ldo, ng, lane, v);    // training the embedding side: what ln_bwd needs
        float mean, rstd;
        row_stats(v, ng, lane, H, eps, mean, rstd);
        if (stats_out && lane == 0) { stats_out[2 * (size_t)row] = mean; stats_out[2 * (size_t)row + 1] = rstd; }
        row_load<float>(gamma, ng, lane, a);
        row_load<float>(beta, ng, lane, b);
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[g][e] = (v[g][e] - mean) * rstd * a[g][e] + b[g][e];
        if (thr16) row_dropout(v, ng, lane, row, H, seed, site, thr16, scale);
        row_store<T>(out + (size_t)row * ldo, ng, lane, v);
    }
}

// gradient of the embedding sum: every token row of dpre is added into the rows of the word / position tables it read
// (--fine_tune_to all, Pretraining; the reference gets this from nn.Embedding's backward).  fp32 atomics.
template <typename T>
__global__ void __launch_bounds__(256) embed_bwd_kernel(const int64_t* __restrict__ ids, int ld_ids, const T* __restrict__ dpre, int ldd,
                                                        float* __restrict__ dword, float* __restrict__ dpos, int n_rows, int S, int H,
                                                        int roberta, int pad_id) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += gridDim.x * 4) {
        const int item = row / S, s = row % S;
        const int64_t* idr = ids + (size_t)item * ld_ids;
        const int64_t idraw = idr[s];
        const int64_t id = idraw < 0 ? -idraw - 1 : idraw;     // (negative ids: see embed_ln_kernel)
        int pid = s;
        if (roberta) {
            int c = 0;
            for (int t = 0; t <= s; ++t) c += (idr[t] != pad_id && idr[t] >= 0);
            pid = (idraw != pad_id && idraw >= 0) ? c + pad_id : pad_id;
        }
        for (int c = lane; c < H; c += 64) {
            const float g = Elem<T>::ld(dpre + (size_t)row * ldd + c);
            if (dword) atomicAdd(dword + (size_t)id * H + c, g);
            if (dpos) atomicAdd(dpos + (size_t)pid * H + c, g);
        }
    }
}

// ------------------------------------------------------------------ fp8 (OCP e4m3fn) row quantisation
// x8[row] = e4m3(x[row] * 448 / amax(row)), scale[row] = amax(row) / 448: one scale per row (token), the A operand form of the
// fp8 a4r_gemm_nt.  v_cvt_pk_fp8_f32 rounds to nearest even; |x * 448 / amax| <= 448 = the e4m3 maximum, so nothing overflows.
A4R_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int NG>
A4R_DEV void row_quant_store(const float (&v)[NG][8], int ng, int lane, unsigned char* q, float* scale_out) {
    float am = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g)
        if (lane + 64 * g < ng) {
#pragma unroll
            for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(v[g][e]));
        }
    am = wave_max(am);
    const float inv = am > 0.f ? __fdiv_rn(448.f, am) : 0.f;         // correctly rounded (once per row): bit-equal with a host restatement
    if (lane == 0) *scale_out = am > 0.f ? __fdiv_rn(am, 448.f) : 1.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int gi = lane + 64 * g;
        if (gi < ng) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = v[g][e] * inv;
            *reinterpret_cast<uint2*>(q + gi * 8) = f32x8_to_fp8(t);
        }
    }
}

constexpr int QMAXG = 8;   // 8-element groups per lane of the standalone quantiser: H <= 64 * 8 * 8 = 4096
template <typename T>
__global__ void __launch_bounds__(256) quant_rows_fp8_kernel(const T* __restrict__ x, int ldx, unsigned char* __restrict__ q, int ldq,
                                                             float* __restrict__ scale, int M, int H) {
    const int lane = threadIdx.x & 63;
    const int ng = H / 8;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
        float v[QMAXG][8];
#pragma unroll
        for (int g = 0; g < QMAXG; ++g) {
            const int gi = lane + 64 * g;
            if (gi < ng) load_vec<T, 8>(x + (size_t)row * ldx + gi * 8, v[g]);
        }
        row_quant_store<QMAXG>(v, ng, lane, q + (size_t)row * ldq, scale + row);
    }
}

// ------------------------------------------------------------------ LN forward
// y (optional when y8 is given) in T; y8 / ys (optional): the same row as e4m3 with its per-row scale (the next GEMM's fp8 A operand,
// quantised from the fp32 result: one rounding instead of bf16 then e4m3).
template <typename T>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const T* __restrict__ vin, int ldv, const float* __restrict__ add, int add_rows,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                     T* __restrict__ y, int ldy, float* __restrict__ stats, int M, int H,
                                                     uint64_t seed, uint32_t site, uint32_t thr16, float scale,
                                                     unsigned char* __restrict__ y8, int ld8, float* __restrict__ ys) {
    const int lane = threadIdx.x & 63;
    const int ng = H / 8;
    float ga[MAXG][8], be[MAXG][8];
    row_load<float>(gamma, ng, lane, ga);
    row_load<float>(beta, ng, lane, be);
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
        float v[MAXG][8];
        row_load<T>(vin + (size_t)row * ldv, ng, lane, v);
        if (add) {
            float a[MAXG][8];
            row_load<float>(add + (size_t)(row % add_rows) * H, ng, lane, a);
#pragma unroll
            for (int g = 0; g < MAXG; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[g][e] += a[g][e];
        }
        float mean, rstd;
        row_stats(v, ng, lane, H, eps, mean, rstd);
        if (stats && lane == 0) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[g][e] = (v[g][e] - mean) * rstd * ga[g][e] + be[g][e];
        if (thr16) row_dropout(v, ng, lane, row, H, seed, site, thr16, scale);
        if (y) row_store<T>(y + (size_t)row * ldy, ng, lane, v);
        if (y8) row_quant_store<MAXG>(v, ng, lane, y8 + (size_t)row * ld8, ys + row);
    }
}

// ------------------------------------------------------------------ lean LN forward / backward (round 4)
// The forms the image tower's un-adapted LayerNorms run (no additive table, no dropout, no e4m3 output / no parameter gradients, no second output):
// gamma (and beta) live in LDS instead of 32 registers, so 6 - 8 waves per SIMD fit instead of 4 - 5, and every wave requests its NEXT row before it works
// on the current one.  With one 1.5 KB row in flight per wave and 5 120 waves on the chip the general kernels moved 4.0 - 4.4 TB/s: the bytes in flight
// (7.7 MB against the ~16 MB that 8 TB/s x 2 us of loaded latency ask for), not the HBM, were the bound.
template <typename T> struct RawRow {
    static constexpr int NQ = sizeof(T) == 2 ? 1 : 2;         // 16-byte chunks per 8-element group
    uint4 q[MAXG][NQ];
    A4R_DEV void request(const T* p, int ng, int lane) {
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int gi = lane + 64 * g;
#pragma unroll
            for (int i = 0; i < NQ; ++i) q[g][i] = gi < ng ? *reinterpret_cast<const uint4*>(p + gi * 8 + i * Elem<T>::PER16) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    A4R_DEV void unpack(int g, float (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < NQ; ++i) Elem<T>::unpack(q[g][i], v + i * Elem<T>::PER16);
    }
    A4R_DEV void forget() {                                  // the compiler loses track of the contents: a second unpack() is done again, not kept as fp32 registers
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
#pragma unroll
            for (int i = 0; i < NQ; ++i) asm volatile("" : "+v"(q[g][i].x), "+v"(q[g][i].y), "+v"(q[g][i].z), "+v"(q[g][i].w));
    }
};
A4R_DEV void lds_vec8(const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

template <typename T>
__global__ void __launch_bounds__(256) ln_fwd_lean_kernel(const T* __restrict__ vin, int ldv, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, T* __restrict__ y, int ldy, float* __restrict__ stats, int M, int H) {
    __shared__ __attribute__((aligned(16))) float gs[1024], bs[1024];
    for (int c = threadIdx.x; c < 1024; c += 256) { gs[c] = c < H ? gamma[c] : 0.f; bs[c] = c < H ? beta[c] : 0.f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, ng = H / 8;
    const int row0 = blockIdx.x * 4 + (threadIdx.x >> 6), step = gridDim.x * 4;
    RawRow<T> cur;
    if (row0 < M) cur.request(vin + (size_t)row0 * ldv, ng, lane);
    for (int row = row0; row < M; row += step) {
        float v[MAXG][8];
#pragma unroll
        for (int g = 0; g < MAXG; ++g) cur.unpack(g, v[g]);
        if (row + step < M) cur.request(vin + (size_t)(row + step) * ldv, ng, lane);      // the wave's next row, in flight under this row's arithmetic
        float mean, rstd;
        row_stats(v, ng, lane, H, eps, mean, rstd);
        if (stats && lane == 0) *reinterpret_cast<float2*>(stats + 2 * (size_t)row) = make_float2(mean, rstd);
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int gi = lane + 64 * g;
            if (gi < ng) {
                float ga[8], be[8];
                lds_vec8(gs + gi * 8, ga);
                lds_vec8(bs + gi * 8, be);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[g][e] = (v[g][e] - mean) * rstd * ga[e] + be[e];
                store_vec<T, 8>(y + (size_t)row * ldy + gi * 8, v[g]);
            }
        }
    }
}

// dv = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)) + dres, dxhat = dy * gamma
template <typename T>
__global__ void __launch_bounds__(256) ln_bwd_lean_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ vin, int ldv, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma, const T* __restrict__ dres, int lddres, T* __restrict__ dv, int lddv,
                                                             int M, int H) {
    __shared__ __attribute__((aligned(16))) float gs[1024];
    for (int c = threadIdx.x; c < 1024; c += 256) gs[c] = c < H ? gamma[c] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, ng = H / 8;
    const int row0 = blockIdx.x * 4 + (threadIdx.x >> 6), step = gridDim.x * 4;
    const float invH = 1.f / (float)H;
    RawRow<T> cv, cd;
    float2 cst = make_float2(0.f, 0.f);
    if (row0 < M) {
        cv.request(vin + (size_t)row0 * ldv, ng, lane);
        cd.request(dy + (size_t)row0 * lddy, ng, lane);
        cst = *reinterpret_cast<const float2*>(stats + 2 * (size_t)row0);
    }
    for (int row = row0; row < M; row += step) {
        const float mean = cst.x, rstd = cst.y;
        RawRow<T> nv = cv, nd = cd, rr;
        if (dres) rr.request(dres + (size_t)row * lddres, ng, lane);      // BEFORE the next row: the counter wait for it then leaves those loads in flight
        if (row + step < M) {                                 // the wave's next row (both operands and its statistics) in flight under this row's arithmetic
            nv.request(vin + (size_t)(row + step) * ldv, ng, lane);
            nd.request(dy + (size_t)(row + step) * lddy, ng, lane);
            cst = *reinterpret_cast<const float2*>(stats + 2 * (size_t)(row + step));
        }
        // two passes over the row as LOADED (16-byte chunks): xhat and dxhat are rebuilt in the second one instead of being held as 32 fp32 registers
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            float ga[8], xh[8], dx[8];
            cv.unpack(g, xh);
            cd.unpack(g, dx);
            lds_vec8(gs + (lane + 64 * g < ng ? lane + 64 * g : 0) * 8, ga);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = dx[e] * ga[e];                // (groups past the row: dy was requested as zero)
                c1 += t;
                c2 += t * ((xh[e] - mean) * rstd);
            }
        }
        c1 = wave_sum(c1) * invH;
        c2 = wave_sum(c2) * invH;
        cv.forget();
        cd.forget();
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int gi = lane + 64 * g;
            if (gi < ng) {
                float ga[8], xh[8], dx[8], r8[8];
                cv.unpack(g, xh);
                cd.unpack(g, dx);
                lds_vec8(gs + gi * 8, ga);
                if (dres) rr.unpack(g, r8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    dx[e] = rstd * (dx[e] * ga[e] - c1 - (xh[e] - mean) * rstd * c2);
                    if (dres) dx[e] += r8[e];
                }
                store_vec<T, 8>(dv + (size_t)row * lddv + gi * 8, dx);
            }
        }
        cv = nv; cd = nd;
    }
}

// The same backward WITH the LayerNorm's parameter gradients (dgamma, dbeta) and the optional second output through another dropout mask --
// trainable LayerNorms of un-adapted sub-layers: full fine-tuning (Pretraining/, --fine_tune_to all) and --finetune_layernorm.  ln_bwd_kernel below
// ran these at 2.3 TB/s (82 us at 40 448 x 768: one row in flight per wave, up to 1024 workgroups each flushing 2 H column sums with atomics onto the
// same addresses).  Here: the lean kernel's loads (gamma in LDS, the wave's next row requested under this row's arithmetic), EIGHT waves per
// workgroup and one workgroup per CU, so that 256 workgroups flush instead of 1024 (0.4 M atomics instead of 1.6 M).
template <typename T>
__global__ void __launch_bounds__(512) ln_bwd_pg_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ vin, int ldv, const float* __restrict__ stats,
                                                        const float* __restrict__ gamma, const T* __restrict__ dres, int lddres, T* __restrict__ dv, int lddv,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int H,
                                                        T* __restrict__ dv2, int lddv2, uint64_t seed2, uint32_t site2, uint32_t thr2, float scale2) {
    __shared__ __attribute__((aligned(16))) float gs[1024];
    __shared__ float red[2][8][1024];
    for (int c = threadIdx.x; c < 1024; c += 512) gs[c] = c < H ? gamma[c] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ng = H / 8;
    const int row0 = blockIdx.x * 8 + wave, step = gridDim.x * 8;
    const float invH = 1.f / (float)H;
    float sg[MAXG][8], sb[MAXG][8];
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) { sg[g][e] = 0.f; sb[g][e] = 0.f; }
    RawRow<T> cv, cd;
    float2 cst = make_float2(0.f, 0.f);
    if (row0 < M) {
        cv.request(vin + (size_t)row0 * ldv, ng, lane);
        cd.request(dy + (size_t)row0 * lddy, ng, lane);
        cst = *reinterpret_cast<const float2*>(stats + 2 * (size_t)row0);
    }
    for (int row = row0; row < M; row += step) {
        const float mean = cst.x, rstd = cst.y;
        RawRow<T> nv = cv, nd = cd, rr;
        if (dres) rr.request(dres + (size_t)row * lddres, ng, lane);
        if (row + step < M) {
            nv.request(vin + (size_t)(row + step) * ldv, ng, lane);
            nd.request(dy + (size_t)(row + step) * lddy, ng, lane);
            cst = *reinterpret_cast<const float2*>(stats + 2 * (size_t)(row + step));
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            float ga[8], xh[8], dx[8];
            cv.unpack(g, xh);
            cd.unpack(g, dx);
            lds_vec8(gs + (lane + 64 * g < ng ? lane + 64 * g : 0) * 8, ga);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float x = (xh[e] - mean) * rstd;        // (groups past the row: dy was requested as zero -> nothing accumulates)
                sg[g][e] += dx[e] * x;
                sb[g][e] += dx[e];
                const float t = dx[e] * ga[e];
                c1 += t;
                c2 += t * x;
            }
        }
        c1 = wave_sum(c1) * invH;
        c2 = wave_sum(c2) * invH;
        cv.forget();
        cd.forget();
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int gi = lane + 64 * g;
            if (gi < ng) {
                float ga[8], xh[8], dx[8], r8[8];
                cv.unpack(g, xh);
                cd.unpack(g, dx);
                lds_vec8(gs + gi * 8, ga);
                if (dres) rr.unpack(g, r8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    dx[e] = rstd * (dx[e] * ga[e] - c1 - (xh[e] - mean) * rstd * c2);
                    if (dres) dx[e] += r8[e];
                }
                store_vec<T, 8>(dv + (size_t)row * lddv + gi * 8, dx);
                if (dv2) {          // second output: dv through ANOTHER dropout mask (row_dropout's element -> lot map)
                    if (thr2) {
                        const uint64_t e0 = (uint64_t)row * (uint64_t)H + (uint64_t)gi * 8;
                        const uint64_t h0 = a4r_hash64(seed2, site2, e0 >> 2), h1 = a4r_hash64(seed2, site2, (e0 >> 2) + 1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            dx[e] = (((uint32_t)(h0 >> (16 * e)) & 0xffffu) >= thr2) ? dx[e] * scale2 : 0.f;
                            dx[e + 4] = (((uint32_t)(h1 >> (16 * e)) & 0xffffu) >= thr2) ? dx[e + 4] * scale2 : 0.f;
                        }
                    }
                    store_vec<T, 8>(dv2 + (size_t)row * lddv2 + gi * 8, dx);
                }
            }
        }
        cv = nv; cd = nd;
    }
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int gi = lane + 64 * g;
        if (gi < ng) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[0][wave][gi * 8 + e] = sg[g][e]; red[1][wave][gi * 8 + e] = sb[g][e]; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 512) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { a += red[0][w][c]; b += red[1][w][c]; }
        if (dgamma) atomicAdd(dgamma + c, a);
        if (dbeta) atomicAdd(dbeta + c, b);
    }
}

// ------------------------------------------------------------------ residual add + LN (round 4: --residual_dtype fp32 on un-adapted sub-layers)
// s = h + res (fp32; res either the fp32 twin of the residual stream or its T tensor); y = LayerNorm(s) from the UNROUNDED sum.  Optional
// outputs: sum (T: what a4r_ln_bwd re-reads as v), sum32 (fp32: the residual operand of a fused adapter launch that follows, Pfeiffer), y32
// (y before its rounding: the next sub-layer's fp32 residual).  The reference under autocast runs exactly this arithmetic: the dense output is
// bf16, `hidden_states + input_tensor` promotes to the fp32 LayerNorm output of the layer below, LayerNorm runs in fp32 (HF BertSelfOutput /
// BertOutput reached from Downstream/Text/model/encoders.py:53).
template <typename T>
__global__ void __launch_bounds__(256) ln_fwd_sum_kernel(const T* __restrict__ h, int ldh, const float* __restrict__ res32, int ldr32, const T* __restrict__ res,
                                                         int ldr, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         T* __restrict__ y, int ldy, T* __restrict__ sum, int lds_, float* __restrict__ sum32, int lds32,
                                                         float* __restrict__ y32, int ldy32, float* __restrict__ stats, int M, int H) {
    const int lane = threadIdx.x & 63;
    const int ng = H / 8;
    float ga[MAXG][8], be[MAXG][8];
    row_load<float>(gamma, ng, lane, ga);
    row_load<float>(beta, ng, lane, be);
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += gridDim.x * 4) {
        float v[MAXG][8], a[MAXG][8];
        row_load<T>(h + (size_t)row * ldh, ng, lane, v);
        if (res32) row_load<float>(res32 + (size_t)row * ldr32, ng, lane, a);
        else row_load<T>(res + (size_t)row * ldr, ng, lane, a);
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[g][e] += a[g][e];
        if (sum) row_store<T>(sum + (size_t)row * lds_, ng, lane, v);
        if (sum32) row_store<float>(sum32 + (size_t)row * lds32, ng, lane, v);
        float mean, rstd;
        row_stats(v, ng, lane, H, eps, mean, rstd);
        if (lane == 0) { stats[2 * (size_t)row] = mean; stats[2 * (size_t)row + 1] = rstd; }
#pragma unroll
        for (int g = 0; g < MAXG; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[g][e] = (v[g][e] - mean) * rstd * ga[g][e] + be[g][e];
        row_store<T>(y + (size_t)row * ldy, ng, lane, v);
        if (y32) row_store<float>(y32 + (size_t)row * ldy32, ng, lane, v);
    }
}

// ------------------------------------------------------------------ LN backward
// dv = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)), dxhat = dy * gamma (dy first goes back through
// the forward's dropout mask).  Column sums (dgamma, dbeta, dbias = sum dv) are kept per lane over the rows a
// wave visits, reduced over the block's 4 waves in LDS, and flushed with one atomic per column per block.
template <typename T, bool WGB, bool WDB>     // WGB: accumulate dgamma/dbeta; WDB: accumulate dbias (unused sums cost 16-32 VGPRs each)
__global__ void __launch_bounds__(256) ln_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ vin, int ldv,
                                                     const float* __restrict__ add, int add_rows, const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, const T* __restrict__ dres, int lddres,
                                                     T* __restrict__ dv, int lddv,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias,
                                                     int M, int H, uint64_t seed, uint32_t site, uint32_t thr16, float scale,
                                                     T* __restrict__ dv2, int lddv2, uint64_t seed2, uint32_t site2, uint32_t thr2, float scale2) {
    __shared__ float red[(WGB ? 2 : 0) + (WDB ? 1 : 0) + 1][4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ng = H / 8;
    float ga[MAXG][8];
    row_load<float>(gamma, ng, lane, ga);
    float sg[MAXG][8], sb[MAXG][8], sv[MAXG][8];
#pragma unroll
    for (int g = 0; g < MAXG; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) { sg[g][e] = 0.f; sb[g][e] = 0.f; sv[g][e] = 0.f; }
    for (int row = blockIdx.x * 4 + wave; row < M; row += gridDim.x * 4) {
        const float mean = stats[2 * (size_t)row], rstd = stats[2 * (size_t)row + 1];      // requested with the row, not after the dropout mask
        float v[MAXG][8], d[MAXG][8];
        row_load<T>(vin + (size_t)row * ldv, ng, lane, v);
        row_load<T>(dy + (size_t)row * lddy, ng, lane, d);
        if (add) {
            float a[MAXG][8];
            row_load<float>(add + (size_t)(row % add_rows) * H, ng, lane, a);
#pragma unroll
            for (int g = 0; g < MAXG; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) v[g][e] += a[g][e];
        }
        if (thr16) row_dropout(d, ng, lane, row, H, seed, site, thr16, scale);
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            if (lane + 64 * g < ng) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (v[g][e] - mean) * rstd;
                    if constexpr (WGB) {
                        sg[g][e] += d[g][e] * xh;
                        sb[g][e] += d[g][e];
                    }
                    const float dx = d[g][e] * ga[g][e];
                    v[g][e] = xh;
                    d[g][e] = dx;
                    c1 += dx;
                    c2 += dx * xh;
                }
            }
        }
        c1 = wave_sum(c1) / (float)H;
        c2 = wave_sum(c2) / (float)H;
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            if (lane + 64 * g < ng) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    d[g][e] = rstd * (d[g][e] - c1 - v[g][e] * c2);
                    if constexpr (WDB) sv[g][e] += d[g][e];
                }
            }
        }
        if (dres) {      // residual branch by-passing this LayerNorm (not part of dbias)
            float rr[MAXG][8];
            row_load<T>(dres + (size_t)row * lddres, ng, lane, rr);
#pragma unroll
            for (int g = 0; g < MAXG; ++g)
#pragma unroll
                for (int e = 0; e < 8; ++e) d[g][e] += rr[g][e];
        }
        row_store<T>(dv + (size_t)row * lddv, ng, lane, d);
        if (dv2) {       // second output: dv through ANOTHER dropout mask (the gradient of a dense output that was dropped out before the residual add)
            if (thr2) row_dropout(d, ng, lane, row, H, seed2, site2, thr2, scale2);
            row_store<T>(dv2 + (size_t)row * lddv2, ng, lane, d);
        }
    }
    if constexpr (!WGB && !WDB) return;
    constexpr int SV = WGB ? 2 : 0;
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        const int gi = lane + 64 * g;
        if (gi < ng) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if constexpr (WGB) {
                    red[0][wave][gi * 8 + e] = sg[g][e];
                    red[1][wave][gi * 8 + e] = sb[g][e];
                }
                if constexpr (WDB) red[SV][wave][gi * 8 + e] = sv[g][e];
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        if constexpr (WGB) {
            if (dgamma) atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
            if (dbeta) atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
        }
        if constexpr (WDB) atomicAdd(dbias + c, red[SV][0][c] + red[SV][1][c] + red[SV][2][c] + red[SV][3][c]);
    }
}

// ------------------------------------------------------------------ row gather / scatter, elementwise
// out[r, :] = in[idx[r], :] (gather) or out[idx[r], :] = in[r, :] (scatter; idx without repeats), 16-byte pieces: the rows of the item slots a batch
// really uses (ragged histories: the pad slots of short users are not encoded, engine.py train_forward)
__global__ void __launch_bounds__(256) rows_idx_copy_kernel(const char* __restrict__ in, size_t ldi_b, char* __restrict__ out, size_t ldo_b,
                                                            const int32_t* __restrict__ idx, int n, int pieces, int scatter) {
    const size_t total = (size_t)n * pieces;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / pieces;
        const size_t c = (i - r * pieces) * 16;
        const size_t other = (size_t)idx[r];
        const size_t ri = scatter ? r : other, ro = scatter ? other : r;
        *reinterpret_cast<uint4*>(out + ro * ldo_b + c) = *reinterpret_cast<const uint4*>(in + ri * ldi_b + c);
    }
}

template <typename T, bool SCATTER>
__global__ void __launch_bounds__(256) rows_copy_kernel(const T* __restrict__ in, int ldi, T* __restrict__ out, int ldo,
                                                        int n, int row_step, int H) {
    constexpr int PER = Elem<T>::PER16;
    const int cpr = H / PER;
    const size_t total = (size_t)n * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / cpr;
        const int c = (int)(i % cpr) * PER;
        const size_t ri = SCATTER ? r : r * row_step, ro = SCATTER ? r * row_step : r;
        *reinterpret_cast<uint4*>(out + ro * ldo + c) = *reinterpret_cast<const uint4*>(in + ri * ldi + c);
    }
}

// out[r] = in[r / step] if r % step == 0 and r / step < n else 0, for r < fill_rows
template <typename T>
__global__ void __launch_bounds__(256) scatter_fill_kernel(const T* __restrict__ in, int ldi, T* __restrict__ out, int ldo,
                                                           int n, int row_step, int H, int fill_rows) {
    constexpr int PER = Elem<T>::PER16;
    const int cpr = H / PER;
    const size_t total = (size_t)fill_rows * cpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / cpr;
        const int c = (int)(i % cpr) * PER;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (r % row_step == 0 && r / row_step < (size_t)n) v = *reinterpret_cast<const uint4*>(in + (r / row_step) * ldi + c);
        *reinterpret_cast<uint4*>(out + r * ldo + c) = v;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) dropout_apply_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int M, int N,
                                                            uint64_t seed, uint32_t site, uint32_t thr16, float scale) {
    const int c8n = N / 8;
    const size_t total = (size_t)M * c8n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / c8n;
        const int col = (int)(i % c8n) * 8;
        float v[8];
        load_vec<T, 8>(x + row * ldx + col, v);
        const uint64_t e0 = (uint64_t)row * (uint64_t)N + (uint64_t)col;
        const uint64_t h0 = a4r_hash64(seed, site, e0 >> 2), h1 = a4r_hash64(seed, site, (e0 >> 2) + 1);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = (((uint32_t)(h0 >> (16 * e)) & 0xffffu) >= thr16) ? v[e] * scale : 0.f;
            v[e + 4] = (((uint32_t)(h1 >> (16 * e)) & 0xffffu) >= thr16) ? v[e + 4] * scale : 0.f;
        }
        store_vec<T, 8>(y + row * ldy + col, v);
    }
}

__global__ void __launch_bounds__(256) act_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ pre,
                                                          float* __restrict__ dx, int64_t n, int act) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        dx[i] = dy[i] * act_bwd(pre[i], act);
}

inline bool bad_dtype(int d) { return d != A4R_F32 && d != A4R_BF16; }
inline bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
inline int row_grid(int rows) { int g = (rows + 3) / 4; return g > 2048 ? 2048 : (g < 1 ? 1 : g); }
// grid of a lean LayerNorm launch: never more workgroups than are resident at once (occupancy x CUs) -- every wave then walks the same number of rows
// (+- 1) with its next row always requested, instead of a second, thin round of workgroups starting when the first has finished
template <auto Kernel> int resident_grid(int rows) {         // (the kernel is a template VALUE: one cached capacity per kernel, not per signature)
    static int caps[16] = {0};                                // per device id (one process per GPU sees one entry; a process that drives several must not mix them)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int& cap = caps[dev];
    if (!cap) {
        int per_cu = 0;
        hipDeviceProp_t pr;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, Kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
        cap = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256) * per_cu;
    }
    const int g = (rows + 3) / 4;
    return g > cap ? cap : (g < 1 ? 1 : g);
}

}  // namespace

extern "C" int a4r_embed_ln(void* stream, const int64_t* ids, int ld_ids, const float* word, const float* pos,
                            const float* type0, const float* gamma, const float* beta, float eps,
                            void* out, int ldo, int n_items, int S, int H, int roberta, int pad_id, int dtype,
                            float drop_p, uint32_t drop_site, uint64_t drop_seed, void* pre_out, float* stats_out, float* key_mask_out) {
    if (!ids || !word || !pos || !type0 || !gamma || !beta || !out) return A4R_EINVAL;
    if (pre_out && misaligned(pre_out)) return A4R_EINVAL;
    if (bad_dtype(dtype) || n_items <= 0 || S <= 0 || H <= 0 || H % 8 || H > 1024 || ld_ids < S || (key_mask_out && ld_ids < 2 * S)) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    if ((ldo * esz) % 16 || ldo < H || misaligned(out) || misaligned(word) || misaligned(pos) || misaligned(type0)) return A4R_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rows = n_items * S;
    const uint32_t thr = a4r_thr16(drop_p);
    const float sc = a4r_keep_scale(drop_p);
    if (dtype == A4R_BF16)
        hipLaunchKernelGGL(embed_ln_kernel<bf16_t>, dim3(row_grid(rows)), dim3(256), 0, s, ids, ld_ids, word, pos, type0, gamma, beta, eps,
                           (bf16_t*)out, ldo, rows, S, H, roberta, pad_id, drop_seed, drop_site, thr, sc, (bf16_t*)pre_out, stats_out, key_mask_out);
    else
        hipLaunchKernelGGL(embed_ln_kernel<float>, dim3(row_grid(rows)), dim3(256), 0, s, ids, ld_ids, word, pos, type0, gamma, beta, eps,
                           (float*)out, ldo, rows, S, H, roberta, pad_id, drop_seed, drop_site, thr, sc, (float*)pre_out, stats_out, key_mask_out);
    return a4r_launch_status();
}

extern "C" int a4r_embed_bwd(void* stream, const int64_t* ids, int ld_ids, const void* dpre, int ldd, float* dword, float* dpos,
                             int n_items, int S, int H, int roberta, int pad_id, int dtype) {
    if (!ids || !dpre || (!dword && !dpos) || bad_dtype(dtype) || n_items <= 0 || S <= 0 || H <= 0 || ld_ids < S || ldd < H) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rows = n_items * S;
    if (dtype == A4R_BF16)
        hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, dim3(row_grid(rows)), dim3(256), 0, s, ids, ld_ids, (const bf16_t*)dpre, ldd, dword, dpos,
                           rows, S, H, roberta, pad_id);
    else
        hipLaunchKernelGGL(embed_bwd_kernel<float>, dim3(row_grid(rows)), dim3(256), 0, s, ids, ld_ids, (const float*)dpre, ldd, dword, dpos,
                           rows, S, H, roberta, pad_id);
    return a4r_launch_status();
}

static int ln_fwd_launch(void* stream, const void* v, int ldv, const float* add, int add_rows, const float* gamma, const float* beta, float eps,
                         void* y, int ldy, float* stats, int M, int H, int dtype, float drop_p, uint32_t drop_site, uint64_t drop_seed,
                         void* y8, int ld8, float* ys) {
    if (!v || !gamma || !beta || (!y && !y8) || bad_dtype(dtype) || M <= 0 || H <= 0 || H % 8 || H > 1024) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    if ((ldv * esz) % 16 || ldv < H || misaligned(v)) return A4R_EINVAL;
    if (y && ((ldy * esz) % 16 || ldy < H || misaligned(y))) return A4R_EINVAL;
    if (y8 && (!ys || ld8 % 8 || ld8 < H || (reinterpret_cast<uintptr_t>(y8) & 7u))) return A4R_EINVAL;
    if (add && (add_rows <= 0 || misaligned(add))) return A4R_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const uint32_t thr = a4r_thr16(drop_p);
    const float sc = a4r_keep_scale(drop_p);
    if (!add && !thr && !y8) {                                // the lean form (the image tower's un-adapted LayerNorms)
        if (dtype == A4R_BF16)
            hipLaunchKernelGGL(ln_fwd_lean_kernel<bf16_t>, dim3(resident_grid<ln_fwd_lean_kernel<bf16_t>>(M)), dim3(256), 0, s, (const bf16_t*)v, ldv, gamma, beta, eps, (bf16_t*)y, ldy, stats, M, H);
        else
            hipLaunchKernelGGL(ln_fwd_lean_kernel<float>, dim3(resident_grid<ln_fwd_lean_kernel<float>>(M)), dim3(256), 0, s, (const float*)v, ldv, gamma, beta, eps, (float*)y, ldy, stats, M, H);
        return a4r_launch_status();
    }
    if (dtype == A4R_BF16)
        hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, dim3(row_grid(M)), dim3(256), 0, s, (const bf16_t*)v, ldv, add, add_rows, gamma, beta, eps,
                           (bf16_t*)y, ldy, stats, M, H, drop_seed, drop_site, thr, sc, (unsigned char*)y8, ld8, ys);
    else
        hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(row_grid(M)), dim3(256), 0, s, (const float*)v, ldv, add, add_rows, gamma, beta, eps,
                           (float*)y, ldy, stats, M, H, drop_seed, drop_site, thr, sc, (unsigned char*)y8, ld8, ys);
    return a4r_launch_status();
}

extern "C" int a4r_ln_fwd(void* stream, const void* v, int ldv, const float* add, int add_rows,
                          const float* gamma, const float* beta, float eps,
                          void* y, int ldy, float* stats, int M, int H, int dtype,
                          float drop_p, uint32_t drop_site, uint64_t drop_seed) {
    if (!y) return A4R_EINVAL;
    return ln_fwd_launch(stream, v, ldv, add, add_rows, gamma, beta, eps, y, ldy, stats, M, H, dtype, drop_p, drop_site, drop_seed, nullptr, 0, nullptr);
}

extern "C" int a4r_ln_fwd_sum(void* stream, const void* h, int ldh, const float* res32, int ldres32, const void* res, int ldres,
                              const float* gamma, const float* beta, float eps, void* y, int ldy, void* sum, int ldsum, float* sum32, int ldsum32,
                              float* y32, int ldy32, float* stats, int M, int H, int dtype) {
    if (!h || (!res32 && !res) || !gamma || !beta || !y || !stats || bad_dtype(dtype) || M <= 0 || H <= 0 || H % 8 || H > 1024) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    if ((ldh * esz) % 16 || ldh < H || misaligned(h) || (ldy * esz) % 16 || ldy < H || misaligned(y)) return A4R_EINVAL;
    if (res32 && (ldres32 % 4 || ldres32 < H || misaligned(res32))) return A4R_EINVAL;
    if (!res32 && ((ldres * esz) % 16 || ldres < H || misaligned(res))) return A4R_EINVAL;
    if (sum && ((ldsum * esz) % 16 || ldsum < H || misaligned(sum))) return A4R_EINVAL;
    if (sum32 && (ldsum32 % 4 || ldsum32 < H || misaligned(sum32))) return A4R_EINVAL;
    if (y32 && (ldy32 % 4 || ldy32 < H || misaligned(y32))) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == A4R_BF16)
        hipLaunchKernelGGL(ln_fwd_sum_kernel<bf16_t>, dim3(row_grid(M)), dim3(256), 0, s, (const bf16_t*)h, ldh, res32, ldres32, (const bf16_t*)res, ldres, gamma,
                           beta, eps, (bf16_t*)y, ldy, (bf16_t*)sum, ldsum, sum32, ldsum32, y32, ldy32, stats, M, H);
    else
        hipLaunchKernelGGL(ln_fwd_sum_kernel<float>, dim3(row_grid(M)), dim3(256), 0, s, (const float*)h, ldh, res32, ldres32, (const float*)res, ldres, gamma,
                           beta, eps, (float*)y, ldy, (float*)sum, ldsum, sum32, ldsum32, y32, ldy32, stats, M, H);
    return a4r_launch_status();
}

extern "C" int a4r_ln_fwd_fp8(void* stream, const void* v, int ldv, const float* add, int add_rows,
                              const float* gamma, const float* beta, float eps,
                              void* y, int ldy, void* y8, int ld8, float* yscale, float* stats, int M, int H, int dtype) {
    if (!y8) return A4R_EINVAL;
    return ln_fwd_launch(stream, v, ldv, add, add_rows, gamma, beta, eps, y, ldy, stats, M, H, dtype, 0.f, 0, 0, y8, ld8, yscale);
}

extern "C" int a4r_quant_rows_fp8(void* stream, const void* x, int ldx, void* q, int ldq, float* scale, int M, int H, int dtype) {
    if (!x || !q || !scale || bad_dtype(dtype) || M <= 0 || H <= 0 || H % 8 || H > 4096) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    if ((ldx * esz) % 16 || ldx < H || misaligned(x) || ldq % 8 || ldq < H || (reinterpret_cast<uintptr_t>(q) & 7u)) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == A4R_BF16)
        hipLaunchKernelGGL(quant_rows_fp8_kernel<bf16_t>, dim3(row_grid(M)), dim3(256), 0, s, (const bf16_t*)x, ldx, (unsigned char*)q, ldq, scale, M, H);
    else
        hipLaunchKernelGGL(quant_rows_fp8_kernel<float>, dim3(row_grid(M)), dim3(256), 0, s, (const float*)x, ldx, (unsigned char*)q, ldq, scale, M, H);
    return a4r_launch_status();
}

extern "C" int a4r_ln_bwd(void* stream, const void* dy, int lddy, const void* v, int ldv, const float* add, int add_rows,
                          const float* stats, const float* gamma, const void* dres, int lddres, void* dv, int lddv,
                          float* dgamma, float* dbeta, float* dbias, int M, int H, int dtype,
                          float drop_p, uint32_t drop_site, uint64_t drop_seed,
                          void* dv2, int lddv2, float drop2_p, uint32_t drop2_site, uint64_t drop2_seed) {
    if (dv2 && (misaligned(dv2) || (lddv2 * (dtype == A4R_F32 ? 4 : 2)) % 16 || lddv2 < H || drop2_p < 0.f || drop2_p >= 1.f)) return A4R_EINVAL;
    if (dres && (misaligned(dres) || (lddres * (dtype == A4R_F32 ? 4 : 2)) % 16 || lddres < H)) return A4R_EINVAL;
    if (!dy || !v || !stats || !gamma || !dv || bad_dtype(dtype) || M <= 0 || H <= 0 || H % 8 || H > 1024) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    if ((lddy * esz) % 16 || (ldv * esz) % 16 || (lddv * esz) % 16 || lddy < H || ldv < H || lddv < H) return A4R_EINVAL;
    if (misaligned(dy) || misaligned(v) || misaligned(dv)) return A4R_EINVAL;
    if (add && (add_rows <= 0 || misaligned(add))) return A4R_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const uint32_t thr = a4r_thr16(drop_p);
    const float sc = a4r_keep_scale(drop_p);
    int grid = row_grid(M);
    const bool wgb = dgamma || dbeta, wdb = dbias != nullptr;
    if (!wgb && !wdb && !add && !thr && !dv2) {               // the lean form (frozen LayerNorm parameters, no dropout: the image tower under LoRA)
        if (dtype == A4R_BF16)
            hipLaunchKernelGGL(ln_bwd_lean_kernel<bf16_t>, dim3(resident_grid<ln_bwd_lean_kernel<bf16_t>>(M)), dim3(256), 0, s, (const bf16_t*)dy, lddy, (const bf16_t*)v, ldv, stats, gamma,
                               (const bf16_t*)dres, lddres, (bf16_t*)dv, lddv, M, H);
        else
            hipLaunchKernelGGL(ln_bwd_lean_kernel<float>, dim3(resident_grid<ln_bwd_lean_kernel<float>>(M)), dim3(256), 0, s, (const float*)dy, lddy, (const float*)v, ldv, stats, gamma,
                               (const float*)dres, lddres, (float*)dv, lddv, M, H);
        return a4r_launch_status();
    }
    static const int pg_on = getenv("A4R_LN_PG") ? atoi(getenv("A4R_LN_PG")) != 0 : 1;          // (A/B runs: 0 = ln_bwd_kernel)
    if (pg_on && wgb && !wdb && !add && !thr && M >= 2048) {     // trainable LayerNorm of an un-adapted sub-layer: parameter gradients on lean loads
        int dev = 0, ncu = 256;
        hipDeviceProp_t pr;
        static int ncu_c[16] = {0};                          // per device id
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
        if (!ncu_c[dev]) ncu_c[dev] = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
        ncu = ncu_c[dev];
        int g8 = (M + 7) / 8; if (g8 > ncu) g8 = ncu;
        const uint32_t thr2 = a4r_thr16(drop2_p);
        const float sc2 = a4r_keep_scale(drop2_p);
        if (dtype == A4R_BF16)
            hipLaunchKernelGGL(ln_bwd_pg_kernel<bf16_t>, dim3(g8), dim3(512), 0, s, (const bf16_t*)dy, lddy, (const bf16_t*)v, ldv, stats, gamma, (const bf16_t*)dres, lddres,
                               (bf16_t*)dv, lddv, dgamma, dbeta, M, H, (bf16_t*)dv2, lddv2, drop2_seed, drop2_site, thr2, sc2);
        else
            hipLaunchKernelGGL(ln_bwd_pg_kernel<float>, dim3(g8), dim3(512), 0, s, (const float*)dy, lddy, (const float*)v, ldv, stats, gamma, (const float*)dres, lddres,
                               (float*)dv, lddv, dgamma, dbeta, M, H, (float*)dv2, lddv2, drop2_seed, drop2_site, thr2, sc2);
        return a4r_launch_status();
    }
    if (grid > 1024) grid = 1024;    // 3-4 blocks per CU (VGPR-limited); bounds the column-sum atomics to 1024 x H per accumulator
#define A4R_LNB(T_, G_, B_)                                                                                                   \
    hipLaunchKernelGGL((ln_bwd_kernel<T_, G_, B_>), dim3(grid), dim3(256), 0, s, (const T_*)dy, lddy, (const T_*)v, ldv, add,    \
                       add_rows, stats, gamma, (const T_*)dres, lddres, (T_*)dv, lddv, dgamma, dbeta, dbias, M, H, drop_seed,  \
                       drop_site, thr, sc, (T_*)dv2, lddv2, drop2_seed, drop2_site, a4r_thr16(drop2_p), a4r_keep_scale(drop2_p))
    if (dtype == A4R_BF16) {
        if (wgb && wdb) A4R_LNB(bf16_t, true, true); else if (wgb) A4R_LNB(bf16_t, true, false);
        else if (wdb) A4R_LNB(bf16_t, false, true); else A4R_LNB(bf16_t, false, false);
    } else {
        if (wgb && wdb) A4R_LNB(float, true, true); else if (wgb) A4R_LNB(float, true, false);
        else if (wdb) A4R_LNB(float, false, true); else A4R_LNB(float, false, false);
    }
#undef A4R_LNB
    return a4r_launch_status();
}

static int rows_copy(void* stream, const void* in, int ldi, void* out, int ldo, int n, int row_step, int H, int dtype, bool scatter) {
    if (!in || !out || bad_dtype(dtype) || n <= 0 || row_step <= 0 || H <= 0) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    if ((H * esz) % 16 || (ldi * esz) % 16 || (ldo * esz) % 16 || misaligned(in) || misaligned(out)) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t total = (size_t)n * (H * esz / 16);
    int grid = (int)((total + 255) / 256); if (grid > 2048) grid = 2048;
    if (dtype == A4R_BF16) {
        if (scatter) hipLaunchKernelGGL((rows_copy_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, s, (const bf16_t*)in, ldi, (bf16_t*)out, ldo, n, row_step, H);
        else hipLaunchKernelGGL((rows_copy_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, s, (const bf16_t*)in, ldi, (bf16_t*)out, ldo, n, row_step, H);
    } else {
        if (scatter) hipLaunchKernelGGL((rows_copy_kernel<float, true>), dim3(grid), dim3(256), 0, s, (const float*)in, ldi, (float*)out, ldo, n, row_step, H);
        else hipLaunchKernelGGL((rows_copy_kernel<float, false>), dim3(grid), dim3(256), 0, s, (const float*)in, ldi, (float*)out, ldo, n, row_step, H);
    }
    return a4r_launch_status();
}
extern "C" int a4r_rows_idx_copy(void* stream, const void* in, int64_t ldi_bytes, void* out, int64_t ldo_bytes, const int32_t* idx, int n,
                                 int64_t row_bytes, int scatter) {
    if (!in || !out || !idx || n <= 0 || row_bytes <= 0 || row_bytes % 16 || ldi_bytes % 16 || ldo_bytes % 16 || ldi_bytes < row_bytes || ldo_bytes < row_bytes ||
        misaligned(in) || misaligned(out) || row_bytes / 16 > 0x7fffffff)
        return A4R_EINVAL;
    const int pieces = (int)(row_bytes / 16);
    const size_t total = (size_t)n * pieces;
    int grid = (int)((total + 255) / 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(rows_idx_copy_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (const char*)in, (size_t)ldi_bytes, (char*)out,
                       (size_t)ldo_bytes, idx, n, pieces, scatter ? 1 : 0);
    return a4r_launch_status();
}
extern "C" int a4r_gather_rows(void* stream, const void* in, int ldi, void* out, int ldo, int n, int row_step, int H, int dtype) {
    return rows_copy(stream, in, ldi, out, ldo, n, row_step, H, dtype, false);
}
extern "C" int a4r_scatter_rows(void* stream, const void* in, int ldi, void* out, int ldo, int n, int row_step, int H, int dtype) {
    return rows_copy(stream, in, ldi, out, ldo, n, row_step, H, dtype, true);
}

extern "C" int a4r_scatter_rows_fill(void* stream, const void* in, int ldi, void* out, int ldo, int n, int row_step, int H, int dtype, int fill_rows) {
    if (!in || !out || bad_dtype(dtype) || n <= 0 || row_step <= 0 || H <= 0 || fill_rows < (n - 1) * row_step + 1) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2, per = 16 / esz;
    if (H % per || (ldi * esz) % 16 || (ldo * esz) % 16 || misaligned(in) || misaligned(out)) return A4R_EINVAL;
    const size_t total = (size_t)fill_rows * (H / per);
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == A4R_BF16) hipLaunchKernelGGL(scatter_fill_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)in, ldi, (bf16_t*)out, ldo, n, row_step, H, fill_rows);
    else hipLaunchKernelGGL(scatter_fill_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)in, ldi, (float*)out, ldo, n, row_step, H, fill_rows);
    return a4r_launch_status();
}

extern "C" int a4r_act_bwd_f32(void* stream, const float* dy, const float* pre, float* dx, int64_t n, int act) {
    if (!dy || !pre || !dx || n <= 0) return A4R_EINVAL;
    int grid = (int)((n + 255) / 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(act_bwd_f32_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dy, pre, dx, n, act);
    return a4r_launch_status();
}

extern "C" int a4r_dropout_apply(void* stream, const void* x, int ldx, void* y, int ldy, int M, int N, int dtype,
                                 float drop_p, uint32_t drop_site, uint64_t drop_seed) {
    if (!x || !y || bad_dtype(dtype) || M <= 0 || N <= 0 || N % 8 || drop_p < 0.f || drop_p >= 1.f) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    if ((ldx * esz) % 16 || (ldy * esz) % 16 || ldx < N || ldy < N || misaligned(x) || misaligned(y)) return A4R_EINVAL;
    const size_t total = (size_t)M * (N / 8);
    int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == A4R_BF16)
        hipLaunchKernelGGL(dropout_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, M, N,
                           drop_seed, drop_site, a4r_thr16(drop_p), a4r_keep_scale(drop_p));
    else
        hipLaunchKernelGGL(dropout_apply_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, M, N,
                           drop_seed, drop_site, a4r_thr16(drop_p), a4r_keep_scale(drop_p));
    return a4r_launch_status();
}
