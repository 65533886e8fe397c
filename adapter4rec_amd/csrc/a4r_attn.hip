// Short-sequence self-attention (S <= 32 keys, head dim 32 or 64): BERT title encoder (S = 30,
// 12 heads x 64) and the SASRec user encoder (S = 20, 2 heads x 32, causal & key mask).
// One 64-lane wave owns one (item, head): the whole score matrix is a 2x2 grid of 16x16 MFMA
// tiles held in registers, softmax is a 16-lane shuffle reduction, and nothing S x S ever reaches
// HBM.  Operands whose contraction index runs along tile rows (V in P.V; dO, Q, K in the backward
// products) are staged once in LDS and gathered down columns; P / dS are re-laid out through LDS.
// The kernel is HBM-bound (reads qkv once, writes ctx once); all softmax arithmetic is fp32.
#include <cstdlib>
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

template <typename T> A4R_DEV uint4 ldg16(const T* p) { return *reinterpret_cast<const uint4*>(p); }

// Every LDS region of these kernels is PRIVATE to one wave (smem + wave * WAVE_LDS): a wave's LDS operations execute in issue order,
// so a read sees the same wave's earlier writes once they have been issued -- what is needed between the write and the read phases is
// an ordering point for the compiler and the wave's own counter, not a workgroup barrier.  (With __syncthreads() the four (item, head)
// pairs of a workgroup ran in lock step: every wave waited for the slowest pair's loads at every phase.)
A4R_DEV void wave_lds_fence() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

template <typename T, int STRIDE> A4R_DEV void lds_put16(char* tile, int row, int ch, const uint4& v) {
    char* d = tile + row * STRIDE + ch * 16;
    if constexpr (STRIDE % 16 == 0) {
        *reinterpret_cast<uint4*>(d) = v;
    } else {   // 8-byte aligned rows
        reinterpret_cast<uint2*>(d)[0] = make_uint2(v.x, v.y);
        reinterpret_cast<uint2*>(d)[1] = make_uint2(v.z, v.w);
    }
}

template <typename T, int DH> struct AttnCfg {
    static constexpr int PER = Elem<T>::PER16;
    static constexpr int KSTEP = Mma<T>::KSTEP;
    static constexpr int KSD = DH / KSTEP;                 // chunk steps over the head dim
    static constexpr int KSP = 32 / KSTEP;                 // chunk steps over 32 keys / queries
    static constexpr int DT = DH / 16;                     // 16-wide d tiles
    static constexpr int CPR = DH / PER;                   // 16-byte chunks per head row
    static constexpr int NLD = 32 * CPR / 64;              // chunks each lane stages per 32-row tile
    static constexpr int PSTRIDE = 32 * (int)sizeof(T);    // [32][32] images, read as 16-byte chunks
    static constexpr int GSTRIDE = DH * (int)sizeof(T) + (sizeof(T) == 2 ? 8 : 16);   // gathered tiles (bank spread)
    static constexpr int OSTRIDE = DH * (int)sizeof(T);    // output staging, read as 16-byte chunks
};

// scores of one (item, head): sc[mt][nt] 16x16 tiles, rows = queries, cols = keys (rows/keys >= S are clamped loads)
template <typename T, int DH>
A4R_DEV void qk_scores(const T* base, int ld, int q_off, int k_off, int S, int lane, f32x4_t (&sc)[2][2]) {
    using C = AttnCfg<T, DH>;
    const int r16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) sc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::KSD; ++ks) {
        uint4 qa[2], kb[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = min(t * 16 + r16, S - 1);
            qa[t] = ldg16(base + (size_t)row * ld + q_off + (ks * 4 + kg) * C::PER);
            kb[t] = ldg16(base + (size_t)row * ld + k_off + (ks * 4 + kg) * C::PER);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) Mma<T>::mma(qa[mt], kb[nt], sc[mt][nt]);
    }
}

// softmax of row q (two key tiles); returns p[nt] and whether each entry is kept by dropout
A4R_DEV void softmax_row(const float (&s)[2], int q, int r16, int S, const float (&km)[2], int causal, float scale, float mask_neg,
                         float (&p)[2]) {
    float x[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int key = nt * 16 + r16;
        const float v = s[nt] * scale;
        const bool allowed = (km[nt] != 0.f) && (!causal || key <= q);
        x[nt] = key < S ? (allowed ? v : v + mask_neg) : -INFINITY;
    }
    const float m = group16_max(fmaxf(x[0], x[1]));
    const float e0 = expf(x[0] - m), e1 = expf(x[1] - m);
    const float inv = 1.f / group16_sum(e0 + e1);
    p[0] = e0 * inv;
    p[1] = e1 * inv;
}

template <typename T, int DH, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) attn_fwd_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                              T* __restrict__ out, int ldo, const float* __restrict__ key_mask,
                                                              int n_items, int S_in, int n_heads, int causal, float scale, float mask_neg,
                                                              uint64_t seed, uint32_t site, uint32_t thr16, float keep_scale,
                                                              const int* __restrict__ offsets) {
    using C = AttnCfg<T, DH>;
    constexpr int WAVE_LDS = 32 * C::PSTRIDE + 32 * C::GSTRIDE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    char* Ps = smem + wave * WAVE_LDS;
    char* Vs = Ps + 32 * C::PSTRIDE;
    const int total = n_items * n_heads;
    int gw = blockIdx.x * WAVES + wave;
    const bool active = gw < total;
    if (!active) gw = total - 1;
    const int item = gw / n_heads, head = gw % n_heads;
    // packed items (a4r_attn_t.offsets): item i owns rows [offsets[i], offsets[i + 1]) -- its own token count, no pad rows in the tensors
    const int row0 = offsets ? __builtin_amdgcn_readfirstlane(offsets[item]) : item * S_in;
    const int S = offsets ? __builtin_amdgcn_readfirstlane(offsets[item + 1]) - row0 : S_in;
    const T* base = qkv + (size_t)row0 * ld + head * DH;

    // (the key mask too: it used to be the third serial round trip of the wave)
    // Loads are UNCONDITIONAL (clamped rows / keys, the out-of-range ones zeroed where they are consumed): with `if (row < S)` around
    // them hipcc waited with vmcnt(0) after every conditional group -- three serialized round trips instead of one.
    float km[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int key = nt * 16 + r16;
        km[nt] = key_mask ? key_mask[(size_t)row0 + min(key, S - 1)] : 1.f;
    }
    // V is requested before the scores are computed (its loads used to be issued only after Q K^T had waited for Q and K)
    uint4 sv[C::NLD];
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) {
        const int id = lane + 64 * i, row = id / C::CPR, ch = id % C::CPR;
        sv[i] = ldg16(base + (size_t)min(row, S - 1) * ld + v_off + ch * C::PER);
    }
    f32x4_t sc[2][2];
    qk_scores<T, DH>(base, ld, q_off, k_off, S, lane, sc);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
        if (nt * 16 + r16 >= S) km[nt] = 0.f;

    // stage V (rows >= S are zero so that 0 * V stays 0)
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) {
        const int id = lane + 64 * i, row = id / C::CPR, ch = id % C::CPR;
        lds_put16<T, C::GSTRIDE>(Vs, row, ch, row < S ? sv[i] : make_uint4(0u, 0u, 0u, 0u));
    }

#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int q = mt * 16 + kg * 4 + rr;
            const float s2[2] = {sc[mt][0][rr], sc[mt][1][rr]};
            float p[2];
            softmax_row(s2, q, r16, S, km, causal, scale, mask_neg, p);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int key = nt * 16 + r16;
                float pv = p[nt];
                if (thr16) pv = dropout_keep(seed, site, ((uint64_t)gw * 32 + q) * 32 + key, thr16) ? pv * keep_scale : 0.f;
                Elem<T>::st(reinterpret_cast<T*>(Ps + q * C::PSTRIDE) + key, pv);
            }
        }
    wave_lds_fence();

    f32x4_t o[2][C::DT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) o[mt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::KSP; ++ks) {
        uint4 pa[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            pa[mt] = *reinterpret_cast<const uint4*>(Ps + (mt * 16 + r16) * C::PSTRIDE + (ks * 4 + kg) * 16);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            const uint4 vb = gather_chunk<T>(Vs, C::GSTRIDE, ks * C::KSTEP, dt * 16, lane);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) Mma<T>::mma(pa[mt], vb, o[mt][dt]);
        }
    }
    wave_lds_fence();
    // stage O over the V tile, then 16-byte row stores
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                Elem<T>::st(reinterpret_cast<T*>(Vs + (mt * 16 + kg * 4 + rr) * C::OSTRIDE) + dt * 16 + r16, o[mt][dt][rr]);
    wave_lds_fence();
    if (active) {
#pragma unroll
        for (int i = 0; i < C::NLD; ++i) {
            const int id = lane + 64 * i, row = id / C::CPR, ch = id % C::CPR;
            if (row < S)
                *reinterpret_cast<uint4*>(out + ((size_t)row0 + row) * ldo + head * DH + ch * C::PER) =
                    *reinterpret_cast<const uint4*>(Vs + row * C::OSTRIDE + ch * 16);
        }
    }
}

// one [32][DH] product  acc[t][dt] = sum_k Aimg[t*16 + r][k] * Gtile[k][dt*16 + c]
template <typename T, int DH>
A4R_DEV void img_times_tile(const char* Aimg, const char* Gtile, int lane, f32x4_t (&acc)[2][AttnCfg<T, DH>::DT]) {
    using C = AttnCfg<T, DH>;
    const int r16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) acc[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C::KSP; ++ks) {
        uint4 a[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const uint4*>(Aimg + (t * 16 + r16) * C::PSTRIDE + (ks * 4 + kg) * 16);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            const uint4 b = gather_chunk<T>(Gtile, C::GSTRIDE, ks * C::KSTEP, dt * 16, lane);
#pragma unroll
            for (int t = 0; t < 2; ++t) Mma<T>::mma(a[t], b, acc[t][dt]);
        }
    }
}

// write a [32][DH] accumulator block into LDS (row-major, OSTRIDE) and copy rows < S to global with 16-byte stores
template <typename T, int DH>
A4R_DEV void store_block(char* stage, const f32x4_t (&acc)[2][AttnCfg<T, DH>::DT], T* gdst, int ldg, int S, int lane, bool active) {
    using C = AttnCfg<T, DH>;
    const int r16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                Elem<T>::st(reinterpret_cast<T*>(stage + (t * 16 + kg * 4 + rr) * C::OSTRIDE) + dt * 16 + r16, acc[t][dt][rr]);
    wave_lds_fence();
    if (active) {
#pragma unroll
        for (int i = 0; i < C::NLD; ++i) {
            const int id = lane + 64 * i, row = id / C::CPR, ch = id % C::CPR;
            if (row < S)
                *reinterpret_cast<uint4*>(gdst + (size_t)row * ldg + ch * C::PER) =
                    *reinterpret_cast<const uint4*>(stage + row * C::OSTRIDE + ch * 16);
        }
    }
}

template <typename T, int DH, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) attn_bwd_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                              const T* __restrict__ dout, int ldo, T* __restrict__ dqkv,
                                                              const float* __restrict__ key_mask, int n_items, int S_in, int n_heads,
                                                              int causal, float scale, float mask_neg,
                                                              uint64_t seed, uint32_t site, uint32_t thr16, float keep_scale,
                                                              const int* __restrict__ offsets) {
    using C = AttnCfg<T, DH>;
    constexpr int TILE = 32 * C::GSTRIDE, IMG = 32 * C::PSTRIDE;
    constexpr int WAVE_LDS = 3 * TILE + 3 * IMG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    char* Qs = smem + wave * WAVE_LDS;
    char* Ks = Qs + TILE;
    char* dOs = Ks + TILE;
    char* PT = dOs + TILE;      // [key][q]  = P' (after dropout)
    char* dS = PT + IMG;        // [q][key]
    char* dST = dS + IMG;       // [key][q]
    const int total = n_items * n_heads;
    int gw = blockIdx.x * WAVES + wave;
    const bool active = gw < total;
    if (!active) gw = total - 1;
    const int item = gw / n_heads, head = gw % n_heads;
    const int row0 = offsets ? __builtin_amdgcn_readfirstlane(offsets[item]) : item * S_in;
    const int S = offsets ? __builtin_amdgcn_readfirstlane(offsets[item + 1]) - row0 : S_in;
    const T* base = qkv + (size_t)row0 * ld + head * DH;
    const T* dbase = dout + (size_t)row0 * ldo + head * DH;
    T* gbase = dqkv + (size_t)row0 * ld + head * DH;

    float km[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int key = nt * 16 + r16;
        km[nt] = key_mask ? key_mask[(size_t)row0 + min(key, S - 1)] : 1.f;      // (unconditional load, clamped; keys >= S zeroed below)
    }
    // Every global read of the pair is requested BEFORE anything is consumed, and every byte ONCE: the row-major operand fragments
    // of Q K^T and dO V^T (lane (r16, kg) holds the 16-byte chunk ks * 4 + kg of rows r16 and 16 + r16) are exactly the chunks the
    // staged copies of Q, K, dO are made of, so the same registers fill the LDS tiles (they used to be loaded a second time in
    // row-chunk order: 28 loads per lane instead of 16).  Rows >= S: Q, K, V clamp to row S - 1 (finite values; their scores are
    // masked / their dS rows are zero because dO is zero there), dO is zero.
    uint4 fq[C::KSD][2], fk[C::KSD][2], fd[C::KSD][2], fv[C::KSD][2];
#pragma unroll
    for (int ks = 0; ks < C::KSD; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = t * 16 + r16, rc = min(row, S - 1);
            fq[ks][t] = ldg16(base + (size_t)rc * ld + q_off + (ks * 4 + kg) * C::PER);
            fk[ks][t] = ldg16(base + (size_t)rc * ld + k_off + (ks * 4 + kg) * C::PER);
            fv[ks][t] = ldg16(base + (size_t)rc * ld + v_off + (ks * 4 + kg) * C::PER);
            fd[ks][t] = ldg16(dbase + (size_t)rc * ldo + (ks * 4 + kg) * C::PER);      // (unconditional: rows >= S are zeroed when staged)
        }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
        if (nt * 16 + r16 >= S) km[nt] = 0.f;
#pragma unroll
    for (int ks = 0; ks < C::KSD; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = t * 16 + r16, ch = ks * 4 + kg;
            const bool in = row < S;                       // staged rows >= S are ZERO (they are contraction terms of dK, dV / dQ)
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            lds_put16<T, C::GSTRIDE>(Qs, row, ch, in ? fq[ks][t] : z);
            lds_put16<T, C::GSTRIDE>(Ks, row, ch, in ? fk[ks][t] : z);
            if (!in) fd[ks][t] = z;
            lds_put16<T, C::GSTRIDE>(dOs, row, ch, fd[ks][t]);
        }

    f32x4_t sc[2][2], dp[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) { sc[mt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[mt][nt] = sc[mt][nt]; }
#pragma unroll
    for (int ks = 0; ks < C::KSD; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                Mma<T>::mma(fq[ks][mt], fk[ks][nt], sc[mt][nt]);
                Mma<T>::mma(fd[ks][mt], fv[ks][nt], dp[mt][nt]);      // dP' = dO . V^T  (dO rows >= S are zero, so dS rows >= S vanish)
            }

#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int q = mt * 16 + kg * 4 + rr;
            const float s2[2] = {sc[mt][0][rr], sc[mt][1][rr]};
            float p[2], g[2], pd[2];
            softmax_row(s2, q, r16, S, km, causal, scale, mask_neg, p);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int key = nt * 16 + r16;
                float keepf = 1.f;
                if (thr16) keepf = dropout_keep(seed, site, ((uint64_t)gw * 32 + q) * 32 + key, thr16) ? keep_scale : 0.f;
                pd[nt] = p[nt] * keepf;
                g[nt] = dp[mt][nt][rr] * keepf;           // d loss / d P
            }
            const float delta = group16_sum(g[0] * p[0] + g[1] * p[1]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int key = nt * 16 + r16;
                const float ds = p[nt] * (g[nt] - delta) * scale;
                Elem<T>::st(reinterpret_cast<T*>(PT + key * C::PSTRIDE) + q, pd[nt]);
                Elem<T>::st(reinterpret_cast<T*>(dS + q * C::PSTRIDE) + key, ds);
                Elem<T>::st(reinterpret_cast<T*>(dST + key * C::PSTRIDE) + q, ds);
            }
        }
    wave_lds_fence();

    f32x4_t acc[2][C::DT];
    img_times_tile<T, DH>(PT, dOs, lane, acc);          // dV[key][d] = sum_q P'[q][key] dO[q][d]
    wave_lds_fence();
    store_block<T, DH>(dOs, acc, gbase + v_off, ld, S, lane, active);
    img_times_tile<T, DH>(dST, Qs, lane, acc);          // dK[key][d] = sum_q dS[q][key] Q[q][d]
    wave_lds_fence();
    store_block<T, DH>(Qs, acc, gbase + k_off, ld, S, lane, active);
    img_times_tile<T, DH>(dS, Ks, lane, acc);           // dQ[q][d] = sum_key dS[q][key] K[key][d]
    wave_lds_fence();
    store_block<T, DH>(Ks, acc, gbase + q_off, ld, S, lane, active);
}

// ------------------------------------------------------------------------------------------------ backward, bf16 (round 2)
#ifndef A4R_ATTN_BWD_ABL_CT
#define A4R_ATTN_BWD_ABL_CT 0      /* timing-only builds (-DA4R_ATTN_BWD_ABL_CT=n; wrong results): 1 no stores, 2 no loads -- compile time: see the load loop */
#endif
// The same mathematics arranged around what timing ablations showed (tools/attn_short_bench.py, A4R_ATTN_BWD_ABL: 1 = no stores,
// 2 = no loads).  With neither loads nor stores the generic kernel above still took ~90 of its ~150 us at the text tower's shape: it
// is INSTRUCTION-bound (a 16-lane shuffle reduction per query row, one dropout hash per element, precise expf, ~300 two-byte LDS
// operations per lane for the P / dS / dS^T images, column gathers, output staging), not HBM-bound.  Here (40 us of work, 100 us total):
//   * the scores are produced TRANSPOSED (S^T = K Q^T: rows = keys, the lane's column = a query), so a query's 32 keys are 8
//     registers x the 4 lanes l, l^16, l^32, l^48: softmax and delta are register sums + two xor-shuffles, 4 consecutive keys share
//     one dropout hash (the lots dropout_keep draws from), exp is v_exp_f32;
//   * a 16 x 16 accumulator tile IS an MFMA operand chunk once the contraction index is permuted -- k-slot (kg, j) <-> row
//     (j >> 2) * 16 + 4 kg + (j & 3) of the tile pair (tile 0 | tile 1) -- the same permutation ds_read_b64_tr_b16 applies to a
//     row-major LDS image read down its columns (a4r_attn_long.hip).  dQ^T = K^T dS^T therefore takes dS^T straight from the
//     accumulator registers and two transposed reads of the K image (the image = plain 16-byte stores of the very registers the
//     score MFMAs consumed);
//   * dV^T = dO^T P' and dK^T = Q^T dS contract over the queries with the KEY on the lane: P'^T and dS^T change hands once through
//     the then-free K image (8 two-byte stores, 4 eight-byte loads per lane);
//   * the three transposed products go back through the images so that rows leave as whole 16-byte chunks, 8 lanes = one 128-byte
//     line: stored straight from the accumulators (8 bytes per lane, 32-byte pieces of 16 rows per instruction) the same bytes took
//     32 us longer.
// 12 KiB of LDS per wave (was 18.75), 140 VGPRs: three workgroups per CU.  Mask, causal and dropout semantics are the generic kernel's.
typedef short bt_v4s_t __attribute__((ext_vector_type(4)));
template <int DH> struct BtGeo {
    static constexpr int ROWB = DH * 2, CPR = DH / 8, KSD = DH / 32, ND = DH / 16, IMG = 32 * ROWB;
    static constexpr int WAVE_LDS = 2 * IMG + (IMG > 4096 ? IMG : 4096);      // Q, dO, K images; the K slot later holds P'^T | dS^T (2 x 2 KiB)
    static A4R_DEV int swz(int row) { return CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
};
template <int DH> A4R_DEV uint4 bt_frag_tr(const char* img, int d0, int lane) {
    using G = BtGeo<DH>;
    const int kg = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int chunk = (d0 >> 3) + (pp >> 1);
    const int r0 = 4 * kg + q, r1 = r0 + 16;
    const char* a0 = img + r0 * G::ROWB + ((chunk ^ G::swz(r0)) << 4) + 8 * (pp & 1);
    const char* a1 = img + r1 * G::ROWB + ((chunk ^ G::swz(r1)) << 4) + 8 * (pp & 1);
    const bt_v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bt_v4s_t*)(a0));
    const bt_v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bt_v4s_t*)(a1));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}
A4R_DEV uint4 bt_pack(const f32x4_t& a, const f32x4_t& b) {       // tile 0 rows | tile 1 rows of one lane column -> operand chunk
    return make_uint4(pack2_bf16(a[0], a[1]), pack2_bf16(a[2], a[3]),
                      pack2_bf16(b[0], b[1]), pack2_bf16(b[2], b[3]));
}
A4R_DEV void bt_store4(bf16_t* p, const f32x4_t& v) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack2_bf16(v[0], v[1]),
                                              pack2_bf16(v[2], v[3]));
}
A4R_DEV float bt_red4(float v, bool mx) {       // over the 4 lanes l, l^16, l^32, l^48 that share a column
    const float a = __shfl_xor(v, 16, 64);
    v = mx ? fmaxf(v, a) : v + a;
    const float b = __shfl_xor(v, 32, 64);
    return mx ? fmaxf(v, b) : v + b;
}

template <int DH, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, 3) attn_bwd_tr_kernel(const bf16_t* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                                 const bf16_t* __restrict__ dout, int ldo, bf16_t* __restrict__ dqkv,
                                                                 const float* __restrict__ key_mask, int n_items, int S_in, int n_heads,
                                                                 int causal, float scale, float mask_neg,
                                                                 uint64_t seed, uint32_t site, uint32_t thr16, float keep_scale,
                                                                 const int* __restrict__ offsets) {
    using T = bf16_t;
    using G = BtGeo<DH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    char* Qi = smem + wave * G::WAVE_LDS;
    char* Oi = Qi + G::IMG;
    char* Ki = Oi + G::IMG;
    const int total = n_items * n_heads;
    const int gw = blockIdx.x * WAVES + wave;
    if (gw >= total) return;                      // whole waves leave (no workgroup barrier below; EXEC stays all ones for the transposed reads)
    const int item = gw / n_heads, head = gw % n_heads;
    const int row0 = offsets ? __builtin_amdgcn_readfirstlane(offsets[item]) : item * S_in;
    const int S = offsets ? __builtin_amdgcn_readfirstlane(offsets[item + 1]) - row0 : S_in;
    const T* base = qkv + (size_t)row0 * ld + head * DH;
    const T* dbase = dout + (size_t)row0 * ldo + head * DH;
    T* gbase = dqkv + (size_t)row0 * ld + head * DH;

    float kmT[2][4];                              // key mask of the keys down the lane's rows
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kt = nt * 16 + kg * 4 + r;
            kmT[nt][r] = (kt < S) ? (key_mask ? key_mask[(size_t)row0 + kt] : 1.f) : 0.f;      // (as a clamped unconditional load hipcc turned each value into a predicate at once: eight serialized waits)
        }
    // every global read of the pair is requested before anything is consumed, every byte once (rows >= S: Q, K, V clamp to row S - 1,
    // finite values whose scores are masked / whose dS rows vanish because dO is zero there)
    uint4 fq[G::KSD][2], fk[G::KSD][2], fd[G::KSD][2], fv[G::KSD][2];
#pragma unroll
    for (int ks = 0; ks < G::KSD; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = t * 16 + r16, rc = min(row, S - 1);
            // (all sixteen loads UNCONDITIONAL and back to back: behind a run-time ablation test, or with dO under `if (row < S)`, hipcc
            // put an s_waitcnt vmcnt(0) after every group of four -- four serialized round trips at the start of every workgroup)
            if (A4R_ATTN_BWD_ABL_CT & 2) { fq[ks][t] = make_uint4(lane, ks, t, 1); fk[ks][t] = fq[ks][t]; fv[ks][t] = fq[ks][t]; fd[ks][t] = fq[ks][t]; continue; }
            fq[ks][t] = ldg16(base + (size_t)rc * ld + q_off + (ks * 4 + kg) * 8);
            fk[ks][t] = ldg16(base + (size_t)rc * ld + k_off + (ks * 4 + kg) * 8);
            fv[ks][t] = ldg16(base + (size_t)rc * ld + v_off + (ks * 4 + kg) * 8);
            fd[ks][t] = ldg16(dbase + (size_t)rc * ldo + (ks * 4 + kg) * 8);      // rows >= S: zeroed below, where it is consumed
        }
#pragma unroll
    for (int ks = 0; ks < G::KSD; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = t * 16 + r16, off = row * G::ROWB + (((ks * 4 + kg) ^ G::swz(row)) << 4);
            if (row >= S) fd[ks][t] = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(Qi + off) = fq[ks][t];
            *reinterpret_cast<uint4*>(Ki + off) = fk[ks][t];
            *reinterpret_cast<uint4*>(Oi + off) = fd[ks][t];
        }
    f32x4_t scT[2][2], dpT[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) { scT[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dpT[a][b] = scT[a][b]; }
#pragma unroll
    for (int ks = 0; ks < G::KSD; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                Mma<T>::mma(fk[ks][nt], fq[ks][mt], scT[nt][mt]);      // S^T  : rows key (4 kg + r of tile nt), column q = r16 of tile mt
                Mma<T>::mma(fv[ks][nt], fd[ks][mt], dpT[nt][mt]);      // dP'^T = V dO^T
            }
    wave_lds_fence();                                     // the images are complete (this wave's own stores)
    // ---- softmax / dropout / dS with the QUERY on the lane: its 32 keys are 8 registers x the lanes l, l^16, l^32, l^48
    uint2 pw[2][2], dw[2][2];                             // P'^T and dS^T tiles as packed bf16 (4 consecutive keys of the lane's query)
    f32x4_t aq[2][G::ND], av[2][G::ND], ak[2][G::ND];     // the three products, transposed: rows = 4 head columns, the lane's column = a token
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int q = mt * 16 + r16;
        float x[2][4];
        float m = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = nt * 16 + kg * 4 + r;
                const float v = scT[nt][mt][r] * scale;
                const bool allowed = (kmT[nt][r] != 0.f) && (!causal || key <= q);
                x[nt][r] = key < S ? (allowed ? v : v + mask_neg) : -INFINITY;
                m = fmaxf(m, x[nt][r]);
            }
        m = bt_red4(m, true);
        float l = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { x[nt][r] = __expf(x[nt][r] - m); l += x[nt][r]; }
        const float inv = 1.f / bt_red4(l, false);
        float dsum = 0.f;
        f32x4_t pk[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            uint64_t hsh = 0;
            if (thr16) hsh = a4r_hash64(seed, site, (((uint64_t)gw * 32 + q) * 32 + nt * 16 + kg * 4) >> 2);      // = dropout_keep's lots
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x[nt][r] *= inv;                                                      // P
                float keepf = 1.f;
                if (thr16) keepf = (((uint32_t)(hsh >> (16 * r)) & 0xffffu) >= thr16) ? keep_scale : 0.f;
                dpT[nt][mt][r] *= keepf;                                              // d loss / d P
                pk[nt][r] = x[nt][r] * keepf;                                         // P'
                dsum += dpT[nt][mt][r] * x[nt][r];
            }
        }
        dsum = bt_red4(dsum, false);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dpT[nt][mt][r] = x[nt][r] * (dpT[nt][mt][r] - dsum) * scale;       // dS^T
            pw[nt][mt] = make_uint2(pack2_bf16(pk[nt][0], pk[nt][1]),
                                    pack2_bf16(pk[nt][2], pk[nt][3]));
            dw[nt][mt] = make_uint2(pack2_bf16(dpT[nt][mt][0], dpT[nt][mt][1]),
                                    pack2_bf16(dpT[nt][mt][2], dpT[nt][mt][3]));
        }
        const uint4 dsB = make_uint4(dw[0][mt].x, dw[0][mt].y, dw[1][mt].x, dw[1][mt].y);
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) {
            aq[mt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            Mma<T>::mma(bt_frag_tr<DH>(Ki, dt * 16, lane), dsB, aq[mt][dt]);       // dQ^T[d][q] = sum_key K[key][d] dS[q][key]
        }
    }
    // ---- the other two products contract over the QUERIES with the key on the lane: the 16 x 16 tiles change hands through the
    // (now free) K image, written [key][query] two bytes at a time and read back as the operand chunk itself (2 x 8 bytes)
    wave_lds_fence();
    unsigned short* Pt = reinterpret_cast<unsigned short*>(Ki);           // [32 keys][32 queries] bf16
    unsigned short* Dt = Pt + 32 * 32;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int q = mt * 16 + r16, k0 = nt * 16 + kg * 4;
            Pt[(k0 + 0) * 32 + q] = (unsigned short)(pw[nt][mt].x & 0xffffu);
            Pt[(k0 + 1) * 32 + q] = (unsigned short)(pw[nt][mt].x >> 16);
            Pt[(k0 + 2) * 32 + q] = (unsigned short)(pw[nt][mt].y & 0xffffu);
            Pt[(k0 + 3) * 32 + q] = (unsigned short)(pw[nt][mt].y >> 16);
            Dt[(k0 + 0) * 32 + q] = (unsigned short)(dw[nt][mt].x & 0xffffu);
            Dt[(k0 + 1) * 32 + q] = (unsigned short)(dw[nt][mt].x >> 16);
            Dt[(k0 + 2) * 32 + q] = (unsigned short)(dw[nt][mt].y & 0xffffu);
            Dt[(k0 + 3) * 32 + q] = (unsigned short)(dw[nt][mt].y >> 16);
        }
    wave_lds_fence();
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int key = nt * 16 + r16;
        const uint2 p0 = *reinterpret_cast<const uint2*>(Pt + key * 32 + kg * 4), p1 = *reinterpret_cast<const uint2*>(Pt + key * 32 + 16 + kg * 4);
        const uint2 d0 = *reinterpret_cast<const uint2*>(Dt + key * 32 + kg * 4), d1 = *reinterpret_cast<const uint2*>(Dt + key * 32 + 16 + kg * 4);
        const uint4 pB = make_uint4(p0.x, p0.y, p1.x, p1.y), dsB = make_uint4(d0.x, d0.y, d1.x, d1.y);
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) {
            av[nt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            ak[nt][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            Mma<T>::mma(bt_frag_tr<DH>(Oi, dt * 16, lane), pB, av[nt][dt]);        // dV^T[d][key] = sum_q dO[q][d] P'[q][key]
            Mma<T>::mma(bt_frag_tr<DH>(Qi, dt * 16, lane), dsB, ak[nt][dt]);       // dK^T[d][key] = sum_q Q[q][d] dS[q][key]
        }
    }
    // ---- out: the three [32][DH] blocks go through the (now free) images so that a row leaves as whole 16-byte chunks, 8 lanes = one
    // 128-byte line.  (Stored straight from the accumulators -- 8 bytes per lane, 32-byte pieces of 16 rows per instruction -- the same
    // bytes took 32 us longer: A4R_ATTN_BWD_ABL=9 in tools/attn_short_bench.py.)
    wave_lds_fence();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tok = t * 16 + r16;
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) {
            const int off = tok * G::ROWB + (((dt * 2 + (kg >> 1)) ^ G::swz(tok)) << 4) + 8 * (kg & 1);
            bt_store4(reinterpret_cast<T*>(Qi + off), aq[t][dt]);
            bt_store4(reinterpret_cast<T*>(Oi + off), av[t][dt]);
            bt_store4(reinterpret_cast<T*>(Ki + off), ak[t][dt]);
        }
    }
    wave_lds_fence();
    if (!(A4R_ATTN_BWD_ABL_CT & 1)) {
#pragma unroll
        for (int i = 0; i < 32 * G::CPR / 64; ++i) {
            const int id = lane + 64 * i, row = id / G::CPR, ch = id % G::CPR;
            if (row < S) {
                const int off = row * G::ROWB + ((ch ^ G::swz(row)) << 4);
                *reinterpret_cast<uint4*>(gbase + (size_t)row * ld + q_off + ch * 8) = *reinterpret_cast<const uint4*>(Qi + off);
                *reinterpret_cast<uint4*>(gbase + (size_t)row * ld + v_off + ch * 8) = *reinterpret_cast<const uint4*>(Oi + off);
                *reinterpret_cast<uint4*>(gbase + (size_t)row * ld + k_off + ch * 8) = *reinterpret_cast<const uint4*>(Ki + off);
            }
        }
    }
}

struct Launch {
    hipStream_t s; const a4r_attn_t* a; uint32_t thr; float ks;
};

template <typename T, int DH, int WAVES>
int launch_fwd(const Launch& L) {
    using C = AttnCfg<T, DH>;
    const a4r_attn_t& a = *L.a;
    constexpr int LDS = WAVES * (32 * C::PSTRIDE + 32 * C::GSTRIDE);
    const int total = a.n_items * a.n_heads;
    hipLaunchKernelGGL((attn_fwd_kernel<T, DH, WAVES>), dim3((total + WAVES - 1) / WAVES), dim3(WAVES * 64), LDS, L.s,
                       (const T*)a.qkv, a.ld, a.q_off, a.k_off, a.v_off, (T*)a.out, a.ldo, a.offsets ? nullptr : a.key_mask, a.n_items, a.S, a.n_heads,
                       a.causal, a.scale, a.mask_neg, a.drop_seed, a.drop_site, L.thr, L.ks, a.offsets);
    return a4r_launch_status();
}
template <typename T, int DH, int WAVES>
int launch_bwd(const Launch& L) {
    using C = AttnCfg<T, DH>;
    const a4r_attn_t& a = *L.a;
    static const int pad = getenv("A4R_ATTN_BWD_LDS_PAD") ? atoi(getenv("A4R_ATTN_BWD_LDS_PAD")) : 0;      // occupancy experiment (tools/)
    const int LDS = WAVES * (3 * 32 * C::GSTRIDE + 3 * 32 * C::PSTRIDE) + pad;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<T, DH, WAVES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int total = a.n_items * a.n_heads;
    hipLaunchKernelGGL((attn_bwd_kernel<T, DH, WAVES>), dim3((total + WAVES - 1) / WAVES), dim3(WAVES * 64), LDS, L.s,
                       (const T*)a.qkv, a.ld, a.q_off, a.k_off, a.v_off, (const T*)a.dout, a.ldo, (T*)a.dqkv, a.offsets ? nullptr : a.key_mask,
                       a.n_items, a.S, a.n_heads, a.causal, a.scale, a.mask_neg, a.drop_seed, a.drop_site, L.thr, L.ks, a.offsets);
    return a4r_launch_status();
}

template <int DH, int WAVES>
int launch_bwd_tr(const Launch& L) {
    const a4r_attn_t& a = *L.a;
    constexpr int LDS = WAVES * BtGeo<DH>::WAVE_LDS;
    static bool attr_set = false;
    if (!attr_set && LDS > 48 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_tr_kernel<DH, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int total = a.n_items * a.n_heads;
    hipLaunchKernelGGL((attn_bwd_tr_kernel<DH, WAVES>), dim3((total + WAVES - 1) / WAVES), dim3(WAVES * 64), LDS, L.s,
                       (const bf16_t*)a.qkv, a.ld, a.q_off, a.k_off, a.v_off, (const bf16_t*)a.dout, a.ldo, (bf16_t*)a.dqkv, a.offsets ? nullptr : a.key_mask,
                       a.n_items, a.S, a.n_heads, a.causal, a.scale, a.mask_neg, a.drop_seed, a.drop_site, L.thr, L.ks, a.offsets);
    return a4r_launch_status();
}
const bool g_attn_bwd_tr = !(getenv("A4R_ATTN_BWD_TR") && atoi(getenv("A4R_ATTN_BWD_TR")) == 0);      // 0: the generic kernel for bf16 too (A/B, tests)

int check(const a4r_attn_t* a, bool bwd) {
    if (!a || !a->qkv) return A4R_EINVAL;
    if (bwd ? (!a->dout || !a->dqkv) : !a->out) return A4R_EINVAL;
    if (a->dtype != A4R_BF16 && a->dtype != A4R_F32) return A4R_EINVAL;
    const bool narrow = a->dh > 0 && a->dh <= 16;          // K-Adapter blocks: scalar kernels of a4r_attn_small.hip
    // head widths 128 / 256 (fp32 only): the user tower at --embedding_dim 256 / 512 with the default 2 heads (Downstream/Text/parameters.py:27-28)
    const bool wide = (a->dh == 128 || a->dh == 256) && a->dtype == A4R_F32 && !a->offsets;
    if (a->n_items <= 0 || a->n_heads <= 0 || a->S <= 0 || a->S > 32 || (a->dh != 32 && a->dh != 64 && !narrow && !wide)) return A4R_EINVAL;
    const int esz = a->dtype == A4R_F32 ? 4 : 2, per = narrow ? 1 : 16 / esz;
    if ((a->ld * esz) % 16 || (a->ldo * esz) % 16 || a->q_off % per || a->k_off % per || a->v_off % per) return A4R_EINVAL;
    if (a->ldo < a->n_heads * a->dh) return A4R_EINVAL;
    if (a->drop_p < 0.f || a->drop_p >= 1.f) return A4R_EINVAL;
    if (a->offsets && narrow) return A4R_EINVAL;        // packed items: the MFMA kernels only
    uintptr_t m = reinterpret_cast<uintptr_t>(a->qkv) | reinterpret_cast<uintptr_t>(a->out) | reinterpret_cast<uintptr_t>(a->dout) |
                  reinterpret_cast<uintptr_t>(a->dqkv);
    if (m & 15u) return A4R_EINVAL;
    return A4R_OK;
}

}  // namespace

int a4r_attn_small(hipStream_t s, const a4r_attn_t* t, bool bwd);   // a4r_attn_small.hip

extern "C" int a4r_attn_fwd(void* stream, const a4r_attn_t* a) {
    if (int e = check(a, false)) return e;
    if (a->dh <= 16) return a4r_attn_small(reinterpret_cast<hipStream_t>(stream), a, false);
    Launch L{reinterpret_cast<hipStream_t>(stream), a, a4r_thr16(a->drop_p), a4r_keep_scale(a->drop_p)};
    if (a->dtype == A4R_BF16) return a->dh == 64 ? launch_fwd<bf16_t, 64, 4>(L) : launch_fwd<bf16_t, 32, 4>(L);
    if (a->dh == 128) return launch_fwd<float, 128, 2>(L);
    if (a->dh == 256) return launch_fwd<float, 256, 1>(L);
    return a->dh == 64 ? launch_fwd<float, 64, 4>(L) : launch_fwd<float, 32, 4>(L);
}

extern "C" int a4r_attn_bwd(void* stream, const a4r_attn_t* a) {
    if (int e = check(a, true)) return e;
    if (a->dh <= 16) return a4r_attn_small(reinterpret_cast<hipStream_t>(stream), a, true);
    Launch L{reinterpret_cast<hipStream_t>(stream), a, a4r_thr16(a->drop_p), a4r_keep_scale(a->drop_p)};
    if (a->dtype == A4R_BF16 && g_attn_bwd_tr) {
        // 4 waves per workgroup: 1 .. 4 measured the same (100 - 106 us at the text tower's shape), 6 and 12 (= all heads of an item in
        // one workgroup) 127 / 134 us -- a workgroup's LDS is only released when its slowest wave is done
        return a->dh == 64 ? launch_bwd_tr<64, 4>(L) : launch_bwd_tr<32, 4>(L);
    }
    if (a->dtype == A4R_BF16) return a->dh == 64 ? launch_bwd<bf16_t, 64, 4>(L) : launch_bwd<bf16_t, 32, 4>(L);
    if (a->dh == 128) return launch_bwd<float, 128, 1>(L);
    if (a->dh == 256) return launch_bwd<float, 256, 1>(L);
    return a->dh == 64 ? launch_bwd<float, 64, 2>(L) : launch_bwd<float, 32, 4>(L);
}
