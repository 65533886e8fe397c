// Short-sequence self-attention (S <= 32 keys, head dim 32 or 64): BERT title encoder (S = 30,
// 12 heads x 64) and the SASRec user encoder (S = 20, 2 heads x 32, causal & key mask).
// One 64-lane wave owns one (item, head): the whole score matrix is a 2x2 grid of 16x16 MFMA
// tiles held in registers, softmax is a 16-lane shuffle reduction, and nothing S x S ever reaches
// HBM.  Operands whose contraction index runs along tile rows (V in P.V; dO, Q, K in the backward
// products) are staged once in LDS and gathered down columns; P / dS are re-laid out through LDS.
// The kernel is HBM-bound (reads qkv once, writes ctx once); all softmax arithmetic is fp32.
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
                                                              int n_items, int S, int n_heads, int causal, float scale, float mask_neg,
                                                              uint64_t seed, uint32_t site, uint32_t thr16, float keep_scale) {
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
    const T* base = qkv + (size_t)item * S * ld + head * DH;

    // (the key mask too: it used to be the third serial round trip of the wave)
    float km[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int key = nt * 16 + r16;
        km[nt] = (key < S) ? (key_mask ? key_mask[(size_t)item * S + key] : 1.f) : 0.f;
    }
    // V is requested before the scores are computed (its loads used to be issued only after Q K^T had waited for Q and K)
    uint4 sv[C::NLD];
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) {
        const int id = lane + 64 * i, row = id / C::CPR, ch = id % C::CPR;
        sv[i] = make_uint4(0u, 0u, 0u, 0u);
        if (row < S) sv[i] = ldg16(base + (size_t)row * ld + v_off + ch * C::PER);
    }
    f32x4_t sc[2][2];
    qk_scores<T, DH>(base, ld, q_off, k_off, S, lane, sc);

    // stage V (rows >= S are zero so that 0 * V stays 0)
#pragma unroll
    for (int i = 0; i < C::NLD; ++i) {
        const int id = lane + 64 * i, row = id / C::CPR, ch = id % C::CPR;
        lds_put16<T, C::GSTRIDE>(Vs, row, ch, sv[i]);
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
                *reinterpret_cast<uint4*>(out + ((size_t)item * S + row) * ldo + head * DH + ch * C::PER) =
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
                                                              const float* __restrict__ key_mask, int n_items, int S, int n_heads,
                                                              int causal, float scale, float mask_neg,
                                                              uint64_t seed, uint32_t site, uint32_t thr16, float keep_scale) {
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
    const T* base = qkv + (size_t)item * S * ld + head * DH;
    const T* dbase = dout + (size_t)item * S * ldo + head * DH;
    T* gbase = dqkv + (size_t)item * S * ld + head * DH;

    float km[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int key = nt * 16 + r16;
        km[nt] = (key < S) ? (key_mask ? key_mask[(size_t)item * S + key] : 1.f) : 0.f;
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
            fd[ks][t] = make_uint4(0u, 0u, 0u, 0u);
            if (row < S) fd[ks][t] = ldg16(dbase + (size_t)row * ldo + (ks * 4 + kg) * C::PER);
        }
#pragma unroll
    for (int ks = 0; ks < C::KSD; ++ks)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = t * 16 + r16, ch = ks * 4 + kg;
            const bool in = row < S;                       // staged rows >= S are ZERO (they are contraction terms of dK, dV / dQ)
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            lds_put16<T, C::GSTRIDE>(Qs, row, ch, in ? fq[ks][t] : z);
            lds_put16<T, C::GSTRIDE>(Ks, row, ch, in ? fk[ks][t] : z);
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
                       (const T*)a.qkv, a.ld, a.q_off, a.k_off, a.v_off, (T*)a.out, a.ldo, a.key_mask, a.n_items, a.S, a.n_heads,
                       a.causal, a.scale, a.mask_neg, a.drop_seed, a.drop_site, L.thr, L.ks);
    return a4r_launch_status();
}
template <typename T, int DH, int WAVES>
int launch_bwd(const Launch& L) {
    using C = AttnCfg<T, DH>;
    const a4r_attn_t& a = *L.a;
    constexpr int LDS = WAVES * (3 * 32 * C::GSTRIDE + 3 * 32 * C::PSTRIDE);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<T, DH, WAVES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int total = a.n_items * a.n_heads;
    hipLaunchKernelGGL((attn_bwd_kernel<T, DH, WAVES>), dim3((total + WAVES - 1) / WAVES), dim3(WAVES * 64), LDS, L.s,
                       (const T*)a.qkv, a.ld, a.q_off, a.k_off, a.v_off, (const T*)a.dout, a.ldo, (T*)a.dqkv, a.key_mask,
                       a.n_items, a.S, a.n_heads, a.causal, a.scale, a.mask_neg, a.drop_seed, a.drop_site, L.thr, L.ks);
    return a4r_launch_status();
}

int check(const a4r_attn_t* a, bool bwd) {
    if (!a || !a->qkv) return A4R_EINVAL;
    if (bwd ? (!a->dout || !a->dqkv) : !a->out) return A4R_EINVAL;
    if (a->dtype != A4R_BF16 && a->dtype != A4R_F32) return A4R_EINVAL;
    const bool narrow = a->dh > 0 && a->dh <= 16;          // K-Adapter blocks: scalar kernels of a4r_attn_small.hip
    if (a->n_items <= 0 || a->n_heads <= 0 || a->S <= 0 || a->S > 32 || (a->dh != 32 && a->dh != 64 && !narrow)) return A4R_EINVAL;
    const int esz = a->dtype == A4R_F32 ? 4 : 2, per = narrow ? 1 : 16 / esz;
    if ((a->ld * esz) % 16 || (a->ldo * esz) % 16 || a->q_off % per || a->k_off % per || a->v_off % per) return A4R_EINVAL;
    if (a->ldo < a->n_heads * a->dh) return A4R_EINVAL;
    if (a->drop_p < 0.f || a->drop_p >= 1.f) return A4R_EINVAL;
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
    return a->dh == 64 ? launch_fwd<float, 64, 4>(L) : launch_fwd<float, 32, 4>(L);
}

extern "C" int a4r_attn_bwd(void* stream, const a4r_attn_t* a) {
    if (int e = check(a, true)) return e;
    if (a->dh <= 16) return a4r_attn_small(reinterpret_cast<hipStream_t>(stream), a, true);
    Launch L{reinterpret_cast<hipStream_t>(stream), a, a4r_thr16(a->drop_p), a4r_keep_scale(a->drop_p)};
    if (a->dtype == A4R_BF16) return a->dh == 64 ? launch_bwd<bf16_t, 64, 4>(L) : launch_bwd<bf16_t, 32, 4>(L);
    return a->dh == 64 ? launch_bwd<float, 64, 2>(L) : launch_bwd<float, 32, 4>(L);
}
