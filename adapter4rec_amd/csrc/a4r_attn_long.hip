// a4r_attn_long_fwd / a4r_attn_long_bwd: un-masked multi-head attention for 32 < S <= 256 tokens per item, head width 64 or 32
// -- the ViT / MAE item tower (S = 197 / 50, dh 64; HF ViTSelfAttention as called from Downstream/CV/model/encoders.py:21-32)
// and the two TransformerBlocks inside each KAdapterBlock of VITKAdaptedCVModel (Downstream/CV/model/model.py:374-404,
// modules.py:24-36,148-187: width 384, 12 heads of 32, all-ones mask, dropout on the probabilities).
// The S <= 32 kernels of a4r_attn.hip keep a whole score matrix in one wave; here one workgroup (8 waves) owns one
// (item, head) pair, stages the key-side matrices of that pair in LDS and never writes anything S x S to HBM.
//
// Everything is the 16x16 "chunk" primitive of a4r_common.h (so the bf16 and exact-fp32 instantiations share all
// addressing).  A score tile is produced TRANSPOSED (key-side fragment first): lane (c = l & 15, kg = l >> 4) then holds,
// for ONE query column c, the 4 consecutive keys 16*kt + 4*kg + r.  The whole key range of a 16-query block lives in
// registers (<= 16 tiles), so the softmax is plain register arithmetic + two cross-lane steps (l ^ 16, l ^ 32), and the
// probabilities feed the next product as an MFMA operand WITHOUT any re-layout by permuting the contraction index:
//     k-slot (kg, j) of chunk step st  <->  key  KSTEP*st + (j >> 2)*16 + 4*kg + (j & 3)
// (two score tiles per step for bf16, one for fp32).  The other operand of such a product needs, per lane, 4 + 4
// consecutive KEYS at one head column -- i.e. V (or K, Q, dO) read TRANSPOSED from its row-major LDS image: ds_read_b64_tr_b16 for
// bf16, four 4-byte gathers for fp32 (frag_T; no second, transposed image of anything).
//
// forward  (per 16-query block): S^T = K Q^T | softmax | O^T = V^T P^T         + lse = max + log(sum) per query (fp32)
// backward, two launches (FlashAttention-2 split, no atomics):
//   dq   (per 16-query block): delta = dO . O (forward output), then per chunk step of keys S^T, dP^T = V dO^T,
//        dS = P (dP - delta) scale, dQ^T += K^T dS^T  (nothing held for the whole key range)
//   dkdv (per 16-key tile, a wave owns its key tiles): S = Q K^T, dP = dO V^T (tiles with the KEY on the lane),
//        dV^T += dO^T P, dK^T += Q^T dS  over all query groups; lse and delta come from LDS.
// HBM-bound in principle (fwd: reads 3 M H, writes M H); measured numbers are in DESIGN.md.
#include <cstdlib>
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

template <typename T> A4R_DEV uint4 ldg16(const T* p) { return *reinterpret_cast<const uint4*>(p); }

// 8 waves per (item, head) workgroup: two workgroups (114 KB of LDS at S = 197) give a CU 4 waves per SIMD to hide the
// staging and Q / dO load latency behind; with 4-wave workgroups the chip sat at 0.4 waves per SIMD (PMC).
// (round 3: 7 waves for S = 197 -- its 13 query blocks / key tiles in two even rounds instead of 8 + 5 -- ran the ViT step 2 % SLOWER: 14 instead of
// 16 waves per CU hide less of the staging latency than the idle second round costs)
template <int NKT> struct WG { static constexpr int NWAVE = NKT <= 4 ? 4 : 8, NTHR = NWAVE * 64; };   // short sequences have <= 4 query blocks

template <typename T, int DH> struct Geo {
    static constexpr int PER = Elem<T>::PER16;              // elements per 16-byte chunk
    static constexpr int KSTEP = Mma<T>::KSTEP;             // contraction length of one chunk step (32 / 16)
    static constexpr int KS = DH / KSTEP;                   // chunk steps over the head width (dh 64: 2 / 4, dh 32: 1 / 2)
    static constexpr int CPR = DH / PER;                    // chunks per row of a [*, DH] matrix (16 / 8 / 4)
    static constexpr int ROWB = DH * (int)sizeof(T);        // row bytes (256 / 128 / 64)
    static constexpr int TPS = KSTEP / 16;                  // score tiles per chunk step (2 / 1)
    static constexpr int ND = DH / 16;                      // 16-column tiles of the head width
    // 16 rows x one 16-byte chunk column per quarter wave: rows that share a 256-byte bank window get distinct chunk slots
    static A4R_DEV int swz(int row) { return CPR == 16 ? (row & 15) : CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }
};

// LDS layout of a workgroup: the staged matrices (lds_main_*), then one output staging block per wave (bf16 only)
template <typename T, int DH> struct STG { static constexpr int BYTES = sizeof(T) == 2 ? 16 * DH * 2 : 0; };
template <typename T, int DH, int NKT> constexpr size_t img_t() { return 0; }   // (round 3: no transposed copy for fp32 either -- with it the fp32 backward needed 174 / 235 KB of LDS at S = 197 and could not run ViT-B/16)
template <typename T, int DH, int NKT> constexpr size_t lds_main_fwd() { return 2 * (size_t)NKT * 16 * Geo<T, DH>::ROWB; }
template <typename T, int DH, int NKT> constexpr size_t lds_main_dq() { return 2 * (size_t)NKT * 16 * Geo<T, DH>::ROWB + img_t<T, DH, NKT>(); }
template <typename T, int DH, int NKT> constexpr size_t lds_main_dkdv() { return 2 * (size_t)NKT * 16 * Geo<T, DH>::ROWB + 2 * img_t<T, DH, NKT>() + 2 * NKT * 16 * sizeof(float); }

// [S][DH] (global, row stride ld) -> LDS row-major [SP][DH], 16-byte chunks XOR-swizzled; rows >= S are zero
template <typename T, int DH> A4R_DEV void stage_rows(char* lds, const T* src, int ld, int S, int SP, int tid, int NTHR) {
    using G = Geo<T, DH>;
    for (int id = tid; id < SP * G::CPR; id += NTHR) {
        const int r = id / G::CPR, c = id % G::CPR;
        const uint4 v = r < S ? ldg16(src + (size_t)r * ld + c * G::PER) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(lds + r * G::ROWB + ((c ^ G::swz(r)) << 4)) = v;
    }
}
// operand chunk from a row-major image: row `row`, chunk step ks
template <typename T, int DH> A4R_DEV uint4 frag_rows(const char* lds, int row, int ks, int kg) {
    using G = Geo<T, DH>;
    return *reinterpret_cast<const uint4*>(lds + row * G::ROWB + (((ks * 4 + kg) ^ G::swz(row)) << 4));
}
// The transposed operand chunk (4 + 4 consecutive keys at one head column) for bf16 WITHOUT a transposed copy: ds_read_b64_tr_b16 gathers, per 16-lane group, a 4-row x 16-column
// block of a ROW-major image and hands lane i the block's column i (4 consecutive rows) -- exactly the 4 + 4 keys the permuted
// contraction index asks for.  Lane 4q + p of a group supplies the address of block row q, columns 4p .. 4p+3.  EXEC must be
// all ones (every call site sits in wave-uniform control flow).
typedef short v4s_t __attribute__((ext_vector_type(4)));
template <int DH> A4R_DEV uint4 frag_tr(const char* lds, int d0, int st, int lane) {
    using G = Geo<bf16_t, DH>;
    const int kg = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int chunk = (d0 >> 3) + (p >> 1);
    const int r0 = 32 * st + 4 * kg + q, r1 = r0 + 16;
    const char* a0 = lds + r0 * G::ROWB + ((chunk ^ G::swz(r0)) << 4) + 8 * (p & 1);
    const char* a1 = lds + r1 * G::ROWB + ((chunk ^ G::swz(r1)) << 4) + 8 * (p & 1);
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a0));
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a1));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}
// one accessor for both element types: bf16 reads the row-major image transposed, fp32 a transposed copy
// fp32 (the parity instantiation): the same 4 consecutive rows at one head column, gathered with four 4-byte reads from the ROW-major
// swizzled image (no transposed copy: two [64][S + 8] fp32 images next to the row-major ones did not fit the LDS at S = 197)
template <int DH> A4R_DEV uint4 frag_gather_f32(const char* lds, int d, int st, int kg) {
    using G = Geo<float, DH>;
    uint4 r;
    uint32_t* o = &r.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = G::KSTEP * st + 4 * kg + j;
        o[j] = *reinterpret_cast<const uint32_t*>(lds + row * G::ROWB + (((d >> 2) ^ G::swz(row)) << 4) + (d & 3) * 4);
    }
    return r;
}
// one accessor for both element types: both read the row-major image transposed
template <typename T, int DH> A4R_DEV uint4 frag_T(const void* img, int SPT, int d0, int st, int lane) {
    if constexpr (sizeof(T) == 2) return frag_tr<DH>(reinterpret_cast<const char*>(img), d0, st, lane);
    else return frag_gather_f32<DH>(reinterpret_cast<const char*>(img), d0 + (lane & 15), st, lane >> 4);
}
// dropout of the probabilities (SelfAttention.dropout, modules.py:35): element (pair = item * heads + head, query, key); the four
// consecutive keys of a transposed score tile share one hash
A4R_DEV uint64_t drop_idx(int pair, int q, int key) { return (((uint64_t)pair * 256 + q) << 8) + key; }

// probabilities / score gradients of chunk step st as an operand chunk (see the k-slot permutation in the header)
template <typename T, int NKT> A4R_DEV uint4 pack_step(const f32x4_t (&t)[NKT], int st) {
    if constexpr (sizeof(T) == 2) {
        const f32x4_t a = t[2 * st], b = t[2 * st + 1];
        // (scalar conversions here on purpose: with the vector form the 14-tile forward kernel needs 31 more registers than its
        // 128-register budget -- the scheduler hoists the conversions of all steps -- and spills: 158 -> 254 us)
        return make_uint4(f32_to_bf16_bits(a[0]) | (f32_to_bf16_bits(a[1]) << 16), f32_to_bf16_bits(a[2]) | (f32_to_bf16_bits(a[3]) << 16),
                          f32_to_bf16_bits(b[0]) | (f32_to_bf16_bits(b[1]) << 16), f32_to_bf16_bits(b[2]) | (f32_to_bf16_bits(b[3]) << 16));
    } else {
        const f32x4_t a = t[st];
        return make_uint4(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3]));
    }
}
// 4 consecutive head columns of one row, fp32 registers -> global
template <typename T> A4R_DEV void store4(T* p, const f32x4_t& v) {
    if constexpr (sizeof(T) == 2)
        *reinterpret_cast<uint2*>(p) = make_uint2(f32_to_bf16_bits(v[0]) | (f32_to_bf16_bits(v[1]) << 16),
                                                  f32_to_bf16_bits(v[2]) | (f32_to_bf16_bits(v[3]) << 16));
    else
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
A4R_DEV float red4(float v, bool mx) {       // over the 4 lanes l, l^16, l^32, l^48 that share a column
    const float a = __shfl_xor(v, 16, 64);
    v = mx ? fmaxf(v, a) : v + a;
    const float b = __shfl_xor(v, 32, 64);
    return mx ? fmaxf(v, b) : v + b;
}

// A wave's [16 tokens][DH] result block, held transposed in accumulators (lane (c, kg): token c, head columns dt * 16 + 4 kg .. + 3),
// leaves through a wave-private LDS block so that every global store is a whole 16-byte chunk and 8 lanes cover a 128-byte line
// (bf16; stored straight from the accumulators -- 8 bytes per lane, 32-byte pieces of 16 rows per instruction -- the short-sequence
// backward moved the same bytes 32 us slower, a4r_attn.hip).  fp32 accumulators are 16 bytes per lane already.
template <typename T, int DH>
A4R_DEV void store_block16(char* stg, const f32x4_t (&o)[Geo<T, DH>::ND], T* g0, size_t ldg, int rows_valid, int lane) {
    using G = Geo<T, DH>;
    const int fr = lane & 15, kg = lane >> 4;
    if constexpr (sizeof(T) == 4) {
        if (fr < rows_valid) {
#pragma unroll
            for (int dt = 0; dt < G::ND; ++dt) store4<T>(g0 + (size_t)fr * ldg + dt * 16 + kg * 4, o[dt]);
        }
    } else {
        constexpr int ROWB = DH * 2, CPR = DH / 8;
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt)
            store4<T>(reinterpret_cast<T*>(stg + fr * ROWB + (((dt * 2 + (kg >> 1)) ^ G::swz(fr)) << 4) + 8 * (kg & 1)), o[dt]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 16 * CPR / 64; ++i) {
            const int id = lane + 64 * i, row = id / CPR, ch = id % CPR;
            if (row < rows_valid)
                *reinterpret_cast<uint4*>(g0 + (size_t)row * ldg + ch * 8) = *reinterpret_cast<const uint4*>(stg + row * ROWB + ((ch ^ G::swz(row)) << 4));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// transposed RAW score tiles of one 16-query block: s[kt][r] = q[query c] . k[key 16 kt + 4 kg + r]  (keys >= S: -inf; the softmax
// scale is folded into the exponent by the caller).  Only tiles that reach past S carry the per-element select.
template <typename T, int DH, int NKT>
A4R_DEV void scores_t(const char* Kr, const uint4 (&qf)[Geo<T, DH>::KS], f32x4_t (&s)[NKT], int S, int fr, int kg) {
    using G = Geo<T, DH>;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) Mma<T>::mma(frag_rows<T, DH>(Kr, kt * 16 + fr, ks, kg), qf[ks], acc);
        if (kt * 16 + 16 > S) {                      // wave-uniform: the last one or two tiles
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = (kt * 16 + kg * 4 + r < S) ? acc[r] : -INFINITY;
        }
        s[kt] = acc;
        if ((kt & 1) == 1) __builtin_amdgcn_sched_barrier(0);      // keeps the scheduler from hoisting all 2 NKT fragment reads (spills)
    }
}

struct Drop { uint64_t seed; uint32_t site, thr16; float keep_scale; int abl; };     // abl: timing ablations (A4R_ATTN_LONG_ABL; wrong results): 1 no stores, 2 no per-block global loads, 4 no staging loads, 8 no softmax arithmetic

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int DH, int NKT>
__global__ void __launch_bounds__(WG<NKT>::NTHR, 4) attn_long_fwd_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                            T* __restrict__ ctx, int ldo, float* __restrict__ lse,
                                                            int S, int nh, float scale, Drop dr) {
    using G = Geo<T, DH>;
    constexpr int NTHR = WG<NKT>::NTHR, NWAVE = WG<NKT>::NWAVE;
    constexpr int SP = NKT * 16, SPT = SP + 8, NST = SP / G::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kr = smem;                                                  // [SP][DH] row-major
    char* Vimg = smem + SP * G::ROWB;                                 // [SP][DH] row-major (read transposed: ds_read_b64_tr_b16 / 4-byte gathers)
    const int item = blockIdx.x / nh, h = blockIdx.x % nh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kg = lane >> 4;
    char* stg = smem + lds_main_fwd<T, DH, NKT>() + wave * STG<T, DH>::BYTES;      // this wave's output staging block (bf16)
    const T* base = qkv + (size_t)item * S * ld + h * DH;
    if (!(dr.abl & 4)) {
        stage_rows<T, DH>(Kr, base + k_off, ld, S, SP, tid, NTHR);
        stage_rows<T, DH>(Vimg, base + v_off, ld, S, SP, tid, NTHR);
    }
    __syncthreads();
    const int nqb = (S + 15) >> 4;
    for (int qb = wave; qb < nqb; qb += NWAVE) {
        const int rq = qb * 16 + fr;
        const bool valid = rq < S;
        uint4 qf[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks)
            qf[ks] = (valid && !(dr.abl & 2)) ? ldg16(base + q_off + (size_t)rq * ld + (ks * 4 + kg) * G::PER) : make_uint4(0, 0, 0, 0);
        f32x4_t s[NKT];
        scores_t<T, DH, NKT>(Kr, qf, s, S, fr, kg);
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kt][r]);
        m = red4(m, true);                                   // max of the RAW scores (scale > 0)
        // p = exp(scale (s - m)) = exp2(s c - m c), c = scale log2(e): one fma + v_exp_f32 per element; the 1 / row-sum factor is
        // applied to the 16 output values of the lane instead of its 4 NKT probabilities
        const float c2 = scale * 1.44269504088896f, mc = m * c2;
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) { if (!(dr.abl & 8)) s[kt][r] = __builtin_amdgcn_exp2f(fmaf(s[kt][r], c2, -mc)); l += s[kt][r]; }
        l = red4(l, false);
        float inv = 1.f / l;
        if (dr.thr16) {                                       // P' = dropout(P): the 4 keys of a lane's tile column share one hash
            inv *= dr.keep_scale;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const uint64_t hsh = a4r_hash64(dr.seed, dr.site, drop_idx(blockIdx.x, rq, kt * 16 + kg * 4) >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] = (((uint32_t)(hsh >> (16 * r)) & 0xffffu) >= dr.thr16) ? s[kt][r] : 0.f;
            }
        }
        if (valid && kg == 0) lse[((size_t)item * nh + h) * S + rq] = m * scale + __logf(l);
        f32x4_t o[G::ND];
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NST; ++st) {                   // probabilities are packed step by step (keeping all NST chunks spilled)
            const uint4 pf = pack_step<T, NKT>(s, st);
#pragma unroll
            for (int dt = 0; dt < G::ND; ++dt) Mma<T>::mma(frag_T<T, DH>(Vimg, SPT, dt * 16, st, lane), pf, o[dt]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= inv;
        if (!(dr.abl & 1)) store_block16<T, DH>(stg, o, ctx + ((size_t)item * S + qb * 16) * ldo + h * DH, ldo, S - qb * 16, lane);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dq + delta
template <typename T, int DH, int NKT>
__global__ void __launch_bounds__(WG<NKT>::NTHR, 4) attn_long_dq_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                           const T* __restrict__ dctx, int ldo, const T* __restrict__ octx,
                                                           const float* __restrict__ lse, float* __restrict__ delta,
                                                           T* __restrict__ dqkv, int S, int nh, float scale, Drop dr) {
    using G = Geo<T, DH>;
    constexpr int NTHR = WG<NKT>::NTHR, NWAVE = WG<NKT>::NWAVE;
    constexpr int SP = NKT * 16, SPT = SP + 8, NST = SP / G::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kr = smem;
    char* Vr = smem + SP * G::ROWB;
    char* Kimg = Kr;                                                  // K's row-major image doubles as the transposed operand
    const int item = blockIdx.x / nh, h = blockIdx.x % nh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kg = lane >> 4;
    char* stg = smem + lds_main_dq<T, DH, NKT>() + wave * STG<T, DH>::BYTES;
    const T* base = qkv + (size_t)item * S * ld + h * DH;
    stage_rows<T, DH>(Kr, base + k_off, ld, S, SP, tid, NTHR);
    stage_rows<T, DH>(Vr, base + v_off, ld, S, SP, tid, NTHR);
    __syncthreads();
    const int nqb = (S + 15) >> 4;
    for (int qb = wave; qb < nqb; qb += NWAVE) {
        const int rq = qb * 16 + fr;
        const bool valid = rq < S;
        const size_t grow = (size_t)item * S + rq;
        uint4 qf[G::KS], dof[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            qf[ks] = valid ? ldg16(base + q_off + (size_t)rq * ld + (ks * 4 + kg) * G::PER) : make_uint4(0, 0, 0, 0);
            dof[ks] = valid ? ldg16(dctx + grow * ldo + h * DH + (ks * 4 + kg) * G::PER) : make_uint4(0, 0, 0, 0);
        }
        const float c2 = scale * 1.44269504088896f;
        const float lq2 = (valid ? lse[((size_t)item * nh + h) * S + rq] : 0.f) * 1.44269504088896f;
        // delta = sum_k P' dP' = dO . O (the forward output, dropout included): one dot product per query instead of a pass over all
        // key tiles, so P, dP and dS are produced and consumed one chunk step (32 / 16 keys) at a time and never held for the whole row
        float dsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            const uint4 of = valid ? ldg16(octx + grow * ldo + h * DH + (ks * 4 + kg) * G::PER) : make_uint4(0, 0, 0, 0);
            float a8[G::PER], b8[G::PER];
            Elem<T>::unpack(of, a8);
            Elem<T>::unpack(dof[ks], b8);
#pragma unroll
            for (int e = 0; e < G::PER; ++e) dsum += a8[e] * b8[e];
        }
        dsum = red4(dsum, false);
        if (valid && kg == 0) delta[((size_t)item * nh + h) * S + rq] = dsum;
        f32x4_t o[G::ND];
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            f32x4_t ds[G::TPS];
#pragma unroll
            for (int t = 0; t < G::TPS; ++t) {
                const int kt = st * G::TPS + t;
                f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    Mma<T>::mma(frag_rows<T, DH>(Kr, kt * 16 + fr, ks, kg), qf[ks], sc);
                    Mma<T>::mma(frag_rows<T, DH>(Vr, kt * 16 + fr, ks, kg), dof[ks], dp);
                }
                uint64_t hsh = 0;
                if (dr.thr16) hsh = a4r_hash64(dr.seed, dr.site, drop_idx(blockIdx.x, rq, kt * 16 + kg * 4) >> 2);
                const bool partial = kt * 16 + 16 > S;                    // wave-uniform
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = __builtin_amdgcn_exp2f(fmaf(sc[r], c2, -lq2));          // P = exp(scale s - lse)
                    if (partial) pv = (kt * 16 + kg * 4 + r < S) ? pv : 0.f;
                    float dpr = dp[r];
                    if (dr.thr16) dpr = (((uint32_t)(hsh >> (16 * r)) & 0xffffu) >= dr.thr16) ? dpr * dr.keep_scale : 0.f;
                    ds[t][r] = pv * (dpr - dsum);                                      // (x scale: applied to the 16 outputs of the lane)
                }
            }
            const uint4 dsf = pack_step<T, G::TPS>(ds, 0);
#pragma unroll
            for (int dt = 0; dt < G::ND; ++dt) Mma<T>::mma(frag_T<T, DH>(Kimg, SPT, dt * 16, st, lane), dsf, o[dt]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[dt][r] *= scale;
        store_block16<T, DH>(stg, o, dqkv + ((size_t)item * S + qb * 16) * ld + q_off + h * DH, ld, S - qb * 16, lane);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dk, dv
template <typename T, int DH, int NKT>
__global__ void __launch_bounds__(WG<NKT>::NTHR, 4) attn_long_dkdv_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                             const T* __restrict__ dctx, int ldo, const float* __restrict__ lse,
                                                             const float* __restrict__ delta, T* __restrict__ dqkv,
                                                             int S, int nh, float scale, Drop dr) {
    using G = Geo<T, DH>;
    constexpr int NTHR = WG<NKT>::NTHR, NWAVE = WG<NKT>::NWAVE;
    constexpr int SP = NKT * 16, SPT = SP + 8, NG = SP / G::KSTEP;          // NG query groups of KSTEP queries
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qr = smem;
    char* Or = smem + SP * G::ROWB;
    char* Qimg = Qr;
    char* Oimg = Or;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * SP * G::ROWB);
    float* del_s = lse_s + SP;
    const int item = blockIdx.x / nh, h = blockIdx.x % nh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kg = lane >> 4;
    char* stg = smem + lds_main_dkdv<T, DH, NKT>() + wave * STG<T, DH>::BYTES;
    const T* base = qkv + (size_t)item * S * ld + h * DH;
    const T* dob = dctx + (size_t)item * S * ldo + h * DH;
    stage_rows<T, DH>(Qr, base + q_off, ld, S, SP, tid, NTHR);
    stage_rows<T, DH>(Or, dob, ldo, S, SP, tid, NTHR);
    for (int i = tid; i < SP; i += NTHR) {
        lse_s[i] = i < S ? lse[((size_t)item * nh + h) * S + i] * 1.44269504088896f : 0.f;
        del_s[i] = i < S ? delta[((size_t)item * nh + h) * S + i] : 0.f;
    }
    __syncthreads();
    const int nkt = (S + 15) >> 4;
    const float c2 = scale * 1.44269504088896f;
    for (int kt = wave; kt < nkt; kt += NWAVE) {
        const int rk = kt * 16 + fr;                          // this lane's key (column of every tile below)
        const bool kvalid = rk < S;
        uint4 kf[G::KS], vf[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            kf[ks] = kvalid ? ldg16(base + k_off + (size_t)rk * ld + (ks * 4 + kg) * G::PER) : make_uint4(0, 0, 0, 0);
            vf[ks] = kvalid ? ldg16(base + v_off + (size_t)rk * ld + (ks * 4 + kg) * G::PER) : make_uint4(0, 0, 0, 0);
        }
        f32x4_t dk[G::ND], dv[G::ND];
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) { dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            f32x4_t p[G::TPS], ds[G::TPS];
#pragma unroll
            for (int t = 0; t < G::TPS; ++t) {
                const int q0 = g * G::KSTEP + t * 16;         // tile rows = queries q0 + 4 kg + r, column = key rk
                f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    Mma<T>::mma(frag_rows<T, DH>(Qr, q0 + fr, ks, kg), kf[ks], sc);
                    Mma<T>::mma(frag_rows<T, DH>(Or, q0 + fr, ks, kg), vf[ks], dpt);
                }
                const bool partial = q0 + 16 > S;                          // wave-uniform: query rows past S (lse_s = 0 there)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q0 + kg * 4 + r;
                    float pv = __builtin_amdgcn_exp2f(fmaf(sc[r], c2, -lse_s[q]));     // lse_s holds lse log2(e)
                    if (partial) pv = q < S ? pv : 0.f;
                    if (!kvalid) pv = 0.f;
                    float keepf = 1.f;                        // here the tile's 4 rows are 4 QUERIES at one key: one hash each
                    if (dr.thr16) keepf = dropout_keep(dr.seed, dr.site, drop_idx(blockIdx.x, q, rk), dr.thr16) ? dr.keep_scale : 0.f;
                    p[t][r] = pv * keepf;
                    ds[t][r] = pv * (dpt[r] * keepf - del_s[q]);                       // (x scale: applied to dK at the end)
                }
            }
            const uint4 pf = pack_step<T, G::TPS>(p, 0), dsf = pack_step<T, G::TPS>(ds, 0);
#pragma unroll
            for (int dt = 0; dt < G::ND; ++dt) {
                Mma<T>::mma(frag_T<T, DH>(Oimg, SPT, dt * 16, g, lane), pf, dv[dt]);
                Mma<T>::mma(frag_T<T, DH>(Qimg, SPT, dt * 16, g, lane), dsf, dk[dt]);
            }
        }
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dk[dt][r] *= scale;
        T* blk = dqkv + ((size_t)item * S + kt * 16) * ld + h * DH;
        store_block16<T, DH>(stg, dk, blk + k_off, ld, S - kt * 16, lane);
        store_block16<T, DH>(stg, dv, blk + v_off, ld, S - kt * 16, lane);
    }
}

int nkt_for(int S) { return S <= 32 ? 2 : S <= 64 ? 4 : S <= 128 ? 8 : S <= 224 ? 14 : 16; }

template <typename T, int DH, int NKT> size_t lds_fwd() { return lds_main_fwd<T, DH, NKT>() + WG<NKT>::NWAVE * STG<T, DH>::BYTES; }
template <typename T, int DH, int NKT> size_t lds_dq() { return lds_main_dq<T, DH, NKT>() + WG<NKT>::NWAVE * STG<T, DH>::BYTES; }
template <typename T, int DH, int NKT> size_t lds_dkdv() { return lds_main_dkdv<T, DH, NKT>() + WG<NKT>::NWAVE * STG<T, DH>::BYTES; }
constexpr size_t LDS_MAX = 160 * 1024;

template <typename K> int set_lds(K kernel, size_t bytes) {
    if (bytes > LDS_MAX) return A4R_EINVAL;
    if (bytes > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return A4R_ELAUNCH;
    return A4R_OK;
}

Drop drop_of(const a4r_attn_t* a) {
    static const int abl = getenv("A4R_ATTN_LONG_ABL") ? atoi(getenv("A4R_ATTN_LONG_ABL")) : 0;
    return Drop{a->drop_seed, a->drop_site, a4r_thr16(a->drop_p), a4r_keep_scale(a->drop_p), abl};
}

template <typename T, int DH, int NKT> int run_fwd(hipStream_t s, const a4r_attn_t* a, float* lse) {
    const size_t lds = lds_fwd<T, DH, NKT>();
    if (int rc = set_lds(attn_long_fwd_kernel<T, DH, NKT>, lds)) return rc;
    hipLaunchKernelGGL((attn_long_fwd_kernel<T, DH, NKT>), dim3(a->n_items * a->n_heads), dim3(WG<NKT>::NTHR), lds, s, (const T*)a->qkv, a->ld, a->q_off,
                       a->k_off, a->v_off, (T*)a->out, a->ldo, lse, a->S, a->n_heads, a->scale, drop_of(a));
    return a4r_launch_status();
}
template <typename T, int DH, int NKT> int run_bwd(hipStream_t s, const a4r_attn_t* a, const float* lse, float* delta) {
    const size_t l1 = lds_dq<T, DH, NKT>(), l2 = lds_dkdv<T, DH, NKT>();
    if (int rc = set_lds(attn_long_dq_kernel<T, DH, NKT>, l1)) return rc;
    if (int rc = set_lds(attn_long_dkdv_kernel<T, DH, NKT>, l2)) return rc;
    const dim3 grid(a->n_items * a->n_heads), block(WG<NKT>::NTHR);
    hipLaunchKernelGGL((attn_long_dq_kernel<T, DH, NKT>), grid, block, l1, s, (const T*)a->qkv, a->ld, a->q_off, a->k_off, a->v_off,
                       (const T*)a->dout, a->ldo, (const T*)a->out, lse, delta, (T*)a->dqkv, a->S, a->n_heads, a->scale, drop_of(a));
    hipLaunchKernelGGL((attn_long_dkdv_kernel<T, DH, NKT>), grid, block, l2, s, (const T*)a->qkv, a->ld, a->q_off, a->k_off, a->v_off,
                       (const T*)a->dout, a->ldo, lse, (const float*)delta, (T*)a->dqkv, a->S, a->n_heads, a->scale, drop_of(a));
    return a4r_launch_status();
}

int check(const a4r_attn_t* a, bool bwd) {
    if (!a || !a->qkv || a->n_items <= 0 || a->S <= 0 || a->S > 256 || (a->dh != 64 && a->dh != 32) || a->n_heads <= 0) return A4R_EINVAL;
    if (a->key_mask || a->causal) return A4R_EINVAL;                               // neither the ViT / MAE tower nor its K-Adapter blocks mask
    if (a->drop_p < 0.f || a->drop_p >= 1.f) return A4R_EINVAL;
    if ((int64_t)a->n_items * a->n_heads >= (1ll << 40)) return A4R_EINVAL;        // dropout counter: 40 + 8 + 8 bits
    if (a->dtype != A4R_BF16 && a->dtype != A4R_F32) return A4R_EINVAL;
    const int es = a->dtype == A4R_BF16 ? 2 : 4;
    if ((a->ld * es) % 16 || (a->ldo * es) % 16 || (a->q_off * es) % 16 || (a->k_off * es) % 16 || (a->v_off * es) % 16) return A4R_EINVAL;
    if (reinterpret_cast<uintptr_t>(a->qkv) & 15u) return A4R_EINVAL;
    if (!bwd && (!a->out || (reinterpret_cast<uintptr_t>(a->out) & 15u))) return A4R_EINVAL;
    if (bwd && (!a->dout || !a->dqkv || !a->out || ((reinterpret_cast<uintptr_t>(a->dout) | reinterpret_cast<uintptr_t>(a->dqkv) | reinterpret_cast<uintptr_t>(a->out)) & 15u)))
        return A4R_EINVAL;                                                         // bwd reads the forward output too (a->out, same ldo)
    return A4R_OK;
}

#define A4R_NKT_SWITCH(T_, D_, CALL_)                   \
    switch (nkt_for(a->S)) {                            \
        case 2: return CALL_(T_, D_, 2);                \
        case 4: return CALL_(T_, D_, 4);                \
        case 8: return CALL_(T_, D_, 8);                \
        case 14: return CALL_(T_, D_, 14);              \
        default: return CALL_(T_, D_, 16);              \
    }
#define A4R_DH_SWITCH(T_, CALL_)                                            \
    if (a->dh == 64) { A4R_NKT_SWITCH(T_, 64, CALL_) } else { A4R_NKT_SWITCH(T_, 32, CALL_) }

}  // namespace

extern "C" int a4r_attn_long_fwd(void* stream, const a4r_attn_t* a, float* lse) {
    if (int rc = check(a, false)) return rc;
    if (!lse) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define A4R_F(T_, D_, N_) run_fwd<T_, D_, N_>(s, a, lse)
    if (a->dtype == A4R_BF16) { A4R_DH_SWITCH(bf16_t, A4R_F) }
    A4R_DH_SWITCH(float, A4R_F)
#undef A4R_F
}

extern "C" int a4r_attn_long_bwd(void* stream, const a4r_attn_t* a, const float* lse, float* delta_ws) {
    if (int rc = check(a, true)) return rc;
    if (!lse || !delta_ws) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define A4R_B(T_, D_, N_) run_bwd<T_, D_, N_>(s, a, lse, delta_ws)
    if (a->dtype == A4R_BF16) { A4R_DH_SWITCH(bf16_t, A4R_B) }
    A4R_DH_SWITCH(float, A4R_B)
#undef A4R_B
}
