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
// backward, two kernels (FlashAttention-2 split, no atomics) -- as two launches above 64 tokens, as ONE launch (attn_long_bwd_kernel: dq half, barrier, dkdv half in the
// same workgroup and LDS) up to 64, where the second half then finds the first half's reads in the caches:
//   dq   (per 16-query block): delta = dO . O (forward output), then per chunk step of keys S^T, dP^T = V dO^T,
//        dS = P (dP - delta) scale, dQ^T += K^T dS^T  (nothing held for the whole key range)
//   dkdv (per 16-key tile, a wave owns its key tiles): S = Q K^T, dP = dO V^T (tiles with the KEY on the lane),
//        dV^T += dO^T P, dK^T += Q^T dS  over all query groups; lse and delta come from LDS.
// HBM-bound in principle (fwd: reads 3 M H, writes M H); measured numbers are in DESIGN.md.
#include <cstdlib>
#include "a4r_common.h"
#include "../../include/a4r.h"

#ifndef A4R_ATTN_W14
#define A4R_ATTN_W14 8
#endif
namespace {

template <typename T> A4R_DEV uint4 ldg16(const T* p) { return *reinterpret_cast<const uint4*>(p); }
// The [S][DH] slice of one (item, head) as a raw buffer: 16-byte loads at a 32-bit byte offset (+ a scalar offset), and every access that starts past
// the slice's last row returns ZERO in hardware -- the rows >= S of the staged images and of the last query / key block need no branch, no select and
// no 64-bit address arithmetic (round 4: those were ~8 vector instructions and an exec-mask branch per 16-byte load).
typedef unsigned int u32x4_raw_t __attribute__((ext_vector_type(4)));
struct RowsView {
    __amdgpu_buffer_rsrc_t r;
    uint32_t ldb;                                            // row stride in bytes
    template <typename T> static A4R_DEV RowsView make(const T* p, int ld, int S, int DH) {
        const uint32_t ldb = (uint32_t)ld * (uint32_t)sizeof(T);
        return RowsView{__builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p), 0, (int)((uint32_t)(S - 1) * ldb + (uint32_t)DH * (uint32_t)sizeof(T)), 0x27000), ldb};
    }
    A4R_DEV uint4 load(uint32_t voff, uint32_t soff = 0) const {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
    }
    A4R_DEV uint4 row_chunk(int row, int chunk) const { return load((uint32_t)row * ldb + (uint32_t)chunk * 16u); }
    A4R_DEV void store(uint32_t voff, const uint4& v) const {                        // (a store that starts past the last row is dropped)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_raw_t, v), r, (int)voff, 0, 0);
    }
    A4R_DEV void store8(uint32_t voff, const uint2& v) const {
        typedef unsigned int u32x2_raw_t __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_raw_t, v), r, (int)voff, 0, 0);
    }
};
// the lane index as a value the compiler cannot see through: the dropout counters are rebuilt from it INSIDE the (wave-uniform) dropout branch, so their
// loop-invariant parts are not hoisted into registers held across the key loops of runs without dropout (round 4: those registers were the spills)
A4R_DEV int opaque_lane(int lane) { asm volatile("" : "+v"(lane)); return lane; }

// 8 waves per (item, head) workgroup: two workgroups (114 KB of LDS at S = 197) give a CU 4 waves per SIMD to hide the
// staging and Q / dO load latency behind; with 4-wave workgroups the chip sat at 0.4 waves per SIMD (PMC).
// (round 3: 7 waves for S = 197 -- its 13 query blocks / key tiles in two even rounds instead of 8 + 5 -- ran the ViT step 2 % SLOWER: 14 instead of
// 16 waves per CU hide less of the staging latency than the idle second round costs)
template <int NKT> struct WG { static constexpr int NWAVE = NKT <= 4 ? 4 : (NKT == 14 ? A4R_ATTN_W14 : 8), NTHR = NWAVE * 64; };   // short sequences have <= 4 query blocks

template <typename T, int DH> struct Geo {
    static constexpr int PER = Elem<T>::PER16;              // elements per 16-byte chunk
    static constexpr int KSTEP = Mma<T>::KSTEP;             // contraction length of one chunk step (32 / 16)
    static constexpr int KS = DH / KSTEP;                   // chunk steps over the head width (dh 64: 2 / 4, dh 32: 1 / 2)
    static constexpr int CPR = DH / PER;                    // chunks per row of a [*, DH] matrix (16 / 8 / 4)
    static constexpr int ROWB = DH * (int)sizeof(T);        // row bytes (256 / 128 / 64)
    static constexpr int TPS = KSTEP / 16;                  // score tiles per chunk step (2 / 1)
    static constexpr int ND = DH / 16;                      // 16-column tiles of the head width
    // 16 rows x one 16-byte chunk column per quarter wave: rows that share a 256-byte bank window get distinct chunk slots
    static A4R_DEV int swz(int row) { return CPR >= 16 ? (row & 15) : CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); }      // (CPR 32: fp32 rows of 128 columns, two bank windows per row)
};

// LDS layout of a workgroup: the staged matrices (lds_main_*), then one output staging block per wave (bf16 only)
template <typename T, int DH> struct STG { static constexpr int BYTES = sizeof(T) == 2 ? 16 * DH * 2 : 0; };
template <typename T, int DH, int NKT> constexpr size_t img_t() { return 0; }   // (round 3: no transposed copy for fp32 either -- with it the fp32 backward needed 174 / 235 KB of LDS at S = 197 and could not run ViT-B/16)
// (+ NKT * 16 floats at the end of every main area: the item's key mask, staged per workgroup -- text towers with --num_words_title > 32, round 5)
template <int NKT> constexpr size_t lds_km() { return (size_t)NKT * 16 * sizeof(float); }
template <typename T, int DH, int NKT> constexpr size_t lds_main_fwd() { return 2 * (size_t)NKT * 16 * Geo<T, DH>::ROWB + lds_km<NKT>(); }
template <typename T, int DH, int NKT> constexpr size_t lds_main_dq() { return 2 * (size_t)NKT * 16 * Geo<T, DH>::ROWB + img_t<T, DH, NKT>() + lds_km<NKT>(); }
template <typename T, int DH, int NKT> constexpr size_t lds_main_dkdv() { return 2 * (size_t)NKT * 16 * Geo<T, DH>::ROWB + 2 * img_t<T, DH, NKT>() + 2 * NKT * 16 * sizeof(float) + lds_km<NKT>(); }

// [S][DH] (global, row stride ld) -> LDS row-major [SP][DH], 16-byte chunks XOR-swizzled; rows >= S are zero.  Two phases, so that a workgroup has
// the rows of BOTH its staged matrices in flight before the first LDS write (round 4: the one-call form compiled to a rolled loop of
// load -> s_waitcnt vmcnt(0) -> ds_write, 4 + 4 dependent HBM round trips before the workgroup's first product).
template <typename T, int DH, int SP, int NTHR> struct Stager {
    using G = Geo<T, DH>;
    static constexpr int TOTAL = SP * G::CPR, NIT = (TOTAL + NTHR - 1) / NTHR;
    uint4 v[NIT];
    A4R_DEV void request(const RowsView& src, int tid) {
        static_assert(NTHR % G::CPR == 0, "a thread keeps its chunk column");
        const uint32_t voff = (uint32_t)(tid / G::CPR) * src.ldb + (uint32_t)(tid % G::CPR) * 16u;
#pragma unroll
        for (int it = 0; it < NIT; ++it) v[it] = src.load(voff, (uint32_t)(it * (NTHR / G::CPR)) * src.ldb);      // (ids >= TOTAL are rows >= SP >= S: zero)
    }
    A4R_DEV void commit(char* lds, int tid) const {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = tid + it * NTHR, r = id / G::CPR, c = id % G::CPR;
            if (id < TOTAL) *reinterpret_cast<uint4*>(lds + r * G::ROWB + ((c ^ G::swz(r)) << 4)) = v[it];
        }
    }
};
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
// (round 4: ONE v_cvt_pk_bf16_f32 per pair -- the scalar form cost 4 vector instructions per pair (two conversions, shift, or); the
// hoisting that made the vector form spill in round 2 is held back by the callers' per-step scheduling barriers)
template <typename T, int NKT> A4R_DEV uint4 pack_step(const f32x4_t (&t)[NKT], int st) {
    if constexpr (sizeof(T) == 2) {
        const f32x4_t a = t[2 * st], b = t[2 * st + 1];
        return make_uint4(pack2_bf16(a[0], a[1]), pack2_bf16(a[2], a[3]), pack2_bf16(b[0], b[1]), pack2_bf16(b[2], b[3]));
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
A4R_DEV void store_block16(char* stg, const f32x4_t (&o)[Geo<T, DH>::ND], const RowsView& dst, int row0, int lane_) {
    using G = Geo<T, DH>;
    const int lane = opaque_lane(lane_);                     // the addresses below are rebuilt here, not kept in registers across the caller's loops
    const int fr = lane & 15, kg = lane >> 4;
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt)
            dst.store((uint32_t)(row0 + fr) * dst.ldb + (uint32_t)(dt * 16 + kg * 4) * 4u,
                      make_uint4(__float_as_uint(o[dt][0]), __float_as_uint(o[dt][1]), __float_as_uint(o[dt][2]), __float_as_uint(o[dt][3])));
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
            dst.store((uint32_t)(row0 + row) * dst.ldb + (uint32_t)ch * 16u, *reinterpret_cast<const uint4*>(stg + row * ROWB + ((ch ^ G::swz(row)) << 4)));
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// first key tile that can reach past S in the NKT instantiation (nkt_for: S > 16 * the previous instantiation's tiles)
template <int NKT> constexpr int kt_partial_lo() { return NKT == 2 ? 0 : NKT == 4 ? 2 : NKT == 8 ? 4 : NKT == 14 ? 8 : 14; }

// -inf on the keys >= S of the score tiles (transposed: lane (c, kg) holds keys 16 kt + 4 kg + r).  Only the tiles that CAN be partial in this
// instantiation are looked at, each behind a real wave-uniform branch (round 4: the per-tile test inside the product loop had been if-converted into
// two selects per element on every tile, their 56 lane masks spilled to a VGPR and read back with v_readlane -- 16 vector instructions per tile).
template <int NKT> A4R_DEV void mask_keys(f32x4_t (&s)[NKT], int S, int kg, float fill) {
    const int lim = S - kg * 4;
#pragma unroll
    for (int kt = kt_partial_lo<NKT>(); kt < NKT; ++kt) {
        if (kt * 16 + 16 > S) {
            asm volatile("" ::: "memory");                   // keeps the branch a branch
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] = (kt * 16 + r < lim) ? s[kt][r] : fill;
        }
    }
}

// transposed RAW score tiles of one 16-query block: s[kt][r] = q[query c] . k[key 16 kt + 4 kg + r]  (keys >= S: -inf; the softmax
// scale is folded into the exponent by the caller).  The key-side fragments of the next tile group are requested before the products of
// the current one (one group = 2 tiles for bf16: two independent accumulator chains), a scheduling barrier per group keeps the order.
template <typename T, int DH, int NKT>
A4R_DEV void scores_t(const char* Kr, const uint4 (&qf)[Geo<T, DH>::KS], f32x4_t (&s)[NKT], int S, int fr, int kg) {
    using G = Geo<T, DH>;
    constexpr int TPG = sizeof(T) == 2 ? 2 : 1, NGRP = NKT / TPG;
    static_assert(NKT % TPG == 0, "tile groups");
    constexpr bool PF = sizeof(T) == 2;                      // (the fp32 parity instantiation has no registers for a second fragment set)
    uint4 kf[PF ? 2 : 1][TPG][G::KS];
    if constexpr (PF) {
#pragma unroll
        for (int t = 0; t < TPG; ++t)
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) kf[0][t][ks] = frag_rows<T, DH>(Kr, t * 16 + fr, ks, kg);
    }
#pragma unroll
    for (int g = 0; g < NGRP; ++g) {
        const int gl = PF ? g + 1 : g;                       // the group whose fragments are requested in this region
        if (gl < NGRP) {
#pragma unroll
            for (int t = 0; t < TPG; ++t)
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) kf[PF ? (gl & 1) : 0][t][ks] = frag_rows<T, DH>(Kr, (gl * TPG + t) * 16 + fr, ks, kg);
        }
        f32x4_t acc[TPG];
#pragma unroll
        for (int t = 0; t < TPG; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
            for (int t = 0; t < TPG; ++t) Mma<T>::mma(kf[PF ? (g & 1) : 0][t][ks], qf[ks], acc[t]);
#pragma unroll
        for (int t = 0; t < TPG; ++t) s[g * TPG + t] = acc[t];
        __builtin_amdgcn_sched_barrier(0);
    }
    mask_keys<NKT>(s, S, kg, -INFINITY);
}

// The item's key mask (HF attention_mask: 1 = attend; Downstream/Text/model/encoders.py:48-57), staged to LDS by the workgroup.  Masked keys get
// probability 0 (HF adds finfo.min to their scores); an item with NO attended key -- the pad item -- attends uniformly over its S keys, as HF's
// softmax over S equal scores does: its raw scores are replaced by 0 in the forward and in both backward kernels, so lse = log S recomputes P = 1 / S.
template <int SP, int NTHR> A4R_DEV void stage_key_mask(float* km, const float* kmask, int item, int S, int tid) {
    for (int i = tid; i < SP; i += NTHR) km[i] = i < S ? kmask[(size_t)item * S + i] : 0.f;
}
// index of the item's first attended key (SP when it has none), the same value in every lane
template <int SP> A4R_DEV int first_attended(const float* km, int lane) {
    int f = SP;
    for (int i = lane; i < SP; i += 64) f = (km[i] != 0.f && i < f) ? i : f;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = __shfl_xor(f, d, 64); f = o < f ? o : f; }
    return f;
}
// CAUSAL (round 5, with a key mask only: the user tower at --max_seq_len > 32, model/modules.py:31-42 with the mask of model/encoders.py:24-28): key k
// is allowed for query q when the mask has it AND k <= q.  A query row without any allowed key (the left-padded positions of a short history: q below
// the first attended key; without causal: an item without attended keys) attends uniformly over the S keys in the forward -- the reference adds -1e9
// to every score of such a row, which leaves them equal in fp32 --; both backward kernels treat such a row the same way (raw scores 0, P = 1 / S over all S
// keys: the softmax Jacobian autograd applies to it).  In the model its output gradient is zero behind the loss mask, so nothing flows from it.

struct Drop { uint64_t seed; uint32_t site, thr16; float keep_scale; };

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int DH, int NKT, bool KM = false>      // KM: the launch carries a key mask (text towers; instantiated for head width 64)
__global__ void __launch_bounds__(WG<NKT>::NTHR, DH > 64 ? 2 : 4) attn_long_fwd_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                            T* __restrict__ ctx, int ldo, float* __restrict__ lse,
                                                            int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
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
    const int nqb = (S + 15) >> 4;
    // the query rows of a block are requested one block ahead: the first before the key side is staged (its latency hides behind the staging's),
    // the next at the top of the current block's arithmetic
    uint4 qn[G::KS];
    const RowsView qview = RowsView::make(base + q_off, ld, S, DH), cview = RowsView::make(ctx + (size_t)item * S * ldo + h * DH, ldo, S, DH);
    auto request_q = [&](int qb) {
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) qn[ks] = qview.row_chunk(qb * 16 + fr, ks * 4 + kg);
    };
    if (wave < nqb) request_q(wave);
    {
        Stager<T, DH, SP, NTHR> sk, sv;
        sk.request(RowsView::make(base + k_off, ld, S, DH), tid);
        sv.request(RowsView::make(base + v_off, ld, S, DH), tid);
        sk.commit(Kr, tid);
        sv.commit(Vimg, tid);
    }
    [[maybe_unused]] float* km = reinterpret_cast<float*>(smem + lds_main_fwd<T, DH, NKT>() - lds_km<NKT>());
    if constexpr (KM) stage_key_mask<SP, NTHR>(km, kmask, item, S, tid);
    __syncthreads();
    int first = 0;
    if constexpr (KM) first = first_attended<SP>(km, lane);
    for (int qb = wave; qb < nqb; qb += NWAVE) {
        const int rq = qb * 16 + fr;
        const bool valid = rq < S;
        uint4 qf[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) qf[ks] = qn[ks];
        if (qb + NWAVE < nqb) request_q(qb + NWAVE);
        f32x4_t s[NKT];
        scores_t<T, DH, NKT>(Kr, qf, s, S, fr, kg);
        if constexpr (KM) {
            const bool empty = causal ? rq < first : first >= S;          // this query has no allowed key
            const int kmax = causal ? rq : SP;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const f32x4_t m4 = *reinterpret_cast<const f32x4_t*>(km + kt * 16 + kg * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + kg * 4 + r;
                    s[kt][r] = empty ? (key < S ? 0.f : -INFINITY) : ((m4[r] != 0.f && key <= kmax) ? s[kt][r] : -INFINITY);
                }
            }
        }
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kt][r]);
        m = red4(m, true);                                   // max of the RAW scores (scale > 0)
        // p = exp(scale (s - m)) = exp2(s c - m c), c = scale log2(e): one (packed) fma + v_exp_f32 per element; the 1 / row-sum factor is
        // applied to the 16 output values of the lane instead of its 4 NKT probabilities
        const float c2 = scale * 1.44269504088896f, mc = m * c2;
        const f32x4_t c2v = {c2, c2, c2, c2}, mcv = {-mc, -mc, -mc, -mc};
        f32x4_t lv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4_t t = __builtin_elementwise_fma(s[kt], c2v, mcv);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = __builtin_amdgcn_exp2f(t[r]);
            s[kt] = t;
            lv += t;
        }
        const float l = red4((lv[0] + lv[1]) + (lv[2] + lv[3]), false);           // >= 1: the row maximum contributes exp2(0)
        float inv = sizeof(T) == 2 ? __builtin_amdgcn_rcpf(l) : 1.f / l;
        if (dr.thr16) {                                       // P' = dropout(P): the 4 keys of a lane's tile column share one hash
            inv *= dr.keep_scale;
            const int ol = opaque_lane(lane), orq = qb * 16 + (ol & 15), okg = ol >> 4;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const uint64_t hsh = a4r_hash64(dr.seed, dr.site, drop_idx(blockIdx.x, orq, kt * 16 + okg * 4) >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] = (((uint32_t)(hsh >> (16 * r)) & 0xffffu) >= dr.thr16) ? s[kt][r] : 0.f;
            }
        }
        if (valid && kg == 0) lse[((size_t)item * nh + h) * S + rq] = m * scale + __builtin_amdgcn_logf(l) * 0.69314718055994531f;   // v_log_f32 = log2; l is a normal number
        f32x4_t o[G::ND];
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // O^T += V^T P^T in regions of HALF head-column tiles: the transposed V fragments of the next region are requested before the products of the
        // current one; probabilities are packed step by step (keeping all NST chunks spilled)
        constexpr int HALF = G::ND >= 4 ? 2 : G::ND, NR = G::ND / HALF, NREG = NST * NR;
        uint4 vf[2][HALF], pf = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < HALF; ++j) vf[0][j] = frag_T<T, DH>(Vimg, SPT, j * 16, 0, lane);
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const int st = i / NR, hr = i % NR;
            if (hr == 0) pf = pack_step<T, NKT>(s, st);
            if (i + 1 < NREG) {
#pragma unroll
                for (int j = 0; j < HALF; ++j) vf[(i + 1) & 1][j] = frag_T<T, DH>(Vimg, SPT, (((i + 1) % NR) * HALF + j) * 16, (i + 1) / NR, lane);
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) Mma<T>::mma(vf[i & 1][j], pf, o[hr * HALF + j]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) o[dt] *= f32x4_t{inv, inv, inv, inv};
        store_block16<T, DH>(stg, o, cview, qb * 16, lane);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dq + delta
template <typename T, int DH, int NKT, bool KM>
A4R_DEV void dq_body(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off, const T* __restrict__ dctx, int ldo, const T* __restrict__ octx,
                     const float* __restrict__ lse, float* delta, T* dqkv, int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
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
    const int nqb = (S + 15) >> 4;
    // the q / dO / O rows and the lse of a wave's FIRST block are requested before the key side is staged (their latency runs under the staging's);
    // a second set for the next block does not fit the registers of the head-width-64 instantiations: with it they spilled inside the key loop, and a
    // scratch reload's vmcnt(0) waits for every prefetch in flight
    uint4 qn[G::KS], don[G::KS], on[G::KS];
    float lqn = 0.f;
    const RowsView qview = RowsView::make(base + q_off, ld, S, DH), doview = RowsView::make(dctx + (size_t)item * S * ldo + h * DH, ldo, S, DH),
                   oview = RowsView::make(octx + (size_t)item * S * ldo + h * DH, ldo, S, DH),
                   dqview = RowsView::make(dqkv + (size_t)item * S * ld + q_off + h * DH, ld, S, DH);
    auto request_rows = [&](int qb) {
        const int ol = opaque_lane(lane), rq = qb * 16 + (ol & 15), okg = ol >> 4;      // (offsets rebuilt per call, not held across the loops)
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            qn[ks] = qview.row_chunk(rq, ks * 4 + okg);
            don[ks] = doview.row_chunk(rq, ks * 4 + okg);
            on[ks] = oview.row_chunk(rq, ks * 4 + okg);
        }
        lqn = rq < S ? lse[((size_t)item * nh + h) * S + rq] : 0.f;
    };
    if (wave < nqb) request_rows(wave);
    {
        Stager<T, DH, SP, NTHR> sk, sv;
        sk.request(RowsView::make(base + k_off, ld, S, DH), tid);
        sv.request(RowsView::make(base + v_off, ld, S, DH), tid);
        sk.commit(Kr, tid);
        sv.commit(Vr, tid);
    }
    [[maybe_unused]] float* km = reinterpret_cast<float*>(smem + lds_main_dq<T, DH, NKT>() - lds_km<NKT>());
    if constexpr (KM) stage_key_mask<SP, NTHR>(km, kmask, item, S, tid);
    __syncthreads();
    int first = 0;
    if constexpr (KM) first = first_attended<SP>(km, lane);
    const float c2 = scale * 1.44269504088896f;
    const f32x4_t c2v = {c2, c2, c2, c2};
    for (int qb = wave; qb < nqb; qb += NWAVE) {
        const int rq = qb * 16 + fr;
        const bool valid = rq < S;
        [[maybe_unused]] const bool empty = KM && (causal ? rq < first : first >= S);
        [[maybe_unused]] const int kmax = causal ? rq : SP;
        if (qb != wave) request_rows(qb);
        uint4 qf[G::KS], dof[G::KS], of[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) { qf[ks] = qn[ks]; dof[ks] = don[ks]; of[ks] = on[ks]; }
        const float lq2 = lqn * 1.44269504088896f;
        // delta = sum_k P' dP' = dO . O (the forward output, dropout included): one dot product per query instead of a pass over all
        // key tiles, so P, dP and dS are produced and consumed one chunk step (32 / 16 keys) at a time and never held for the whole row
        float dsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            if constexpr (sizeof(T) == 2) {                   // v_dot2c_f32_bf16: two exact products and the fp32 sums per instruction
                const uint32_t a4[4] = {of[ks].x, of[ks].y, of[ks].z, of[ks].w}, b4[4] = {dof[ks].x, dof[ks].y, dof[ks].z, dof[ks].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) dsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_raw_t, a4[e]), __builtin_bit_cast(bf16x2_raw_t, b4[e]), dsum, false);
            } else {
                float a8[G::PER], b8[G::PER];
                Elem<T>::unpack(of[ks], a8);
                Elem<T>::unpack(dof[ks], b8);
#pragma unroll
                for (int e = 0; e < G::PER; ++e) dsum += a8[e] * b8[e];
            }
        }
        dsum = red4(dsum, false);
        if (valid && kg == 0) delta[((size_t)item * nh + h) * S + rq] = dsum;
        const f32x4_t lqv = {-lq2, -lq2, -lq2, -lq2}, ndsv = {-dsum, -dsum, -dsum, -dsum};
        const int lim = S - kg * 4;
        f32x4_t o[G::ND];
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // per key tile: S^T and dP^T (row fragments of K and V), then the element-wise stage; the next tile's fragments are requested between the
        // two (and the transposed K fragments of the step's dQ product before its first tile), so the LDS latency runs under vector work
        uint4 rk[G::KS], rv[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) { rk[ks] = frag_rows<T, DH>(Kr, fr, ks, kg); rv[ks] = frag_rows<T, DH>(Vr, fr, ks, kg); }
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            f32x4_t ds[G::TPS];
            constexpr int HALF = G::ND >= 4 ? 2 : G::ND;
            uint4 tf[2][HALF];                                  // transposed K fragments, two slots of HALF head-column tiles
#pragma unroll
            for (int t = 0; t < G::TPS; ++t) {
                const int kt = st * G::TPS + t;
                f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = ndsv;      // dP - delta straight from the matrix pipe: the accumulator starts at -delta
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    Mma<T>::mma(rk[ks], qf[ks], sc);
                    Mma<T>::mma(rv[ks], dof[ks], dp);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kt + 1 < NKT) {
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) {
                        rk[ks] = frag_rows<T, DH>(Kr, (kt + 1) * 16 + fr, ks, kg);
                        rv[ks] = frag_rows<T, DH>(Vr, (kt + 1) * 16 + fr, ks, kg);
                    }
                }
                if (t == G::TPS - 1) {
#pragma unroll
                    for (int j = 0; j < HALF; ++j) tf[0][j] = frag_T<T, DH>(Kimg, SPT, j * 16, st, lane);
                }
                if constexpr (KM) { if (empty) sc = f32x4_t{0.f, 0.f, 0.f, 0.f}; }       // (a query without allowed keys: the forward's convention)
                f32x4_t pv = __builtin_elementwise_fma(sc, c2v, lqv);                   // P = exp(scale s - lse)
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(pv[r]);
                if constexpr (KM) if (!empty) {
                    const f32x4_t m4 = *reinterpret_cast<const f32x4_t*>(km + kt * 16 + kg * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = (m4[r] != 0.f && kt * 16 + kg * 4 + r <= kmax) ? pv[r] : 0.f;
                }
                if (kt >= kt_partial_lo<NKT>() && kt * 16 + 16 > S) {                    // wave-uniform: the last one or two tiles
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = (kt * 16 + r < lim) ? pv[r] : 0.f;
                }
                if (dr.thr16) {
                    const int ol = opaque_lane(lane);
                    const uint64_t hsh = a4r_hash64(dr.seed, dr.site, drop_idx(blockIdx.x, qb * 16 + (ol & 15), kt * 16 + (ol >> 4) * 4) >> 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dp[r] = ((((uint32_t)(hsh >> (16 * r)) & 0xffffu) >= dr.thr16) ? (dp[r] + dsum) * dr.keep_scale : 0.f) - dsum;
                }
                ds[t] = pv * dp;                                                       // (x scale: applied to the 16 outputs of the lane)
                __builtin_amdgcn_sched_barrier(0);
            }
            const uint4 dsf = pack_step<T, G::TPS>(ds, 0);
            if constexpr (G::ND > HALF) {
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[1][j] = frag_T<T, DH>(Kimg, SPT, (HALF + j) * 16, st, lane);
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[0][j], dsf, o[j]);
            if constexpr (G::ND > HALF) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[1][j], dsf, o[HALF + j]);
            }
            if constexpr (G::ND > 2 * HALF) {                       // head width 128 (fp32, the user tower): the column tiles behind the first four, unscheduled
#pragma unroll
                for (int j = 2 * HALF; j < G::ND; ++j) Mma<T>::mma(frag_T<T, DH>(Kimg, SPT, j * 16, st, lane), dsf, o[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) o[dt] *= f32x4_t{scale, scale, scale, scale};
        store_block16<T, DH>(stg, o, dqview, qb * 16, lane);
    }
}

template <typename T, int DH, int NKT, bool KM = false>      // KM: the launch carries a key mask (text towers / the user tower)
__global__ void __launch_bounds__(WG<NKT>::NTHR, DH > 64 ? 2 : 4) attn_long_dq_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                           const T* __restrict__ dctx, int ldo, const T* __restrict__ octx,
                                                           const float* __restrict__ lse, float* __restrict__ delta,
                                                           T* __restrict__ dqkv, int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
    dq_body<T, DH, NKT, KM>(qkv, ld, q_off, k_off, v_off, dctx, ldo, octx, lse, delta, dqkv, S, nh, scale, dr, kmask, causal);
}

// ------------------------------------------------------------------------------------------------ backward: dk, dv
template <typename T, int DH, int NKT, bool KM>
A4R_DEV void dkdv_body(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off, const T* __restrict__ dctx, int ldo, const float* __restrict__ lse,
                       const float* delta, T* dqkv, int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
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
    const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    char* stg = smem + lds_main_dkdv<T, DH, NKT>() + wave * STG<T, DH>::BYTES;
    const T* base = qkv + (size_t)item * S * ld + h * DH;
    const T* dob = dctx + (size_t)item * S * ldo + h * DH;
    const int nkt = (S + 15) >> 4;
    // the K / V rows of a wave's first key tile are requested before the query side is staged
    uint4 kn[G::KS], vn[G::KS];
    const RowsView kview = RowsView::make(base + k_off, ld, S, DH), vview = RowsView::make(base + v_off, ld, S, DH),
                   dkview = RowsView::make(dqkv + (size_t)item * S * ld + k_off + h * DH, ld, S, DH),
                   dvview = RowsView::make(dqkv + (size_t)item * S * ld + v_off + h * DH, ld, S, DH);
    auto request_kv = [&](int kt) {
        const int ol = opaque_lane(lane0), rk = kt * 16 + (ol & 15), okg = ol >> 4;      // (offsets rebuilt per call, not held across the loops)
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            kn[ks] = kview.row_chunk(rk, ks * 4 + okg);
            vn[ks] = vview.row_chunk(rk, ks * 4 + okg);
        }
    };
    if (wave < nkt) request_kv(wave);
    {
        Stager<T, DH, SP, NTHR> sq, so;
        sq.request(RowsView::make(base + q_off, ld, S, DH), tid);
        so.request(RowsView::make(dob, ldo, S, DH), tid);
        static_assert(SP <= NTHR, "one row statistic per thread");
        const bool rv = tid < S;
        const float lv = rv ? lse[((size_t)item * nh + h) * S + tid] * -1.44269504088896f : 0.f;
        const float dv0 = rv ? -delta[((size_t)item * nh + h) * S + tid] : 0.f;
        sq.commit(Qr, tid);
        so.commit(Or, tid);
        if (tid < SP) { lse_s[tid] = lv; del_s[tid] = dv0; }
    }
    [[maybe_unused]] float* km = reinterpret_cast<float*>(smem + lds_main_dkdv<T, DH, NKT>() - lds_km<NKT>());
    if constexpr (KM) stage_key_mask<SP, NTHR>(km, kmask, item, S, tid);
    __syncthreads();
    int first = 0;
    if constexpr (KM) first = first_attended<SP>(km, lane0);
    const float c2 = scale * 1.44269504088896f;
    const f32x4_t c2v = {c2, c2, c2, c2};
    for (int kt = wave; kt < nkt; kt += NWAVE) {
        // (lane-derived addresses are rebuilt per key tile from an opaque copy of the lane index: held across this loop they were spilled and reloaded)
        const int lane = opaque_lane(lane0), fr = lane & 15, kg = lane >> 4;
        if (kt != wave) request_kv(kt);
        uint4 kf[G::KS], vf[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
        f32x4_t dk[G::ND], dv[G::ND];
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) { dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        // per 16-query tile: S and dP (row fragments of Q and dO; the next tile's are requested right after the products), then the element-wise
        // stage.  Query rows >= S need no select: their Q / dO rows, lse and delta are zero in LDS, so P = 1 meets dO = 0 and dS = 0; key columns >= S
        // (zero K / V rows) only reach output columns that are never stored.
        uint4 rq_[G::KS], ro[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) { rq_[ks] = frag_rows<T, DH>(Qr, fr, ks, kg); ro[ks] = frag_rows<T, DH>(Or, fr, ks, kg); }
#pragma unroll                                          // (round 4: rolled, the loop rebuilt its LDS offsets every group: unrolled -3 % on the backward at S = 197)
        for (int g = 0; g < NG; ++g) {
            constexpr int HALF = G::ND >= 4 ? 2 : G::ND;
            uint4 tf[2][HALF];                                  // transposed fragments, two slots of HALF head-column tiles
            uint32_t pw[4], dw[4];                            // the step's P and dS operand chunks, packed tile by tile
#pragma unroll
            for (int t = 0; t < G::TPS; ++t) {
                const int q0 = g * G::KSTEP + t * 16;         // tile rows = queries q0 + 4 kg + r, column = key rk
                // (del_s holds -delta: dP - delta comes straight from the matrix pipe, the accumulator starts there; lse_s holds -lse log2(e))
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_s + q0 + kg * 4), d4 = *reinterpret_cast<const f32x4_t*>(del_s + q0 + kg * 4);
                f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dpt = d4;
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    Mma<T>::mma(rq_[ks], kf[ks], sc);
                    Mma<T>::mma(ro[ks], vf[ks], dpt);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int qn0 = q0 + 16 < SP ? q0 + 16 : 0;       // (the last tile wraps to rows that are simply not used)
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) { rq_[ks] = frag_rows<T, DH>(Qr, qn0 + fr, ks, kg); ro[ks] = frag_rows<T, DH>(Or, qn0 + fr, ks, kg); }
                }
                if (t == G::TPS - 1) {
#pragma unroll
                    for (int j = 0; j < HALF; ++j) tf[0][j] = frag_T<T, DH>(Oimg, SPT, j * 16, g, lane);
                }
                if constexpr (KM) {                            // rows = queries q0 + 4 kg + r, this lane's key = 16 kt + fr
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (causal ? q0 + kg * 4 + r < first : first >= S) sc[r] = 0.f;
                }
                f32x4_t pv = __builtin_elementwise_fma(sc, c2v, l4), dsv;
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(pv[r]);
                if constexpr (KM) {
                    const int key = kt * 16 + fr;
                    const bool kon = km[key] != 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = q0 + kg * 4 + r;
                        const bool empty = causal ? q < first : first >= S;
                        if (!empty && !(kon && (!causal || key <= q))) pv[r] = 0.f;
                    }
                }
                if (dr.thr16) {                               // here the tile's 4 rows are 4 QUERIES at one key: one hash each
                    const int ol = opaque_lane(lane), ork = kt * 16 + (ol & 15), okg = ol >> 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float keepf = dropout_keep(dr.seed, dr.site, drop_idx(blockIdx.x, q0 + okg * 4 + r, ork), dr.thr16) ? dr.keep_scale : 0.f;
                        dsv[r] = pv[r] * ((dpt[r] - d4[r]) * keepf + d4[r]);
                        pv[r] *= keepf;
                    }
                } else {
                    dsv = pv * dpt;                                                     // (x scale: applied to dK at the end)
                }
                if constexpr (sizeof(T) == 2) {
                    pw[2 * t] = pack2_bf16(pv[0], pv[1]); pw[2 * t + 1] = pack2_bf16(pv[2], pv[3]);
                    dw[2 * t] = pack2_bf16(dsv[0], dsv[1]); dw[2 * t + 1] = pack2_bf16(dsv[2], dsv[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pw[r] = __float_as_uint(pv[r]); dw[r] = __float_as_uint(dsv[r]); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const uint4 pf = make_uint4(pw[0], pw[1], pw[2], pw[3]), dsf = make_uint4(dw[0], dw[1], dw[2], dw[3]);
            // dV^T += dO^T P, dK^T += Q^T dS, HALF head-column tiles at a time: a slot is refilled as soon as its products are issued
            if constexpr (G::ND > HALF) {
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[1][j] = frag_T<T, DH>(Oimg, SPT, (HALF + j) * 16, g, lane);
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[0][j], pf, dv[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < HALF; ++j) tf[0][j] = frag_T<T, DH>(Qimg, SPT, j * 16, g, lane);
            if constexpr (G::ND > HALF) {
#pragma unroll
                for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[1][j], pf, dv[HALF + j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[1][j] = frag_T<T, DH>(Qimg, SPT, (HALF + j) * 16, g, lane);
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[0][j], dsf, dk[j]);
            if constexpr (G::ND > HALF) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[1][j], dsf, dk[HALF + j]);
            }
            if constexpr (G::ND > 2 * HALF) {                       // head width 128: the column tiles behind the first four, unscheduled
#pragma unroll
                for (int j = 2 * HALF; j < G::ND; ++j) {
                    Mma<T>::mma(frag_T<T, DH>(Oimg, SPT, j * 16, g, lane), pf, dv[j]);
                    Mma<T>::mma(frag_T<T, DH>(Qimg, SPT, j * 16, g, lane), dsf, dk[j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) dk[dt] *= f32x4_t{scale, scale, scale, scale};
        store_block16<T, DH>(stg, dk, dkview, kt * 16, lane);
        store_block16<T, DH>(stg, dv, dvview, kt * 16, lane);
    }
}

template <typename T, int DH, int NKT, bool KM = false>
__global__ void __launch_bounds__(WG<NKT>::NTHR, DH > 64 ? 2 : 4) attn_long_dkdv_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                             const T* __restrict__ dctx, int ldo, const float* __restrict__ lse,
                                                             const float* __restrict__ delta, T* __restrict__ dqkv,
                                                             int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
    dkdv_body<T, DH, NKT, KM>(qkv, ld, q_off, k_off, v_off, dctx, ldo, lse, delta, dqkv, S, nh, scale, dr, kmask, causal);
}

// Both halves of the backward in ONE launch per (item, head): dq + delta first, then -- same workgroup, same LDS -- dk and dv.  The second half's reads of
// Q / K / V / dO repeat what this workgroup read tens of microseconds earlier (L2 / Infinity Cache instead of HBM: as two launches over all items the dkdv
// kernel found nothing of the dq kernel's stream left in a 256-MB cache at ViT-B/16's 500 MB), and delta goes through global memory within the workgroup.
template <typename T, int DH, int NKT, bool KM = false>
__global__ void __launch_bounds__(WG<NKT>::NTHR, DH > 64 ? 2 : 4) attn_long_bwd_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                            const T* __restrict__ dctx, int ldo, const T* __restrict__ octx,
                                                            const float* __restrict__ lse, float* delta, T* dqkv,
                                                            int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
    dq_body<T, DH, NKT, KM>(qkv, ld, q_off, k_off, v_off, dctx, ldo, octx, lse, delta, dqkv, S, nh, scale, dr, kmask, causal);
    __threadfence_block();                                  // delta written by this workgroup's waves is read by others of it below
    __syncthreads();                                        // ... and the LDS images of the first half are dead
    dkdv_body<T, DH, NKT, KM>(qkv, ld, q_off, k_off, v_off, dctx, ldo, lse, delta, dqkv, S, nh, scale, dr, kmask, causal);
}

int nkt_for(int S) { return S <= 32 ? 2 : S <= 64 ? 4 : S <= 128 ? 8 : S <= 224 ? 14 : 16; }
// kt_partial_lo<NKT>() (which key tiles can reach past S) restates this map: a launch whose S does not fit its instantiation's assumption is refused
template <int NKT> bool s_fits(int S) { return S <= NKT * 16 && S > 16 * kt_partial_lo<NKT>(); }

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
    return Drop{a->drop_seed, a->drop_site, a4r_thr16(a->drop_p), a4r_keep_scale(a->drop_p)};
}

template <typename T, int DH, int NKT, bool KM> int run_fwd_km(hipStream_t s, const a4r_attn_t* a, float* lse) {
    const size_t lds = lds_fwd<T, DH, NKT>();
    if (int rc = set_lds(attn_long_fwd_kernel<T, DH, NKT, KM>, lds)) return rc;
    hipLaunchKernelGGL((attn_long_fwd_kernel<T, DH, NKT, KM>), dim3(a->n_items * a->n_heads), dim3(WG<NKT>::NTHR), lds, s, (const T*)a->qkv, a->ld, a->q_off,
                       a->k_off, a->v_off, (T*)a->out, a->ldo, lse, a->S, a->n_heads, a->scale, drop_of(a), (const float*)a->key_mask, a->causal);
    return a4r_launch_status();
}
template <typename T, int DH, int NKT> int run_fwd(hipStream_t s, const a4r_attn_t* a, float* lse) {
    if (!s_fits<NKT>(a->S)) return A4R_EINVAL;
    if (a->key_mask) return run_fwd_km<T, DH, NKT, true>(s, a, lse);
    return run_fwd_km<T, DH, NKT, false>(s, a, lse);
}
template <typename T, int DH, int NKT, bool KM> int run_bwd_km(hipStream_t s, const a4r_attn_t* a, const float* lse, float* delta) {
    const size_t l1 = lds_dq<T, DH, NKT>(), l2 = lds_dkdv<T, DH, NKT>();
    if (int rc = set_lds(attn_long_dq_kernel<T, DH, NKT, KM>, l1)) return rc;
    if (int rc = set_lds(attn_long_dkdv_kernel<T, DH, NKT, KM>, l2)) return rc;
    const dim3 grid(a->n_items * a->n_heads), block(WG<NKT>::NTHR);
    // one launch up to 64 tokens (ViT-MAE's 50: 56 -> 48 us, the step -0.6 % bf16 / -0.9 % fp8); above, the two launches stay -- at ViT-B/16's 197 the one-launch
    // form measured the same (351 vs 340 - 354 us, step -0.1 %) with 7 spilled registers instead of 2 (profiles/r05_q_attn_bwd_fused.txt).  A4R_ATTN_BWD_FUSED=0 / 1 forces either.
    static const int knob = []{ const char* e = getenv("A4R_ATTN_BWD_FUSED"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    const bool fused = knob < 0 ? NKT <= 4 : knob != 0;
    if (fused) {
        const size_t l = l1 > l2 ? l1 : l2;
        if (int rc = set_lds(attn_long_bwd_kernel<T, DH, NKT, KM>, l)) return rc;
        hipLaunchKernelGGL((attn_long_bwd_kernel<T, DH, NKT, KM>), grid, block, l, s, (const T*)a->qkv, a->ld, a->q_off, a->k_off, a->v_off,
                           (const T*)a->dout, a->ldo, (const T*)a->out, lse, delta, (T*)a->dqkv, a->S, a->n_heads, a->scale, drop_of(a), (const float*)a->key_mask, a->causal);
        return a4r_launch_status();
    }
    hipLaunchKernelGGL((attn_long_dq_kernel<T, DH, NKT, KM>), grid, block, l1, s, (const T*)a->qkv, a->ld, a->q_off, a->k_off, a->v_off,
                       (const T*)a->dout, a->ldo, (const T*)a->out, lse, delta, (T*)a->dqkv, a->S, a->n_heads, a->scale, drop_of(a), (const float*)a->key_mask, a->causal);
    hipLaunchKernelGGL((attn_long_dkdv_kernel<T, DH, NKT, KM>), grid, block, l2, s, (const T*)a->qkv, a->ld, a->q_off, a->k_off, a->v_off,
                       (const T*)a->dout, a->ldo, lse, (const float*)delta, (T*)a->dqkv, a->S, a->n_heads, a->scale, drop_of(a), (const float*)a->key_mask, a->causal);
    return a4r_launch_status();
}
template <typename T, int DH, int NKT> int run_bwd(hipStream_t s, const a4r_attn_t* a, const float* lse, float* delta) {
    if (!s_fits<NKT>(a->S)) return A4R_EINVAL;
    if (a->key_mask) return run_bwd_km<T, DH, NKT, true>(s, a, lse, delta);
    return run_bwd_km<T, DH, NKT, false>(s, a, lse, delta);
}

int check(const a4r_attn_t* a, bool bwd) {
    if (!a || !a->qkv || a->n_items <= 0 || a->S <= 0 || a->S > 256 || (a->dh != 64 && a->dh != 32 && a->dh != 128) || a->n_heads <= 0) return A4R_EINVAL;
    if (a->dh == 128 && (a->dtype != A4R_F32 || a->S > 128)) return A4R_EINVAL;     // head width 128: fp32, up to 128 tokens (two [128, 128] fp32 images = 128 KB of LDS)
    if (a->offsets || (a->causal && !a->key_mask)) return A4R_EINVAL;              // no packed items; causal only together with a key mask (the user tower); key_mask (fp32 [n_items, S], optional)
    if (a->key_mask && (reinterpret_cast<uintptr_t>(a->key_mask) & 3u)) return A4R_EINVAL;
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
// head width 128 (fp32, S <= 128: the user tower at the parser's default --embedding_dim 256 with two heads and --max_seq_len above 32)
#define A4R_DH128_SWITCH(CALL_)                         \
    if (a->dh == 128) switch (nkt_for(a->S)) {          \
        case 2: return CALL_(float, 128, 2);            \
        case 4: return CALL_(float, 128, 4);            \
        default: return CALL_(float, 128, 8);           \
    }

}  // namespace

extern "C" int a4r_attn_long_fwd(void* stream, const a4r_attn_t* a, float* lse) {
    if (int rc = check(a, false)) return rc;
    if (!lse) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define A4R_F(T_, D_, N_) run_fwd<T_, D_, N_>(s, a, lse)
    if (a->dtype == A4R_BF16) { A4R_DH_SWITCH(bf16_t, A4R_F) }
    A4R_DH128_SWITCH(A4R_F)
    A4R_DH_SWITCH(float, A4R_F)
#undef A4R_F
}

extern "C" int a4r_attn_long_bwd(void* stream, const a4r_attn_t* a, const float* lse, float* delta_ws) {
    if (int rc = check(a, true)) return rc;
    if (!lse || !delta_ws) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define A4R_B(T_, D_, N_) run_bwd<T_, D_, N_>(s, a, lse, delta_ws)
    if (a->dtype == A4R_BF16) { A4R_DH_SWITCH(bf16_t, A4R_B) }
    A4R_DH128_SWITCH(A4R_B)
    A4R_DH_SWITCH(float, A4R_B)
#undef A4R_B
}
