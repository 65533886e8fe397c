// Shared pieces of the long-attention kernels (a4r_attn_long.hip: forward + the two-launch backward; a4r_attn_long1.hip: the one-pass backward):
// geometry, raw-buffer row views, LDS staging, operand fragment accessors, the store path.  Everything sits in an anonymous namespace of the
// including translation unit.
#pragma once
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

}  // namespace
