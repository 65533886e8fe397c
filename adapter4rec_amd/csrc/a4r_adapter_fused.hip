// Fused bottleneck adapter + residual(s) + LayerNorm, forward and backward, bf16 (the north_star's "LDS-staged adapter
// bottleneck down/up projections fused with residual + LayerNorm").
//
// Replaces, per adapter, the launch sequences
//   forward : skinny64 (zp = A Wd^T + bd, z = act(zp)) | skinnyk (v = z Wu^T + bu + residuals) | ln_fwd (y = LN(v))
//   backward: ln_bwd (dv) | skinny64 (dzp = (dv Wu) * act'(zp)) | skinnyk (dh = dzp Wd + dv, dropout)
// of BertAdaptedSelfOutput.forward / AdapterBlock.forward (Downstream/Text/model/model.py:292-297, modules.py:116-134; the
// Compacter form model.py:696-720 without the inner residual; the Pfeiffer form model.py:321-329 with A = LN(h + input))
// by ONE launch each: every activation row is read once and written once.  Algorithmic HBM bytes per launch, bf16:
//   forward  read A, O (2 M H) | write v, y (2 M H) | zp, z (2 M 64)                      = (4 H + 128) 2 M  bytes
//   backward read dy, v (2 M H) [+ dres] | write dv, dh (2 M H) | zp in, dzp out (2 M 64) = (4 H + 128) 2 M  bytes
// against ~7 M H x 2 for the three-launch forms (h / dv re-read by two kernels, v re-read by ln_fwd).
//
// Structure.  One workgroup of NW waves per CU (persistent over 16-row tiles); wave w owns columns [w CW, (w+1) CW) of the
// width H = NW x CW for every row of the tile.  Both weight matrices stay ON THE CU for the whole launch as MFMA operand
// fragments (forward: both in registers, 48 + 48 VGPRs at CW = 96; backward: Wu in registers, Wd as a fragment-ordered LDS
// image, its registers go to the column sums): the 192 KB of weights are read once per CU, not once per tile -- streaming
// them per 16 rows is what made the first fused kernel (round 1) slower than the un-fused launches.
//   * a lane holds, of row (lane & 15), the 16-byte pieces at columns c0 + 32 s + 8 (lane >> 4) + [0, 8), s < CW / 32:
//     exactly the MFMA operand of the down-projection (contraction over H: the wave's CW columns are K-steps), so a
//     global_load_dwordx4 is an operand with no LDS round trip;
//   * the [16, 64] partial products of the NW waves are summed through LDS (34 KB), bias + activation applied once, the
//     bottleneck z goes back to LDS as the bf16 operand of the up-projection (K = 64: two MFMA steps);
//   * the up-projection's weight rows are PERMUTED when the fragments are loaded so that accumulator register (tile 2s+h, r)
//     of a lane is column c0 + 32 s + 8 (lane >> 4) + 4 h + r -- the very columns of the pieces the lane loaded: residual
//     adds, LayerNorm and the 16-byte stores happen in registers in the load layout;
//   * row statistics: per-wave (mean, centred sum of squares) over its CW columns, combined across waves exactly (Chan);
//   * the next tile's rows are requested before the current tile is processed (static vmcnt schedule: the last tile
//     re-requests itself instead of branching).
// Three workgroup barriers per tile.  The kernels are HBM-bound: 24 MFMAs per wave per 96 KB of traffic.
#include <cstdlib>
#include <type_traits>
#include "a4r_common.h"
#include "../../include/a4r.h"

#ifdef A4R_STAMP
// diagnostic build only (tools/adapter_timeline.py): s_memrealtime of waves 0 and 7 of every workgroup around the three barriers of its
// THIRD backward tile: [wg][wave 0 / 7][top, before / after barrier 1, before / after barrier 2, before / after barrier 3, end]
__device__ unsigned long long g_a4r_ad_stamps[256 * 2 * 8];
extern "C" int a4r_debug_adapter_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_a4r_ad_stamps), sizeof(g_a4r_ad_stamps)) == hipSuccess ? 0 : -2;
}
#define A4R_AD_ST_(k_)                                                                                                 \
    if (lane == 0 && (wave == 0 || wave == NW - 1) && tile == (int)blockIdx.x + 2 * (int)gridDim.x && blockIdx.x < 256)  \
        g_a4r_ad_stamps[(blockIdx.x * 2 + (wave ? 1 : 0)) * 8 + (k_)] = __builtin_amdgcn_s_memrealtime();
#if A4R_STAMP == 2          /* -DA4R_STAMP=2: the FORWARD kernel's tile (top, before / after each of its three barriers, end); 1: the backward's */
#define A4R_ADF_ST(k_) A4R_AD_ST_(k_)
#define A4R_AD_ST(k_)
#else
#define A4R_ADF_ST(k_)
#define A4R_AD_ST(k_) A4R_AD_ST_(k_)
#endif
#if A4R_STAMP == 3          /* -DA4R_STAMP=3: the forward LAUNCH: wave 0 of every workgroup at entry, after the prologue (weights, first rows requested, parameters in LDS), after its last tile */
#undef A4R_ADF_ST
#undef A4R_AD_ST
#define A4R_ADF_ST(k_)
#define A4R_AD_ST(k_)
#define A4R_ADF_LAUNCH(k_) if (lane == 0 && (wave == 0 || wave == NW - 1) && blockIdx.x < 256) g_a4r_ad_stamps[blockIdx.x * 16 + (wave ? 8 : 0) + (k_)] = __builtin_amdgcn_s_memrealtime();      /* slots 0-7: wave 0, 8-15: the last wave */
#else
#define A4R_ADF_LAUNCH(k_)
#endif
#if A4R_STAMP == 4          /* -DA4R_STAMP=4: the same for the backward launch (entry, prologue done, first tile done, last tile done) */
#undef A4R_ADF_ST
#undef A4R_AD_ST
#define A4R_ADF_ST(k_)
#define A4R_AD_ST(k_)
#define A4R_ADB_LAUNCH(k_) if (lane == 0 && wave == 0 && blockIdx.x < 256) g_a4r_ad_stamps[blockIdx.x * 16 + (k_)] = __builtin_amdgcn_s_memrealtime();
#else
#define A4R_ADB_LAUNCH(k_)
#endif
#else
#define A4R_ADB_LAUNCH(k_)
#define A4R_AD_ST(k_)
#define A4R_ADF_ST(k_)
#define A4R_ADF_LAUNCH(k_)
#endif

#ifndef A4R_AD_ABL
#define A4R_AD_ABL 0          /* timing-only diagnostic builds (-DA4R_AD_ABL=n): 1 no tile stores, 2 no tile loads, 4 no column-sum flush (atomics) at the end of the backward; forward: 8 no row loads in the tile loop (register form), 16 no stores */
#endif

namespace {

constexpr int ZLD = 68;            // floats per row of a wave's partial [16][64] tile in LDS (272-byte rows: bank spread)

#define A4R_LDS_BARRIER()                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    __builtin_amdgcn_s_barrier();                            \
    asm volatile("" ::: "memory")

A4R_DEV float kg_sum(float v) {    // over the 4 lanes (lane & 15 fixed) that hold one row
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
A4R_DEV float bf16_round(float x) { return bf16_bits_to_f32(f32_to_bf16_bits(x)); }

struct AdFwdArgs {
    const bf16_t* A; const bf16_t* O; int lda, ldo, a_in_resid;
    const bf16_t* Wd; const bf16_t* Wu; const float* bd; const float* bu; const float* gamma; const float* beta;
    float eps; int act;
    bf16_t* zp; bf16_t* z; bf16_t* v; bf16_t* y; int ldv, ldy; float* stats; int M;
    unsigned char* y8; int ld8; float* ys;      // optional: the LayerNorm output as OCP e4m3 + per-row scale (the next GEMM's fp8 A operand)
    // --residual_dtype fp32 (round 4): the residual stream between sub-layers in fp32, as under the reference's autocast (its LayerNorm
    // outputs fp32 and the residual add promotes to it): O32 replaces O as the residual operand, y32 is y before its bf16 rounding
    const float* O32; int ldo32; float* y32; int ldy32;
    // --residual_dtype bf24 (round 6): the same twins as ONE BYTE per element beside the bf16 tensor -- the next 8 mantissa bits of the fp32 value
    // as a signed offset from its bf16 rounding (lo8_of / lo8_join below): O + Olo is read as a 24-bit float, y + ylo written as one
    const signed char* Olo; int ldolo; signed char* ylo; int ldylo;
    int lo4;                                    // the planes hold 4 bits per element (w_frag bit 2): --residual_dtype bf20
    int wfrag;                                  // Wd / Wu in fragment order
};

// sum of the NW partial [16][64] tiles for EPT consecutive bottleneck columns of one row (thread t: element t * EPT)
template <int NW, int EPT>
A4R_DEV void reduce_partials(const float (*zpart)[16][ZLD], int row, int zd, float (&s)[EPT]) {
#pragma unroll
    for (int i = 0; i < EPT; ++i) s[i] = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if constexpr (EPT == 2) {
            const float2 q = *reinterpret_cast<const float2*>(&zpart[w][row][zd]);
            s[0] += q.x; s[1] += q.y;
        } else {
            const float4 q = *reinterpret_cast<const float4*>(&zpart[w][row][zd]);
            s[0] += q.x; s[1] += q.y; s[2] += q.z; s[3] += q.w;
        }
    }
}
template <int EPT>
A4R_DEV void store_bf16_n(bf16_t* dst, const float (&v)[EPT]) {
    if constexpr (EPT == 2) *reinterpret_cast<uint32_t*>(dst) = pack2_bf16(v[0], v[1]);
    else *reinterpret_cast<uint2*>(dst) = make_uint2(pack2_bf16(v[0], v[1]),
                                                      pack2_bf16(v[2], v[3]));
}
// byte offset of bottleneck column zd of row r in the swizzled bf16 [16][64] operand image (16-byte chunk c at c ^ ((r >> 1) & 7))
A4R_DEV int zbf_off(int r, int zd) { return r * 128 + ((((zd >> 3) ^ ((r >> 1) & 7))) << 4) + (zd & 7) * 2; }

// The 24-bit residual stream: the bf16 tensor stays the RNE rounding the GEMMs read; a byte plane beside it holds the SIGNED byte
// d = (bits(x) - (bits(bf16) << 16)) >> 8  (the next 8 mantissa bits of the fp32 value x, truncated, as an offset in [-128, 127] from its bf16 rounding:
// bit-pattern arithmetic, binade crossings need no case; an exact tie rounded down, d = +128, is stored as 127) and
// x24 = (bits(bf16) << 16) + (d << 8) restores x to 2^-15 relative (7 + 8 explicit mantissa bits).  A zero byte is a zero offset: rows nobody wrote (padding) read as their bf16 value.
// Both directions are byte permutes: 2 (join) / ~4 (split) vector instructions per element.
A4R_DEV float lo8_join(uint32_t hw, uint32_t lw, int j) {     // element j (0 .. 7) of a piece: hw = the word of its bf16 pair, lw = the word of its byte quad
    const uint32_t sel = ((j & 1) ? 0x07060000u : 0x05040000u) | ((uint32_t)(j & 3) << 8) | 0x0cu;
    return __uint_as_float((__builtin_amdgcn_perm(hw, lw, sel) ^ 0x8000u) - 0x8000u);      // (hi16 | d << 8) with d sign-extended into the upper half
}
A4R_DEV uint32_t lo8_split4(const float (&x)[8], int m, uint32_t hw0, uint32_t hw1) {   // the byte quad of elements 4 m .. 4 m + 3 (bf16 pairs hw0, hw1 as stored)
    auto off = [](float v, uint32_t hi16) { const int d = (int)(__float_as_uint(v) - hi16); return (uint32_t)(d < 0x7FFF ? d : 0x7FFF); };
    const uint32_t d0 = off(x[4 * m], hw0 << 16), d1 = off(x[4 * m + 1], hw0 & 0xFFFF0000u);
    const uint32_t d2 = off(x[4 * m + 2], hw1 << 16), d3 = off(x[4 * m + 3], hw1 & 0xFFFF0000u);
    const uint32_t p01 = __builtin_amdgcn_perm(d1, d0, 0x0c0c0501u), p23 = __builtin_amdgcn_perm(d3, d2, 0x0c0c0501u);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u);
}

// The 20-bit form (w_frag bit 2, --residual_dtype bf20): FOUR more mantissa bits, rounded to nearest, as a signed nibble n = clamp((bits(x) - (bits(bf16) << 16) + 0x800) >> 12, -8, 7);
// x20 = (bits(bf16) << 16) + (n << 12).  Eight consecutive elements share one 32-bit word, element j in bits [4 j, 4 j + 4): half the plane bytes of the
// 24-bit form.  What the residual stream needs beyond bf16 is ~2 bits (its rounding error enters the rms distance in quadrature: 2^-13 against the 2^-9 of the
// other bf16 tensors of a sub-layer); rounding, not truncation: a truncated nibble shrinks every element by 2^-13 of itself, coherently over 24 sub-layers.
A4R_DEV float lo4_join(uint32_t hw, uint32_t lw, int j) {
    const uint32_t hi16 = (j & 1) ? (hw & 0xFFFF0000u) : (hw << 16);
    const int n = __builtin_amdgcn_sbfe((int)lw, 4 * j, 4);
    return __uint_as_float(hi16 + ((uint32_t)n << 12));
}
A4R_DEV uint32_t lo4_split8(const float (&x)[8], const uint32_t (&hw)[4]) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t hi16 = (j & 1) ? (hw[j >> 1] & 0xFFFF0000u) : (hw[j >> 1] << 16);
        int d = (int)(__float_as_uint(x[j]) - hi16) + 0x800;
        d = d < 0x7FFF ? d : 0x7FFF;
        w |= __builtin_amdgcn_ubfe((uint32_t)d, 12, 4) << (4 * j);
    }
    return w;
}

// The nibble plane is laid out in the kernel's LANE order within a row (it is written and read by this kernel only; whole rows may be moved): the KS words of lane
// (wave, kg) -- its pieces s = 0 .. KS - 1, columns wave CW + 32 s + 8 kg + [0, 8) -- sit together at word (wave 4 + kg) KS, so a lane moves them with ONE
// memory instruction.  (Measured first in row-major order: three 4-byte requests per lane and tile cost what the byte plane's three 8-byte ones do -- the
// fused forward pays per memory INSTRUCTION through its address unit, not per byte: profiles/r06_h_residual_bf20.txt.)
template <int KS> struct LoWords { uint32_t w[KS]; };

// RM = 1: the residual operand is the fp32 tensor O32 (two 16-byte pieces per 8 columns instead of one); RM = 2: the bf16 tensor O + its byte plane Olo; RM = 3: ... + its nibble plane
template <int CW, int NW, int RM = 0>
__global__ void __launch_bounds__(NW * 64) adapter_ln_fwd_kernel(const AdFwdArgs p) {
    constexpr int KS = CW / 32, H = CW * NW, NT = NW * 64, EPT = 1024 / NT, OP = RM ? 2 : 1;
    constexpr bool R32 = RM == 1;
    __shared__ __attribute__((aligned(16))) float zpart[NW][16][ZLD];
    __shared__ __attribute__((aligned(16))) char zbf[16 * 128];
    __shared__ float red[NW][16][2];
    __shared__ float red8[NW][16];
    __shared__ __attribute__((aligned(16))) float par[3][H];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kg = lane >> 4;
    const int c0 = wave * CW;
    const int cl = c0 + kg * 8;                              // + 32 s: first column of the lane's piece s
    A4R_ADF_LAUNCH(0)
    const int ntiles = p.M / 16;
    // the per-column parameters are REQUESTED here and written to LDS after the weight and first-tile requests below (round 4: as a plain copy loop this
    // compiled to H / NT dependent rounds of load -> wait -> ds_write ahead of everything else the workgroup asks for)
    constexpr int NPAR = (H + NT - 1) / NT;
    float par_r[NPAR][3];
#pragma unroll
    for (int i = 0; i < NPAR; ++i) {
        const int c = tid + i * NT;
        const bool in = c < H;
        par_r[i][0] = in ? p.bu[c] : 0.f; par_r[i][1] = in ? p.gamma[c] : 0.f; par_r[i][2] = in ? p.beta[c] : 0.f;
    }

    // weight fragments (W side of the MFMA: a lane supplies weight row (lane & 15), 8 contraction elements at lane >> 4).
    // wfrag (round 5): the matrices arrive in FRAGMENT order (a4r_pack_matrices layouts 1 / 2, include/a4r.h): a wave instruction then reads 1 KiB
    // contiguous instead of 16 row pieces of 64 bytes, which the CU's address unit takes twice as long over -- all 8 waves' requests issued after
    // 2.9 instead of 4.6 us, first tile 3 us earlier (tools/adapter_launch_timeline.py; profiles/r05_m_*)
    uint4 wd[KS][4], wu[2 * KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            wd[s][nt] = *reinterpret_cast<const uint4*>(p.wfrag ? p.Wd + (size_t)(((wave * KS + s) * 4 + nt) * 64 + lane) * 8
                                                                 : p.Wd + (size_t)(nt * 16 + fr) * H + cl + s * 32);
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)      // accumulator register (kg', r) of tile 2s+h <- weight row c0 + 32 s + 8 kg' + 4 h + r
                wu[2 * s + h][ks] = *reinterpret_cast<const uint4*>(p.wfrag ? p.Wu + (size_t)((((wave * KS + s) * 2 + h) * 2 + ks) * 64 + lane) * 8
                                                                             : p.Wu + (size_t)(c0 + s * 32 + (fr >> 2) * 8 + h * 4 + (fr & 3)) * 64 + ks * 32 + kg * 8);

    const int e0 = tid * EPT, rrow = e0 >> 6, rzd = e0 & 63;  // this thread's share of the [16][64] reduction
    float bd_r[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) bd_r[i] = p.bd[rzd + i];

    // (rows are requested ONE tile ahead.  Two tiles ahead -- a third register set, ~96 KB in flight per CU -- measured slower in round 4:
    // 54.1 - 55.3 us against 51.4 us per launch at M = 40 448, 246 VGPRs; profiles/r04_e_adapter_ab.txt)
    uint4 a_cur[KS], o_cur[KS * OP], a_nxt[KS], o_nxt[KS * OP];
#define A4R_AD_LOAD_O(dst_, row_)                                                                                       \
    if constexpr (R32) {                                                                                               \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                               \
            dst_[2 * s] = *reinterpret_cast<const uint4*>(p.O32 + (row_) * p.ldo32 + cl + s * 32);                      \
            dst_[2 * s + 1] = *reinterpret_cast<const uint4*>(p.O32 + (row_) * p.ldo32 + cl + s * 32 + 4);              \
        }                                                                                                              \
    } else if constexpr (RM == 2) {                                                                                    \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                               \
            dst_[2 * s] = *reinterpret_cast<const uint4*>(p.O + (row_) * p.ldo + cl + s * 32);                          \
            const uint2 l_ = *reinterpret_cast<const uint2*>(p.Olo + (row_) * p.ldolo + cl + s * 32);                   \
            dst_[2 * s + 1] = make_uint4(l_.x, l_.y, 0u, 0u);                                                           \
        }                                                                                                              \
    } else if constexpr (RM == 3) {                                                                                    \
        const LoWords<KS> l_ = *reinterpret_cast<const LoWords<KS>*>(p.Olo + (row_) * p.ldolo + (wave * 4 + kg) * KS * 4); \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                               \
            dst_[2 * s] = *reinterpret_cast<const uint4*>(p.O + (row_) * p.ldo + cl + s * 32);                          \
            dst_[2 * s + 1] = make_uint4(l_.w[s], 0u, 0u, 0u);                                                          \
        }                                                                                                              \
    } else {                                                                                                           \
        _Pragma("unroll") for (int s = 0; s < KS; ++s) dst_[s] = *reinterpret_cast<const uint4*>(p.O + (row_) * p.ldo + cl + s * 32); \
    }
    {
        const size_t row = (size_t)blockIdx.x * 16 + fr;
#pragma unroll
        for (int s = 0; s < KS; ++s) a_cur[s] = *reinterpret_cast<const uint4*>(p.A + row * p.lda + cl + s * 32);
        A4R_AD_LOAD_O(o_cur, row)
    }
    A4R_ADF_LAUNCH(5)          // every request of the prologue issued
#pragma unroll
    for (int i = 0; i < NPAR; ++i) {
        const int c = tid + i * NT;
        if (c < H) { par[0][c] = par_r[i][0]; par[1][c] = par_r[i][1]; par[2][c] = par_r[i][2]; }
    }
    A4R_ADF_LAUNCH(6)          // the parameters (the first loads of the launch) have arrived
    A4R_LDS_BARRIER();                                       // par[] visible
    A4R_ADF_LAUNCH(1)
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        {   // request the next tile's rows now (the last tile re-requests itself: the vmcnt schedule stays static)
            const int tn = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;
            const size_t row = (size_t)tn * 16 + fr;
            if (A4R_AD_ABL & 8) {
#pragma unroll
                for (int s = 0; s < KS; ++s) { a_nxt[s] = a_cur[s]; o_nxt[s] = o_cur[s]; }
            } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) a_nxt[s] = *reinterpret_cast<const uint4*>(p.A + row * p.lda + cl + s * 32);
            A4R_AD_LOAD_O(o_nxt, row)
            }
        }
        A4R_ADF_ST(0)
        const size_t row = (size_t)tile * 16 + fr;
        // ---- down-projection: partial over this wave's CW columns
        f32x4_t zacc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) zacc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) Mma<bf16_t>::mma(wd[s][nt], a_cur[s], zacc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4_t*>(&zpart[wave][fr][nt * 16 + kg * 4]) = zacc[nt];
        A4R_ADF_ST(1)
        A4R_LDS_BARRIER();
        A4R_ADF_ST(2)
        if (it == 0) { A4R_ADF_LAUNCH(3) }
        // ---- sum over waves, bias, activation: zp (pre-activation) and z to HBM, z as the next MFMA's operand to LDS
        {
            float s_[EPT], a_[EPT];
            reduce_partials<NW, EPT>(zpart, rrow, rzd, s_);
#pragma unroll
            for (int i = 0; i < EPT; ++i) { s_[i] += bd_r[i]; a_[i] = act_fwd(s_[i], p.act); }
            const size_t g = ((size_t)tile * 16 + rrow) * 64 + rzd;
            if (!(A4R_AD_ABL & 16)) {
            store_bf16_n<EPT>(p.zp + g, s_);
            store_bf16_n<EPT>(p.z + g, a_);
            }
            store_bf16_n<EPT>(reinterpret_cast<bf16_t*>(zbf + zbf_off(rrow, rzd)), a_);
        }
        A4R_ADF_ST(3)
        A4R_LDS_BARRIER();
        A4R_ADF_ST(4)
        // ---- up-projection (K = 64) into the load layout
        uint4 zf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) zf[ks] = *reinterpret_cast<const uint4*>(zbf + fr * 128 + (((ks * 4 + kg) ^ ((fr >> 1) & 7)) << 4));
        f32x4_t acc[2 * KS];
#pragma unroll
        for (int t = 0; t < 2 * KS; ++t) {
            acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) Mma<bf16_t>::mma(wu[t][ks], zf[ks], acc[t]);
        }
        // ---- v = up + bias + residual(s) in fp32.  The LayerNorm runs on the UNROUNDED sum (round 4): rounding v to its bf16 storage first
        // (rounds 1 - 3: "what backward will re-read") was one rounding of an O(1) tensor per sub-layer that the reference's autocast path
        // does not have -- its LayerNorm reads the fp32 sum -- and cost 1.2 - 2.2x its distance from fp32 on scores / embeddings at BERT-base
        // (tests/test_parity_base_gpu.py).  Backward: xhat = (v_bf16 - mean) rstd or (y - beta) / gamma, either way within bf16 rounding.
        float vv[KS][8];
        float s1 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float af[8], of[8], bu8[8];
            Elem<bf16_t>::unpack(a_cur[s], af);
            if constexpr (R32) {
                *reinterpret_cast<uint4*>(of) = o_cur[2 * s];
                *reinterpret_cast<uint4*>(of + 4) = o_cur[2 * s + 1];
            } else if constexpr (RM == 2) {
                const uint32_t hw[4] = {o_cur[2 * s].x, o_cur[2 * s].y, o_cur[2 * s].z, o_cur[2 * s].w}, lw[2] = {o_cur[2 * s + 1].x, o_cur[2 * s + 1].y};
#pragma unroll
                for (int j = 0; j < 8; ++j) of[j] = lo8_join(hw[j >> 1], lw[j >> 2], j);
            } else if constexpr (RM == 3) {
                const uint32_t hw[4] = {o_cur[2 * s].x, o_cur[2 * s].y, o_cur[2 * s].z, o_cur[2 * s].w};
#pragma unroll
                for (int j = 0; j < 8; ++j) of[j] = lo4_join(hw[j >> 1], o_cur[2 * s + 1].x, j);
            } else {
                Elem<bf16_t>::unpack(o_cur[s], of);
            }
            *reinterpret_cast<float4*>(bu8) = *reinterpret_cast<const float4*>(&par[0][cl + s * 32]);
            *reinterpret_cast<float4*>(bu8 + 4) = *reinterpret_cast<const float4*>(&par[0][cl + s * 32 + 4]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = acc[2 * s + (j >> 2)][j & 3] + bu8[j] + of[j];
                if (p.a_in_resid) x += af[j];
                vv[s][j] = x;
            }
            if (p.v && !(A4R_AD_ABL & 16)) *reinterpret_cast<uint4*>(p.v + row * p.ldv + cl + s * 32) = Elem<bf16_t>::pack(vv[s]);      // (null: backward rebuilds xhat from y, see FY below)
#pragma unroll
            for (int j = 0; j < 8; ++j) s1 += vv[s][j];
        }
        // ---- row statistics: this wave's (mean, centred sum of squares) over CW columns; exact combination across waves
        const float mean_w = kg_sum(s1) * (1.f / CW);
        float q = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = vv[s][j] - mean_w; q += d * d; }
        q = kg_sum(q);
        if (kg == 0) { red[wave][fr][0] = mean_w; red[wave][fr][1] = q; }
        A4R_ADF_ST(5)
        A4R_LDS_BARRIER();
        A4R_ADF_ST(6)
        float mean = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) mean += red[w][fr][0];
        mean *= (1.f / NW);
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const float d = red[w][fr][0] - mean; m2 += red[w][fr][1] + (float)CW * d * d; }
        const float rstd = rsqrtf(m2 * (1.f / H) + p.eps);
        if (wave == 0 && kg == 0) { p.stats[2 * row] = mean; p.stats[2 * row + 1] = rstd; }
        float yv[KS][8];
        float am = 0.f;
        LoWords<KS> ylw;                                     // (nibble plane of y: the lane's KS words, stored together after the loop)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float g8[8], b8[8];
            *reinterpret_cast<float4*>(g8) = *reinterpret_cast<const float4*>(&par[1][cl + s * 32]);
            *reinterpret_cast<float4*>(g8 + 4) = *reinterpret_cast<const float4*>(&par[1][cl + s * 32 + 4]);
            *reinterpret_cast<float4*>(b8) = *reinterpret_cast<const float4*>(&par[2][cl + s * 32]);
            *reinterpret_cast<float4*>(b8 + 4) = *reinterpret_cast<const float4*>(&par[2][cl + s * 32 + 4]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { yv[s][j] = (vv[s][j] - mean) * rstd * g8[j] + b8[j]; am = fmaxf(am, fabsf(yv[s][j])); }
            if (p.y && (!(A4R_AD_ABL & 16) || yv[s][0] == 123.456f)) *reinterpret_cast<uint4*>(p.y + row * p.ldy + cl + s * 32) = Elem<bf16_t>::pack(yv[s]);
            if (p.ylo) {                                     // the byte plane of y: offsets from the bf16 values just stored
                const uint4 pk = Elem<bf16_t>::pack(yv[s]);
                const uint32_t hw[4] = {pk.x, pk.y, pk.z, pk.w};
                if (p.lo4) ylw.w[s] = lo4_split8(yv[s], hw);
                else *reinterpret_cast<uint2*>(p.ylo + row * p.ldylo + cl + s * 32) = make_uint2(lo8_split4(yv[s], 0, hw[0], hw[1]), lo8_split4(yv[s], 1, hw[2], hw[3]));
            }
            if (p.y32) {
                *reinterpret_cast<float4*>(p.y32 + row * p.ldy32 + cl + s * 32) = make_float4(yv[s][0], yv[s][1], yv[s][2], yv[s][3]);
                *reinterpret_cast<float4*>(p.y32 + row * p.ldy32 + cl + s * 32 + 4) = make_float4(yv[s][4], yv[s][5], yv[s][6], yv[s][7]);
            }
        }
        if (p.ylo && p.lo4) *reinterpret_cast<LoWords<KS>*>(p.ylo + row * p.ldylo + (wave * 4 + kg) * KS * 4) = ylw;
        if (p.y8) {      // e4m3 row = y * 448 / max|y| from the fp32 values (one rounding; the same arithmetic as a4r_ln_fwd_fp8), scale = max|y| / 448
            am = fmaxf(am, __shfl_xor(am, 16, 64));
            am = fmaxf(am, __shfl_xor(am, 32, 64));
            if (kg == 0) red8[wave][fr] = am;
            A4R_LDS_BARRIER();
            am = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) am = fmaxf(am, red8[w][fr]);
            const float inv = am > 0.f ? __fdiv_rn(448.f, am) : 0.f;
            if (wave == 0 && kg == 0) p.ys[row] = am > 0.f ? __fdiv_rn(am, 448.f) : 1.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                int lo = 0, hi = 0;
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(yv[s][0] * inv, yv[s][1] * inv, lo, false);
                lo = __builtin_amdgcn_cvt_pk_fp8_f32(yv[s][2] * inv, yv[s][3] * inv, lo, true);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(yv[s][4] * inv, yv[s][5] * inv, hi, false);
                hi = __builtin_amdgcn_cvt_pk_fp8_f32(yv[s][6] * inv, yv[s][7] * inv, hi, true);
                *reinterpret_cast<uint2*>(p.y8 + row * p.ld8 + cl + s * 32) = make_uint2((unsigned)lo, (unsigned)hi);
            }
        }
        A4R_ADF_ST(7)
        if (it == 0) { A4R_ADF_LAUNCH(4) }
#pragma unroll
        for (int s = 0; s < KS; ++s) a_cur[s] = a_nxt[s];
#pragma unroll
        for (int s = 0; s < KS * OP; ++s) o_cur[s] = o_nxt[s];
    }
    A4R_ADF_LAUNCH(2)
#undef A4R_AD_LOAD_O
}

struct AdBwdArgs {
    const bf16_t* dy; const bf16_t* v; const bf16_t* dres; int lddy, ldv, lddres;
    const float* stats; const float* gamma; const float* beta;      // beta != null (FY): `v` holds y = LN(v), xhat = (y - beta) / gamma
    const bf16_t* zp; int act;
    const bf16_t* WuT; const bf16_t* WdT; int inner_res;
    bf16_t* dv; bf16_t* dzp; bf16_t* dh; int lddv, lddh;
    float* dgamma; float* dbeta; float* dbias; float* dbd;
    int M, bias_total, wfrag;
    uint64_t seed; uint32_t site, thr16; float keep_scale;
};

// WGB: accumulate dgamma / dbeta (trainable LayerNorm: Pfeiffer's LN_new, --finetune_layernorm); WDB: dbias = column sums of dv
// BEFORE dres is added (the up-projection bias sits inside the LayerNorm input).
// DRES: a gradient arrives along the residual stream as well (pre-LN towers) and is read row by row like dy and v.
// Every global load and store of the tile loop is UNCONDITIONAL (template flags, the timing-only A4R_AD_ABL macro): with run-time tests
// around them (`if (p.dres)`, an ablation knob read from the arguments) the number of operations behind a load differs by path and hipcc falls
// back to s_waitcnt vmcnt(0) -- at the top of every tile and in front of the act' operand -- which drains the tile's own stores and the
// requests for the next tiles every time (adapter_ln_bwd 67 -> 74 us in the step when the knob went in).
// FY: the forward did not keep v (the LayerNorm's input sum) but only y = LN(v), which the next GEMM reads anyway: xhat = (y - beta) / gamma
// per column (1 / gamma and -beta / gamma sit in LDS next to gamma), rstd from the saved statistics -- 62 MB less written per forward launch.
template <int CW, int NW, bool WGB, bool WDB, bool DRES, bool FY = false>
__global__ void __launch_bounds__(NW * 64) adapter_ln_bwd_kernel(const AdBwdArgs p) {
    constexpr int KS = CW / 32, H = CW * NW, NT = NW * 64, EPT = 1024 / NT;
    __shared__ __attribute__((aligned(16))) float zpart[NW][16][ZLD];
    __shared__ __attribute__((aligned(16))) char zbf[16 * 128];
    __shared__ float red[NW][16][2];
    __shared__ __attribute__((aligned(16))) float par[H];                 // gamma; re-used for the column sums at the end
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kg = lane >> 4;
    const int c0 = wave * CW;
    const int cl = c0 + kg * 8;
    A4R_ADB_LAUNCH(0)
    __shared__ __attribute__((aligned(16))) float par_fy[FY ? 2 : 1][FY ? H : 4];      // 1 / gamma, -beta / gamma
    // the per-column parameters are REQUESTED here and written to LDS behind every other request of the prologue (round 5: written at once, the
    // wave sat out a whole memory round trip before it asked for its weight fragments and first rows: `entry -> every request issued` 4.4 us)
    constexpr int NPAR = (H + NT - 1) / NT;
    float g_r[NPAR], b_r[NPAR];
#pragma unroll
    for (int i = 0; i < NPAR; ++i) {
        const int c = tid + i * NT;
        g_r[i] = c < H ? p.gamma[c] : 1.f;
        b_r[i] = (FY && c < H) ? p.beta[c] : 0.f;
    }

    // dz = dv . Wu (contraction over H: WuT [64, H]) keeps its fragments in registers; dh = dzp . Wd (contraction over 64:
    // WdT [H, 64]) keeps them in LDS in fragment order (a wave's private 2 KS x 2 KiB, read back as conflict-free
    // ds_read_b128: 12 reads per lane per tile) -- the column-sum accumulators need the 48 registers more.
    __shared__ __attribute__((aligned(16))) uint4 wdl[NW][2 * KS][2][64];
    uint4 wu[KS][4];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            wu[s][nt] = *reinterpret_cast<const uint4*>(p.wfrag ? p.WuT + (size_t)(((wave * KS + s) * 4 + nt) * 64 + lane) * 8      // (fragment order: see the forward)
                                                                 : p.WuT + (size_t)(nt * 16 + fr) * H + cl + s * 32);
    // (round 5) the image is written by LDS-DMA: a lane's 16 bytes land at piece + lane * 16, which IS wdl[wave][2 s + h][ks][lane] -- no trip through
    // the registers, no wait in front of a ds_write: the launch's prologue 8.6 -> ? us (tools/adapter_launch_timeline.py bwd).  hipcc does not see
    // these requests: the wait in front of the prologue's barrier is counted by hand (A4R_AD_WDL_DMA=0 at compile time: the register form, A/B)
#ifndef A4R_AD_WDL_DMA
#define A4R_AD_WDL_DMA 1
#endif
#if A4R_AD_WDL_DMA
    {
        const uint32_t img = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&wdl[wave][0][0][0];
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const uint32_t voff = p.wfrag ? (uint32_t)((((((wave * KS + s) * 2 + h) * 2 + ks) * 64 + lane) * 8) * 2)
                                                  : (uint32_t)(((c0 + s * 32 + (fr >> 2) * 8 + h * 4 + (fr & 3)) * 64 + ks * 32 + kg * 8) * 2);
                    uint32_t keep;
                    asm volatile(
                        "s_mov_b32 %0, m0\n\t"
                        "s_mov_b32 m0, %3\n\t"
                        "s_nop 0\n\t"
                        "global_load_lds_dwordx4 %1, %2\n\t"
                        "s_mov_b32 m0, %0"
                        : "=&s"(keep)
                        : "v"(voff), "s"(p.WdT), "s"(img + (uint32_t)(((2 * s + h) * 2 + ks) * 1024))
                        : "memory");
                }
    }
#else
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wdl[wave][2 * s + h][ks][lane] =
                    *reinterpret_cast<const uint4*>(p.wfrag ? p.WdT + (size_t)((((wave * KS + s) * 2 + h) * 2 + ks) * 64 + lane) * 8
                                                            : p.WdT + (size_t)(c0 + s * 32 + (fr >> 2) * 8 + h * 4 + (fr & 3)) * 64 + ks * 32 + kg * 8);
#endif

    const int e0 = tid * EPT, rrow = e0 >> 6, rzd = e0 & 63;
    float sd[EPT];                          // column sums of dzp over this workgroup's tiles (the down-projection's bias gradient)
#pragma unroll
    for (int i = 0; i < EPT; ++i) sd[i] = 0.f;
    float sg[KS][8], sb[KS][8], sv[KS][8];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) { sg[s][j] = 0.f; sb[s][j] = 0.f; sv[s][j] = 0.f; }

    const int ntiles = p.M / 16;
    // rows are requested TWO tiles ahead where the registers allow it (DEEP): with ~12 MB in flight chip-wide a request takes longer than the
    // ~5 us of arithmetic of one tile.  The thread's EPT pre-activations (operand of act') and the DRES rows travel with them.
    constexpr bool DEEP = CW < 96 || (!WGB && !DRES);
    typedef typename std::conditional<EPT == 2, uint32_t, uint2>::type zp_t;
    struct Rows { uint4 d[KS], v[KS], r[DRES ? KS : 1]; float2 st; zp_t zp; };
    auto request = [&](Rows& R, int t) {
        const size_t row = (size_t)t * 16 + fr;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (A4R_AD_ABL & 2) { R.d[s] = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); R.v[s] = R.d[s]; if constexpr (DRES) R.r[s] = R.d[s]; continue; }
            R.d[s] = *reinterpret_cast<const uint4*>(p.dy + row * p.lddy + cl + s * 32);
            R.v[s] = *reinterpret_cast<const uint4*>(p.v + row * p.ldv + cl + s * 32);
            if constexpr (DRES) R.r[s] = *reinterpret_cast<const uint4*>(p.dres + row * p.lddres + cl + s * 32);
        }
        R.st = (A4R_AD_ABL & 2) ? make_float2(0.f, 1.f) : *reinterpret_cast<const float2*>(p.stats + 2 * row);
        R.zp = *reinterpret_cast<const zp_t*>(p.zp + ((size_t)t * 16 + rrow) * 64 + rzd);
    };
    const int G = (int)gridDim.x;
    Rows cur, nxt, nn;
    request(cur, (int)blockIdx.x);
    if constexpr (DEEP) request(nxt, (int)blockIdx.x + G < ntiles ? (int)blockIdx.x + G : (int)blockIdx.x);
#pragma unroll
    for (int i = 0; i < NPAR; ++i) {
        const int c = tid + i * NT;
        if (c < H) {
            par[c] = g_r[i];
            if constexpr (FY) { const float ig = 1.f / g_r[i]; par_fy[0][c] = ig; par_fy[1][c] = -b_r[i] * ig; }
        }
    }
    A4R_ADB_LAUNCH(5)
#if A4R_AD_WDL_DMA
    // the 4 KS pieces of the image have landed once at most the requests issued BEHIND them are outstanding (vmcnt counts in order): every wave
    // issues the first tile's rows (2 KS + 2, + KS with a residual-stream gradient) and, DEEP, the second tile's; the parameter loads in between
    // may be skipped by a whole wave and are not counted (a lower bound only over-waits)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((2 * KS + 2 + (DRES ? KS : 0)) * (DEEP ? 2 : 1)) : "memory");
#endif
    A4R_LDS_BARRIER();
    A4R_ADB_LAUNCH(1)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        {
            const int ahead = (DEEP ? 2 : 1) * G;
            request(nn, tile + ahead < ntiles ? tile + ahead : tile);       // (past the end: re-request this tile, the schedule stays static)
        }
        A4R_AD_ST(0)
        const size_t row = (size_t)tile * 16 + fr;
        const float mean = cur.st.x, rstd = cur.st.y;
        // ---- LayerNorm backward, part 1: xhat, g = dy * gamma, the two row means
        float xh[KS][8], g[KS][8];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float d8[8], ga8[8];
            Elem<bf16_t>::unpack(cur.d[s], d8);
            Elem<bf16_t>::unpack(cur.v[s], xh[s]);
            *reinterpret_cast<float4*>(ga8) = *reinterpret_cast<const float4*>(&par[cl + s * 32]);
            *reinterpret_cast<float4*>(ga8 + 4) = *reinterpret_cast<const float4*>(&par[cl + s * 32 + 4]);
            float ig8[8], ib8[8];
            if constexpr (FY) {
                *reinterpret_cast<float4*>(ig8) = *reinterpret_cast<const float4*>(&par_fy[0][cl + s * 32]);
                *reinterpret_cast<float4*>(ig8 + 4) = *reinterpret_cast<const float4*>(&par_fy[0][cl + s * 32 + 4]);
                *reinterpret_cast<float4*>(ib8) = *reinterpret_cast<const float4*>(&par_fy[1][cl + s * 32]);
                *reinterpret_cast<float4*>(ib8 + 4) = *reinterpret_cast<const float4*>(&par_fy[1][cl + s * 32 + 4]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = FY ? fmaf(xh[s][j], ig8[j], ib8[j]) : (xh[s][j] - mean) * rstd;
                if constexpr (WGB) { sg[s][j] += d8[j] * x; sb[s][j] += d8[j]; }
                const float gg = d8[j] * ga8[j];
                xh[s][j] = x;
                g[s][j] = gg;
                c1 += gg;
                c2 += gg * x;
            }
        }
        c1 = kg_sum(c1);
        c2 = kg_sum(c2);
        if (kg == 0) { red[wave][fr][0] = c1; red[wave][fr][1] = c2; }
        A4R_AD_ST(1)
        A4R_LDS_BARRIER();
        A4R_AD_ST(2)
        c1 = 0.f; c2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { c1 += red[w][fr][0]; c2 += red[w][fr][1]; }
        c1 *= (1.f / H);
        c2 *= (1.f / H);
        // ---- part 2: dv (+ dres), rounded to its bf16 storage = the operand of dz = dv . Wu
        uint4 dvp[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            float d8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                d8[j] = rstd * (g[s][j] - c1 - xh[s][j] * c2);
                if constexpr (WDB) { if (!p.bias_total) sv[s][j] += d8[j]; }
            }
            if constexpr (DRES) {
                float r8[8];
                Elem<bf16_t>::unpack(cur.r[s], r8);
#pragma unroll
                for (int j = 0; j < 8; ++j) d8[j] += r8[j];
            }
            if constexpr (WDB) {             // pre-LN towers (ViT): the bias sits in the residual stream, its gradient is the TOTAL dv
                if (p.bias_total) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) sv[s][j] += d8[j];
                }
            }
            dvp[s] = Elem<bf16_t>::pack(d8);
            if (!(A4R_AD_ABL & 1)) *reinterpret_cast<uint4*>(p.dv + row * p.lddv + cl + s * 32) = dvp[s];
        }
        // ---- dz partial over this wave's columns, summed through LDS; dzp = dz * act'(zp)
        f32x4_t zacc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) zacc[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) Mma<bf16_t>::mma(wu[s][nt], dvp[s], zacc[nt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4_t*>(&zpart[wave][fr][nt * 16 + kg * 4]) = zacc[nt];
        A4R_AD_ST(3)
        A4R_LDS_BARRIER();
        A4R_AD_ST(4)
        {
            float s_[EPT];
            reduce_partials<NW, EPT>(zpart, rrow, rzd, s_);
            const size_t gi = ((size_t)tile * 16 + rrow) * 64 + rzd;
            float pre[EPT];
            if constexpr (EPT == 2) {
                const uint32_t u = cur.zp;
                pre[0] = bf16_bits_to_f32(u & 0xffffu); pre[1] = bf16_bits_to_f32(u >> 16);
            } else {
                const uint2 u = cur.zp;
                pre[0] = bf16_bits_to_f32(u.x & 0xffffu); pre[1] = bf16_bits_to_f32(u.x >> 16);
                pre[2] = bf16_bits_to_f32(u.y & 0xffffu); pre[3] = bf16_bits_to_f32(u.y >> 16);
            }
#pragma unroll
            for (int i = 0; i < EPT; ++i) { s_[i] *= act_bwd(pre[i], p.act); sd[i] += s_[i]; }
            store_bf16_n<EPT>(p.dzp + gi, s_);
            store_bf16_n<EPT>(reinterpret_cast<bf16_t*>(zbf + zbf_off(rrow, rzd)), s_);
        }
        A4R_AD_ST(5)
        A4R_LDS_BARRIER();
        A4R_AD_ST(6)
        // ---- dh = dzp . Wd (+ dv: the adapter's inner residual), through the dense output's dropout mask
        uint4 zf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) zf[ks] = *reinterpret_cast<const uint4*>(zbf + fr * 128 + (((ks * 4 + kg) ^ ((fr >> 1) & 7)) << 4));
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            f32x4_t acc[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                acc[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) Mma<bf16_t>::mma(wdl[wave][2 * s + h][ks][lane], zf[ks], acc[h]);
            }
            float o8[8], dv8[8];
            Elem<bf16_t>::unpack(dvp[s], dv8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o8[j] = acc[j >> 2][j & 3] + (p.inner_res ? dv8[j] : 0.f);
            if (p.thr16) {
                const uint64_t el = (uint64_t)row * (uint64_t)H + (uint64_t)(cl + s * 32);
                const uint64_t h0 = a4r_hash64(p.seed, p.site, el >> 2), h1 = a4r_hash64(p.seed, p.site, (el >> 2) + 1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o8[e] = (((uint32_t)(h0 >> (16 * e)) & 0xffffu) >= p.thr16) ? o8[e] * p.keep_scale : 0.f;
                    o8[e + 4] = (((uint32_t)(h1 >> (16 * e)) & 0xffffu) >= p.thr16) ? o8[e + 4] * p.keep_scale : 0.f;
                }
            }
            if (!(A4R_AD_ABL & 1)) *reinterpret_cast<uint4*>(p.dh + row * p.lddh + cl + s * 32) = Elem<bf16_t>::pack(o8);
            else if (o8[0] == 12345.f) p.dbias[0] = o8[1];
        }
        A4R_AD_ST(7)
        if (tile == (int)blockIdx.x) { A4R_ADB_LAUNCH(4) }
        if constexpr (DEEP) { cur = nxt; nxt = nn; }
        else cur = nn;
    }
    A4R_ADB_LAUNCH(2)
    // ---- db_down = column sums of dzp: thread t holds columns (t EPT) & 63 of row (t EPT) >> 6 -> through LDS, one atomic per column
    if (A4R_AD_ABL & 4) return;                 // (timing only: no column-sum flush)
    if (p.dbd) {
        A4R_LDS_BARRIER();
#pragma unroll
        for (int i = 0; i < EPT; ++i) zpart[0][rrow][rzd + i] = sd[i];
        A4R_LDS_BARRIER();
        if (tid < 64) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += zpart[0][r][tid];
            atomicAdd(p.dbd + tid, t);
        }
    }
    // ---- column sums: over the 16 rows a lane group holds (lane & 15), then one atomic per column per workgroup
    if constexpr (WGB || WDB) {
        auto flush = [&](float (&acc)[KS][8], float* dst) {
            A4R_LDS_BARRIER();                                            // par[] free (gamma no longer read / previous flush done)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = group16_sum(acc[s][j]);
                    if (fr == 0) par[cl + s * 32 + j] = t;
                }
            A4R_LDS_BARRIER();
            if (dst)
                for (int c = tid; c < H; c += NT) atomicAdd(dst + c, par[c]);
        };
        if constexpr (WGB) { flush(sg, p.dgamma); flush(sb, p.dbeta); }
        if constexpr (WDB) flush(sv, p.dbias);
    }
}

inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

template <int CW, int NW>
int launch_fwd(hipStream_t s, const AdFwdArgs& a, int grid) {
    if (a.O32) hipLaunchKernelGGL((adapter_ln_fwd_kernel<CW, NW, 1>), dim3(grid), dim3(NW * 64), 0, s, a);
    else if (a.Olo && a.lo4) hipLaunchKernelGGL((adapter_ln_fwd_kernel<CW, NW, 3>), dim3(grid), dim3(NW * 64), 0, s, a);
    else if (a.Olo) hipLaunchKernelGGL((adapter_ln_fwd_kernel<CW, NW, 2>), dim3(grid), dim3(NW * 64), 0, s, a);
    else hipLaunchKernelGGL((adapter_ln_fwd_kernel<CW, NW, 0>), dim3(grid), dim3(NW * 64), 0, s, a);
    return a4r_launch_status();
}
template <int CW, int NW, bool DRES>
int launch_bwd_d(hipStream_t s, const AdBwdArgs& a, int grid) {
    const bool wgb = a.dgamma || a.dbeta, wdb = a.dbias != nullptr;
    if (a.beta) {                                  // xhat from y: instantiated for the frozen-LayerNorm post-LN form the text tower runs
        if constexpr (!DRES) {
            if (!wgb && wdb) { hipLaunchKernelGGL((adapter_ln_bwd_kernel<CW, NW, false, true, false, true>), dim3(grid), dim3(NW * 64), 0, s, a); return a4r_launch_status(); }
            if (!wgb && !wdb) { hipLaunchKernelGGL((adapter_ln_bwd_kernel<CW, NW, false, false, false, true>), dim3(grid), dim3(NW * 64), 0, s, a); return a4r_launch_status(); }
        }
        return A4R_EINVAL;
    }
    if (wgb && wdb) hipLaunchKernelGGL((adapter_ln_bwd_kernel<CW, NW, true, true, DRES>), dim3(grid), dim3(NW * 64), 0, s, a);
    else if (wgb) hipLaunchKernelGGL((adapter_ln_bwd_kernel<CW, NW, true, false, DRES>), dim3(grid), dim3(NW * 64), 0, s, a);
    else if (wdb) hipLaunchKernelGGL((adapter_ln_bwd_kernel<CW, NW, false, true, DRES>), dim3(grid), dim3(NW * 64), 0, s, a);
    else hipLaunchKernelGGL((adapter_ln_bwd_kernel<CW, NW, false, false, DRES>), dim3(grid), dim3(NW * 64), 0, s, a);
    return a4r_launch_status();
}
template <int CW, int NW>
int launch_bwd(hipStream_t s, const AdBwdArgs& a, int grid) {
    return a.dres ? launch_bwd_d<CW, NW, true>(s, a, grid) : launch_bwd_d<CW, NW, false>(s, a, grid);
}

}  // namespace

int a4r_cu_count();       // a4r_gemm256.hip

extern "C" int a4r_adapter_ln_fwd(void* stream, const void* A, int lda, const void* R1, int ldr1, const void* R2, int ldr2,
                                  const void* Wd, const float* bd, const void* Wu, const float* bu,
                                  const float* gamma, const float* beta, float eps, int act,
                                  void* zp, void* z, void* v, int ldv, void* y, int ldy, float* stats, int M, int H, int d, int dtype,
                                  void* y8, int ld8, float* ys, const float* res32, int ldres32, float* y32, int ldy32, int w_frag) {
    if (!A || !R1 || !Wd || !bd || !Wu || !bu || !gamma || !beta || !zp || !z || (!v && !y) || (!y && !y8) || !stats) return A4R_EINVAL;      // v may be null when y is kept
    const bool lo8 = (w_frag & 2) != 0;                  // res32 / y32 are BYTE planes (the 24-bit residual stream): 8-byte pieces, leading dimension in bytes
    if ((w_frag & 4) && !lo8) return A4R_EINVAL;
    if (lo8) {
        const int hb = (w_frag & 4) ? H / 2 : H, al = (w_frag & 4) ? 4 : 8;      // bytes of a row of the plane; its pieces are 4 (nibbles) / 8 bytes
        if ((res32 && (ldres32 % al || ldres32 < hb || (reinterpret_cast<uintptr_t>(res32) & (al - 1)))) || (y32 && (ldy32 % al || ldy32 < hb || (reinterpret_cast<uintptr_t>(y32) & (al - 1))))) return A4R_EINVAL;
    } else if ((res32 && (ldres32 % 4 || ldres32 < H || misaligned16(res32))) || (y32 && (ldy32 % 4 || ldy32 < H || misaligned16(y32)))) return A4R_EINVAL;
    if (y8 && (!ys || ld8 % 8 || ld8 < H || (reinterpret_cast<uintptr_t>(y8) & 7u))) return A4R_EINVAL;
    if (dtype != A4R_BF16 || d != 64 || M <= 0 || M % 16) return A4R_EINVAL;
    if (lda % 8 || ldr1 % 8 || (R2 && ldr2 % 8) || (v && ldv % 8) || (y && ldy % 8)) return A4R_EINVAL;
    if (misaligned16(A) || misaligned16(R1) || misaligned16(R2) || misaligned16(Wd) || misaligned16(Wu) || misaligned16(v) || misaligned16(y) ||
        misaligned16(zp) || misaligned16(z) || (reinterpret_cast<uintptr_t>(stats) & 7u))
        return A4R_EINVAL;
    // the residual sum is R1 + R2; the kernel streams TWO tensors: the down-projection input A and one other
    AdFwdArgs a{};
    a.A = reinterpret_cast<const bf16_t*>(A); a.lda = lda;
    if (R1 == A && ldr1 == lda && R2) { a.O = reinterpret_cast<const bf16_t*>(R2); a.ldo = ldr2; a.a_in_resid = 1; }          // Houlsby: up + h + input
    else if (R2 == A && ldr2 == lda) { a.O = reinterpret_cast<const bf16_t*>(R1); a.ldo = ldr1; a.a_in_resid = 1; }             // parallel form
    else if (!R2 && R1 != A) { a.O = reinterpret_cast<const bf16_t*>(R1); a.ldo = ldr1; a.a_in_resid = 0; }                     // Compacter / Pfeiffer
    else return A4R_EINVAL;
    a.Wd = reinterpret_cast<const bf16_t*>(Wd); a.Wu = reinterpret_cast<const bf16_t*>(Wu); a.bd = bd; a.bu = bu;
    a.gamma = gamma; a.beta = beta; a.eps = eps; a.act = act;
    a.zp = reinterpret_cast<bf16_t*>(zp); a.z = reinterpret_cast<bf16_t*>(z); a.v = reinterpret_cast<bf16_t*>(v); a.y = reinterpret_cast<bf16_t*>(y);
    a.ldv = ldv; a.ldy = ldy; a.stats = stats; a.M = M;
    a.y8 = reinterpret_cast<unsigned char*>(y8); a.ld8 = ld8; a.ys = ys;
    a.lo4 = (w_frag & 4) != 0;
    if (lo8) { a.Olo = reinterpret_cast<const signed char*>(res32); a.ldolo = ldres32; a.ylo = reinterpret_cast<signed char*>(y32); a.ldylo = ldy32; }
    else { a.O32 = res32; a.ldo32 = ldres32; a.y32 = y32; a.ldy32 = ldy32; }      // (res32: the fp32 twin of the residual operand that is not A)
    a.wfrag = (w_frag & 1) != 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int ntiles = M / 16, ncu = a4r_cu_count();
    const int grid = ntiles < ncu ? ntiles : ncu;
    switch (H) {
        case 128: return launch_fwd<32, 4>(s, a, grid);
        case 256: return launch_fwd<32, 8>(s, a, grid);
        case 512: return launch_fwd<64, 8>(s, a, grid);
        case 768: return launch_fwd<96, 8>(s, a, grid);
        case 1024: return launch_fwd<128, 8>(s, a, grid);
        default: return A4R_EINVAL;
    }
}

extern "C" int a4r_adapter_ln_bwd(void* stream, const void* dy, int lddy, const void* v, int ldv, const float* stats, const float* gamma,
                                  const void* dres, int lddres, const void* zp, int act, const void* WuT, const void* WdT, int inner_res,
                                  void* dv, int lddv, void* dzp, void* dh, int lddh, float* dgamma, float* dbeta, float* dbias,
                                  int M, int H, int d, int dtype, float drop_p, uint32_t drop_site, uint64_t drop_seed, float* dbd, int flags,
                                  const float* beta_y) {
    if (!dy || !v || !stats || !gamma || !zp || !WuT || !WdT || !dv || !dzp || !dh) return A4R_EINVAL;
    if (dtype != A4R_BF16 || d != 64 || M <= 0 || M % 16) return A4R_EINVAL;
    if (lddy % 8 || ldv % 8 || (dres && lddres % 8) || lddv % 8 || lddh % 8) return A4R_EINVAL;
    if (misaligned16(dy) || misaligned16(v) || misaligned16(dres) || misaligned16(WuT) || misaligned16(WdT) || misaligned16(dv) || misaligned16(dh) ||
        misaligned16(zp) || misaligned16(dzp) || (reinterpret_cast<uintptr_t>(stats) & 7u))
        return A4R_EINVAL;
    AdBwdArgs a{};
    a.dy = reinterpret_cast<const bf16_t*>(dy); a.v = reinterpret_cast<const bf16_t*>(v); a.dres = reinterpret_cast<const bf16_t*>(dres);
    a.lddy = lddy; a.ldv = ldv; a.lddres = lddres; a.stats = stats; a.gamma = gamma; a.beta = beta_y;
    a.zp = reinterpret_cast<const bf16_t*>(zp); a.act = act;
    a.WuT = reinterpret_cast<const bf16_t*>(WuT); a.WdT = reinterpret_cast<const bf16_t*>(WdT); a.inner_res = inner_res;
    a.dv = reinterpret_cast<bf16_t*>(dv); a.dzp = reinterpret_cast<bf16_t*>(dzp); a.dh = reinterpret_cast<bf16_t*>(dh);
    a.lddv = lddv; a.lddh = lddh; a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias; a.dbd = dbd; a.M = M; a.bias_total = flags & 1; a.wfrag = (flags >> 1) & 1;
    a.seed = drop_seed; a.site = drop_site; a.thr16 = a4r_thr16(drop_p); a.keep_scale = a4r_keep_scale(drop_p);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int ntiles = M / 16, ncu = a4r_cu_count();
    const int grid = ntiles < ncu ? ntiles : ncu;
    switch (H) {
        case 128: return launch_bwd<32, 4>(s, a, grid);
        case 256: return launch_bwd<32, 8>(s, a, grid);
        case 512: return launch_bwd<64, 8>(s, a, grid);
        case 768: return launch_bwd<96, 8>(s, a, grid);
        // H = 1024: the fragment image of Wd (128 KB) + the partial tiles exceed the 160 KB of LDS -> the three-launch form
        default: return A4R_EINVAL;
    }
}
