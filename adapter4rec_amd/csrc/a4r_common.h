// Common device helpers for the a4r HIP kernels (gfx950 / CDNA4 only).
//
// Everything matmul-shaped in this library is expressed through ONE tile primitive:
// a 16x16 output tile accumulated from "16-byte chunks": lane l of the wave supplies, for
// operand A, 16 bytes of row (l & 15) at chunk (l >> 4) of the current K step, and the same
// for operand B (B is always addressed [n][k], i.e. the "NT" form y = x W^T).
//   T = __bf16 : chunk = 8 elements, one v_mfma_f32_16x16x32_bf16 covers K = 32
//   T = float  : chunk = 4 elements, four v_mfma_f32_16x16x4_f32 cover K = 16
//                (step s uses element s of every lane's chunk: k = 4*kg + s, a bijection on 16 k's)
// so bf16 and fp32 instantiations of a kernel share all addressing (rows of 16-byte chunks).
// Accumulator layout of the 16x16 tile (both forms): col = lane & 15, row = (lane >> 4) * 4 + reg.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define A4R_DEV __device__ __forceinline__

typedef __bf16 bf16_t;
struct fp8_t { unsigned char v; };      // OCP e4m3fn (gfx950's fp8; NOT MI300's fnuz), storage tag for the fp8 GEMM operands
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

enum { A4R_BF16 = 0, A4R_F32 = 1, A4R_FP8 = 2 };      // A4R_FP8: OCP e4m3fn operands of a4r_gemm_nt (one byte per element)
enum { A4R_ACT_NONE = 0, A4R_ACT_RELU = 1, A4R_ACT_GELU = 2, A4R_ACT_GELU_TANH = 3, A4R_ACT_LEAKY = 4, A4R_DACT_MULQ8_ = 14, A4R_DACT_MUL_ = 15 };

// ---------------------------------------------------------------- error codes (C ABI)
#ifndef A4R_OK            // (also in include/a4r.h, the public copy)
#define A4R_OK 0
#define A4R_EINVAL (-1)   // bad shape / alignment / dtype
#define A4R_ELAUNCH (-2)  // hipGetLastError() != hipSuccess after launch
#endif

static inline int a4r_launch_status() { return hipGetLastError() == hipSuccess ? A4R_OK : A4R_ELAUNCH; }

// ---------------------------------------------------------------- element traits
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int PER16 = 4;
    static A4R_DEV float ld(const float* p) { return *p; }
    static A4R_DEV void st(float* p, float v) { *p = v; }
    static A4R_DEV void unpack(const uint4& v, float* o) {
        o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
    }
    static A4R_DEV uint4 pack(const float* o) {
        return make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3]));
    }
};
A4R_DEV float bf16_bits_to_f32(unsigned u) { return __uint_as_float(u << 16); }
A4R_DEV unsigned f32_to_bf16_bits(float f) {       // plain cast: v_cvt_pk_bf16_f32, RNE, NaN stays NaN
    bf16_t b = (bf16_t)f;
    return (unsigned)__builtin_bit_cast(unsigned short, b);
}
// two fp32 -> one dword of two bf16 (a in the low half): written as a VECTOR conversion so that it is ONE v_cvt_pk_bf16_f32 with
// the operands in this order.  The scalar form  bits(a) | bits(b) << 16  let the compiler pair conversions by register adjacency and
// re-shuffle the halves afterwards (v_and / v_lshlrev / v_or_b32_sdwa: 10 extra vector instructions per 8 outputs in every epilogue).
typedef __bf16 bf16x2_raw_t __attribute__((ext_vector_type(2)));
typedef float f32x2_raw_t __attribute__((ext_vector_type(2)));
A4R_DEV unsigned pack2_bf16(float a, float b) {
    const f32x2_raw_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_raw_t));
}
template <> struct Elem<bf16_t> {
    static constexpr int PER16 = 8;
    static A4R_DEV float ld(const bf16_t* p) { return (float)(*p); }
    static A4R_DEV void st(bf16_t* p, float v) { *p = (bf16_t)v; }
    static A4R_DEV void unpack(const uint4& v, float* o) {
        o[0] = bf16_bits_to_f32(v.x & 0xffffu); o[1] = bf16_bits_to_f32(v.x >> 16);
        o[2] = bf16_bits_to_f32(v.y & 0xffffu); o[3] = bf16_bits_to_f32(v.y >> 16);
        o[4] = bf16_bits_to_f32(v.z & 0xffffu); o[5] = bf16_bits_to_f32(v.z >> 16);
        o[6] = bf16_bits_to_f32(v.w & 0xffffu); o[7] = bf16_bits_to_f32(v.w >> 16);
    }
    static A4R_DEV uint4 pack(const float* o) {
        return make_uint4(pack2_bf16(o[0], o[1]),
                          pack2_bf16(o[2], o[3]),
                          pack2_bf16(o[4], o[5]),
                          pack2_bf16(o[6], o[7]));
    }
};

// load / store N consecutive elements of T (N a multiple of PER16) as fp32, 16 bytes at a time
template <typename T, int N> A4R_DEV void load_vec(const T* p, float* o) {
#pragma unroll
    for (int i = 0; i < N / Elem<T>::PER16; ++i) {
        uint4 v = *reinterpret_cast<const uint4*>(p + i * Elem<T>::PER16);
        Elem<T>::unpack(v, o + i * Elem<T>::PER16);
    }
}
template <typename T, int N> A4R_DEV void store_vec(T* p, const float* o) {
#pragma unroll
    for (int i = 0; i < N / Elem<T>::PER16; ++i)
        *reinterpret_cast<uint4*>(p + i * Elem<T>::PER16) = Elem<T>::pack(o + i * Elem<T>::PER16);
}

// ---------------------------------------------------------------- the tile primitive
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static constexpr int KSTEP = 32;   // K covered by one chunk step (4 lane groups x 8 elements)
    static A4R_DEV void mma(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static constexpr int KSTEP = 16;
    static A4R_DEV void mma(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};

template <> struct Mma<fp8_t> {
    // a 16-byte chunk holds 16 e4m3 elements: two v_mfma_f32_16x16x32_fp8_fp8 (8 bytes per lane each, K = 32) cover K = 64.  Both
    // operands use the same byte -> contraction-slot map, so the products pair the right elements whatever the slot order.  Same
    // cycles per instruction as the bf16 form: twice the flops per LDS / DMA byte, the same flops per MFMA cycle.
    static constexpr int KSTEP = 64;
    static A4R_DEV void mma(const uint4& a, const uint4& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)(((unsigned long)a.y << 32) | a.x), (long)(((unsigned long)b.y << 32) | b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)(((unsigned long)a.w << 32) | a.z), (long)(((unsigned long)b.w << 32) | b.z), c, 0, 0, 0);
    }
};

// 8 fp32 -> 8 OCP e4m3 bytes (v_cvt_pk_fp8_f32: round to nearest even; the caller keeps |v| <= 448)
A4R_DEV uint2 f32x8_to_fp8(const float (&v)[8]) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}

// Gather one operand chunk DOWN a column of a row-major LDS tile (the "transposed" operand form:
// the contraction index runs along tile rows).  Lane (i = l & 15, kg = l >> 4) collects
// tile[k0 + kg*PER16 + j][col0 + i], j = 0..PER16-1.  stride_b = tile row stride in bytes.
template <typename T> A4R_DEV uint4 gather_chunk(const char* tile, int stride_b, int k0, int col0, int lane) {
    constexpr int PER = Elem<T>::PER16;
    const char* p = tile + (k0 + (lane >> 4) * PER) * stride_b + (col0 + (lane & 15)) * (int)sizeof(T);
    if constexpr (sizeof(T) == 4) {
        uint4 r;
        r.x = *reinterpret_cast<const unsigned*>(p);
        r.y = *reinterpret_cast<const unsigned*>(p + stride_b);
        r.z = *reinterpret_cast<const unsigned*>(p + 2 * stride_b);
        r.w = *reinterpret_cast<const unsigned*>(p + 3 * stride_b);
        return r;
    } else {
        unsigned e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = *reinterpret_cast<const unsigned short*>(p + j * stride_b);
        return make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
}

// ---------------------------------------------------------------- activations
// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, below fp32 GEMM noise): one v_exp + one v_rcp instead of
// the libm erff polynomial ladder.  The GELU epilogue of the two FFN GEMMs touches 2 x M x 3072 elements per layer
// and was ~30 % of those launches with erff.  e = exp(-z*z) is shared with the GELU derivative's Gaussian.
A4R_DEV float erf_as(float z, float e_neg_z2) {
    const float az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * az);   // v_rcp_f32 (1 ulp)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.f - poly * e_neg_z2;                          // >= 0
    return __builtin_copysignf(r, z);                               // (one v_bfi_b32; the compare + select form cost a v_cmp, a hazard nop and a v_cndmask)
}
A4R_DEV float gelu_erf_fwd(float x) {
    const float z = x * 0.70710678118654752440f;
    return 0.5f * x * (1.f + erf_as(z, __expf(-z * z)));
}
A4R_DEV float gelu_erf_bwd(float x) {      // Phi(x) + x * phi(x)
    const float z = x * 0.70710678118654752440f;
    const float e = __expf(-z * z);         // = exp(-x^2 / 2)
    return 0.5f * (1.f + erf_as(z, e)) + x * 0.3989422804014327f * e;
}
A4R_DEV void gelu_erf_both(float x, float& g, float& dg) {     // one exp, one rcp for value and derivative
    const float z = x * 0.70710678118654752440f;
    const float e = __expf(-z * z);
    const float cdf = 0.5f * (1.f + erf_as(z, e));
    g = x * cdf;
    dg = cdf + x * 0.3989422804014327f * e;
}
// The same for N values (N even), sweep by sweep over N / 2 packed pairs.  Element by element hipcc runs each pair's dependent chain of packed-fp32
// instructions to its end before it starts the next pair's, and a packed result consumed by the very next instruction costs a wait state
// (~790 s_nop per 256 x 256 tile epilogue and wave); the empty asm after each sweep ties the pairs together so that a result is consumed N / 2
// instructions later.
typedef float a4r_f2_t __attribute__((ext_vector_type(2)));
#ifndef A4R_GELU_V2
#define A4R_GELU_V2 1      /* 0 (A/B builds): round 3's form -- z = x / sqrt 2 formed, exp through a multiply by log2(e) */
#endif
template <int N>
A4R_DEV void gelu_erf_both_n(float (&x)[N], float (&dg)[N]) {
    static_assert(N % 2 == 0 && N <= 8, "pairs");
    constexpr int P = N / 2;
    a4r_f2_t xv[P], z[P], e[P], t[P], pl[P], xc[P];
#define A4R_TIE(a_) if constexpr (P == 4) asm volatile("" : "+v"(a_[0]), "+v"(a_[1]), "+v"(a_[2]), "+v"(a_[3])); else if constexpr (P == 2) asm volatile("" : "+v"(a_[0]), "+v"(a_[1]));
#if A4R_GELU_V2
    // The constants of the chain folded into each other (round 4: the epilogue is vector-issue bound, every instruction counts):
    //   exp(-x^2 / 2) = exp2(-(c x)^2), c = sqrt(log2(e) / 2): v_exp_f32 IS exp2 -- no multiply by log2(e) per element;
    //   t = 1 / (1 + p |z|), z = x / sqrt 2  ->  1 / (1 + (p / sqrt 2) |x|): one fma on |x|, z itself is never formed;
    //   sign(erf(z)) = sign(x).
#pragma unroll
    for (int j = 0; j < P; ++j) { xv[j] = a4r_f2_t{x[2 * j], x[2 * j + 1]}; z[j] = xv[j] * 0.84932180028801904272f; xc[j] = xv[j] * 0.3989422804014327f; }
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const a4r_f2_t q = -z[j] * z[j];
        e[j] = a4r_f2_t{__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
        t[j] = a4r_f2_t{__builtin_amdgcn_rcpf(__builtin_fmaf(fabsf(xv[j].x), 0.23164190423296285f, 1.f)),
                        __builtin_amdgcn_rcpf(__builtin_fmaf(fabsf(xv[j].y), 0.23164190423296285f, 1.f))};
        z[j] = xv[j];                   // (only the sign is read below)
    }
#else
#pragma unroll
    for (int j = 0; j < P; ++j) { xv[j] = a4r_f2_t{x[2 * j], x[2 * j + 1]}; z[j] = xv[j] * 0.70710678118654752440f; xc[j] = xv[j] * 0.3989422804014327f; }
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const a4r_f2_t q = -z[j] * z[j];
        e[j] = a4r_f2_t{__expf(q.x), __expf(q.y)};
        t[j] = a4r_f2_t{__builtin_amdgcn_rcpf(1.f + 0.3275911f * fabsf(z[j].x)), __builtin_amdgcn_rcpf(1.f + 0.3275911f * fabsf(z[j].y))};
    }
#endif
    A4R_TIE(t)
#pragma unroll
    for (int j = 0; j < P; ++j) pl[j] = t[j] * 1.061405429f + (-1.453152027f);
    A4R_TIE(pl)
#pragma unroll
    for (int j = 0; j < P; ++j) pl[j] = t[j] * pl[j] + 1.421413741f;
    A4R_TIE(pl)
#pragma unroll
    for (int j = 0; j < P; ++j) pl[j] = t[j] * pl[j] + (-0.284496736f);
    A4R_TIE(pl)
#pragma unroll
    for (int j = 0; j < P; ++j) pl[j] = t[j] * pl[j] + 0.254829592f;
    A4R_TIE(pl)
#pragma unroll
    for (int j = 0; j < P; ++j) pl[j] = t[j] * pl[j];
    A4R_TIE(pl)
#pragma unroll
    for (int j = 0; j < P; ++j) pl[j] = 1.f - pl[j] * e[j];
    A4R_TIE(pl)
#pragma unroll
    for (int j = 0; j < P; ++j) pl[j] = a4r_f2_t{__builtin_copysignf(pl[j].x, z[j].x), __builtin_copysignf(pl[j].y, z[j].y)} * 0.5f + 0.5f;      // cdf
    A4R_TIE(pl)
#pragma unroll
    for (int j = 0; j < P; ++j) e[j] = xc[j] * e[j] + pl[j];      // derivative
    A4R_TIE(e)
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const a4r_f2_t g = xv[j] * pl[j];
        x[2 * j] = g.x; x[2 * j + 1] = g.y;
        dg[2 * j] = e[j].x; dg[2 * j + 1] = e[j].y;
    }
#undef A4R_TIE
}
A4R_DEV float act_fwd(float x, int act) {
    switch (act) {
        case A4R_ACT_RELU: return x > 0.f ? x : 0.f;
        case A4R_ACT_GELU: return gelu_erf_fwd(x);
        case A4R_ACT_GELU_TANH: {
            float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
            return 0.5f * x * (1.f + tanhf(u));
        }
        case A4R_ACT_LEAKY: return x > 0.f ? x : 0.01f * x;
        default: return x;
    }
}
A4R_DEV float act_bwd(float x, int act) {   // d act(x) / dx
    switch (act) {
        case A4R_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case A4R_ACT_GELU: return gelu_erf_bwd(x);
        case A4R_ACT_GELU_TANH: {
            float x2 = x * x;
            float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
            float t = tanhf(u);
            float du = 0.7978845608028654f * (1.f + 3.f * 0.044715f * x2);
            return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * du;
        }
        case A4R_ACT_LEAKY: return x > 0.f ? 1.f : 0.01f;
        default: return 1.f;
    }
}

// ---------------------------------------------------------------- counter-based dropout
// One hash yields four 16-bit lots; element e keeps iff lot(e) >= thr16 (thr16 = round(p * 65536)).
// The mask is a pure function of (seed, site, e), so backward regenerates it instead of storing it.
// The hash is two 32-bit integer finalisers (xorshift-multiply, the "lowbias32" constants) over decorrelated 32-bit counters:
// 4 quarter-rate multiplies per four elements.  The splitmix64 it replaces needed three 64-bit multiplies = 12 of them plus
// carries, and the dropout epilogue of a 256 x 256 GEMM tile cost ~7 us of VALU time for ~20 us of main loop.
A4R_DEV uint32_t a4r_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
A4R_DEV uint64_t a4r_hash64(uint64_t seed, uint32_t site, uint64_t idx) {
    const uint32_t s0 = (uint32_t)seed ^ (site * 0x9E3779B1u), s1 = (uint32_t)(seed >> 32) + site * 0x85EBCA77u;
    const uint32_t c = (uint32_t)idx ^ ((uint32_t)(idx >> 32) * 0xC2B2AE3Du);
    const uint32_t lo = a4r_mix32(c * 2u + s0);
    const uint32_t hi = a4r_mix32((c * 2u + 1u) ^ s1 ^ lo);
    return ((uint64_t)hi << 32) | lo;
}
A4R_DEV bool dropout_keep(uint64_t seed, uint32_t site, uint64_t e, uint32_t thr16) {
    uint64_t h = a4r_hash64(seed, site, e >> 2);
    return ((uint32_t)(h >> (16 * (e & 3))) & 0xffffu) >= thr16;
}
static inline uint32_t a4r_thr16(float p) {
    if (p <= 0.f) return 0;
    float t = p * 65536.f + 0.5f;
    return t >= 65535.f ? 65535u : (uint32_t)t;
}
static inline float a4r_keep_scale(float p) { return p > 0.f ? 1.f / (1.f - (float)a4r_thr16(p) / 65536.f) : 1.f; }

// ---------------------------------------------------------------- wave reductions (64 lanes)
A4R_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
A4R_DEV float group16_sum(float v) {   // across the 16 lanes that share (lane >> 4)
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
A4R_DEV float group16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
