// Shared GEMM epilogue: 8 consecutive output columns of one row, fp32 in registers ->
// alpha, bias, [copy of the pre-activation], activation, act'(Pre), [dropout], residuals, [dropout], store.
#pragma once
#include "a4r_common.h"
#include "../../include/a4r.h"

template <typename TO>
struct GemmEpi {
    TO* C; TO* C2; const TO* R1; const TO* R2; const TO* Pre; const float* bias;
    int ldc, ldc2, ldr1, ldr2, ldpre, N, act, dact, drop_first;
    float alpha, keep_scale;
    uint32_t thr16, drop_site;
    uint64_t drop_seed;
};

template <typename TO>
A4R_DEV GemmEpi<TO> make_epi(const a4r_gemm_t& p, uint32_t thr16, float keep_scale) {
    GemmEpi<TO> e;
    e.C = reinterpret_cast<TO*>(p.C); e.C2 = reinterpret_cast<TO*>(p.C2);
    e.R1 = reinterpret_cast<const TO*>(p.R1); e.R2 = reinterpret_cast<const TO*>(p.R2); e.Pre = reinterpret_cast<const TO*>(p.Pre);
    e.bias = p.bias;
    e.ldc = p.ldc; e.ldc2 = p.ldc2; e.ldr1 = p.ldr1; e.ldr2 = p.ldr2; e.ldpre = p.ldpre; e.N = p.N;
    e.act = p.act; e.dact = p.dact; e.drop_first = p.drop_first;
    e.alpha = p.alpha; e.keep_scale = keep_scale; e.thr16 = thr16; e.drop_site = p.drop_site; e.drop_seed = p.drop_seed;
    return e;
}

A4R_DEV void epi_dropout8(float (&v)[8], uint64_t e0, uint64_t seed, uint32_t site, uint32_t thr16, float keep_scale) {
    const uint64_t h0 = a4r_hash64(seed, site, e0 >> 2);          // e0 % 8 == 0
    const uint64_t h1 = a4r_hash64(seed, site, (e0 >> 2) + 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = (((uint32_t)(h0 >> (16 * e)) & 0xffffu) >= thr16) ? v[e] * keep_scale : 0.f;
        v[e + 4] = (((uint32_t)(h1 >> (16 * e)) & 0xffffu) >= thr16) ? v[e + 4] * keep_scale : 0.f;
    }
}

// v: raw accumulator values of columns gcol .. gcol+7 of row grow; bias8: the 8 bias values (zeros if none)
template <typename TO>
A4R_DEV void epilogue8(float (&v)[8], const float (&bias8)[8], size_t grow, int gcol, const GemmEpi<TO>& e) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = v[i] * e.alpha + bias8[i];
    if (e.C2) store_vec<TO, 8>(e.C2 + grow * e.ldc2 + gcol, v);
    if (e.act != A4R_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = act_fwd(v[i], e.act);
    }
    if (e.dact != A4R_ACT_NONE) {
        float pre[8];
        load_vec<TO, 8>(e.Pre + grow * e.ldpre + gcol, pre);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= act_bwd(pre[i], e.dact);
    }
    const uint64_t e0 = (uint64_t)grow * (uint64_t)e.N + (uint64_t)gcol;
    if (e.thr16 && e.drop_first) epi_dropout8(v, e0, e.drop_seed, e.drop_site, e.thr16, e.keep_scale);
    if (e.R1) {
        float t[8];
        load_vec<TO, 8>(e.R1 + grow * e.ldr1 + gcol, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += t[i];
    }
    if (e.R2) {
        float t[8];
        load_vec<TO, 8>(e.R2 + grow * e.ldr2 + gcol, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += t[i];
    }
    if (e.thr16 && !e.drop_first) epi_dropout8(v, e0, e.drop_seed, e.drop_site, e.thr16, e.keep_scale);
    store_vec<TO, 8>(e.C + grow * e.ldc + gcol, v);
}

// ---- 4-column form (one lane's accumulator registers of a TRANSPOSED 16x16 MFMA tile: 4 consecutive columns of one row)
template <typename T> A4R_DEV void load4(const T* p, float* o) {
    if constexpr (sizeof(T) == 4) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    } else {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = bf16_bits_to_f32(v.x & 0xffffu); o[1] = bf16_bits_to_f32(v.x >> 16);
        o[2] = bf16_bits_to_f32(v.y & 0xffffu); o[3] = bf16_bits_to_f32(v.y >> 16);
    }
}
template <typename T> A4R_DEV void store4(T* p, const float* o) {
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        *reinterpret_cast<uint2*>(p) = make_uint2(f32_to_bf16_bits(o[0]) | (f32_to_bf16_bits(o[1]) << 16),
                                                  f32_to_bf16_bits(o[2]) | (f32_to_bf16_bits(o[3]) << 16));
    }
}

template <typename TO>
A4R_DEV void epilogue4(float (&v)[4], const float4& b4, size_t grow, int gcol, const GemmEpi<TO>& e) {
    v[0] = v[0] * e.alpha + b4.x; v[1] = v[1] * e.alpha + b4.y; v[2] = v[2] * e.alpha + b4.z; v[3] = v[3] * e.alpha + b4.w;
    if (e.C2) store4<TO>(e.C2 + grow * e.ldc2 + gcol, v);
    if (e.act != A4R_ACT_NONE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = act_fwd(v[i], e.act);
    }
    if (e.dact != A4R_ACT_NONE) {
        float pre[4];
        load4<TO>(e.Pre + grow * e.ldpre + gcol, pre);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] *= act_bwd(pre[i], e.dact);
    }
    const uint64_t e0 = (uint64_t)grow * (uint64_t)e.N + (uint64_t)gcol;      // gcol % 4 == 0: one hash, four 16-bit lots
    if (e.thr16 && e.drop_first) {
        const uint64_t h = a4r_hash64(e.drop_seed, e.drop_site, e0 >> 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (((uint32_t)(h >> (16 * i)) & 0xffffu) >= e.thr16) ? v[i] * e.keep_scale : 0.f;
    }
    if (e.R1) {
        float t[4];
        load4<TO>(e.R1 + grow * e.ldr1 + gcol, t);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += t[i];
    }
    if (e.R2) {
        float t[4];
        load4<TO>(e.R2 + grow * e.ldr2 + gcol, t);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += t[i];
    }
    if (e.thr16 && !e.drop_first) {
        const uint64_t h = a4r_hash64(e.drop_seed, e.drop_site, e0 >> 2);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (((uint32_t)(h >> (16 * i)) & 0xffffu) >= e.thr16) ? v[i] * e.keep_scale : 0.f;
    }
    store4<TO>(e.C + grow * e.ldc + gcol, v);
}
