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
