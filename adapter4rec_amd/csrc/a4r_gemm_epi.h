// Shared GEMM epilogue: NC (8 or 4) consecutive output columns of one row, fp32 in registers ->
// alpha, bias, [C2 = pre-activation or its derivative], activation, act'(Pre) / * Pre, [dropout], residuals, [dropout], store.
// ACT / DACT template arguments >= 0 fix the activation at compile time (no per-element switch, small code: the
// 256-tile kernel unrolls this 16 times per wave and a runtime switch blew its code up to 280 KB); -1 = runtime value.
#pragma once
#include "a4r_common.h"
#include "../../include/a4r.h"
#ifndef A4R_ABL
#define A4R_ABL 0      /* timing-only diagnostic builds (tools/gemm_abl.sh, tools/epi_abl.sh); epilogue bits: 64 no GELU arithmetic, 128 no C2 store, 256 no Pre operand, 512 no C store, 1024 non-temporal C / C2 stores, 2048 / 4096 write-through (sc1 / sc0 sc1) C / C2 stores */
#endif

template <typename TO>
struct GemmEpi {
    TO* C; TO* C2; const TO* R1; const TO* R2; const TO* Pre; const float* bias;
    int ldc, ldc2, ldr1, ldr2, ldpre, N, act, dact, drop_first, c2_mode;
    float alpha, keep_scale;
    uint32_t thr16, drop_site;
    uint64_t drop_seed, row0;
};

template <typename TO>
A4R_DEV GemmEpi<TO> make_epi(const a4r_gemm_t& p, uint32_t thr16, float keep_scale) {
    GemmEpi<TO> e;
    e.C = reinterpret_cast<TO*>(p.C); e.C2 = reinterpret_cast<TO*>(p.C2);
    e.R1 = reinterpret_cast<const TO*>(p.R1); e.R2 = reinterpret_cast<const TO*>(p.R2); e.Pre = reinterpret_cast<const TO*>(p.Pre);
    e.bias = p.bias;
    e.ldc = p.ldc; e.ldc2 = p.ldc2; e.ldr1 = p.ldr1; e.ldr2 = p.ldr2; e.ldpre = p.ldpre; e.N = p.N;
    e.act = p.act; e.dact = p.dact; e.drop_first = p.drop_first; e.c2_mode = p.c2_mode;
    e.alpha = p.alpha; e.keep_scale = keep_scale; e.thr16 = thr16; e.drop_site = p.drop_site; e.drop_seed = p.drop_seed;
    e.row0 = (uint64_t)p.drop_row0;
    return e;
}

// A stored GELU derivative as 8-bit fixed point (c2_mode 2 / A4R_DACT_MUL_Q8, include/a4r.h): gelu'(x) lies in [-0.1289, 1.1289];
// 255 steps of 0.0049326 over that range, |error| <= 0.0025 -- the size of bf16's rounding error for a derivative near 1 (2^-9 ..
// 2^-8) -- at half the bytes of the FFN-up GEMM's second output.
#define A4R_Q8_OFF 0.1289f
#define A4R_Q8_STEP 0.0049326f
#ifndef A4R_Q8_V2
#define A4R_Q8_V2 1        /* 0 (A/B builds): round 3's form (add, multiply, v_rndne, convert per element) */
#endif
template <int NC> A4R_DEV void store_q8(uint8_t* p, const float* d) {
    uint32_t w[NC / 4];
#if A4R_Q8_V2
    // v_cvt_pk_u8_f32 itself rounds to nearest even and saturates at 0 / 255 (probed on gfx950, build/probe_cvt.hip: 0.5 -> 0, 1.5 -> 2,
    // 2.5 -> 2, 254.5 -> 254, 300 -> 255, -0.6 -> 0): no v_rndne in front of it, and the add / multiply run as packed pairs -- 2 + 1
    // instructions per element down to 1/2 + 1/2 + 1, the stored byte unchanged.
    typedef float f2_ __attribute__((ext_vector_type(2)));
    f2_ t[NC / 2];
#pragma unroll
    for (int j = 0; j < NC / 2; ++j) t[j] = (f2_{d[2 * j], d[2 * j + 1]} + A4R_Q8_OFF) * (1.f / A4R_Q8_STEP);
#pragma unroll
    for (int g = 0; g < NC / 4; ++g) {
        w[g] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) w[g] = __builtin_amdgcn_cvt_pk_u8_f32(i & 1 ? t[2 * g + (i >> 1)].y : t[2 * g + (i >> 1)].x, (uint32_t)i, w[g]);
    }
#else
#pragma unroll
    for (int g = 0; g < NC / 4; ++g) {
        w[g] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)      // v_cvt_pk_u8_f32: saturating float -> byte i of the word (of an integer-valued float: no rounding question)
            w[g] = __builtin_amdgcn_cvt_pk_u8_f32(rintf((d[4 * g + i] + A4R_Q8_OFF) * (1.f / A4R_Q8_STEP)), (uint32_t)i, w[g]);
    }
#endif
    if constexpr (NC == 8) *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
    else *reinterpret_cast<uint32_t*>(p) = w[0];
}
template <int NC> A4R_DEV void unpack_q8(const uint32_t* w, float* o) {
#pragma unroll
    for (int i = 0; i < NC; ++i) o[i] = (float)((w[i >> 2] >> (8 * (i & 3))) & 0xffu) * A4R_Q8_STEP - A4R_Q8_OFF;
}

// NC consecutive elements <-> fp32 registers (16 B per bf16x8 / fp32x4, 8 B per bf16x4)
template <typename T, int NC> A4R_DEV void load_n(const T* p, float* o) {
    if constexpr (NC == 8) load_vec<T, 8>(p, o);
    else if constexpr (sizeof(T) == 4) load_vec<T, 4>(p, o);
    else {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
        o[0] = bf16_bits_to_f32(v.x & 0xffffu); o[1] = bf16_bits_to_f32(v.x >> 16);
        o[2] = bf16_bits_to_f32(v.y & 0xffffu); o[3] = bf16_bits_to_f32(v.y >> 16);
    }
}
template <typename T, int NC> A4R_DEV void store_n(T* p, const float* o) {
    // A4R_ABL & 1024: non-temporal output stores.  Kernel by kernel the hint was neutral or worse except on the '* 8-bit derivative'
    // dgrad (240 against 262 - 274 us); end to end the step was 0.23 ms SLOWER with it there: the next GEMM reads that output.
    if constexpr (NC == 8 && sizeof(T) == 2 && (A4R_ABL & 1024) != 0) {
        const uint4 w = Elem<T>::pack(o);
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        v4u x = {w.x, w.y, w.z, w.w};
        __builtin_nontemporal_store(x, reinterpret_cast<v4u*>(p));
        return;
    }
    if constexpr (NC == 8 && sizeof(T) == 2 && (A4R_ABL & 6144) != 0) {        // 2048: write-through (sc1) stores, 4096: sc0 sc1 -- the end-of-kernel release finds nothing dirty
        const uint4 w = Elem<T>::pack(o);
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        v4u x = {w.x, w.y, w.z, w.w};
        // (s_nop: a VMEM store of more than 64 bits followed by a vector write of its data registers needs a wait state; hipcc cannot see into the asm)
        if constexpr ((A4R_ABL & 4096) != 0) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
        return;
    }
    if constexpr (NC == 8) store_vec<T, 8>(p, o);
    else if constexpr (sizeof(T) == 4) store_vec<T, 4>(p, o);
    else
        *reinterpret_cast<uint2*>(p) = make_uint2(pack2_bf16(o[0], o[1]),
                                                  pack2_bf16(o[2], o[3]));
}

template <int NC>
A4R_DEV void epi_dropout(float (&v)[NC], uint64_t e0, uint64_t seed, uint32_t site, uint32_t thr16, float keep_scale) {
#pragma unroll
    for (int g = 0; g < NC / 4; ++g) {                       // e0 % 4 == 0: one hash per 4 elements, four 16-bit lots
        const uint64_t h = a4r_hash64(seed, site, (e0 >> 2) + g);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[4 * g + i] = (((uint32_t)(h >> (16 * i)) & 0xffffu) >= thr16) ? v[4 * g + i] * keep_scale : 0.f;
    }
}

// pre_ld != nullptr: the NC elements of Pre were requested earlier by the caller (16-byte pieces; see load_pre_n)
// Q8: the Pre operand is the one-byte-per-element derivative tensor (compile-time: a run-time test here made the callers' register
// arrays addressable and sent them to scratch)
template <typename TO, int NC, bool Q8 = false> A4R_DEV void load_pre_n(uint4* q, uint32_t grow, int gcol, const GemmEpi<TO>& e) {
    if constexpr (Q8) {
        const uint8_t* p8 = reinterpret_cast<const uint8_t*>(e.Pre) + (size_t)grow * (uint32_t)e.ldpre + gcol;
        if constexpr (NC == 8) { const uint2 w = *reinterpret_cast<const uint2*>(p8); q[0] = make_uint4(w.x, w.y, 0u, 0u); }
        else q[0] = make_uint4(*reinterpret_cast<const uint32_t*>(p8), 0u, 0u, 0u);
        return;
    }
#pragma unroll
    for (int s = 0; s < NC * (int)sizeof(TO) / 16; ++s)
        q[s] = *reinterpret_cast<const uint4*>(e.Pre + (size_t)grow * (uint32_t)e.ldpre + gcol + s * (16 / (int)sizeof(TO)));
}

// R1PF: the caller ALWAYS passes r1_ld and has filled it whenever e.R1 is set (a run-time null test on a register array would send
// the array to scratch)
// EF >= 0: the caller states at compile time which optional pieces CAN be present (bit 1 dropout, 2 R1, 4 R2, 8 C2); the tests of the
// others (a uniform branch each, 16 groups per tile) and their code are not emitted.  EF < 0: every piece behind its run-time test.
template <typename TO, int NC, int ACT = -1, int DACT = -1, bool R1PF = false, bool FAST = false, int EF = -1>
A4R_DEV void epilogue_n(float (&v)[NC], const float* bias, uint32_t grow, int gcol, const GemmEpi<TO>& e, const uint4* pre_ld = nullptr,
                        const uint4* r1_ld = nullptr, const uint4* r2_ld = nullptr, TO* cdst = nullptr, uint64_t e0v = 0, float cmul = 1.f,
                        uint8_t* q8dst = nullptr) {      // q8dst: where this group's 8-bit derivative goes when not row-major (a4r_gemm_t.q8_tiled)
    // FAST (cdst, e0v): the caller formed the address of C[grow][gcol] / the dropout element index itself (the 256-tile kernel: a uniform
    // tile base + a per-lane 32-bit offset instead of a 64-bit multiply per group)
    const int act = ACT >= 0 ? ACT : e.act;
    const int dact = DACT >= 0 ? DACT : e.dact;
    const uint32_t thr16 = (EF < 0 || (EF & 1)) ? e.thr16 : 0u;
    // EF bit 64 (with 8): the second output IS the 8-bit GELU derivative (c2_mode 2) -- stated by the dispatcher, so neither the pointer nor
    // the mode is tested per group (16 uniform branches per tile that also kept hipcc from scheduling across the groups)
    constexpr bool C2Q8 = EF >= 0 && (EF & 64) != 0;
    const bool has_r1 = R1PF || ((EF < 0 || (EF & 2)) && e.R1), has_r2 = (EF < 0 || (EF & 4)) && e.R2, has_c2 = C2Q8 || ((EF < 0 || (EF & 8)) && e.C2);
    const int c2_mode = C2Q8 ? 2 : e.c2_mode;
#pragma unroll
    for (int i = 0; i < NC; ++i) v[i] = v[i] * e.alpha + bias[i];
    if (act == A4R_ACT_GELU && has_c2 && c2_mode) {          // value and derivative from one exp + one rcp
        float d[NC];
        if (A4R_ABL & 64) {
#pragma unroll
            for (int i = 0; i < NC; ++i) d[i] = v[i];
        } else {
            gelu_erf_both_n<NC>(v, d);
        }
        if (c2_mode == 2) { if (!(A4R_ABL & 128)) store_q8<NC>(q8dst ? q8dst : reinterpret_cast<uint8_t*>(e.C2) + (size_t)grow * (uint32_t)e.ldc2 + gcol, d); }
        else if (!(A4R_ABL & 128)) store_n<TO, NC>(e.C2 + (size_t)grow * (uint32_t)e.ldc2 + gcol, d);
    } else {
        if (has_c2) {
            if (c2_mode) {
                float d[NC];
#pragma unroll
                for (int i = 0; i < NC; ++i) d[i] = act_bwd(v[i], act);
                store_n<TO, NC>(e.C2 + (size_t)grow * (uint32_t)e.ldc2 + gcol, d);
            } else {
                store_n<TO, NC>(e.C2 + (size_t)grow * (uint32_t)e.ldc2 + gcol, v);
            }
        }
        if (act != A4R_ACT_NONE) {
#pragma unroll
            for (int i = 0; i < NC; ++i) v[i] = act_fwd(v[i], act);
        }
    }
    if (dact != A4R_ACT_NONE && !(A4R_ABL & 256)) {
        float pre[NC];
        // (compile-time split: with the 8-bit form behind a run-time test the callers' prefetch arrays went to scratch)
        constexpr bool Q8_CT = DACT == A4R_DACT_MULQ8_, Q8_RT = DACT < 0;
        bool q8 = Q8_CT;
        if constexpr (Q8_RT) q8 = dact == A4R_DACT_MULQ8_;
        if constexpr (Q8_CT || Q8_RT) {
            if (q8) {
                uint4 w;
                if (pre_ld) w = pre_ld[0];
                else load_pre_n<TO, NC, true>(&w, grow, gcol, e);
                unpack_q8<NC>(&w.x, pre);
            }
        }
        if (!q8) {
            if (pre_ld) {
#pragma unroll
                for (int s = 0; s < NC / Elem<TO>::PER16; ++s) Elem<TO>::unpack(pre_ld[s], pre + s * Elem<TO>::PER16);
            } else {
                load_n<TO, NC>(e.Pre + (size_t)grow * (uint32_t)e.ldpre + gcol, pre);
            }
        }
        if (dact == A4R_DACT_MUL_ || dact == A4R_DACT_MULQ8_) {
#pragma unroll
            for (int i = 0; i < NC; ++i) v[i] *= pre[i];
        } else {
#pragma unroll
            for (int i = 0; i < NC; ++i) v[i] *= act_bwd(pre[i], dact);
        }
    }
    // (the element index of the dropout mask is only formed when a mask is drawn: it is 64-bit arithmetic per group)
    if (thr16 && e.drop_first)
        epi_dropout<NC>(v, FAST ? e0v : ((uint64_t)grow + e.row0) * (uint64_t)e.N + (uint64_t)gcol, e.drop_seed, e.drop_site, thr16, e.keep_scale);
    if (has_r1) {
        float t[NC];
        if (R1PF || r1_ld) {
#pragma unroll
            for (int s = 0; s < NC / Elem<TO>::PER16; ++s) Elem<TO>::unpack(r1_ld[s], t + s * Elem<TO>::PER16);
        } else {
            load_n<TO, NC>(e.R1 + (size_t)grow * (uint32_t)e.ldr1 + gcol, t);
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) v[i] += t[i];
    }
    if (has_r2) {
        float t[NC];
        if (r2_ld) {
#pragma unroll
            for (int s = 0; s < NC / Elem<TO>::PER16; ++s) Elem<TO>::unpack(r2_ld[s], t + s * Elem<TO>::PER16);
        } else {
            load_n<TO, NC>(e.R2 + (size_t)grow * (uint32_t)e.ldr2 + gcol, t);
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) v[i] += t[i];
    }
    if (thr16 && !e.drop_first)
        epi_dropout<NC>(v, FAST ? e0v : ((uint64_t)grow + e.row0) * (uint64_t)e.N + (uint64_t)gcol, e.drop_seed, e.drop_site, thr16, e.keep_scale);
    if constexpr (EF >= 0 && (EF & 16) != 0) {               // C as OCP e4m3 bytes (a4r_gemm_t.c_fp8): v * cmul, saturated at +-448, 8 bytes per lane
        static_assert(NC == 8 && FAST, "e4m3 output: the 256-tile kernel's form");
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_fmed3f(v[i] * cmul, -448.f, 448.f);
        *reinterpret_cast<uint2*>(cdst) = f32x8_to_fp8(t);
        return;
    }
    if (!(A4R_ABL & 512) || v[0] == 12345.678f) store_n<TO, NC>(FAST ? cdst : e.C + (size_t)grow * (uint32_t)e.ldc + gcol, v);
}
// the NC elements of a residual operand as 16-byte pieces, for a caller that requests them ahead of use (r1_ld / r2_ld above)
template <typename TO, int NC> A4R_DEV void load_res_n(uint4* q, const TO* R, int ldr, size_t grow, int gcol) {
#pragma unroll
    for (int s = 0; s < NC * (int)sizeof(TO) / 16; ++s)
        q[s] = *reinterpret_cast<const uint4*>(R + grow * ldr + gcol + s * (16 / (int)sizeof(TO)));
}

// ---- two-stage form of the same epilogue: the loads of a group (Pre, R1, R2: 16 B per operand for bf16, 32 B for fp32) are issued
// by epi_issue() and consumed by epi_finish().  A caller that issues row mi + 1 BEFORE it finishes row mi keeps a row of loads in
// flight behind the stores; with the single-stage epilogue_n() every group waited for its own loads (they cannot be hoisted above
// the previous group's store: C may alias R1 for all the compiler knows), i.e. one L2 round trip per group.
template <typename TO>
struct EpiLoads {
    static constexpr int S = (int)sizeof(TO) / 2;       // uint4 per 8 elements
    uint4 pre[S], r1[S], r2[S];
};

template <typename TO, int DACT = -1>
A4R_DEV void epi_issue(EpiLoads<TO>& L, size_t grow, int gcol, const GemmEpi<TO>& e) {
    constexpr int S = EpiLoads<TO>::S, PER = 16 / (int)sizeof(TO);
    const int dact = DACT >= 0 ? DACT : e.dact;
    if (dact == A4R_DACT_MULQ8_) {
        const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(e.Pre) + grow * e.ldpre + gcol);
        L.pre[0].x = w.x; L.pre[0].y = w.y;
    } else if (dact != A4R_ACT_NONE) {
#pragma unroll
        for (int s = 0; s < S; ++s) L.pre[s] = *reinterpret_cast<const uint4*>(e.Pre + grow * e.ldpre + gcol + s * PER);
    }
    if (e.R1) {
#pragma unroll
        for (int s = 0; s < S; ++s) L.r1[s] = *reinterpret_cast<const uint4*>(e.R1 + grow * e.ldr1 + gcol + s * PER);
    }
    if (e.R2) {
#pragma unroll
        for (int s = 0; s < S; ++s) L.r2[s] = *reinterpret_cast<const uint4*>(e.R2 + grow * e.ldr2 + gcol + s * PER);
    }
}

template <typename TO>
A4R_DEV void epi_unpack8(const uint4* q, float* o) {
#pragma unroll
    for (int s = 0; s < EpiLoads<TO>::S; ++s) Elem<TO>::unpack(q[s], o + s * Elem<TO>::PER16);
}

template <typename TO, int ACT = -1, int DACT = -1>
A4R_DEV void epi_finish(float (&v)[8], const float* bias, const EpiLoads<TO>& L, size_t grow, int gcol, const GemmEpi<TO>& e) {
    constexpr int NC = 8;
    const int act = ACT >= 0 ? ACT : e.act;
    const int dact = DACT >= 0 ? DACT : e.dact;
#pragma unroll
    for (int i = 0; i < NC; ++i) v[i] = v[i] * e.alpha + bias[i];
    if (act == A4R_ACT_GELU && e.C2 && e.c2_mode) {          // value and derivative from one exp + one rcp
        float d[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) gelu_erf_both(v[i], v[i], d[i]);
        if (e.c2_mode == 2) store_q8<NC>(reinterpret_cast<uint8_t*>(e.C2) + grow * e.ldc2 + gcol, d);
        else store_n<TO, NC>(e.C2 + grow * e.ldc2 + gcol, d);
    } else {
        if (e.C2) {
            if (e.c2_mode) {
                float d[NC];
#pragma unroll
                for (int i = 0; i < NC; ++i) d[i] = act_bwd(v[i], act);
                store_n<TO, NC>(e.C2 + grow * e.ldc2 + gcol, d);
            } else {
                store_n<TO, NC>(e.C2 + grow * e.ldc2 + gcol, v);
            }
        }
        if (act != A4R_ACT_NONE) {
#pragma unroll
            for (int i = 0; i < NC; ++i) v[i] = act_fwd(v[i], act);
        }
    }
    if (dact != A4R_ACT_NONE) {
        float pre[NC];
        if (dact == A4R_DACT_MULQ8_) unpack_q8<NC>(&L.pre[0].x, pre);
        else epi_unpack8<TO>(L.pre, pre);
        if (dact == A4R_DACT_MUL_ || dact == A4R_DACT_MULQ8_) {
#pragma unroll
            for (int i = 0; i < NC; ++i) v[i] *= pre[i];
        } else {
#pragma unroll
            for (int i = 0; i < NC; ++i) v[i] *= act_bwd(pre[i], dact);
        }
    }
    const uint64_t e0 = ((uint64_t)grow + e.row0) * (uint64_t)e.N + (uint64_t)gcol;
    if (e.thr16 && e.drop_first) epi_dropout<NC>(v, e0, e.drop_seed, e.drop_site, e.thr16, e.keep_scale);
    if (e.R1) {
        float t[NC];
        epi_unpack8<TO>(L.r1, t);
#pragma unroll
        for (int i = 0; i < NC; ++i) v[i] += t[i];
    }
    if (e.R2) {
        float t[NC];
        epi_unpack8<TO>(L.r2, t);
#pragma unroll
        for (int i = 0; i < NC; ++i) v[i] += t[i];
    }
    if (e.thr16 && !e.drop_first) epi_dropout<NC>(v, e0, e.drop_seed, e.drop_site, e.thr16, e.keep_scale);
    store_n<TO, NC>(e.C + grow * e.ldc + gcol, v);
}
