// skinny64_kernel: C[M, 64] = epilogue(A[M, K] . B[64, K]^T) -- the adapter down-projections (fc_down of AdapterBlock,
// model/modules.py:116-134) and their backward counterparts dz = dy . W_up (44 launches of 40 448 x 64 x 768 per training step).
//
// The op is a stream over A (62 MB for 2.6 GFLOP): what matters is bytes in flight per CU, not the matrix pipe.  The 128-row tile
// of a4r_gemm.hip gave 316 workgroups with one 24 KiB stage in flight each (2.4 TB/s).  Here a workgroup owns 64 rows
// (632 workgroups, three resident per CU at 48 KiB of LDS), 4 waves x (16 rows x 64 columns), and a 3-stage ring of
// 16 KiB K-tiles (A 64 rows x 128 B | B 64 rows x 128 B) filled by LDS-DMA that stays in flight across the one barrier per
// K-tile: ~96 KiB in flight per CU.  Per K-tile: s_waitcnt vmcnt(4) (own DMA of this stage) -> s_barrier (everyone's DMA landed;
// everyone is done reading the stage about to be refilled) -> 4 LDS-DMA for K-tile t + 2 -> 10 ds_read_b128 + 8 MFMA.
// Operands swapped in the MFMA (B fragment first) so that a lane ends up with consecutive columns of one row: the epilogue runs
// straight from the accumulators (v_permlane16_swap pairs two 16-column tiles into 16-byte stores), as in a4r_gemm256.hip.
#include "a4r_gemm_epi.h"

namespace {

constexpr int SK_STAGE = 16384, SK_NST = 3;

A4R_DEV void sk_glds16(const void* base, uint32_t voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) skinny64_kernel(const a4r_gemm_t p, uint32_t thr16, float keep_scale) {
    constexpr int ROWB = 128;
    constexpr int KT = ROWB / (int)sizeof(TI);
    __shared__ __attribute__((aligned(16))) char lds[SK_NST * SK_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lda = p.lda, ldb = p.ldb;
    const int nk = p.K / KT;
    const char* Abase = reinterpret_cast<const char*>(reinterpret_cast<const TI*>(p.A) + (size_t)blockIdx.x * 64 * lda);
    const char* Bbase = reinterpret_cast<const char*>(p.B);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // wave w stages rows 16w .. 16w+15 of A and of B: two 1-KiB pieces each (8 rows x 128 B, swizzled on the source side)
    uint32_t voffA[2], voffB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ul = 8 * (2 * wave + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((ul >> 1) & 7);
        voffA[i] = (uint32_t)(ul * lda * (int)sizeof(TI) + c * 16);
        voffB[i] = (uint32_t)(ul * ldb * (int)sizeof(TI) + c * 16);
    }
    const uint32_t dst0 = lds0 + (uint32_t)(2 * wave) * 1024u;
    auto issue = [&](int t) {
        if (t < nk) {
            const uint32_t d = dst0 + (uint32_t)(t % SK_NST) * SK_STAGE;
            const char* a = Abase + (size_t)t * ROWB;
            const char* b = Bbase + (size_t)t * ROWB;
            sk_glds16(a, voffA[0], d);
            sk_glds16(a, voffA[1], d + 1024u);
            sk_glds16(b, voffB[0], d + 8192u);
            sk_glds16(b, voffB[1], d + 8192u + 1024u);
        }
    };
    issue(0);
    issue(1);

    const int fr = lane & 15, kg = lane >> 4;
    // bias and the Pre operand of the dgrad form (dz = dy . W_up * act'(zpre)) are requested now, not in the epilogue
    const GemmEpi<TO> epi = make_epi<TO>(p, thr16, keep_scale);
    const size_t grow = (size_t)blockIdx.x * 64 + 16 * wave + fr;
    const int gcolp = (kg & 1) * 16 + (kg >> 1) * 8;                       // + pair * 32
    float b8[2][8];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[pr][e] = epi.bias ? epi.bias[gcolp + pr * 32 + e] : 0.f;
    constexpr int PS = 8 * (int)sizeof(TO) / 16;
    uint4 pre_ld0[PS] = {}, pre_ld1[PS] = {};      // two objects, always handed to the epilogue (a run-time null / index made one array addressable: scratch)
    if (epi.dact != A4R_ACT_NONE) {
        load_pre_n<TO, 8>(pre_ld0, grow, gcolp, epi);
        load_pre_n<TO, 8>(pre_ld1, grow, gcolp + 32, epi);
    }
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ra = 16 * wave + fr, ch = ks * 4 + kg;
        a_off[ks] = ra * ROWB + ((ch ^ ((ra >> 1) & 7)) << 4);
        b_off[ks] = 8192 + fr * ROWB + ((ch ^ ((fr >> 1) & 7)) << 4);      // + ni * 2048 (16 rows; the swizzle term repeats every 16 rows)
    }
    f32x4_t acc[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // K-tile t landed, t + 1 may still be in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(t + 2);
        const char* st = lds + (t % SK_NST) * SK_STAGE;
        uint4 af[2], bf[4][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            af[ks] = *reinterpret_cast<const uint4*>(st + a_off[ks]);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bf[ni][ks] = *reinterpret_cast<const uint4*>(st + b_off[ks] + ni * 2048);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) Mma<TI>::mma(bf[ni][ks], af[ks], acc[ni]);
    }

    // ---- epilogue from the (transposed) accumulators: lane (fr, kg) holds row fr, columns ni*16 + kg*4 .. +3 of tile ni
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * pr][r]), __float_as_uint(acc[2 * pr + 1][r]), false, false);
            v[r] = __uint_as_float(sw[0]);
            v[4 + r] = __uint_as_float(sw[1]);
        }
        epilogue_n<TO, 8>(v, b8[pr], grow, gcolp + pr * 32, epi, pr == 0 ? pre_ld0 : pre_ld1);
    }
}

// skinnyk_kernel: C[M, N] = epilogue(A[M, 64] . B[N, 64]^T), ONE K-tile -- the adapter up-projections fc_up (+ bias + the two
// residuals of BertAdaptedSelfOutput, model/model.py:292-297) and the dgrad through fc_down (44 launches of 40 448 x 768 x 64 per
// step, 191 MB each for 4 GFLOP: a stream over R1, R2 and C).  Workgroup = 64 rows x 128 columns, 4 waves x (16 rows x 128
// columns); every residual load of the tile is requested FIRST (they do not depend on the product), then the 24 KiB of operands
// by LDS-DMA, one wait, 16 MFMAs per wave, epilogue from registers.  ~150 registers, 24 KiB of LDS: three workgroups per CU,
// ~170 KB in flight per CU.  (The 256-tile kernel ran this shape at 47 us: one workgroup per CU, residual loads waited for
// where they are issued.)
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) skinnyk_kernel(const a4r_gemm_t p, int ntn, uint32_t thr16, float keep_scale) {
    constexpr int ROWB = 128;
    __shared__ __attribute__((aligned(16))) char lds[8192 + 16384];       // A 64 rows | B 128 rows, 128 B each
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;               // column tiles of a row block are neighbours: A stays in L2
    const int lda = p.lda, ldb = p.ldb;
    const GemmEpi<TO> epi = make_epi<TO>(p, thr16, keep_scale);
    const int fr = lane & 15, kg = lane >> 4;
    const size_t grow = (size_t)tm * 64 + 16 * wave + fr;
    const int gcolp = tn * 128 + (kg & 1) * 16 + (kg >> 1) * 8;           // + pair * 32

    constexpr int S = 8 * (int)sizeof(TO) / 16;
    uint4 r1[4][S], r2[4][S];
    if (epi.R1) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) load_res_n<TO, 8>(r1[pr], epi.R1, epi.ldr1, grow, gcolp + pr * 32);
    }
    if (epi.R2) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) load_res_n<TO, 8>(r2[pr], epi.R2, epi.ldr2, grow, gcolp + pr * 32);
    }
    float b8[4][8];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr)
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[pr][e] = epi.bias ? epi.bias[gcolp + pr * 32 + e] : 0.f;

    const char* Abase = reinterpret_cast<const char*>(reinterpret_cast<const TI*>(p.A) + (size_t)tm * 64 * lda);
    const char* Bbase = reinterpret_cast<const char*>(reinterpret_cast<const TI*>(p.B) + (size_t)tn * 128 * ldb);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
#pragma unroll
    for (int i = 0; i < 6; ++i) {                                          // 24 one-KiB pieces: A 0..7, B 8..23; wave w takes 6w .. 6w+5
        const int q = 6 * wave + i;
        const bool isA = q < 8;
        const int ul = 8 * (isA ? q : q - 8) + (lane >> 3);
        const int c = (lane & 7) ^ ((ul >> 1) & 7);
        const uint32_t voff = (uint32_t)(ul * (isA ? lda : ldb) * (int)sizeof(TI) + c * 16);
        sk_glds16(isA ? Abase : Bbase, voff, lds0 + (uint32_t)q * 1024u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    uint4 af[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ra = 16 * wave + fr, ch = ks * 4 + kg;
        af[ks] = *reinterpret_cast<const uint4*>(lds + ra * ROWB + ((ch ^ ((ra >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        f32x4_t acc[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            acc[h] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int rb = (2 * pr + h) * 16 + fr, ch = ks * 4 + kg;
                const uint4 bf = *reinterpret_cast<const uint4*>(lds + 8192 + rb * ROWB + ((ch ^ ((rb >> 1) & 7)) << 4));
                Mma<TI>::mma(bf, af[ks], acc[h]);
            }
        }
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][r]), __float_as_uint(acc[1][r]), false, false);
            v[r] = __uint_as_float(sw[0]);
            v[4 + r] = __uint_as_float(sw[1]);
        }
        epilogue_n<TO, 8>(v, b8[pr], grow, gcolp + pr * 32, epi, nullptr, r1[pr], r2[pr]);
    }
}

}  // namespace

// called by a4r_gemm_nt for one-K-tile products (K * sizeof == 128) after argument validation; 1 = not instantiated
int a4r_gemm_nt_skinnyk(hipStream_t s, const a4r_gemm_t& g) {
    if (g.K != 64 || g.N % 128 || g.M % 64 || g.in_dtype != A4R_BF16 || g.out_dtype != A4R_BF16) return 1;
    if (g.dact != A4R_ACT_NONE) return 1;                                  // Pre is not prefetched here
    if (g.bias && (reinterpret_cast<uintptr_t>(g.bias) & 3u)) return 1;
    const int ntn = g.N / 128;
    hipLaunchKernelGGL((skinnyk_kernel<bf16_t, bf16_t>), dim3((g.M / 64) * ntn), dim3(256), 0, s, g, ntn, a4r_thr16(g.drop_p), a4r_keep_scale(g.drop_p));
    return a4r_launch_status();
}

// called by a4r_gemm_nt for N == 64 after argument validation; returns 1 when the combination is not instantiated
int a4r_gemm_nt_skinny64(hipStream_t s, const a4r_gemm_t& g) {
    if (g.N != 64 || g.M % 64 || (g.K * 2) % 128 || g.in_dtype != A4R_BF16) return 1;
    if (g.bias && (reinterpret_cast<uintptr_t>(g.bias) & 3u)) return 1;
    const dim3 grid(g.M / 64), block(256);
    if (g.out_dtype == A4R_BF16)
        hipLaunchKernelGGL((skinny64_kernel<bf16_t, bf16_t>), grid, block, 0, s, g, a4r_thr16(g.drop_p), a4r_keep_scale(g.drop_p));
    else
        hipLaunchKernelGGL((skinny64_kernel<bf16_t, float>), grid, block, 0, s, g, a4r_thr16(g.drop_p), a4r_keep_scale(g.drop_p));
    return a4r_launch_status();
}
