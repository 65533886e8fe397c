// a4r_gemm_nt: C[M,N] = epilogue(A[M,K] . B[N,K]^T) on the gfx950 matrix cores.
//
// Tile 128 x BN (BN = 128 or 64) x 128 bytes of K per stage (64 bf16 / 32 fp32), 256 threads =
// 4 waves laid out 2 (M) x 2 (N); each wave owns a 64 x BN/2 block of 16x16 MFMA tiles.
// LDS image of a stage: rows of 128 bytes = eight 16-byte chunks, chunk slot c of row r holds
// global chunk c ^ ((r >> 1) & 7): with ds_read_b128's 16-lane groups every fragment read
// (16 rows x 4 chunk columns) touches 16 distinct 16-byte slots of the 256-byte bank row.
//
// K pipeline (GLDS = true, default): stages are filled by global_load_lds_dwordx4 (no VGPR round
// trip; one wave-instruction writes 8 tile rows = 1 KiB linearly, so the swizzle is applied to the
// per-lane SOURCE address).  The two stages are two DISTINCT __shared__ objects and the K loop is
// unrolled by two, so that the compiler can prove the LDS-DMA into one stage does not alias the
// ds_reads of the other -- with a single array indexed by (kt & 1) hipcc puts s_waitcnt vmcnt(0) in
// front of the first ds_read of every K step and the loads no longer overlap the MFMAs.
// One __syncthreads() per K step (its fence waits for the DMA: vmcnt(0) then s_barrier).
// GLDS = false keeps the register-staged pipeline (loads issued before the MFMAs, ds_write after).
//
// The epilogue goes through LDS (fp32, 64 rows per stage object) so that bias / activation /
// residual / dropout run on 8 consecutive columns per thread and every global access is 16 B/lane.
// Workgroup -> tile map is XCD-aware: blocks that share an XCD (blockIdx % 8) walk consecutive
// N-tiles of the same 128-row A panel, which therefore stays in that XCD's L2.
#include "a4r_gemm_epi.h"

namespace {

constexpr int ROWB = 128;   // bytes of K per tile row per stage

// one 1-KiB LDS-DMA: LDS[lds_dst + lane*16 .. +16) <- global[base + voff .. +16).  Inline asm, as in a4r_gemm256.hip: the compiler neither
// counts these loads nor knows that they write LDS, so it inserts no s_waitcnt of its own -- the K loop's counted waits are the only ones
// (with the builtin form it drained the whole ring once per turn: it cannot order a ring of in-flight DMAs against the reads).
A4R_DEV void glds16_nt(const void* base, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base), "s"(lds_dst)
        : "memory");
}

template <typename TI, int BN>
A4R_DEV void stage_glds(char* stage, const TI* __restrict__ A, int lda, const TI* __restrict__ B, int ldb, int k0, int wave, int lane) {
    constexpr int PER = Elem<TI>::PER16;
    constexpr int NLB = BN / 32;
    const int glr = lane >> 3, glc = lane & 7;
    const uint32_t s0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)stage;
    const TI* Ak = A + k0;                                  // uniform bases; per-lane byte offsets are 32-bit (128 tile rows x ld)
    const TI* Bk = B + k0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = 4 * wave + i, r = 8 * q + glr, c = glc ^ ((r >> 1) & 7);
        glds16_nt(Ak, (uint32_t)((r * lda + c * PER) * (int)sizeof(TI)), s0 + (uint32_t)(q * 1024));
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
        const int q = NLB * wave + i, r = 8 * q + glr, c = glc ^ ((r >> 1) & 7);
        glds16_nt(Bk, (uint32_t)((r * ldb + c * PER) * (int)sizeof(TI)), s0 + (uint32_t)(128 * ROWB + q * 1024));
    }
}

template <typename TI, int BN>
A4R_DEV void compute_stage(const char* stage, int wm, int wn, int lane, f32x4_t (&acc)[4][BN / 32]) {
    constexpr int NI = BN / 32;
    const char* As = stage;
    const char* Bs = stage + 128 * ROWB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ch = ks * 4 + (lane >> 4);
        uint4 af[4], bf[NI];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = wm * 64 + mi * 16 + (lane & 15);
            af[mi] = *reinterpret_cast<const uint4*>(As + row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int row = wn * (BN / 2) + ni * 16 + (lane & 15);
            bf[ni] = *reinterpret_cast<const uint4*>(Bs + row * ROWB + ((ch ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) Mma<TI>::mma(af[mi], bf[ni], acc[mi][ni]);
    }
}

template <typename TI, typename TO, int BN, bool GLDS>
__global__ void __launch_bounds__(256) gemm_nt_kernel(const a4r_gemm_t p, int ntm, int ntn, uint32_t thr16, float keep_scale) {
    constexpr int BM = 128;
    constexpr int PER = Elem<TI>::PER16;
    constexpr int KT = ROWB / (int)sizeof(TI);
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int NI = BN / 32;          // 16-column tiles per wave
    constexpr int NLB = BN / 32;         // 16-byte chunks of B each thread stages (BN*8/256)
    constexpr int SBYTES = (STAGE > 64 * BN * 4) ? STAGE : 64 * BN * 4;   // a stage also holds 64 fp32 epilogue rows
    __shared__ __attribute__((aligned(16))) char lds0[SBYTES];
    __shared__ __attribute__((aligned(16))) char lds1[SBYTES];
    // GLDS: a FOUR-deep ring (distinct objects again: the compiler must see that a DMA into one does not alias the reads of another).  Three
    // stages are in flight while one is multiplied: with two, every 128-byte K stage paid a full DMA round trip (~2 us: the tail-panel
    // launches of the image tower, 12 - 48 stages each, were 8.5 % of its step; the SASRec tower's K = 256 launches 17 - 25 us)
    __shared__ __attribute__((aligned(16))) char lds2[GLDS ? STAGE : 16];
    __shared__ __attribute__((aligned(16))) char lds3[GLDS ? STAGE : 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: the LDS-DMA destination goes through M0
    const int wm = wave >> 1, wn = wave & 1;

    // bijective XCD-aware remap of the 1-D grid
    const int nt = ntm * ntn;
    const int bid = blockIdx.x, xcd = bid & 7, j = bid >> 3, q = nt >> 3, r = nt & 7;
    const int Lt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    const int tm = Lt / ntn, tn = Lt % ntn;

    const int lda = p.lda, ldb = p.ldb;
    const int Kdim = p.K;
    const TI* __restrict__ A = reinterpret_cast<const TI*>(p.A) + (size_t)tm * BM * lda;
    const TI* __restrict__ B = reinterpret_cast<const TI*>(p.B) + (size_t)tn * BN * ldb;

    f32x4_t acc[4][NI];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nk = Kdim / KT;

    if constexpr (GLDS) {
        // step i (stage i lives in object i % 4): wait for THIS wave's DMAs of stage i -- a counted vmcnt: the up to two younger stages stay
        // in flight --, barrier (every wave's part of stage i has landed; every wave is done reading stage i - 1), request stage i + 3 into
        // the object stage i - 1 was read from, multiply stage i.
        constexpr int DPS = 4 + NLB;                                    // DMA instructions per stage per wave
#define A4R_WAIT_STAGE(i_)                                                                          \
        {                                                                                           \
            const int after_ = nk - 1 - (i_);                          /* stages requested after stage i_ */ \
            if (after_ >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * DPS) : "memory");         \
            else if (after_ == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPS) : "memory");        \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
            __builtin_amdgcn_s_barrier();                                                           \
            asm volatile("" ::: "memory");                                                          \
        }
#define A4R_STEP(i_, cur_, refill_)                                                                 \
        if ((i_) < nk) {                                                                            \
            A4R_WAIT_STAGE(i_)                                                                      \
            if ((i_) + 3 < nk) stage_glds<TI, BN>(refill_, A, lda, B, ldb, ((i_) + 3) * KT, wave, lane); \
            compute_stage<TI, BN>(cur_, wm, wn, lane, acc);                                         \
        }
        stage_glds<TI, BN>(lds0, A, lda, B, ldb, 0, wave, lane);
        if (nk > 1) stage_glds<TI, BN>(lds1, A, lda, B, ldb, KT, wave, lane);
        if (nk > 2) stage_glds<TI, BN>(lds2, A, lda, B, ldb, 2 * KT, wave, lane);
        for (int kt = 0; kt < nk; kt += 4) {
            A4R_STEP(kt, lds0, lds3)
            A4R_STEP(kt + 1, lds1, lds0)
            A4R_STEP(kt + 2, lds2, lds1)
            A4R_STEP(kt + 3, lds3, lds2)
        }
#undef A4R_STEP
#undef A4R_WAIT_STAGE
        __syncthreads();                                                // all reads done: lds0 / lds1 become the epilogue's staging rows
    } else {
        // register staging of one K stage (explicit scalars: arrays captured by reference went to scratch)
        uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
        const int srow = tid >> 3, sch = tid & 7;                          // chunk id = tid + 256*i -> row = srow + 32*i
        const int soff = srow * ROWB + ((sch ^ ((srow >> 1) & 7)) << 4);   // ((srow + 32 i) >> 1) & 7 == (srow >> 1) & 7
        const TI* __restrict__ Ag = A + (size_t)srow * lda + sch * PER;
        const TI* __restrict__ Bg = B + (size_t)srow * ldb + sch * PER;
#define A4R_GLOAD(kt_)                                                                         \
    {                                                                                          \
        const int k0_ = (kt_) * KT;                                                            \
        ra0 = *reinterpret_cast<const uint4*>(Ag + k0_);                                       \
        ra1 = *reinterpret_cast<const uint4*>(Ag + (size_t)32 * lda + k0_);                    \
        ra2 = *reinterpret_cast<const uint4*>(Ag + (size_t)64 * lda + k0_);                    \
        ra3 = *reinterpret_cast<const uint4*>(Ag + (size_t)96 * lda + k0_);                    \
        rb0 = *reinterpret_cast<const uint4*>(Bg + k0_);                                       \
        rb1 = *reinterpret_cast<const uint4*>(Bg + (size_t)32 * ldb + k0_);                    \
        if constexpr (NLB == 4) {                                                              \
            rb2 = *reinterpret_cast<const uint4*>(Bg + (size_t)64 * ldb + k0_);                \
            rb3 = *reinterpret_cast<const uint4*>(Bg + (size_t)96 * ldb + k0_);                \
        }                                                                                      \
    }
#define A4R_SWRITE(st_)                                                                        \
    {                                                                                          \
        char* As_ = (st_) + soff;                                                              \
        char* Bs_ = As_ + A_BYTES;                                                             \
        *reinterpret_cast<uint4*>(As_) = ra0;                                                  \
        *reinterpret_cast<uint4*>(As_ + 32 * ROWB) = ra1;                                      \
        *reinterpret_cast<uint4*>(As_ + 64 * ROWB) = ra2;                                      \
        *reinterpret_cast<uint4*>(As_ + 96 * ROWB) = ra3;                                      \
        *reinterpret_cast<uint4*>(Bs_) = rb0;                                                  \
        *reinterpret_cast<uint4*>(Bs_ + 32 * ROWB) = rb1;                                      \
        if constexpr (NLB == 4) {                                                              \
            *reinterpret_cast<uint4*>(Bs_ + 64 * ROWB) = rb2;                                  \
            *reinterpret_cast<uint4*>(Bs_ + 96 * ROWB) = rb3;                                  \
        }                                                                                      \
    }
        A4R_GLOAD(0);
        A4R_SWRITE(lds0);
        __syncthreads();
        for (int kt = 0; kt < nk; kt += 2) {
            if (kt + 1 < nk) A4R_GLOAD(kt + 1);
            compute_stage<TI, BN>(lds0, wm, wn, lane, acc);
            if (kt + 1 < nk) A4R_SWRITE(lds1);
            __syncthreads();
            if (kt + 1 < nk) {
                if (kt + 2 < nk) A4R_GLOAD(kt + 2);
                compute_stage<TI, BN>(lds1, wm, wn, lane, acc);
                if (kt + 2 < nk) A4R_SWRITE(lds0);
                __syncthreads();
            }
        }
#undef A4R_GLOAD
#undef A4R_SWRITE
    }

    // ---- epilogue through LDS: tile rows 0..63 live in lds0, rows 64..127 in lds1 (fp32 [64][BN] each)
    {
        float* Cs = reinterpret_cast<float*>(wm == 0 ? lds0 : lds1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int row = mi * 16 + (lane >> 4) * 4 + rr;
                    const int col = wn * (BN / 2) + ni * 16 + (lane & 15);
                    Cs[row * BN + col] = acc[mi][ni][rr];
                }
    }
    __syncthreads();
    constexpr int TPR = BN / 8, RPP = 256 / TPR;
    const GemmEpi<TO> epi = make_epi<TO>(p, thr16, keep_scale);
    const int c8 = (tid % TPR) * 8;
    const int gcol = tn * BN + c8;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = epi.bias ? epi.bias[gcol + e] : 0.f;
#pragma unroll 1
    for (int pass = 0; pass < BM / RPP; ++pass) {
        const int row = pass * RPP + tid / TPR;
        const size_t grow = (size_t)tm * BM + row;
        const float* Cs = reinterpret_cast<const float*>(row < 64 ? lds0 : lds1) + (row & 63) * BN + c8;
        float v[8];
        const float4 lo = *reinterpret_cast<const float4*>(Cs);
        const float4 hi = *reinterpret_cast<const float4*>(Cs + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        epilogue_n<TO, 8>(v, bias, grow, gcol, epi);
    }
}

int g_variant = 2;   // 0 = 128-tile register staging, 1 = 128-tile direct-to-LDS, 2 = 256-tile ring where it applies (default)

template <typename TI, typename TO, int BN, bool GLDS>
int launch_v(hipStream_t s, const a4r_gemm_t& g) {
    const int ntm = g.M / 128, ntn = g.N / BN;
    hipLaunchKernelGGL((gemm_nt_kernel<TI, TO, BN, GLDS>), dim3(ntm * ntn), dim3(256), 0, s, g, ntm, ntn,
                       a4r_thr16(g.drop_p), a4r_keep_scale(g.drop_p));
    return a4r_launch_status();
}

template <typename TI, typename TO, int BN>
int launch(hipStream_t s, const a4r_gemm_t& g) {
    return g_variant != 0 ? launch_v<TI, TO, BN, true>(s, g) : launch_v<TI, TO, BN, false>(s, g);
}

template <typename TI, typename TO>
int launch_bn(hipStream_t s, const a4r_gemm_t& g) {
    return (g.N % 128 == 0) ? launch<TI, TO, 128>(s, g) : launch<TI, TO, 64>(s, g);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

int a4r_gemm_nt_256(hipStream_t s, const a4r_gemm_t& g);   // a4r_gemm256.hip
int a4r_gemm_nt_skinny64(hipStream_t s, const a4r_gemm_t& g); // a4r_gemm_skinny.hip (N == 64, bf16 in: the adapter down-projections)
int a4r_gemm_nt_skinnyk(hipStream_t s, const a4r_gemm_t& g);  // a4r_gemm_skinny.hip (K == 64, bf16: the adapter up-projections)
static int run_256(hipStream_t s, const a4r_gemm_t& g);
int a4r_cu_count();                                          // a4r_gemm256.hip: CU count rounded down to a multiple of 8
extern "C" int a4r_gemm_tail_plan(int M, int N, int* p_full, int* kp);   // a4r_gemm256.hip: 1 = the launch cuts its last partial round into short tiles

static int run_256(hipStream_t s, const a4r_gemm_t& g) { return a4r_gemm_nt_256(s, g); }

extern int g_tn256_on;                                       // a4r_gemm_tn256.hip: 0 = large weight gradients on the 64-tile kernels too (tests)
#ifdef A4R_WITH_W4
int a4r_gemm_w4(int v);                                      // a4r_gemm256.hip (tools-only build: make W4=1)
#endif
extern int g_tn_variant;                                     // a4r_gemm_tn.hip: 0 = register-staged weight-gradient kernel for bf16 too

// Leading rows (a multiple of 256) of an [M, N] output that the 256 x 256-tile kernel computes; the rows behind them go to the 128-tile
// kernel.  A function of (M, N), the CU count and a4r_gemm_variant only -- NOT of the operand type: a tile-native 8-bit derivative
// (a4r_gemm_t.q8_tiled) is tile-native on exactly these rows and row-major behind them, for every launch that writes or reads it.
//   * no more 256-tiles than a QUARTER of the CUs (the CLS-only last layer: 18 tiles): the 128-tile kernel fills the chip better -> 0.
//     (Rounds 1 - 4 drew the line at half the CUs, from round-2 measurements; re-measured on round-5 code, tools/gemm_small_m.py ->
//     profiles/r05_e_gemm_small_m.txt: at 90 - 120 tiles the 256-tile kernel is 1.2 - 1.6 x faster at every K -- 8 users, N = 768: 20.8 vs
//     25.9 us at K = 768, 57.1 vs 68.2 at K = 3072 --, at <= 60 tiles it wins at K = 768 and loses 15 - 20 % at K >= 2304.)
//   * a partial last round that a4r_gemm_tail_plan cuts into short tiles runs inside the same launch -> M;
//   * a last round of only a few whole row panels (ViT-B/16 at 8 users: 777 tiles = 3 rounds + 9 tiles) goes to the 128-tile kernel
//     as a second launch instead of costing a full round (N = 768, K = 3072: 4 -> 3 rounds + ~1/4) -> the rows in front of it.
extern "C" int a4r_gemm_rows_256(int M, int N) {
    if (g_variant < 2 || M <= 0 || N <= 0 || M % 256 || N % 256) return 0;
    const int ntm = M / 256, ntn = N / 256, tiles = ntm * ntn, ncu = a4r_cu_count();
    const int rem = tiles % ncu;
    if (tiles * 4 <= ncu && g_variant == 2) return 0;
    int pf_ = 0, kp_ = 0;
    if (a4r_gemm_tail_plan(M, N, &pf_, &kp_)) return M;
    if (tiles > ncu && rem > 0 && rem * 4 <= ncu && rem % ntn == 0 && g_variant < 4) return (ntm - rem / ntn) * 256;
    return M;
}

// g -> g1 (rows [0, head_rows), keeps q8_tiled) + g2 (the rows behind them, 8-bit derivative row-major: whole row panels, so the byte
// offset of row head_rows is the same in both orders)
static void split_rows(const a4r_gemm_t& g, int64_t head_rows, int isz, int osz, int c2sz, int presz, a4r_gemm_t& g1, a4r_gemm_t& g2) {
    g1 = g;
    g2 = g;
    g1.M = (int)head_rows;
    g2.M = g.M - (int)head_rows;
    g2.q8_tiled = 0;
    g2.drop_row0 = g.drop_row0 + head_rows;
    auto adv = [&](const void* p, int64_t ld, int sz) { return p ? (const void*)((const char*)p + head_rows * ld * sz) : nullptr; };
    g2.A = adv(g.A, g.lda, isz);
    g2.C = const_cast<void*>(adv(g.C, g.ldc, g.c_fp8 ? 1 : osz));
    g2.C2 = const_cast<void*>(adv(g.C2, g.ldc2, c2sz));
    g2.R1 = adv(g.R1, g.ldr1, osz);
    g2.R2 = adv(g.R2, g.ldr2, osz);
    g2.Pre = adv(g.Pre, g.ldpre, presz);
    if (g.scale_a) g2.scale_a = g.scale_a + head_rows;
    if (g.c_scale_out) g2.c_scale_out = g.c_scale_out + head_rows;
}

extern "C" int a4r_gemm_variant(int v) {
    const int old = g_variant;
    if (v == 3 || v == 5) return -1;         // the four-wave forms were measured slower and are no longer part of the library
#ifdef A4R_WITH_W4
    if (v == 8 || v == 9) { a4r_gemm_w4(v == 9); return old; }       // tools-only build (make W4=1): the four-wave hand-scheduled 256-tile kernel of tools/w4/: 8 = off, 9 = on
#else
    if (v == 8 || v == 9) return -1;         // the four-wave hand-scheduled kernel is not part of the library (tools/w4/README.md)
#endif
    if (v == 6 || v == 7) { g_tn256_on = v == 7; return old; }      // large weight gradients (a4r_gemm_tn256.hip): 6 = on the 64-tile kernels, 7 = default
    if (v >= 0 && v <= 4) g_tn_variant = v != 0;
    if (v >= 0 && v <= 4) g_variant = v;     // 0/1: 128-tile kernels, 2: automatic (default), 4: eight-wave 256 tile forced
    return old;
}

extern "C" int a4r_gemm_nt(void* stream, const a4r_gemm_t* gp) {
    if (!gp) return A4R_EINVAL;
    const a4r_gemm_t& g = *gp;
    if (!g.A || !g.B || !g.C) return A4R_EINVAL;
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.M % 128 || g.N % 64 || g.K % 64) return A4R_EINVAL;
    const int isz = g.in_dtype == A4R_F32 ? 4 : (g.in_dtype == A4R_FP8 ? 1 : 2), osz = g.out_dtype == A4R_F32 ? 4 : 2;
    if ((g.in_dtype != A4R_F32 && g.in_dtype != A4R_BF16 && g.in_dtype != A4R_FP8) || (g.out_dtype != A4R_F32 && g.out_dtype != A4R_BF16)) return A4R_EINVAL;
    if (g.in_dtype == A4R_FP8) {          // e4m3 operands: the 256-tile kernel only, scales required
        if (g.M % 256 || g.N % 256 || g.K % 128 || g.out_dtype != A4R_BF16 || !g.scale_a || !g.scale_b) return A4R_EINVAL;
        if (g.lda < g.K || g.ldb < g.K || g.ldc < g.N || g.lda % 16 || g.ldb % 16 || (g.ldc * osz) % 16) return A4R_EINVAL;
        if (g.c2_mode < 0 || g.c2_mode > 2 || (g.C2 && g.c2_mode == 2 && g.act != A4R_ACT_GELU)) return A4R_EINVAL;
        if (!aligned16(g.A) || !aligned16(g.B) || !aligned16(g.C) || (g.C2 && (!aligned16(g.C2) || (g.ldc2 * (g.c2_mode == 2 ? 1 : osz)) % 16 || g.ldc2 < g.N))) return A4R_EINVAL;
        if ((g.R1 && (!aligned16(g.R1) || (g.ldr1 * osz) % 16)) || (g.R2 && (!aligned16(g.R2) || (g.ldr2 * osz) % 16))) return A4R_EINVAL;
        if (g.dact != A4R_ACT_NONE && (g.dact != A4R_DACT_MULQ8_ || !g.Pre || !aligned16(g.Pre) || g.ldpre % 16 || g.ldpre < g.N)) return A4R_EINVAL;   // (the 8-bit derivative form only)
        if (g.c_fp8 && (g.ldc % 16 || g.R1 || g.R2)) return A4R_EINVAL;                                     // e4m3 C: ldc counts bytes
        if (g.q8_tiled && ((g.C2 && g.c2_mode == 2 && g.ldc2 != g.N) || (g.dact == A4R_DACT_MULQ8_ && g.ldpre != g.N))) return A4R_EINVAL;
        if (g.drop_p < 0.f || g.drop_p >= 1.f) return A4R_EINVAL;
        hipStream_t s8 = reinterpret_cast<hipStream_t>(stream);
        const bool q8op = (g.C2 && g.c2_mode == 2) || g.dact == A4R_DACT_MULQ8_;
        if (g.q8_tiled && q8op) {
            // The tile-native 8-bit derivative covers the rows a bf16 launch over the same [M, N] would give the 256-tile kernel
            // (a4r_gemm_rows_256); the rows behind them are row-major.  e4m3 operands always run on the 256-tile kernel, so the launch
            // is cut at the same row: its writer or reader on the other side may be a bf16 launch (fp8 FFN-up + bf16 dgrad).
            const int head_rows = a4r_gemm_rows_256(g.M, g.N);
            if (head_rows < g.M) {
                a4r_gemm_t g1, g2;
                split_rows(g, head_rows, 1, osz, g.c2_mode == 2 ? 1 : osz, 1, g1, g2);
                if (head_rows > 0) {
                    const int rc = a4r_gemm_nt_256(s8, g1);
                    if (rc != 0) return rc == 1 ? A4R_EINVAL : rc;
                }
                const int rc = a4r_gemm_nt_256(s8, g2);
                return rc == 1 ? A4R_EINVAL : rc;
            }
        }
        const int rc = a4r_gemm_nt_256(s8, g);
        return rc == 1 ? A4R_EINVAL : rc;
    }
    if (g.lda < g.K || g.ldb < g.K || g.ldc < g.N) return A4R_EINVAL;
    if ((g.lda * isz) % 16 || (g.ldb * isz) % 16 || (g.ldc * osz) % 16) return A4R_EINVAL;
    if (!aligned16(g.A) || !aligned16(g.B) || !aligned16(g.C)) return A4R_EINVAL;
    // 8-bit stored derivative (c2_mode 2 / A4R_DACT_MUL_Q8): one byte per element, bf16 GEMMs only, GELU only
    const bool c2q8 = g.C2 && g.c2_mode == 2, preq8 = g.dact == A4R_DACT_MULQ8_;
    if ((c2q8 && (g.out_dtype != A4R_BF16 || g.act != A4R_ACT_GELU)) || (preq8 && g.out_dtype != A4R_BF16) || g.c2_mode < 0 || g.c2_mode > 2) return A4R_EINVAL;
    const int c2sz = c2q8 ? 1 : osz, presz = preq8 ? 1 : osz;
    if (g.C2 && (!aligned16(g.C2) || (g.ldc2 * c2sz) % 16 || g.ldc2 < g.N)) return A4R_EINVAL;
    if (g.R1 && (!aligned16(g.R1) || (g.ldr1 * osz) % 16 || g.ldr1 < g.N)) return A4R_EINVAL;
    if (g.R2 && (!aligned16(g.R2) || (g.ldr2 * osz) % 16 || g.ldr2 < g.N)) return A4R_EINVAL;
    if (g.dact != A4R_ACT_NONE && (!g.Pre || !aligned16(g.Pre) || (g.ldpre * presz) % 16 || g.ldpre < g.N)) return A4R_EINVAL;
    if (g.drop_p < 0.f || g.drop_p >= 1.f) return A4R_EINVAL;
    if (g.dact == A4R_DACT_MUL_ && !g.Pre) return A4R_EINVAL;
    if (g.q8_tiled && ((c2q8 && g.ldc2 != g.N) || (preq8 && g.ldpre != g.N))) return A4R_EINVAL;      // tile-native 8-bit derivative: a dense [M, N] byte tensor
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (g_variant == 2 && g.K == 64 && g.N >= 256) {
        const int rc = a4r_gemm_nt_skinnyk(s, g);
        if (rc != 1) return rc;
    }
    if (g_variant >= 2 && g.M % 256 == 0 && g.N % 256 == 0 && (g.K * isz) % 128 == 0) {
        // How many leading rows the 256-tile kernel takes is a4r_gemm_rows_256(M, N) -- shared with the fp8 branch above, so that the
        // writer and the reader of a tile-native 8-bit derivative agree on its layout whatever their operand types.
        const int head_rows = a4r_gemm_rows_256(g.M, g.N);
        if (head_rows == 0) goto small_tiles;
        if (head_rows < g.M) {
            a4r_gemm_t g1, g2;
            split_rows(g, head_rows, isz, osz, c2sz, presz, g1, g2);
            const int rc = run_256(s, g1);
            if (rc == 0) {
                if (g.in_dtype == A4R_BF16 && g.out_dtype == A4R_BF16) return launch_bn<bf16_t, bf16_t>(s, g2);
                if (g.in_dtype == A4R_F32 && g.out_dtype == A4R_F32) return launch_bn<float, float>(s, g2);
                if (g.in_dtype == A4R_BF16 && g.out_dtype == A4R_F32) return launch_bn<bf16_t, float>(s, g2);
                return launch_bn<float, bf16_t>(s, g2);
            }
            if (rc != 1) return rc;
        } else {
            const int rc = run_256(s, g);        // (a partial last round may run as short tiles inside the same launch: a4r_gemm_tail_plan)
            if (rc != 1) return rc;              // 1 = this (dtype, act, dact) combination has no large-tile instantiation
        }
    }
small_tiles:
    if (g.N == 64 && g_variant >= 2) {
        const int rc = a4r_gemm_nt_skinny64(s, g);
        if (rc != 1) return rc;
    }
    if (g.in_dtype == A4R_BF16 && g.out_dtype == A4R_BF16) return launch_bn<bf16_t, bf16_t>(s, g);
    if (g.in_dtype == A4R_F32 && g.out_dtype == A4R_F32) return launch_bn<float, float>(s, g);
    if (g.in_dtype == A4R_BF16 && g.out_dtype == A4R_F32) return launch_bn<bf16_t, float>(s, g);
    return launch_bn<float, bf16_t>(s, g);
}
