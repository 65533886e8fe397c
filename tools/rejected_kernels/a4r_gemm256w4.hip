// gemm_nt_256w4_kernel: the 256 x 256 persistent tile of a4r_gemm256.hip with FOUR waves (2 x 2, one per SIMD) instead of eight.
// Each wave owns 128 x 128 of the tile = 8 x 8 MFMA 16x16 tiles = 256 accumulator registers (the compiler keeps them in the
// AGPR half of the 512-register budget of a one-wave-per-SIMD kernel).  Why: per K-tile a wave now reads 16 + 16 operand chunks
// for 128 MFMAs (0.25 ds_read_b128 per MFMA instead of 0.375) and the CU reads 128 KiB of LDS instead of 192 KiB, there are
// half as many waves to hold at each of the 4 barriers per K-tile, and every phase carries 32 MFMAs to hide its reads, its DMA
// issue and the barrier behind.  Everything else is the 8-wave kernel: four 16 KiB units per K-tile (A_lo, B_lo, B_hi, A_hi;
// here BOTH matrices are cut as rows {0-63, 128-191} | {64-127, 192-255}) in a 2-deep ring, LDS-DMA kept in flight across
// barriers with counted vmcnt (4 DMA instructions per wave and unit => "all but the newest 4 units" is vmcnt(16)), B_lo / B_hi
// roles alternating per K-tile so the next phase's fragments never land in registers the current MFMAs read, reads and DMA
// issue interleaved with the MFMAs, next tile's prologue in flight during the epilogue, transposed accumulators + permlane16
// pairing for 16-byte stores, compile-time ACT / DACT.
#include "a4r_gemm_epi.h"

namespace {

constexpr int UNIT_BYTES = 16384;
enum { U_ALO = 0, U_BLO = 1, U_BHI = 2, U_AHI = 3 };

A4R_DEV void glds16(const void* base, uint32_t voff, uint32_t lds_dst) {
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

template <typename TI, typename TO, int ACT, int DACT>
__global__ void __launch_bounds__(256, 1) gemm_nt_256w4_kernel(const a4r_gemm_t p, int ntm, int ntn, uint32_t thr16, float keep_scale) {
    constexpr int ROWB = 128;
    constexpr int KT = ROWB / (int)sizeof(TI);
    __shared__ __attribute__((aligned(16))) char lds[8 * UNIT_BYTES];      // [buffer 2][unit 4][128 rows][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nt = ntm * ntn;
    const int q8 = nt >> 3, r8 = nt & 7;
    auto tile_of = [&](int vb) {      // bijective XCD-aware remap (see a4r_gemm256.hip)
        const int xcd = vb & 7, j = vb >> 3;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    };
    int vb = blockIdx.x;
    int Lt = tile_of(vb);
    int tm = Lt / ntn, tn = Lt % ntn;

    const int lda = p.lda, ldb = p.ldb;
    const int nk = p.K / KT;
    const TI* Ap = reinterpret_cast<const TI*>(p.A);
    const TI* Bp = reinterpret_cast<const TI*>(p.B);
    const char* Abase = reinterpret_cast<const char*>(Ap + (size_t)tm * 256 * lda);
    const char* Bbase = reinterpret_cast<const char*>(Bp + (size_t)tn * 256 * ldb);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // per-lane source offsets of this wave's four DMA instructions of each unit kind (bytes from the tile base)
    uint32_t offA_lo[4], offA_hi[4], offB_lo[4], offB_hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ul = 8 * (4 * wave + i) + (lane >> 3);                 // unit row 0..127
        const int c = (lane & 7) ^ ((ul >> 1) & 7);                      // source chunk for linear LDS slot (lane & 7)
        const int r = ul + (ul >> 6) * 64;                               // *_lo tile row; *_hi = + 64
        offA_lo[i] = (uint32_t)(r * lda * (int)sizeof(TI) + c * 16);
        offA_hi[i] = (uint32_t)((r + 64) * lda * (int)sizeof(TI) + c * 16);
        offB_lo[i] = (uint32_t)(r * ldb * (int)sizeof(TI) + c * 16);
        offB_hi[i] = (uint32_t)((r + 64) * ldb * (int)sizeof(TI) + c * 16);
    }
    const uint32_t dma_dst = lds0 + (uint32_t)(4 * wave) * 1024u;       // + buffer*4*UNIT + kind*UNIT + i*1024

#define W4_ISSUE(kind_, tile_, base_, off_)                                                                          \
    if ((tile_) < nk) {                                                                                              \
        const char* src_ = (base_) + (size_t)(tile_) * ROWB;                                                         \
        const uint32_t dst_ = dma_dst + (uint32_t)((((tile_) & 1) * 4 + (kind_)) * UNIT_BYTES);                       \
        glds16(src_, off_[0], dst_);                                                                                 \
        glds16(src_, off_[1], dst_ + 1024u);                                                                         \
        glds16(src_, off_[2], dst_ + 2048u);                                                                         \
        glds16(src_, off_[3], dst_ + 3072u);                                                                         \
    }
#define W4_WAIT_BARRIER(steady_)                                                                                     \
    if (steady_) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");                                        \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                 \
    __builtin_amdgcn_s_barrier();                                                                                    \
    asm volatile("" ::: "memory");
    // MFMA j of a phase (0..31): ks = j >> 4, mi = (j >> 2) & 3, ni = j & 3
#define W4_MFMA_J(ax_, bx_, m0_, n0_, j_) \
    Mma<TI>::mma(bx_[(j_) & 3][(j_) >> 4], ax_[((j_) >> 2) & 3][(j_) >> 4], acc[(m0_) + (((j_) >> 2) & 3)][(n0_) + ((j_) & 3)]);
#define W4_RD(dst_, off_, buf_, unit_, r_) \
    dst_[(r_) >> 1][(r_) & 1] = *reinterpret_cast<const uint4*>(lds + ((buf_) * 4 + (unit_)) * UNIT_BYTES + off_[(r_) >> 1][(r_) & 1]);
    // one phase: barrier | 2 MFMAs | this phase's 4 LDS-DMA | (2 MFMAs + 1 fragment read of the NEXT phase) x 8 | 14 MFMAs
#define W4_PHASE(steady_, issue_, dst_, off_, buf_, unit_, ax_, bx_, m0_, n0_)                         \
    W4_WAIT_BARRIER(steady_)                                                                          \
    _Pragma("unroll") for (int k_ = 0; k_ < 16; ++k_) {                                               \
        W4_MFMA_J(ax_, bx_, m0_, n0_, 2 * k_)                                                         \
        W4_MFMA_J(ax_, bx_, m0_, n0_, 2 * k_ + 1)                                                     \
        if (k_ == 0) { issue_ }                                                                       \
        if (k_ >= 1 && k_ <= 8) { W4_RD(dst_, off_, buf_, unit_, k_ - 1) }                            \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }

    f32x4_t acc[8][8];
    const int fr = lane & 15, kg = lane >> 4;
    int a_off[4][2], b_off[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int ra = wm * 64 + t * 16 + fr, rb = wn * 64 + t * 16 + fr, ch = ks * 4 + kg;
            a_off[t][ks] = ra * ROWB + ((ch ^ ((ra >> 1) & 7)) << 4);
            b_off[t][ks] = rb * ROWB + ((ch ^ ((rb >> 1) & 7)) << 4);
        }

    // 7 units in stream order (K-tiles 0 and 1; the odd K-tile streams B_hi before B_lo)
#define W4_PROLOGUE()                          \
    W4_ISSUE(U_ALO, 0, Abase, offA_lo)         \
    W4_ISSUE(U_BLO, 0, Bbase, offB_lo)         \
    W4_ISSUE(U_BHI, 0, Bbase, offB_hi)         \
    W4_ISSUE(U_AHI, 0, Abase, offA_hi)         \
    W4_ISSUE(U_ALO, 1, Abase, offA_lo)         \
    W4_ISSUE(U_BHI, 1, Bbase, offB_hi)         \
    W4_ISSUE(U_BLO, 1, Bbase, offB_lo)
    W4_PROLOGUE()
    if (nk >= 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");        // 28 issued: A_lo(0), B_lo(0) have landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const GemmEpi<TO> epi = make_epi<TO>(p, thr16, keep_scale);

  for (;;) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    uint4 af[4][2], a1[4][2], b0[4][2], b1[4][2];
#pragma unroll
    for (int r = 0; r < 8; ++r) { W4_RD(af, a_off, 0, U_ALO, r) W4_RD(b0, b_off, 0, U_BLO, r) }
    // even K-tile: (A_lo,B_lo) (A_lo,B_hi) (A_hi,B_hi) (A_hi,B_lo)      odd: (A_lo,B_hi) (A_lo,B_lo) (A_hi,B_lo) (A_hi,B_hi)
    // (on the last K-tile the reads of the "next" tile fetch stale LDS bytes into registers nobody uses: harmless)
    for (int u = 0; u < nk; u += 2) {
        {
            const bool steady = (u + 2 < nk);
            W4_PHASE(steady, W4_ISSUE(U_AHI, u + 1, Abase, offA_hi), b1, b_off, 0, U_BHI, af, b0, 0, 0)
            W4_PHASE(steady, W4_ISSUE(U_ALO, u + 2, Abase, offA_lo), a1, a_off, 0, U_AHI, af, b1, 0, 4)
            W4_PHASE(steady, W4_ISSUE(U_BLO, u + 2, Bbase, offB_lo), af, a_off, 1, U_ALO, a1, b1, 4, 4)
            W4_PHASE(steady, W4_ISSUE(U_BHI, u + 2, Bbase, offB_hi), b1, b_off, 1, U_BHI, a1, b0, 4, 0)
        }
        if (u + 1 < nk) {
            const bool steady = (u + 3 < nk);
            W4_PHASE(steady, W4_ISSUE(U_AHI, u + 2, Abase, offA_hi), b0, b_off, 1, U_BLO, af, b1, 0, 4)
            W4_PHASE(steady, W4_ISSUE(U_ALO, u + 3, Abase, offA_lo), a1, a_off, 1, U_AHI, af, b0, 0, 0)
            W4_PHASE(steady, W4_ISSUE(U_BHI, u + 3, Bbase, offB_hi), af, a_off, 0, U_ALO, a1, b0, 4, 0)
            W4_PHASE(steady, W4_ISSUE(U_BLO, u + 3, Bbase, offB_lo), b0, b_off, 0, U_BLO, a1, b1, 4, 4)
        }
    }

    // ---- epilogue straight from the (transposed) accumulators, next tile's first units already in flight
    const int tm_done = tm, tn_done = tn;
    vb += gridDim.x;
    const bool more = vb < nt;
    if (more) {
        Lt = tile_of(vb);
        tm = Lt / ntn;
        tn = Lt % ntn;
        Abase = reinterpret_cast<const char*>(Ap + (size_t)tm * 256 * lda);
        Bbase = reinterpret_cast<const char*>(Bp + (size_t)tn * 256 * ldb);
        W4_PROLOGUE()
    }
    const size_t grow0 = (size_t)tm_done * 256 + wm * 128 + fr;
    const int gcolp = tn_done * 256 + wn * 128 + (kg & 1) * 16 + (kg >> 1) * 8;      // + pair * 32
#define W4_EPI_PAIR(mi_, pr_)                                                                                               \
    {                                                                                                                       \
        float v_[8], b_[8];                                                                                                 \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                                                  \
            const auto sw_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[mi_][2 * (pr_)][r_]),                     \
                                                              __float_as_uint(acc[mi_][2 * (pr_) + 1][r_]), false, false);  \
            v_[r_] = __uint_as_float(sw_[0]);                                                                               \
            v_[4 + r_] = __uint_as_float(sw_[1]);                                                                           \
        }                                                                                                                   \
        _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) b_[e_] = epi.bias ? epi.bias[gcolp + (pr_) * 32 + e_] : 0.f;       \
        epilogue_n<TO, 8, ACT, DACT>(v_, b_, grow0 + (mi_) * 16, gcolp + (pr_) * 32, epi);                                  \
    }
#define W4_EPI_ROW(mi_) W4_EPI_PAIR(mi_, 0) W4_EPI_PAIR(mi_, 1) W4_EPI_PAIR(mi_, 2) W4_EPI_PAIR(mi_, 3)
    W4_EPI_ROW(0) W4_EPI_ROW(1) W4_EPI_ROW(2) W4_EPI_ROW(3) W4_EPI_ROW(4) W4_EPI_ROW(5) W4_EPI_ROW(6) W4_EPI_ROW(7)
#undef W4_EPI_ROW
#undef W4_EPI_PAIR
    if (!more) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#undef W4_PROLOGUE
#undef W4_ISSUE
#undef W4_WAIT_BARRIER
#undef W4_MFMA_J
#undef W4_RD
#undef W4_PHASE
}

}  // namespace

int a4r_cu_count();

namespace {

template <typename TI, typename TO, int ACT, int DACT>
int launch_w4(hipStream_t s, const a4r_gemm_t& g) {
    const int ntm = g.M / 256, ntn = g.N / 256;
    const int n_cu = a4r_cu_count();
    const int grid = ntm * ntn < n_cu ? ntm * ntn : n_cu;
    hipLaunchKernelGGL((gemm_nt_256w4_kernel<TI, TO, ACT, DACT>), dim3(grid), dim3(256), 0, s, g, ntm, ntn,
                       a4r_thr16(g.drop_p), a4r_keep_scale(g.drop_p));
    return a4r_launch_status();
}

template <typename T>
int dispatch_same(hipStream_t s, const a4r_gemm_t& g) {
    if (g.act == A4R_ACT_NONE && g.dact == A4R_ACT_NONE) return launch_w4<T, T, A4R_ACT_NONE, A4R_ACT_NONE>(s, g);
    if (g.act == A4R_ACT_GELU && g.dact == A4R_ACT_NONE) return launch_w4<T, T, A4R_ACT_GELU, A4R_ACT_NONE>(s, g);
    if (g.act == A4R_ACT_NONE && g.dact == A4R_DACT_MUL_) return launch_w4<T, T, A4R_ACT_NONE, A4R_DACT_MUL_>(s, g);
    return 1;
}

}  // namespace

// bf16 in / bf16 out only (the training step's big GEMMs); returns 1 when the combination is not instantiated
int a4r_gemm_nt_256w4(hipStream_t s, const a4r_gemm_t& g) {
    if (g.bias && (reinterpret_cast<uintptr_t>(g.bias) & 3u)) return 1;
    if (g.in_dtype == A4R_BF16 && g.out_dtype == A4R_BF16) return dispatch_same<bf16_t>(s, g);
    return 1;
}
