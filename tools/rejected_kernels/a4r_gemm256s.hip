// gemm_nt_256s_kernel: 256 x 256 persistent output tile, FOUR waves (2 x 2, one per SIMD, 128 x 128 = 8 x 8 MFMA 16x16 tiles
// = 256 accumulator registers each), K-tile-granular double buffer, ONE continuous LDS-DMA stream.
//
// Why another schedule of the four-wave tile (a4r_gemm256w4.hip measured 518 TF/s against 875 for the eight-wave kernel): with
// one wave per SIMD nothing but the wave's own instruction order hides anything, and that kernel issued its four LDS-DMA of a
// phase back to back behind a barrier (each with an m0 save / restore and an SGPR reload) -- an LDS-DMA issued into a busy
// texture-address path holds the wave's single in-order issue port for ~100 cycles, i.e. 6 MFMA slots, four times per 32 MFMAs.
// Here the 128 MFMAs of a K-tile form one straight line and everything else is threaded through it one instruction at a time:
//
//   MFMA   0.. 63  contraction half ks = 0 (fragments F0, read during the previous K-tile)
//          1.. 16  16 x ds_read_b128, one per MFMA: fragments F1 (ks = 1) of THIS K-tile                     [buffer g & 1]
//         20       s_waitcnt lgkmcnt(0); s_barrier     -- every wave is done reading buffer g & 1 (WAR)
//         21.. 56  8 LDS-DMA (1 KiB each), one per five MFMAs: A rows of stream position g + 2               -> buffer g & 1
//   MFMA  64..127  contraction half ks = 1 (fragments F1)
//         61.. 96  8 LDS-DMA: B rows of stream position g + 2                                                -> buffer g & 1
//         97       s_waitcnt vmcnt(16); s_barrier      -- position g + 1 has landed for every wave (RAW; not in K-tile 0)
//         98..113  16 x ds_read_b128: fragments F0 of the NEXT K-tile                                        [buffer (g+1) & 1]
// The positions are template parameters (R1S, WB, D0, DS, RW, R0S; -DA4R_SCHED_SWEEP + tools/gemm_sched_sweep.py).  Measured rule:
// an LDS-DMA issued while ds_reads are in flight is expensive -- every schedule that let the DMA window overlap a read burst
// lost 30-35 % (688-732 vs 922-1063 TF/s at K = 3072) -- so the K-tile is three disjoint windows: reads, DMA, reads.
//
// (two barriers per K-tile; m0 is written by one s_add_u32 per DMA -- the compiler never allocates it -- and the source is an
// SGPR base + one precomputed VGPR offset per DMA).  The DMA stream does not know about output tiles: position g + 2 of the last
// two K-tiles of an output tile are the first two K-tiles of the workgroup's NEXT output tile, so the ring never drains, the
// epilogue runs with 64 KiB already in flight, and the vmcnt arithmetic is the same in every K-tile (stores and residual loads
// of the epilogue are older than the DMAs that follow them, so "all but the newest 16" still covers them).
// LDS image, swizzle, fragment layout, transposed accumulators and the epilogue are those of a4r_gemm256.hip / a4r_gemm256w4.hip.
#include "a4r_gemm_epi.h"

namespace {

constexpr uint32_t S_BUF = 65536, S_BOFF = 32768;

// one 1-KiB LDS-DMA: LDS[lds_base + IMM + lane*16 .. +16) <- global[base + voff .. +16)
template <int IMM>
A4R_DEV void sdma(const void* base, uint32_t voff, uint32_t lds_base) {
    asm volatile("s_add_u32 m0, %2, %3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_base), "n"(IMM) : "memory", "scc");
}

// Schedule parameters (MFMA indices inside a K-tile): R1S = spacing of the F1 reads (from MFMA 1), WB = WAR barrier,
// D0 / DS = first DMA slot and slot spacing (16 slots), RW = RAW wait + barrier (the F0 reads of the next K-tile follow, one per two MFMAs).
template <int N_> A4R_DEV void s_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <typename TI, typename TO, int ACT, int DACT, int R1S, int WB, int D0, int DS, int RW, int R0S>
__global__ void __launch_bounds__(256, 1) gemm_nt_256s_kernel(const a4r_gemm_t p, int ntm, int ntn, uint32_t thr16, float keep_scale) {
    constexpr int ROWB = 128;
    constexpr int KT = ROWB / (int)sizeof(TI);
    __shared__ __attribute__((aligned(16))) char lds[2 * S_BUF];          // [buffer 2][A 256 rows | B 256 rows][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int nt = ntm * ntn;
    const int q8 = nt >> 3, r8 = nt & 7;
    auto tile_of = [&](int vb) {      // bijective XCD-aware remap (see a4r_gemm256.hip)
        const int xcd = vb & 7, j = vb >> 3;
        return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    };
    const int lda = p.lda, ldb = p.ldb;
    const int nk = p.K / KT;
    const TI* Ap = reinterpret_cast<const TI*>(p.A);
    const TI* Bp = reinterpret_cast<const TI*>(p.B);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // ---- the DMA stream: (output tile, K-tile) positions in the order the workgroup consumes them
    // DMA i (0..7) of a wave stages tile rows 64*wave + 8*i + (lane >> 3); the swizzle of row r is (r >> 1) & 7 = (4*(i & 1) + (lane >> 4)) & 7,
    // so two per-lane offsets (i even / odd) + a scalar row-block base (wave and i >> 1) address all eight.
    uint32_t voffA[2], voffB[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int ul = 8 * b + (lane >> 3);
        const int c = (lane & 7) ^ ((ul >> 1) & 7);                      // source chunk for linear LDS slot (lane & 7)
        voffA[b] = (uint32_t)(ul * lda * (int)sizeof(TI) + c * 16);
        voffB[b] = (uint32_t)(ul * ldb * (int)sizeof(TI) + c * 16);
    }
    const size_t strA = (size_t)16 * lda * sizeof(TI), strB = (size_t)16 * ldb * sizeof(TI);     // bytes per 16 tile rows
    int d_vb = blockIdx.x, d_kt = 0;
    bool d_valid = true;
    const char* dA;
    const char* dB;
    uint32_t d_dst = lds0 + (uint32_t)(8 * wave) * 1024u;               // + parity * S_BUF (toggled by d_advance)
    auto d_seek = [&]() {
        const int Lt = tile_of(d_vb);
        dA = reinterpret_cast<const char*>(Ap + ((size_t)(Lt / ntn) * 256 + 64 * wave) * lda);
        dB = reinterpret_cast<const char*>(Bp + ((size_t)(Lt % ntn) * 256 + 64 * wave) * ldb);
    };
    auto d_advance = [&]() {
        d_dst ^= S_BUF;
        dA += ROWB;
        dB += ROWB;
        if (++d_kt == nk) {
            d_kt = 0;
            d_vb += gridDim.x;
            d_valid = d_vb < nt;
            if (d_valid) d_seek();
        }
    };
    // (i_ is a constant after unrolling; the switch only turns it into a template argument = an asm immediate)
#define S_DMA_(i_, off_, src_, str_, voff_)                                                            \
    if (d_valid) switch (i_) {                                                                         \
        case 0: sdma<(off_) + 0 * 1024>(src_, voff_[0], d_dst); break;                                 \
        case 1: sdma<(off_) + 1 * 1024>(src_, voff_[1], d_dst); break;                                 \
        case 2: sdma<(off_) + 2 * 1024>(src_ + str_, voff_[0], d_dst); break;                          \
        case 3: sdma<(off_) + 3 * 1024>(src_ + str_, voff_[1], d_dst); break;                          \
        case 4: sdma<(off_) + 4 * 1024>(src_ + 2 * str_, voff_[0], d_dst); break;                      \
        case 5: sdma<(off_) + 5 * 1024>(src_ + 2 * str_, voff_[1], d_dst); break;                      \
        case 6: sdma<(off_) + 6 * 1024>(src_ + 3 * str_, voff_[0], d_dst); break;                      \
        default: sdma<(off_) + 7 * 1024>(src_ + 3 * str_, voff_[1], d_dst); break;                     \
    }
#define S_DMA_A(i_) S_DMA_(i_, 0, dA, strA, voffA)
#define S_DMA_B(i_) S_DMA_(i_, (int)S_BOFF, dB, strB, voffB)
    d_seek();
#pragma unroll
    for (int i = 0; i < 8; ++i) { S_DMA_A(i) }
#pragma unroll
    for (int i = 0; i < 8; ++i) { S_DMA_B(i) }
    d_advance();
#pragma unroll
    for (int i = 0; i < 8; ++i) { S_DMA_A(i) }
#pragma unroll
    for (int i = 0; i < 8; ++i) { S_DMA_B(i) }
    d_advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // invariant at the start of every output tile: positions g and g + 1 have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- fragment addressing: A rows wm*128 + mi*16 + (lane & 15) -> + mi * 2048 B; B likewise behind S_BOFF
    const int fr = lane & 15, kg = lane >> 4;
    int adA[2], adB[2];                                                  // [ks], buffer parity added at the point of use
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int ra = wm * 128 + fr, rb = wn * 128 + fr, ch = ks * 4 + kg;
        adA[ks] = ra * ROWB + ((ch ^ ((ra >> 1) & 7)) << 4);
        adB[ks] = (int)S_BOFF + rb * ROWB + ((ch ^ ((rb >> 1) & 7)) << 4);
    }
    uint4 fa[2][8], fb[2][8];
    f32x4_t acc[8][8];
    // read r (0..15) of a fragment set: B0, A0..A7, B1..B7 -- the order in which the MFMAs of a half first touch them
#define S_RD(ks_, r_, par_)                                                                                           \
    if ((r_) == 0) fb[ks_][0] = *reinterpret_cast<const uint4*>(lds + (par_) + adB[ks_]);                             \
    else if ((r_) <= 8) fa[ks_][(r_) - 1] = *reinterpret_cast<const uint4*>(lds + (par_) + adA[ks_] + ((r_) - 1) * 2048); \
    else fb[ks_][(r_) - 8] = *reinterpret_cast<const uint4*>(lds + (par_) + adB[ks_] + ((r_) - 8) * 2048);
    int par = 0;                                                         // byte offset of the buffer the current K-tile reads
#pragma unroll
    for (int r = 0; r < 16; ++r) { S_RD(0, r, 0) }

    int vb = blockIdx.x;
    const GemmEpi<TO> epi = make_epi<TO>(p, thr16, keep_scale);
  for (;;) {                                              // ---- output tiles of this workgroup
    const int Lt = tile_of(vb);
    const int tm = Lt / ntn, tn = Lt % ntn;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            acc[mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            asm volatile("" : "+a"(acc[mi][ni]));         // opaque zeros, in AGPRs: no peeled first K-tile with constant-0 accumulators
        }

    for (int kt = 0; kt < nk; ++kt) {
        const int npar = par ^ (int)S_BUF;
        bool issued = false;
#pragma unroll
        for (int j = 0; j < 128; ++j) {
            const int ks = j >> 6, ni = (j >> 3) & 7, mi = j & 7;
            Mma<TI>::mma(fb[ks][ni], fa[ks][mi], acc[mi][ni]);
            if (j >= 1 && (j - 1) % R1S == 0 && (j - 1) / R1S < 16) { S_RD(1, (j - 1) / R1S, par) }
            if (j == WB) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issued = d_valid;
            }
            if (j >= D0 && (j - D0) % DS == 0 && (j - D0) / DS < 8) { S_DMA_A((j - D0) / DS) }
            if (j >= D0 && (j - D0) % DS == 0 && (j - D0) / DS >= 8 && (j - D0) / DS < 16) { S_DMA_B((j - D0) / DS - 8) }
            if (j == RW) {
                constexpr int n_before = RW < D0 ? 0 : ((RW - D0) / DS + 1 > 16 ? 16 : (RW - D0) / DS + 1);   // slots at or before RW
                if (kt != 0) {                            // (K-tile 0: position g + 1 landed before the previous epilogue, see below)
                    if (issued) s_wait_vm<n_before>();
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
            }
            if (j == D0 + 15 * DS) { d_advance(); }       // after the last slot
            if (j > RW && (j - RW - 1) % R0S == 0 && (j - RW - 1) / R0S < 16) { S_RD(0, (j - RW - 1) / R0S, npar) }
            __builtin_amdgcn_sched_barrier(0);
        }
        par = npar;
    }

    // Drain the DMA stream BEFORE the stores of the epilogue enter the queue (vmcnt counts loads and stores in one order:
    // waiting for a DMA that follows 64 stores means waiting for the stores).  The two positions in flight are the next
    // tile's K-tiles 0 and 1, so its first K-tile needs no vmcnt wait at all and its second waits for stores that are
    // by then more than 200 MFMAs old.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- epilogue straight from the (transposed) accumulators
    const size_t grow0 = (size_t)tm * 256 + wm * 128 + fr;
    const int gcolp = tn * 256 + wn * 128 + (kg & 1) * 16 + (kg >> 1) * 8;      // + pair * 32
    // bias of this lane's 32 columns once per tile; Pre / R1 / R2 of half-row h + 1 (two 8-column groups) are requested before
    // half-row h is finished and stored (epi_issue / epi_finish): with one wave per SIMD a load waited for where it is issued
    // costs a full L2 round trip per group (measured: 10 us per tile for a bias-only epilogue).
    float bias32[4][8];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr)
#pragma unroll
        for (int e = 0; e < 8; ++e) bias32[pr][e] = epi.bias ? epi.bias[gcolp + pr * 32 + e] : 0.f;
    EpiLoads<TO> eld[2][2];
    epi_issue<TO, DACT>(eld[0][0], grow0, gcolp, epi);
    epi_issue<TO, DACT>(eld[0][1], grow0, gcolp + 32, epi);
#define S_EPI_PAIR(mi_, pr_)                                                                                                \
    {                                                                                                                       \
        float v_[8];                                                                                                        \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                                                  \
            const auto sw_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[mi_][2 * (pr_)][r_]),                     \
                                                              __float_as_uint(acc[mi_][2 * (pr_) + 1][r_]), false, false);  \
            v_[r_] = __uint_as_float(sw_[0]);                                                                               \
            v_[4 + r_] = __uint_as_float(sw_[1]);                                                                           \
        }                                                                                                                   \
        epi_finish<TO, ACT, DACT>(v_, bias32[pr_], eld[((pr_) >> 1) & 1][(pr_) & 1], grow0 + (mi_) * 16, gcolp + (pr_) * 32, epi); \
    }
    // half-row h = 2 * mi + (pr >> 1) lives in eld[h & 1]; its successor is requested first
#define S_EPI_ROW(mi_)                                                                                                      \
    epi_issue<TO, DACT>(eld[1][0], grow0 + (mi_) * 16, gcolp + 64, epi);                                                    \
    epi_issue<TO, DACT>(eld[1][1], grow0 + (mi_) * 16, gcolp + 96, epi);                                                    \
    S_EPI_PAIR(mi_, 0) S_EPI_PAIR(mi_, 1)                                                                                   \
    if ((mi_) < 7) {                                                                                                        \
        epi_issue<TO, DACT>(eld[0][0], grow0 + ((mi_) + 1) * 16, gcolp, epi);                                               \
        epi_issue<TO, DACT>(eld[0][1], grow0 + ((mi_) + 1) * 16, gcolp + 32, epi);                                          \
    }                                                                                                                       \
    S_EPI_PAIR(mi_, 2) S_EPI_PAIR(mi_, 3)
    S_EPI_ROW(0) S_EPI_ROW(1) S_EPI_ROW(2) S_EPI_ROW(3) S_EPI_ROW(4) S_EPI_ROW(5) S_EPI_ROW(6) S_EPI_ROW(7)
#undef S_EPI_ROW
#undef S_EPI_PAIR
    vb += gridDim.x;
    if (vb >= nt) break;
  }
#undef S_DMA_A
#undef S_DMA_
#undef S_DMA_B
#undef S_RD
}

}  // namespace

int a4r_cu_count();

namespace {

int g_sched = 0;

template <typename TI, typename TO, int ACT, int DACT, int R1S, int WB, int D0, int DS, int RW, int R0S>
int launch_s_(hipStream_t s, const a4r_gemm_t& g) {
    const int ntm = g.M / 256, ntn = g.N / 256;
    const int n_cu = a4r_cu_count();
    const int grid = ntm * ntn < n_cu ? ntm * ntn : n_cu;
    static_assert(D0 > WB && D0 + 15 * DS <= 127 && RW + 1 + 15 * R0S <= 127 && 1 + 15 * R1S < WB, "schedule out of range");
    hipLaunchKernelGGL((gemm_nt_256s_kernel<TI, TO, ACT, DACT, R1S, WB, D0, DS, RW, R0S>), dim3(grid), dim3(256), 0, s, g, ntm, ntn,
                       a4r_thr16(g.drop_p), a4r_keep_scale(g.drop_p));
    return a4r_launch_status();
}

template <typename TI, typename TO, int ACT, int DACT>
int launch_s(hipStream_t s, const a4r_gemm_t& g) {
#ifdef A4R_SCHED_SWEEP
    if (ACT == 0 && DACT == 0) {
        if (g_sched == 1) return launch_s_<TI, TO, 0, 0, 1, 26, 27, 5, 103, 1>(s, g);
        if (g_sched == 2) return launch_s_<TI, TO, 0, 0, 1, 26, 27, 4, 91, 2>(s, g);
        if (g_sched == 3) return launch_s_<TI, TO, 0, 0, 2, 40, 41, 3, 89, 2>(s, g);
        if (g_sched == 4) return launch_s_<TI, TO, 0, 0, 1, 26, 27, 5, 111, 1>(s, g);
    }
#endif
    return launch_s_<TI, TO, ACT, DACT, 1, 20, 21, 5, 97, 1>(s, g);
}

template <typename T>
int dispatch_same(hipStream_t s, const a4r_gemm_t& g) {
    if (g.act == A4R_ACT_NONE && g.dact == A4R_ACT_NONE) return launch_s<T, T, A4R_ACT_NONE, A4R_ACT_NONE>(s, g);
    if (g.act == A4R_ACT_GELU && g.dact == A4R_ACT_NONE) return launch_s<T, T, A4R_ACT_GELU, A4R_ACT_NONE>(s, g);
    if (g.act == A4R_ACT_NONE && g.dact == A4R_DACT_MUL_) return launch_s<T, T, A4R_ACT_NONE, A4R_DACT_MUL_>(s, g);
    return 1;
}

}  // namespace

#ifdef A4R_SCHED_SWEEP
extern "C" int a4r_gemm_sched(int v) { const int o = g_sched; g_sched = v; return o; }   // schedule sweep (diagnostic builds only)
#endif

// bf16 in / bf16 out only (the training step's big GEMMs); returns 1 when the combination is not instantiated
int a4r_gemm_nt_256s(hipStream_t s, const a4r_gemm_t& g) {
    if (g.bias && (reinterpret_cast<uintptr_t>(g.bias) & 3u)) return 1;
    if (g.in_dtype == A4R_BF16 && g.out_dtype == A4R_BF16) return dispatch_same<bf16_t>(s, g);
    return 1;
}
