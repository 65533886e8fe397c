// gemm_nt_256_kernel: the large-tile path of a4r_gemm_nt (M % 256 == 0, N % 256 == 0, K a multiple of one 128-byte K-tile).
//
// 256 x 256 output tile, 512 threads = 8 waves laid out 2 (M) x 4 (N); each wave owns 128 x 64 = 8 x 4 MFMA 16x16 tiles
// (128 accumulator registers).  One workgroup per CU, two waves per SIMD (wave w and w + 4).
//
// Data movement.  A K-tile (128 B of K per row) is cut into four 16 KiB "units"
//     A_lo = tile rows {0-63, 128-191}   (the upper 64 rows of both M-halves of waves)      A_hi = rows {64-127, 192-255}
//     B_lo = B rows 64w + [0,32)                                                             B_hi = 64w + [32,64)
// held in a 2-deep ring (2 x 64 KiB of LDS) filled by LDS-DMA (global_load_lds_dwordx4 through inline asm: hipcc neither counts
// it nor drains it) that stays IN FLIGHT ACROSS BARRIERS behind COUNTED vmcnt waits -- all but the newest three or four units have landed --
// so a unit has 3 - 4 phases (~1 K-tile) to arrive.  Stream order A_lo(t), B_lo(t), B_hi(t), A_hi(t), A_lo(t+1), ...; since round 3 the
// issues sit in the two phases with the fewest fragment reads (phase 1: B_hi, A_hi of t+1; phase 3: A_lo, B_lo of t+2; A4R_DMA_SCHED below --
// 0 = one unit per phase, the round-2 schedule the hazard notes of this header were first written for).  The XOR swizzle c ^ ((r >> 1) & 7) is applied on the DMA SOURCE address and
// on the fragment read (the LDS image itself is lane-linear), which makes the ds_read_b128 of the fragments conflict-free.
//
// Schedule: a PING-PONG between the two waves of every SIMD.  A K-tile is four phases, one output quadrant of the wave each:
//     (A_lo,B_lo) (A_lo,B_hi) (A_hi,B_hi) (A_hi,B_lo)
// and a phase is, for every wave, the same straight code
//     LOAD segment : ds_read this phase's new fragments (12 / 4 / 8 / 0 reads) | issue 0 / 2 / 0 / 2 units' DMA | counted s_waitcnt vmcnt
//     s_barrier | s_waitcnt lgkmcnt(0) | s_setprio 1 | 16 MFMA | s_setprio 0 | s_barrier
// Waves 4-7 execute ONE extra s_barrier before the loop (waves 0-3 one after it), so the two halves run exactly one barrier
// apart: while waves 0-3 are in their MFMA segment, waves 4-7 are in their LOAD segment and vice versa.  Each SIMD's matrix
// pipe therefore always has exactly ONE wave issuing MFMAs back to back (no arbitration between lock-step partners, which
// cost the previous all-waves-in-phase schedule half of the pipe: 46 % MFMA duty), and a wave's LDS latency is covered by
// its PARTNER's MFMAs, so fragments need no second register set (64 instead of 96 fragment registers).
// Hazards (reads of phase p are issued in LOAD_p; the other half runs one barrier later):
//   RAW  a unit read in phase p is retired by every wave's counted wait in LOAD_(p-1): both halves execute that wait before a
//        barrier the reader passes before LOAD_p.  Phase 0 reads A_lo, B_lo (retired in phase 3 of the previous K-tile),
//        phase 1 B_hi (phase 0), phase 2 A_hi (phase 1).
//   WAR  a slot is re-filled no earlier than two phases after its last read: phase 1 issues B_hi(t+1) (slot last read in phase 1
//        of t-1) and A_hi(t+1) (phase 2 of t-1), phase 3 A_lo(t+2) and B_lo(t+2) (both last read in phase 0 of t).
//   RAW  under A4R_DMA_SCHED 1: see the counted waits next to A4R_KTILE.
//
// Between output tiles (a workgroup is persistent and walks 2 - 8 tiles per launch; tools/gemm_timeline.py stamps this part):
//   * the unit stream does NOT stop at the end of a tile's K range: with an even K-tile count the issue slots of the last two
//     K-tiles fetch K-tiles 0 and 1 (first six units) of the workgroup's NEXT output tile into the same ring positions (same issue
//     pattern as the steady state, so the RAW / WAR arguments above hold unchanged); odd counts issue a six-unit prologue instead;
//   * the accumulators are zeroed inside the LOAD segments of the first K-tile (the compiler peels it into MFMAs with C = 0);
//   * the epilogue (straight from the accumulators, no LDS) is instantiated per set of optional pieces (EF: dropout, R1, R2, C2), reads
//     its Pre / R1 operands of the whole tile up front, and is followed by a counted vmcnt(16): the stores drain behind the next K loop;
//   * workgroups that own one tile fewer than the busiest of their XCD start a fraction of a tile period late (A4R_GEMM_STAGGER).
#include <stdlib.h>
#include "a4r_gemm_epi.h"

#ifndef A4R_C2Q8_CT
#define A4R_C2Q8_CT 1      /* 0 (A/B builds): the GELU + 8-bit-derivative launch tests c2_mode / the C2 pointer per group at run time (round 3) */
#endif
#ifndef A4R_PF_Q8
#define A4R_PF_Q8 8        /* rows of 16 the 8-bit Pre operand is requested ahead of its use (1, 2, 4 or 8 = the whole tile up front) */
#endif
#ifndef A4R_PF_R1
#define A4R_PF_R1 8        /* likewise the residual operand R1 */
#endif
#ifdef A4R_STAMP
// diagnostic build only (-DA4R_STAMP, tools/gemm_stamps.py): (s_memtime, s_memrealtime) of wave 0 of every workgroup at the start and the
// end of the K loop of its first four output tiles -> shader cycles per K loop and the clock the chip holds (cycles / (realtime ticks / 100 MHz)).
// Written to a buffer of its own, never read by the kernel.
__device__ unsigned long long g_a4r_stamps[256 * 4 * 4];
extern "C" int a4r_debug_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_a4r_stamps), sizeof(g_a4r_stamps)) == hipSuccess ? 0 : -2;
}
#define A4R_LOOP_STAMP(k_)                                                                                                   \
    if (tid == 0 && tile_no_ < 4 && blockIdx.x < 256) {                                                                     \
        g_a4r_stamps[(blockIdx.x * 4 + tile_no_) * 4 + 2 * (k_)] = __builtin_amdgcn_s_memtime();                             \
        g_a4r_stamps[(blockIdx.x * 4 + tile_no_) * 4 + 2 * (k_) + 1] = __builtin_amdgcn_s_memrealtime();                     \
    }
// timeline (tools/gemm_timeline.py): s_memrealtime at kernel entry [0], per tile t < 3 at K-loop start / K-loop end / last store issued
// [1 + 3t ..], and after the last tile's stores have drained [10]
__device__ unsigned long long g_a4r_timeline[256 * 12];
extern "C" int a4r_debug_timeline(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_a4r_timeline), sizeof(g_a4r_timeline)) == hipSuccess ? 0 : -2;
}
#define A4R_TL(slot_)                                                                                                        \
    if (tid == 0 && (slot_) < 12 && blockIdx.x < 256) g_a4r_timeline[blockIdx.x * 12 + (slot_)] = __builtin_amdgcn_s_memrealtime();
#define A4R_TLT(k_) if (tile_no_ < 3) { A4R_TL((k_) + 3 * tile_no_) }
#else
#define A4R_LOOP_STAMP(k_)
#define A4R_TL(slot_)
#define A4R_TLT(k_)
#endif
#ifdef A4R_PHASE_STAMP
// diagnostic build only (-DA4R_PHASE_STAMP, tools/gemm_phase_stamps.py): s_memtime at seven points of every phase of K-tiles 4 and 5 of the first
// full tile, waves 0 and 4 (one per ping-pong half) of workgroups 0..63: [wg][half][phase 0..7][point 0..6 (+1 pad)].  Points: 0 phase start,
// 1 reads + DMA issued, 2 counted vmcnt passed, 3 barrier passed, 4 fragments arrived (lgkmcnt 0), 5 last MFMA issued, 6 trailing barrier passed.
// The stamps are SMEM returns collected by ONE extra lgkmcnt(0) at the end of the phase; nothing in the kernel reads them.
__device__ unsigned long long g_a4r_phase_stamps[64 * 2 * 8 * 8];
extern "C" int a4r_debug_phase_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_a4r_phase_stamps), sizeof(g_a4r_phase_stamps)) == hipSuccess ? 0 : -2;
}
#define A4R_ST_BEGIN(u_) const bool st_on_ = !TAIL && st_tile_ == 0 && (u_) >= 4 && (u_) < 6 && (wave & 3) == 0 && blockIdx.x < 64; \
    int st_idx_ = ((u_) - 4) * 4; unsigned long long st_t0 = 0, st_t1 = 0, st_t2 = 0, st_t3 = 0, st_t4 = 0, st_t5 = 0, st_t6 = 0;
#define A4R_ST(k_) if (st_on_) asm volatile("s_memtime %0" : "=s"(st_t##k_));
#define A4R_ST_NEXT                                                                                                              \
    if (st_on_) {                                                                                                                \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
        unsigned long long* d_ = g_a4r_phase_stamps + (size_t)(((blockIdx.x * 2 + (wave >> 2)) * 8 + st_idx_) * 8);              \
        d_[0] = st_t0; d_[1] = st_t1; d_[2] = st_t2; d_[3] = st_t3; d_[4] = st_t4; d_[5] = st_t5; d_[6] = st_t6;                  \
    }                                                                                                                            \
    ++st_idx_;
#define A4R_ST_TILE_DONE ++st_tile_;
#else
#define A4R_ST_BEGIN(u_)
#define A4R_ST(k_)
#define A4R_ST_NEXT
#define A4R_ST_TILE_DONE
#endif

namespace {

constexpr int UNIT_BYTES = 16384;
enum { U_ALO = 0, U_BLO = 1, U_BHI = 2, U_AHI = 3 };

// one 1-KiB LDS-DMA: LDS[lds_dst + lane*16 .. +16) <- global[base + voff .. +16)
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

// a unit's two pieces of one wave behind ONE m0 write: the instruction offset moves the LDS destination AND the source (tools/_probe/
// lds_dma_offset_probe.hip), so the second piece carries offset:1024 and its source offset is passed 1024 bytes low.  6 instructions for 2 KiB
// instead of 10 (A4R_GLDS_PAIR=0: two glds16, A/B builds).
#ifndef A4R_GLDS_PAIR
#define A4R_GLDS_PAIR 1
#endif
template <bool PAIR>
A4R_DEV void glds16x2(const void* base, uint32_t voff0, uint32_t voff1, uint32_t lds_dst) {
  if constexpr (PAIR && A4R_GLDS_PAIR) {      // (PAIR false: short tiles -- their clamped rows sit at offsets < 1024, which the second piece's 1024-low source offset cannot express)
    uint32_t keep;
    const uint32_t v1 = voff1 - 1024u;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %3\n\t"
        "global_load_lds_dwordx4 %2, %3 offset:1024\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff0), "v"(v1), "s"(base), "s"(lds_dst)
        : "memory");
  } else {
    glds16(base, voff0, lds_dst);
    glds16(base, voff1, lds_dst + 1024u);
  }
}

typedef int i32x8_t __attribute__((ext_vector_type(8)));
A4R_DEV f32x4_t mma_mx8(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1, f32x4_t c) {
    const i32x8_t a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    const i32x8_t b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    // cbsz = blgp = 0: both operands OCP e4m3; scales: E8M0 127 = 2^0 in every byte
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

// DMA source offsets (bytes from a short tile's first row) of unit row ul of the A_lo / A_hi units: half (ul >> 6) owns rows half * 16 kp + local,
// local = ul & 63 (+ 64 in A_hi); unit rows whose local row is past 16 kp are fetched from the tile's first row (valid memory, never used).
A4R_DEV void tail_a_offsets(int ul, int c, int kp, int lda_bytes, uint32_t& lo, uint32_t& hi) {
    const int half = ul >> 6, loc = ul & 63, act = 16 * kp;
    lo = (uint32_t)((loc < act ? half * act + loc : 0) * lda_bytes + c * 16);
    hi = (uint32_t)((loc + 64 < act ? half * act + loc + 64 : 0) * lda_bytes + c * 16);
}

// The tiles of one workgroup.  TAIL = false: the full 256 x 256 tiles of the persistent tile map (ntm row panels).  TAIL = true: ONE short tile of
// 32 * t_kp rows (t_kp = 1..7 MFMA row tiles per wave instead of 8) at row t_row0, column panel t_tn -- the rows past the last full ROUND of
// 256-row tiles are cut into short tiles, one per CU, so that every CU finishes together instead of 0 < f < 1 of the CUs running a whole
// extra tile (launch256 below).  A short tile uses the same LDS images, unit stream and phases; wave half wm owns rows
// t_row0 + wm * 16 t_kp + [0, 16 t_kp): unit rows past 16 t_kp of a half are DMA'd from the tile's first row (never used), the MFMA
// segments run the first t_kp row tiles only and the epilogue leaves after row t_kp - 1.  The non-TAIL instantiation puts the first six
// units of the workgroup's short tile in flight at the start of its last epilogue (t_ready).
template <typename TI, typename TO, int ACT, int DACT, int EF, bool TAIL>
A4R_DEV void gemm256_tiles(const a4r_gemm_t& p, char* lds, int ntm, int ntn, int gn_flags, uint32_t thr16, float keep_scale,
                           const int t_row0, const int t_tn, const int t_kp, bool& t_ready) {
    const int gn = gn_flags & 0xffff;                     // band width of the tile map; bit 16: A4R_GEMM_NO_STREAM=1 (A/B switch)
    constexpr int ROWB = 128;
    constexpr int KT = ROWB / (int)sizeof(TI);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // persistent workgroups: virtual block id vb = blockIdx.x + i * gridDim.x; gridDim.x is a multiple of 8, so vb & 7 -- the XCD
    // group under round-robin placement (speed only) -- is the same for every tile a workgroup visits, and t = vb >> 3 counts the
    // tiles of that XCD.  Two bijective tile maps:
    //   gn == 0: XCD x owns a contiguous chunk of the panel-major order (tile = tm * ntn + tn): its 32 workgroups walk all
    //            N-tiles of ~32 / ntn row panels together;
    //   gn  > 0: XCD x owns whole row panels (ntm / 8, the first ntm % 8 XCDs one more) and walks them in BANDS of gn N-tiles:
    //            band, then panel, then N-tile inside the band.  The B tiles of a band (gn x 256 x K) stay in the XCD's 4 MiB
    //            L2 for all its panels and every A panel is fetched by ONE XCD (once per band): for N = 3072, K = 768 the
    //            un-banded map re-fetched the 4.7 MB of weights every round (PMC: 449 MB read per launch against 67).
    const int nt = ntm * ntn;
    const int q8 = nt >> 3, r8 = nt & 7;
    const int xcd = blockIdx.x & 7;
    const int pq = ntm >> 3, pr = ntm & 7;
    const int np_x = pq + (xcd < pr ? 1 : 0), p0_x = xcd * pq + (xcd < pr ? xcd : pr);      // banded map: this XCD's panels
    const int len_x = gn > 0 ? np_x * ntn : (q8 + (xcd < r8 ? 1 : 0));                       // tiles of this XCD
    const int base_x = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    auto tile_of = [&](int t, int& tm_, int& tn_) {
        if (gn > 0) {
            const int nfull = ntn / gn, per_band = np_x * gn;               // full bands first, then the narrower last band
            const int b = t / per_band < nfull ? t / per_band : nfull;
            const int w = b < nfull ? gn : ntn - nfull * gn;
            const int r = t - b * per_band;
            tm_ = p0_x + r / w;
            tn_ = b * gn + r % w;
        } else {
            const int Lt_ = base_x + t;
            tm_ = Lt_ / ntn;
            tn_ = Lt_ % ntn;
        }
    };
    int t_loc = blockIdx.x >> 3;
    A4R_TL(0)
    if (!TAIL && t_loc >= len_x) return;
    if constexpr (!TAIL) {   // Staggered start (gn_flags >> 17 = delay in 10-ns ticks, A4R_GEMM_STAGGER = percent of a tile period, default 50, 0 = off):
        // workgroups that own one tile fewer than the busiest of their XCD have a tile period of slack; started late, their epilogue
        // store bursts fall into the other workgroups' K loops instead of on top of their bursts (tools/gemm_timeline.py: the K loop of
        // the N = 2304 launch 18.6 -> 16.3 us per tile).  Same-box step: -1.5 % on the slower boxes of the pool, neutral on the fastest.
        const int delay = (gn_flags >> 17) & 0x3fff, stride0 = (int)(gridDim.x >> 3);
        // (round 3 measured the delay on EVERY second workgroup, slack or not: 244 vs 244 us on the GELU launch, 245 - 255 vs 232 - 253 on the
        // '* derivative' dgrad -- the store bursts of workgroups in phase are not what the epilogue costs)
        if (delay > 0 && (len_x - 1 - t_loc) / stride0 < (len_x - 1) / stride0) {
            const uint64_t t_end = __builtin_amdgcn_s_memrealtime() + (uint64_t)delay;
            while (__builtin_amdgcn_s_memrealtime() < t_end) __builtin_amdgcn_s_sleep(8);
        }
    }
    int tm = 0, tn = t_tn;
    if constexpr (!TAIL) tile_of(t_loc, tm, tn);
    // short tile: rows per wave half, active MFMA row tiles in the A_lo / A_hi phases
    const int hrow = TAIL ? 16 * t_kp : 128;
    const int kp_lo = t_kp < 4 ? t_kp : 4, kp_hi = t_kp - 4;

    const int lda = p.lda, ldb = p.ldb;
    const int nk = p.K / KT;
    __builtin_assume(nk >= 1);                            // (host-checked; without it the accumulators count as live across the K loop and the epilogue copies every one before its in-place lane swap)
    const TI* Ap = reinterpret_cast<const TI*>(p.A);
    const TI* Bp = reinterpret_cast<const TI*>(p.B);
    const char* Abase = reinterpret_cast<const char*>(Ap + (size_t)(TAIL ? t_row0 : tm * 256) * lda);
    const char* Bbase = reinterpret_cast<const char*>(Bp + (size_t)tn * 256 * ldb);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // per-lane source offsets of this wave's two DMA instructions of each unit kind (bytes from the tile base)
    uint32_t offA_lo[2], offA_hi[2], offB_lo[2], offB_hi[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ul = 8 * (2 * wave + i) + (lane >> 3);                 // unit row 0..127
        const int c = (lane & 7) ^ ((ul >> 1) & 7);                      // source chunk for linear LDS slot (lane & 7)
        const int ra = ul + (ul >> 6) * 64;                              // A_lo tile row; A_hi = + 64
        const int rb = ul + (ul >> 5) * 32;                              // B_lo tile row; B_hi = + 32
        offA_lo[i] = (uint32_t)(ra * lda * (int)sizeof(TI) + c * 16);
        offA_hi[i] = (uint32_t)((ra + 64) * lda * (int)sizeof(TI) + c * 16);
        if constexpr (TAIL) tail_a_offsets(ul, c, t_kp, lda * (int)sizeof(TI), offA_lo[i], offA_hi[i]);
        offB_lo[i] = (uint32_t)(rb * ldb * (int)sizeof(TI) + c * 16);
        offB_hi[i] = (uint32_t)((rb + 32) * ldb * (int)sizeof(TI) + c * 16);
    }
    const uint32_t dma_dst = lds0 + (uint32_t)(2 * wave) * 1024u;       // + buffer*4*UNIT + kind*UNIT + i*1024

    // K-tiles past the end of this output tile's K range are the first K-tiles of the workgroup's NEXT output tile (has_next: nk is
    // even, so ring-buffer parity carries over): the unit stream never stops between tiles, the next K loop starts on data that is
    // already in LDS and the epilogue's own loads (bias, Pre, R1) do not queue behind a 96 KiB prologue burst.
#define A4R_PAIRED (!TAIL)
#define A4R_ISSUE(kind_, tile_, base_, off_)                                                                         \
    if ((tile_) < nk || has_next) {                                                                                  \
        const char* src_ = (tile_) < nk ? (base_) + (size_t)(tile_) * ROWB : (base_##_nx) + (size_t)((tile_) - nk) * ROWB; \
        const uint32_t dst_ = dma_dst + (uint32_t)((((tile_) & 1) * 4 + (kind_)) * UNIT_BYTES);                       \
        glds16x2<A4R_PAIRED>(src_, off_[0], off_[1], dst_);                                                          \
    }
    // LOAD-segment pieces and the MFMA segment of a phase (see the header).  sched_barrier(0) pins the order hipcc emits.
#define A4R_RD_A(dst_, buf_, unit_)                                                                   \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
            dst_[mi][ks] = *reinterpret_cast<const uint4*>((lds + a_base[buf_][ks]) + ((unit_) * UNIT_BYTES + mi * 2048));
#define A4R_RD_B(dst_, buf_, unit_)                                                                   \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                              \
            dst_[ni][ks] = *reinterpret_cast<const uint4*>((lds + b_base[buf_][ks]) + ((unit_) * UNIT_BYTES + ni * 2048));
    // e4m3 operands: ONE block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 per output tile and K-tile (the 2 x 16 bytes a lane holds of a
    // 128-byte K-tile row are its 32-element K block; both operands use the same byte -> contraction-slot map, so the products pair the
    // right elements).  The E8M0 block scales are all 2^0: the per-token / per-channel fp32 scales stay in the epilogue, the instruction
    // is used for its issue rate -- 32 cycles for K = 128 against 4 x 16 for the non-scaled K = 32 form, i.e. twice the bf16 rate
    // (A4R_FP8_MX=0 at compile time: the non-scaled form, A/B builds).
#ifndef A4R_FP8_MX
#define A4R_FP8_MX 1
#endif
#define A4R_MFMA16(ax_, bx_, m0_, n0_, lim_)                                                          \
    if constexpr (TAIL) {              /* short tile: the first lim_ row tiles of this phase's half */  \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) {                                            \
            if (mi >= (lim_)) break;                                                                  \
            if constexpr (sizeof(TI) == 1 && A4R_FP8_MX) {                                            \
                _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                      \
                    acc[(m0_) + mi][(n0_) + ni] = mma_mx8(bx_[ni][0], bx_[ni][1], ax_[mi][0], ax_[mi][1], acc[(m0_) + mi][(n0_) + ni]); \
            } else {                                                                                  \
                _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                      \
                    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                  \
                        Mma<TI>::mma(bx_[ni][ks], ax_[mi][ks], acc[(m0_) + mi][(n0_) + ni]);          \
            }                                                                                         \
        }                                                                                             \
    } else if constexpr (sizeof(TI) == 1 && A4R_FP8_MX) {                                             \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                              \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                          \
                acc[(m0_) + mi][(n0_) + ni] = mma_mx8(bx_[ni][0], bx_[ni][1], ax_[mi][0], ax_[mi][1], acc[(m0_) + mi][(n0_) + ni]); \
    } else {                                                                                          \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                  \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                              \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                          \
                Mma<TI>::mma(bx_[ni][ks], ax_[mi][ks], acc[(m0_) + mi][(n0_) + ni]);                  \
    }
    // full_: the units this phase's counted wait leaves in flight were all issued -> vmcnt(8) (four units may stay in flight);
    // else the shorter count tail_ of the end of the K range.  ONE copy of every phase (duplicated phase bodies spill).
#ifndef A4R_LGKM_ALL
#define A4R_LGKM_ALL 1     /* 0 (A/B builds): no lgkmcnt(0) in front of the MFMA segment -- hipcc's own counted waits let the first MFMAs start on the first fragments */
#endif
#ifndef A4R_Z0_NOWAIT
#define A4R_Z0_NOWAIT 1    /* 0: the counted vmcnt(8) also in the first K-tile of an output tile (A/B builds) */
#endif
#ifndef A4R_ABL
#define A4R_ABL 0          /* timing-only diagnostic builds (tools/gemm_abl.sh): 1 no DMA, 2 no MFMA, 4 no fragment reads, 8 no barriers, 16 no setprio */
#endif
#define A4R_PHASE(reads_, issue_, full_, cnt_, tail_, ax_, bx_, m0_, n0_, z_, lim_)                    \
    A4R_ST(0)                                                                                         \
    if (z_) {                      /* first K-tile of an output tile: this phase's quarter of the accumulators starts from zero */ \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                              \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) acc[(m0_) + mi][(n0_) + ni] = f32x4_t{0.f, 0.f, 0.f, 0.f}; \
    }                                                                                                 \
    if (!(A4R_ABL & 4)) { reads_ }                                                                    \
    if (!(A4R_ABL & 1)) { issue_ }                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    A4R_ST(1)                                                                                         \
    if (z_ && A4R_Z0_NOWAIT) { /* first K-tile of an output tile: every unit read before K-tile 1's phase 1 was issued BEFORE the previous */ \
    } /* tile's stores and has landed (counted wait + barrier at the top of the tile): no wait, the stores keep draining for these 4 phases */ \
    else if (full_) asm volatile("s_waitcnt vmcnt(" cnt_ ")" ::: "memory");                           \
    else asm volatile("s_waitcnt vmcnt(" tail_ ")" ::: "memory");                                     \
    A4R_ST(2)                                                                                         \
    if (!(A4R_ABL & 8)) __builtin_amdgcn_s_barrier();                                                 \
    asm volatile("" ::: "memory");                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    A4R_ST(3)                                                                                         \
    if (A4R_LGKM_ALL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                              \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    A4R_ST(4)                                                                                         \
    if (!(A4R_ABL & 16)) __builtin_amdgcn_s_setprio(1);                                               \
    if (!(A4R_ABL & 2)) { A4R_MFMA16(ax_, bx_, m0_, n0_, lim_) }                                      \
    if (!(A4R_ABL & 16)) __builtin_amdgcn_s_setprio(0);                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    A4R_ST(5)                                                                                         \
    if (!(A4R_ABL & 8)) __builtin_amdgcn_s_barrier();                                                 \
    asm volatile("" ::: "memory");                                                                    \
    A4R_ST(6)                                                                                         \
    A4R_ST_NEXT
    // K-tile u from ring buffer buf_ (compile-time).  n1 = a K-tile u+1 exists, n2 = u+2 exists (A4R_ISSUE skips what does not).
#ifndef A4R_DMA_SCHED
#define A4R_DMA_SCHED 1
#endif
#if A4R_DMA_SCHED == 0
#define A4R_KTILE4(u_, buf_)                                                                                                         \
    {                                                                                                                               \
        const bool n1 = (u_) + 1 < nk || has_next, n2 = (u_) + 2 < nk || has_next;                                                                        \
        const bool z0 = (buf_) == 0 && (u_) == 0;                                                                                   \
        A4R_ST_BEGIN(u_)                                                                                                            \
        asm volatile("" : "+v"(a_base[buf_][0]), "+v"(a_base[buf_][1]), "+v"(b_base[buf_][0]), "+v"(b_base[buf_][1]));              \
        A4R_PHASE(A4R_RD_B(b0, buf_, U_BLO) A4R_RD_A(af, buf_, U_ALO), A4R_ISSUE(U_BHI, (u_) + 1, Bbase, offB_hi), n1, "8", "2", af, b0, 0, 0, z0, kp_lo) \
        A4R_PHASE(A4R_RD_B(b1, buf_, U_BHI), A4R_ISSUE(U_AHI, (u_) + 1, Abase, offA_hi), n1, "8", "0", af, b1, 0, 2, z0, kp_lo)     \
        A4R_PHASE(A4R_RD_A(af, buf_, U_AHI), A4R_ISSUE(U_ALO, (u_) + 2, Abase, offA_lo), n2, "8", "4", af, b1, 4, 2, z0, kp_hi)     \
        A4R_PHASE(, A4R_ISSUE(U_BLO, (u_) + 2, Bbase, offB_lo), n2, "8", "4", af, b0, 4, 0, z0, kp_hi)                              \
    }
#else
    // A4R_DMA_SCHED 1 (A/B builds): the DMA issues sit in the two phases with the fewest fragment reads (4 and 0 instead of 12 / 4 / 8 / 0):
    // phase 1 issues B_hi, A_hi of K-tile u + 1, phase 3 A_lo, B_lo of u + 2 -- the same unit stream.  Counted waits (queue after the phase's
    // issues, oldest first): phase 0 [B_hi A_hi (u+1) | A_lo B_lo (u+2)] must retire B_hi -> vmcnt(6); phase 1 [A_hi(u+1) A_lo B_lo (u+2) B_hi A_hi (u+2)]
    // must retire A_hi -> vmcnt(8); phase 2 reads nothing new in phase 3 -> no wait; phase 3 [B_hi A_hi (u+1) A_lo B_lo (u+2)] must retire
    // A_lo, B_lo (u+1) -> vmcnt(8).  WAR: every slot is re-filled >= 2 phases after its last read (B_hi: phase 1 of u-1, A_hi: 2 of u-1, A_lo / B_lo: 0 of u).
#define A4R_KTILE4(u_, buf_)                                                                                                         \
    {                                                                                                                               \
        const bool n1 = (u_) + 1 < nk || has_next, n2 = (u_) + 2 < nk || has_next;                                                                        \
        const bool z0 = (buf_) == 0 && (u_) == 0;                                                                                   \
        A4R_ST_BEGIN(u_)                                                                                                            \
        asm volatile("" : "+v"(a_base[buf_][0]), "+v"(a_base[buf_][1]), "+v"(b_base[buf_][0]), "+v"(b_base[buf_][1]));              \
        A4R_PHASE(A4R_RD_B(b0, buf_, U_BLO) A4R_RD_A(af, buf_, U_ALO), , n1, "6", "2", af, b0, 0, 0, z0, kp_lo)                     \
        A4R_PHASE(A4R_RD_B(b1, buf_, U_BHI), A4R_ISSUE(U_BHI, (u_) + 1, Bbase, offB_hi) A4R_ISSUE(U_AHI, (u_) + 1, Abase, offA_hi), n1, "8", "0", af, b1, 0, 2, z0, kp_lo) \
        A4R_PHASE(A4R_RD_A(af, buf_, U_AHI), , true, "63", "63", af, b1, 4, 2, z0, kp_hi)                                           \
        A4R_PHASE(, A4R_ISSUE(U_ALO, (u_) + 2, Abase, offA_lo) A4R_ISSUE(U_BLO, (u_) + 2, Bbase, offB_lo), n2, "8", "0", af, b0, 4, 0, z0, kp_hi) \
    }
#endif
#ifndef A4R_PHASES
#define A4R_PHASES 2       /* phases per K-tile for bf16 / fp32 operands: 2 (round 4) or 4 (rounds 2 - 3; A/B builds) */
#endif
#ifndef A4R_PRIO
#define A4R_PRIO 1         /* two-phase schedule: s_setprio 1 around the MFMA segment (1), around the LOAD segment (2), nowhere (0) -- A/B builds */
#endif
#ifndef A4R_PHASES_FP8
#define A4R_PHASES_FP8 2   /* likewise for e4m3 operands (same-box: BERT-base fp8 2 006 -> 2 058 user-seq/s, MAE + Compacter fp8 926 -> 952; profiles/r04_g_fp8_ab.txt) */
#endif
    // TWO phases per K-tile (round 4): the quadrant pairs (A_lo,B_lo)+(A_lo,B_hi) and (A_hi,B_hi)+(A_hi,B_lo) run as ONE
    // MFMA segment of 32 each -- half the barriers per K-tile and 512-cycle matrix segments for the partner wave's LOAD segment (16 / 8 fragment
    // reads + 4 DMA pieces + the counted wait) to hide behind, instead of 256.  Same-box A/B against the four-phase schedule
    // (profiles/r04_f_ab_forms_p2.txt): qkv -6 %, attention-output -3 %, the K = 2304 / 3072 dgrads -2 %, the epilogue-heavy FFN launches
    // -1 %; step 17.51 -> 17.39 ms.  Same unit stream and prologue as A4R_DMA_SCHED 1:
    //   LOAD_A(u): read A_lo, B_lo, B_hi (u) | issue B_hi, A_hi (u+1) | wait: A_hi(u) landed     -> vmcnt(8)  [A_lo B_lo (u+1) + the 4 just issued stay]
    //   LOAD_B(u): read A_hi (u)             | issue A_lo, B_lo (u+2) | wait: B_hi(u+1) landed   -> vmcnt(6)  [A_hi(u+1) + the 4 just issued stay]
    // RAW: a wait sits in front of a barrier that every reader of the unit passes before its LOAD segment.  WAR: B_hi / A_lo / B_lo slots are
    // re-filled >= 2 intervals after their last read; A_hi(u+1)'s slot was last read in LOAD_B(u-1), by the other half ONE interval earlier --
    // so the fragment reads are retired (lgkmcnt(0)) BEFORE the barrier that ends a LOAD segment, not behind it.
#define A4R_PHASE2(reads_, issue_, waitstmt_, ax_, bxa_, na_, bxb_, nb_, m0_, z_, lim_)                 \
    if (z_) {                                                                                           \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                              \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) acc[(m0_) + mi][ni] = f32x4_t{0.f, 0.f, 0.f, 0.f}; \
    }                                                                                                 \
    if (A4R_PRIO == 2) __builtin_amdgcn_s_setprio(1);                                                 \
    if (!(A4R_ABL & 4)) { reads_ }                                                                    \
    if (!(A4R_ABL & 1)) { issue_ }                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    waitstmt_                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                \
    if (A4R_PRIO == 2) __builtin_amdgcn_s_setprio(0);                                                 \
    if (!(A4R_ABL & 8)) __builtin_amdgcn_s_barrier();                                                 \
    asm volatile("" ::: "memory");                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (A4R_PRIO == 1 && !(A4R_ABL & 16)) __builtin_amdgcn_s_setprio(1);                              \
    if (!(A4R_ABL & 2)) { A4R_MFMA16(ax_, bxa_, m0_, na_, lim_) A4R_MFMA16(ax_, bxb_, m0_, nb_, lim_) } \
    if (A4R_PRIO == 1 && !(A4R_ABL & 16)) __builtin_amdgcn_s_setprio(0);                              \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (!(A4R_ABL & 8)) __builtin_amdgcn_s_barrier();                                                 \
    asm volatile("" ::: "memory");
    // A4R_DMA_INTERLEAVE (round 5, from the four-wave kernel's measurements -- LOG part E: a 1-KiB piece costs its wave ~60 cycles of issue when the
    // four waves of a ping-pong half ask the CU's one address unit at the same instant, ~16 when they do not): inside a LOAD segment the pieces are
    // issued BETWEEN groups of fragment reads instead of in one burst behind them, and wave w of the half starts (w & 3) x 16 cycles late, so that the
    // four waves' pieces reach the address unit one after the other.  Same pieces, same counted waits (they sit behind the segment's last issue).
    // MEASURED SLOWER here (profiles/r05_i_w8_dma_interleave_ab.txt, same box: step 17.22 -> 17.99 ms, every shape 5 - 15 % slower): with two waves per
    // SIMD the burst's issue stall is the PARTNER's MFMA time anyway, while reads queued behind a piece reach the LDS later and lengthen the segment.
    // 0 = the burst behind the reads (rounds 3 - 4), kept as the default; 1 for A/B builds.
#ifndef A4R_DMA_INTERLEAVE
#define A4R_DMA_INTERLEAVE 0
#endif
#define A4R_ISSUE1(kind_, tile_, base_, off_, i_)                                                                    \
    if ((tile_) < nk || has_next) {                                                                                  \
        const char* src_ = (tile_) < nk ? (base_) + (size_t)(tile_) * ROWB : (base_##_nx) + (size_t)((tile_) - nk) * ROWB; \
        glds16(src_, off_[i_], dma_dst + (uint32_t)((((tile_) & 1) * 4 + (kind_)) * UNIT_BYTES) + (i_) * 1024u);      \
    }                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);
#define A4R_RD_B1(dst_, buf_, unit_, ni_)                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                  \
        dst_[ni_][ks] = *reinterpret_cast<const uint4*>((lds + b_base[buf_][ks]) + ((unit_) * UNIT_BYTES + (ni_) * 2048)); \
    __builtin_amdgcn_sched_barrier(0);
#define A4R_RD_A1(dst_, buf_, unit_, mi_)                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                  \
        dst_[mi_][ks] = *reinterpret_cast<const uint4*>((lds + a_base[buf_][ks]) + ((unit_) * UNIT_BYTES + (mi_) * 2048)); \
    __builtin_amdgcn_sched_barrier(0);
#define A4R_WAVE_DELAY() for (int d_ = 0; d_ < (wave & 3); ++d_) asm volatile("s_nop 3");
#if A4R_DMA_INTERLEAVE
#define A4R_LOAD_A(u_, buf_)                                                                                                        \
        A4R_WAVE_DELAY()                                                                                                            \
        A4R_ISSUE1(U_BHI, (u_) + 1, Bbase, offB_hi, 0) A4R_RD_B1(b0, buf_, U_BLO, 0) A4R_RD_B1(b0, buf_, U_BLO, 1)                  \
        A4R_ISSUE1(U_BHI, (u_) + 1, Bbase, offB_hi, 1) A4R_RD_B1(b1, buf_, U_BHI, 0) A4R_RD_B1(b1, buf_, U_BHI, 1)                  \
        A4R_ISSUE1(U_AHI, (u_) + 1, Abase, offA_hi, 0) A4R_RD_A1(af, buf_, U_ALO, 0) A4R_RD_A1(af, buf_, U_ALO, 1)                  \
        A4R_ISSUE1(U_AHI, (u_) + 1, Abase, offA_hi, 1) A4R_RD_A1(af, buf_, U_ALO, 2) A4R_RD_A1(af, buf_, U_ALO, 3)
#define A4R_LOAD_B(u_, buf_)                                                                                                        \
        A4R_WAVE_DELAY()                                                                                                            \
        A4R_ISSUE1(U_ALO, (u_) + 2, Abase, offA_lo, 0) A4R_RD_A1(af, buf_, U_AHI, 0)                                                \
        A4R_ISSUE1(U_ALO, (u_) + 2, Abase, offA_lo, 1) A4R_RD_A1(af, buf_, U_AHI, 1)                                                \
        A4R_ISSUE1(U_BLO, (u_) + 2, Bbase, offB_lo, 0) A4R_RD_A1(af, buf_, U_AHI, 2)                                                \
        A4R_ISSUE1(U_BLO, (u_) + 2, Bbase, offB_lo, 1) A4R_RD_A1(af, buf_, U_AHI, 3)
#else
#define A4R_LOAD_A(u_, buf_) A4R_RD_B(b0, buf_, U_BLO) A4R_RD_B(b1, buf_, U_BHI) A4R_RD_A(af, buf_, U_ALO) A4R_ISSUE(U_BHI, (u_) + 1, Bbase, offB_hi) A4R_ISSUE(U_AHI, (u_) + 1, Abase, offA_hi)
#define A4R_LOAD_B(u_, buf_) A4R_RD_A(af, buf_, U_AHI) A4R_ISSUE(U_ALO, (u_) + 2, Abase, offA_lo) A4R_ISSUE(U_BLO, (u_) + 2, Bbase, offB_lo)
#endif
#define A4R_KTILE2(u_, buf_)                                                                                                        \
    {                                                                                                                               \
        const bool n1 = (u_) + 1 < nk || has_next, n2 = (u_) + 2 < nk || has_next;                                                  \
        const bool z0 = (buf_) == 0 && (u_) == 0;                                                                                   \
        asm volatile("" : "+v"(a_base[buf_][0]), "+v"(a_base[buf_][1]), "+v"(b_base[buf_][0]), "+v"(b_base[buf_][1]));              \
        A4R_PHASE2(A4R_LOAD_A(u_, buf_), ,                                                                                          \
                   if (z0) { } else if (n1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");, \
                   af, b0, 0, b1, 2, 0, z0, kp_lo)                                                                                  \
        A4R_PHASE2(A4R_LOAD_B(u_, buf_), ,                                                                                          \
                   if (n2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else if (n1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");, \
                   af, b1, 2, b0, 0, 4, z0, kp_hi)                                                                                  \
    }
    constexpr bool PH2 = (sizeof(TI) == 1 ? A4R_PHASES_FP8 : A4R_PHASES) == 2;
#define A4R_KTILE(u_, buf_) if constexpr (PH2) { A4R_KTILE2(u_, buf_) } else { A4R_KTILE4(u_, buf_) }

    f32x4_t acc[8][4];
#ifdef A4R_PHASE_STAMP
    int st_tile_ = 0;
#endif

    // fragment addressing (unit-local): A rows wm*64 + mi*16 + (lane&15), B rows wn*32 + ni*16 + (lane&15).  The swizzle term (row >> 1) & 7 only
    // depends on lane & 15 (the wave's and the tile's row offsets are multiples of 16), so a fragment's address is ONE per-lane base per K-half
    // + a compile-time constant (ring buffer, unit, mi / ni * 2048): the constants ride in the ds_read offset field instead of a v_add per read
    // in every LOAD segment (12 address registers and ~10 vector instructions per phase, issued next to the partner wave's MFMAs).
    const int fr = lane & 15, kg = lane >> 4;
    // (one base per ring buffer: the offset field holds 16 bits.  The bases are laundered once per K-tile: as loop invariants, hipcc hoists
    // every base + constant into a register of its own)
    uint32_t a_base[2][2], b_base[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sw = (((ks * 4 + kg) ^ ((fr >> 1) & 7)) << 4);
        a_base[0][ks] = (uint32_t)((wm * 64 + fr) * ROWB + sw);
        b_base[0][ks] = (uint32_t)((wn * 32 + fr) * ROWB + sw);
        a_base[1][ks] = a_base[0][ks] + 4u * UNIT_BYTES;
        b_base[1][ks] = b_base[0][ks] + 4u * UNIT_BYTES;
    }

    // ---- prologue of a tile: the first 6 units in stream order (K-tile 0 and A_lo, B_lo of K-tile 1)
#define A4R_PROLOGUE_AT(Ab_, alo_, ahi_, Bb_)    \
    A4R_ISSUE(U_ALO, 0, Ab_, alo_)              \
    A4R_ISSUE(U_BLO, 0, Bb_, offB_lo)           \
    A4R_ISSUE(U_BHI, 0, Bb_, offB_hi)           \
    A4R_ISSUE(U_AHI, 0, Ab_, ahi_)              \
    A4R_ISSUE(U_ALO, 1, Ab_, alo_)              \
    A4R_ISSUE(U_BLO, 1, Bb_, offB_lo)
#define A4R_PROLOGUE() A4R_PROLOGUE_AT(Abase, offA_lo, offA_hi, Bbase)
    bool has_next = false;
    const int tile_stride = (int)(gridDim.x >> 3);        // read once (behind the asm memory clobbers it was re-loaded from the dispatch packet per tile)
    const char* Abase_nx = Abase;
    const char* Bbase_nx = Bbase;
    // store instructions a wave issues per output tile: 8 rows x 2 pairs of 16-byte (bf16) / 2 x 16-byte (fp32) stores, + the second output
    // (a LOWER bound: the generic EF < 0 instantiation may or may not carry a second output)
    constexpr int EPI_STORES = 16 * ((int)sizeof(TO) / 2) + ((EF >= 0 && (EF & 8)) ? 16 : 0);
    if (TAIL && t_ready) {     // the six units were issued at the start of the last full tile's epilogue, BEFORE its >= EPI_STORES stores
        if constexpr (EPI_STORES >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
        A4R_PROLOGUE()
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const GemmEpi<TO> epi = make_epi<TO>(p, thr16, keep_scale);
    const bool stream = !(nk & 1) && !((gn_flags >> 16) & 1);
    // the tile after the current one: found before the current tile's K loop needs it, outside the barrier-to-barrier path (the scalar
    // divisions of tile_of() sat between the top-of-tile barrier and the first phase)
    int tm_nx = tm, tn_nx = tn;
    bool more;
#define A4R_NEXT_TILE()                                                                      \
    t_loc += tile_stride;                                                                    \
    more = !TAIL && t_loc < len_x;                                                           \
    if (more) {                                                                              \
        tile_of(t_loc, tm_nx, tn_nx);                                                        \
        Abase_nx = reinterpret_cast<const char*>(Ap + (size_t)tm_nx * 256 * lda);            \
        Bbase_nx = reinterpret_cast<const char*>(Bp + (size_t)tn_nx * 256 * ldb);            \
    }
    A4R_NEXT_TILE()

#ifdef A4R_STAMP
  int tile_no_ = 0;
#endif
  for (;;) {                                              // ---- tiles of this workgroup
    __builtin_amdgcn_s_barrier();                         // every wave's prologue units have landed (vmcnt(0) above / below)
    asm volatile("" ::: "memory");
    // (the accumulators are zeroed quarter by quarter inside the LOAD segments of the first K-tile's four phases, next to the partner
    // wave's MFMA segment: 128 v_mov per wave in front of the loop were 0.5 us per tile during which neither wave of a SIMD issued MFMAs)
    const int tm_done = tm, tn_done = tn;
    // EF bit 32 (the `* 8-bit derivative` dgrad with a tile-native derivative tensor): the tile's whole Pre operand -- 16 x 8 bytes per lane --
    // is requested HERE, in front of the K loop, and waits in 32 registers: its HBM time (124 MB per launch at B = 32) falls into the K loop,
    // which leaves HBM idle, instead of into the epilogue, where the matrix pipe idles.  In order with the unit stream: the counted wait of
    // K-tile 1 retires these loads too (they are ~1.4 us old by then).
    constexpr bool PRE_TOP = !TAIL && EF >= 0 && (EF & 32) != 0 && DACT == A4R_DACT_MULQ8_ && sizeof(TO) == 2;
    uint2 pt_[16];
    if constexpr (PRE_TOP) {
        const uint8_t* const src_ = reinterpret_cast<const uint8_t*>(p.Pre) + ((size_t)(tm * ntn + tn) * 8 + wave) * 8192 + (size_t)lane * 8;
#pragma unroll
        for (int g_ = 0; g_ < 16; ++g_) pt_[g_] = *reinterpret_cast<const uint2*>(src_ + g_ * 512);
    }
    has_next = more && stream;                            // its first six units are issued by the last two K-tiles of this tile's loop

    uint4 af[4][2], b0[2][2], b1[2][2];
    A4R_LOOP_STAMP(0)
    A4R_TLT(1)
    if (wave >= 4 && !(A4R_ABL & 8)) __builtin_amdgcn_s_barrier();          // waves 4-7 run one barrier behind waves 0-3 from here on
    for (int u = 0; u < nk; u += 2) {
        A4R_KTILE(u, 0)
        if (u + 1 < nk) A4R_KTILE(u + 1, 1)
    }
    if (wave < 4 && !(A4R_ABL & 8)) __builtin_amdgcn_s_barrier();           // re-align: all LDS reads of this tile are complete, the ring is free
    A4R_ST_TILE_DONE
    asm volatile("" ::: "memory");
    A4R_LOOP_STAMP(1)
    A4R_TLT(2)

    // ---- epilogue straight from the accumulators.  The MFMA operands are swapped (B fragment first), so the tile is
    // produced transposed: a lane's 4 registers of tile (mi, ni) are 4 CONSECUTIVE COLUMNS of one output row,
    //     C[wave row mi*16 + (lane & 15)][wave col ni*16 + (lane >> 4)*4 + 0..3],
    // i.e. 8 B (bf16) / 16 B (fp32) per lane and 32 B / 64 B runs per row -- no LDS round trip, no barriers, and the 32
    // independent (mi, ni) groups give the memory system all the parallelism it needs (the LDS-staged form cost ~25 us per
    // tile with one workgroup per CU).
    // Every LDS read of this tile completed before the last barrier.  The NEXT tile's first six units are already on their way (issued
    // by this tile's last two K-tiles) -- or, with an odd K-tile count, are put in flight here -- and land while the accumulators are
    // being written out.
    has_next = false;
    Abase = Abase_nx;
    Bbase = Bbase_nx;
    tm = tm_nx;
    tn = tn_nx;
    if (more && !stream) {                                // odd K-tile count: the next tile's first units are put in flight here
        A4R_PROLOGUE()
    }
    if (!TAIL && !more && t_kp > 0) {                     // last full tile: the first units of this workgroup's short tile
        const char* const At = reinterpret_cast<const char*>(Ap + (size_t)t_row0 * lda);
        const char* const Bt = reinterpret_cast<const char*>(Bp + (size_t)t_tn * 256 * ldb);
        const char* const At_nx = At;
        const char* const Bt_nx = Bt;
        uint32_t ta_lo[2], ta_hi[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ul = 8 * (2 * wave + i) + (lane >> 3);
            tail_a_offsets(ul, (lane & 7) ^ ((ul >> 1) & 7), t_kp, lda * (int)sizeof(TI), ta_lo[i], ta_hi[i]);
        }
#undef A4R_PAIRED
#define A4R_PAIRED false          /* the short tile's A offsets (clamped rows) */
        A4R_PROLOGUE_AT(At, ta_lo, ta_hi, Bt)
#undef A4R_PAIRED
#define A4R_PAIRED (!TAIL)
        t_ready = true;
    }
#define A4R_ACC_LOAD4(dst_, mi_, ni_) _Pragma("unroll") for (int r4_ = 0; r4_ < 4; ++r4_) dst_[r4_] = acc[mi_][ni_][r4_];
#include "a4r_gemm256_epi.inc"
#undef A4R_ACC_LOAD4
    A4R_TLT(3)
#ifdef A4R_STAMP
    ++tile_no_;
    if (!more) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        A4R_TL(10)
    }
#endif
    if (!more) break;
    A4R_NEXT_TILE()
    // the next tile's 6 prologue units were issued BEFORE this tile's stores: all of them have landed once at most the 16 youngest
    // operations (>= 16 stores per wave follow the DMAs) are still outstanding.  The stores themselves drain behind the next K loop.
    // (vmcnt counts stores and is in order: a smaller count than the epilogue's store count would wait for stores here)
    if constexpr (EPI_STORES >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
#undef A4R_NEXT_TILE
#undef A4R_PROLOGUE
#undef A4R_ISSUE
#undef A4R_PAIRED
#undef A4R_RD_A
#undef A4R_RD_B
#undef A4R_MFMA16
#undef A4R_PHASE
#undef A4R_PHASE2
#undef A4R_KTILE
#undef A4R_KTILE2
#undef A4R_KTILE4
#undef A4R_LOAD_A
#undef A4R_LOAD_B
#undef A4R_ISSUE1
#undef A4R_RD_A1
#undef A4R_RD_B1
#undef A4R_WAVE_DELAY
}

// ntm = row panels of FULL tiles; the tail_rows rows behind them (0 = none) are cut into short tiles of 32 * tail_kp rows, tile j (row-panel
// major) on workgroup (j / wg_per_xcd, j % wg_per_xcd) = (XCD, index): the short tiles of one row panel share an XCD's L2.
template <typename TI, typename TO, int ACT, int DACT, int EF>
__global__ void __launch_bounds__(512, 2) gemm_nt_256_kernel(const a4r_gemm_t p, int ntm, int ntn, int gn_flags, uint32_t thr16, float keep_scale,
                                                             int tail_kp, int tail_rows) {
    __shared__ __attribute__((aligned(16))) char lds[8 * UNIT_BYTES];      // [buffer 2][unit 4][128 rows][128 B]
    int t_row0 = 0, t_tn = 0, t_kp = 0;
    if (tail_kp > 0) {
        const int h = 32 * tail_kp;
        const int jt = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
        if (jt < (tail_rows + h - 1) / h * ntn) {
            const int jp = jt / ntn, left = tail_rows - jp * h;
            t_tn = jt - jp * ntn;
            t_row0 = ntm * 256 + jp * h;
            t_kp = left < h ? left / 32 : tail_kp;           // (the last row panel may be shorter)
        }
    }
    bool t_ready = false;
    if (ntm > 0) gemm256_tiles<TI, TO, ACT, DACT, EF, false>(p, lds, ntm, ntn, gn_flags, thr16, keep_scale, t_row0, t_tn, t_kp, t_ready);
    if (t_kp > 0) gemm256_tiles<TI, TO, ACT, DACT, EF, true>(p, lds, ntm, ntn, gn_flags, thr16, keep_scale, t_row0, t_tn, t_kp, t_ready);
}

}  // namespace

int a4r_cu_count() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
        n_cu &= ~7;                                        // multiple of 8: a workgroup stays on one XCD group
        if (n_cu < 8) n_cu = 8;
    }
    return n_cu;
}

// Short-tile tail of a launch.  tiles = R full rounds of the persistent grid + a partial one: the first p_full = floor(R * grid / ntn) row
// panels keep 256-row tiles (<= R per workgroup), the rows behind them are cut into tiles of 32 * kp rows with the smallest kp that needs at
// most one tile per workgroup -- R + f(kp) tile periods instead of R + 1.  Measured (tools/gemm_tail_probe.py, one tile per CU, all CUs):
// f = 0.61 - 0.70 at kp = 1, 0.81 at kp = 4, 0.96 - 0.98 at kp = 7 -- a K-tile's DMA issue, fragment reads and barriers do not shrink with
// the MFMA count -- so the split is used for kp <= 3 only (A4R_GEMM_TAIL = the largest kp, default 3, 0 = never; at kp = 4 - 7 the BERT step
// lost 2 %: the gain is below what the banded tile map, which a tail excludes, is worth on the N = 3072 launches).  A function of (M, N)
// and the CU count only: the launch that writes a tile-native 8-bit derivative and the one that reads it agree on the split.
static int g_tail_max = -1;
extern "C" int a4r_gemm_tail_max(int k) {
    if (g_tail_max < 0) g_tail_max = getenv("A4R_GEMM_TAIL") ? atoi(getenv("A4R_GEMM_TAIL")) : 3;
    const int old = g_tail_max;
    if (k >= 0) g_tail_max = k < 7 ? k : 7;
    return old;
}
extern "C" int a4r_gemm_tail_plan(int M, int N, int* p_full, int* kp) {
    if (!p_full || !kp || M <= 0 || N < 256) { if (p_full) *p_full = M > 0 ? M / 256 : 0; if (kp) *kp = 0; return 0; }
    const int on = a4r_gemm_tail_max(-1);
    const int ntm = M / 256, ntn = N / 256, grid = a4r_cu_count(), nt = ntm * ntn;
    *p_full = ntm;
    *kp = 0;
    if (!on || ntn > grid || nt % grid == 0) return 0;
    const int pf = (nt / grid) * grid / ntn, rows = (ntm - pf) * 256;
    // (round 5) an UNDER-FILLED launch -- fewer tiles than CUs, pf == 0 -- takes any kp up to 7: its one round then costs f(kp) of a tile period on
    // (nearly) every CU instead of a whole period on a fraction of them (8 users, N = 768: 120 full tiles on 256 CUs -> 240 tiles of 128 rows)
    const int kmax = pf == 0 ? 7 : (on < 7 ? on : 7);
    for (int k = 1; k <= kmax; ++k)
        if ((rows + 32 * k - 1) / (32 * k) * ntn <= grid) {
            *p_full = pf;
            *kp = k;
            return 1;
        }
    return 0;
}

// tools/w4/a4r_gemm256w4.hip: the four-wave, hand-scheduled form of this kernel (round 5: bit-equal, at the matrix pipe's cycle floor, NOT faster end
// to end on a power-limited chip).  It is an experiment, not product: compiled in only by `make W4=1` (tools/w4/README.md), never into the shipped library.
#ifdef A4R_WITH_W4
int a4r_gemm_nt_256w4(hipStream_t s, const a4r_gemm_t& g, int to_f32, int act, int dact, int ef, int ntm, int ntn, int gn, int grid);
static int g_w4 = -1;
int a4r_gemm_w4(int v) {            // v = 0 / 1 sets, anything else queries (a4r_gemm_variant 8 / 9; initial value: A4R_GEMM_W4 or 0)
    if (g_w4 < 0) g_w4 = getenv("A4R_GEMM_W4") ? atoi(getenv("A4R_GEMM_W4")) != 0 : 0;
    const int old = g_w4;
    if (v == 0 || v == 1) g_w4 = v;
    return old;
}
#endif

namespace {

int g_band = -1;        // A4R_GEMM_BAND: -1 = automatic, 0 = panel-major map always, n > 0 = bands of n N-tiles wherever the banded map applies

// Band width of the banded tile map, or 0 for the panel-major map.  Banding needs whole panels per XCD without costing a round:
// the slowest XCD must not run more rounds than the balanced map would.
static int band_for(const a4r_gemm_t& g, int ntm, int ntn, int grid, int isz) {
    if (g_band < 0) {
        const char* e = getenv("A4R_GEMM_BAND");
        g_band = e ? atoi(e) + 1000 : 999;                 // 999 = automatic (resolved per shape below)
    }
    if (g_band == 1000 || grid % 8) return 0;
    const int wg_x = grid / 8;
    const int max_len = (ntm / 8 + (ntm % 8 ? 1 : 0)) * ntn;
    const int rounds_bal = (ntm * ntn + grid - 1) / grid, rounds_band = (max_len + wg_x - 1) / wg_x;
    if (ntm < 8 || rounds_band > rounds_bal) return 0;
    if (g_band > 1000) return g_band - 1000 < ntn ? g_band - 1000 : ntn;
    // automatic, long outputs (>= 128 row panels = 16 per XCD: the text tower at 32 users, the image tower): whole row panels per XCD walked in
    // bands of THREE N-tiles for every N (round 4, same-box sweeps of A4R_GEMM_BAND over all workloads, profiles/r04_m_band_sweep.txt: BERT-base
    // 17.21 -> 17.00 ms, RoBERTa 17.55 -> 17.27, BERT fp8 15.32 -> 15.17, ViT + LoRA +-0; at 66 row panels -- ViT-MAE -- the same policy costs
    // 1.4 %, so shorter outputs keep the rule below).  A 3-tile B band is 1.2 MB at K = 768: it stays in the XCD's L2 whatever the A stream does,
    // and the A panels it re-reads (4 x at N = 3072) come from the Infinity Cache.
    static const int long_band = getenv("A4R_GEMM_BAND_LONG") ? atoi(getenv("A4R_GEMM_BAND_LONG")) : 3;
    if (long_band > 0 && ntm >= 128) return long_band < ntn ? long_band : ntn;
    // shorter outputs: band only when the whole B operand does not sit in an XCD's L2 next to the streaming A panels
    // (A4R_GEMM_BAND_FIT / A4R_GEMM_BAND_BYTES: the two thresholds, bytes -- A/B sweeps)
    static const double fit = getenv("A4R_GEMM_BAND_FIT") ? atof(getenv("A4R_GEMM_BAND_FIT")) : 4.0e6;
    static const double budget = getenv("A4R_GEMM_BAND_BYTES") ? atof(getenv("A4R_GEMM_BAND_BYTES")) : 2.5e6;
    const double b_tile = 256.0 * g.K * isz;
    if (b_tile * ntn <= fit) return 0;                      // (N = 2304, K = 768: 3.5 MB still shares an L2 with the A stream: PMC 250 MB read un-banded vs 339 banded)
    int gn = (int)(budget / b_tile);
    if (gn < 2) return 0;                                   // a band of one tile re-reads A once per N-tile: panel-major (A read once) is the better map
    return gn < ntn ? gn : ntn;
}

template <typename TI, typename TO, int ACT, int DACT, int EF = -1>
int launch256(hipStream_t s, const a4r_gemm_t& g) {
    int ntm = g.M / 256;
    const int ntn = g.N / 256;
    const int n_cu = a4r_cu_count();
    int grid = ntm * ntn < n_cu ? ((ntm * ntn + 7) & ~7) : n_cu;       // a multiple of 8 (workgroups past an XCD's tile count exit at once)
    int p_full = ntm, tail_kp = 0;
    a4r_gemm_tail_plan(g.M, g.N, &p_full, &tail_kp);
    const int tail_rows = (ntm - p_full) * 256;
    if (tail_kp > 0) {
        const int n_tail = (tail_rows + 32 * tail_kp - 1) / (32 * tail_kp) * ntn;
        if (p_full == 0) grid = (n_tail + 7) & ~7;
        ntm = p_full;
    }
    static const int no_stream = getenv("A4R_GEMM_NO_STREAM") ? atoi(getenv("A4R_GEMM_NO_STREAM")) != 0 : 0;
    static const int stagger_pct = getenv("A4R_GEMM_STAGGER") ? atoi(getenv("A4R_GEMM_STAGGER")) : 50;     // (30 until the end of round 4; re-swept after the band policy: LOG.md part D)
    int delay = 0;
    if (stagger_pct > 0 && ntm * ntn > 2 * grid) {         // (tile period in 10-ns ticks ~ 145 per K-tile of 128 B + 400)
        delay = (int)((g.K * (int)sizeof(TI) / 128 * 145 + 400) * stagger_pct / 100);
        if (delay > 16383) delay = 16383;
    }
    // (with a short-tile tail the panel-major map -- whole panels per XCD would leave the XCDs uneven numbers of full tiles --, EXCEPT when the full
    // panels divide evenly over the 8 XCDs: the image tower's 259 panels = 256 full + a tail, 32 per XCD; round 4, A4R_GEMM_BAND_TAIL=0 = never)
    static const int band_tail = getenv("A4R_GEMM_BAND_TAIL") ? atoi(getenv("A4R_GEMM_BAND_TAIL")) != 0 : 1;
    const bool band_ok = tail_kp == 0 || (band_tail && sizeof(TI) == 2 && ntm > 0 && ntm % 8 == 0 && grid == n_cu);   // (ViT + LoRA same box: bf16 30.16 -> 30.05 ms; e4m3 +0.4 %: bf16 only)
    const int gn = (band_ok ? band_for(g, ntm, ntn, grid, (int)sizeof(TI)) : 0) | (no_stream << 16) | (delay << 17);
#ifdef A4R_WITH_W4
    if constexpr (sizeof(TI) == 2) {                       // bf16 operands: the four-wave kernel (a4r_gemm256w4.hip) where it applies and is switched on
        const int nk = g.K / 64;
        if (a4r_gemm_w4(-1) && tail_kp == 0 && ntm > 0 && nk >= 4 && !(nk & 1)) {
            const int r = a4r_gemm_nt_256w4(s, g, sizeof(TO) == 4, ACT, DACT, EF, ntm, ntn, gn, grid);
            if (r != 1) return r;                          // (1: this epilogue form is not instantiated there)
        }
    }
#endif
    hipLaunchKernelGGL((gemm_nt_256_kernel<TI, TO, ACT, DACT, EF>), dim3(grid), dim3(512), 0, s, g, ntm, ntn, gn,
                       a4r_thr16(g.drop_p), a4r_keep_scale(g.drop_p), tail_kp, tail_rows);
    return a4r_launch_status();
}

// which optional epilogue pieces a launch carries (the EF template argument of epilogue_n)
static int epi_mask(const a4r_gemm_t& g) {
    return (g.drop_p > 0.f ? 1 : 0) | (g.R1 ? 2 : 0) | (g.R2 ? 4 : 0) | (g.C2 ? 8 : 0);
}

template <typename T>
int dispatch_same(hipStream_t s, const a4r_gemm_t& g) {   // in == out dtype: the activation forms the training step uses
    // bf16 (the training step): the epilogue forms the step launches by the dozen get instantiations WITHOUT the run-time tests of the
    // pieces they do not carry (A4R_GEMM_GENERIC_EPI=1: the all-purpose instantiation always)
    static const int generic = getenv("A4R_GEMM_GENERIC_EPI") ? atoi(getenv("A4R_GEMM_GENERIC_EPI")) != 0 : 0;
    if constexpr (sizeof(T) == 2) {
        const int m = generic ? -1 : epi_mask(g);
        if (g.act == A4R_ACT_NONE && g.dact == A4R_ACT_NONE) {
            if (m == 0) return launch256<T, T, A4R_ACT_NONE, A4R_ACT_NONE, 0>(s, g);
            if (m == 1) return launch256<T, T, A4R_ACT_NONE, A4R_ACT_NONE, 1>(s, g);
            if (m == 2) return launch256<T, T, A4R_ACT_NONE, A4R_ACT_NONE, 2>(s, g);
            if (m == 3) return launch256<T, T, A4R_ACT_NONE, A4R_ACT_NONE, 3>(s, g);      // dense + dropout + residual: an un-adapted BertSelfOutput / BertOutput (Pfeiffer, LoRA)
        }
        if (g.act == A4R_ACT_GELU && g.dact == A4R_ACT_NONE && m == 8)               // (64: the second output is the 8-bit derivative, known at compile time)
            return (g.c2_mode == 2 && A4R_C2Q8_CT) ? launch256<T, T, A4R_ACT_GELU, A4R_ACT_NONE, 8 | 64 * A4R_C2Q8_CT>(s, g) : launch256<T, T, A4R_ACT_GELU, A4R_ACT_NONE, 8>(s, g);
        if (g.act == A4R_ACT_NONE && g.dact == A4R_DACT_MULQ8_ && m == 0)            // (32: tile-native derivative, requested in front of the K loop)
            return g.q8_tiled ? launch256<T, T, A4R_ACT_NONE, A4R_DACT_MULQ8_, 32>(s, g) : launch256<T, T, A4R_ACT_NONE, A4R_DACT_MULQ8_, 0>(s, g);
        if (g.act == A4R_ACT_NONE && g.dact == A4R_DACT_MUL_ && m == 0) return launch256<T, T, A4R_ACT_NONE, A4R_DACT_MUL_, 0>(s, g);
    }
    if (g.act == A4R_ACT_NONE && g.dact == A4R_ACT_NONE) return launch256<T, T, A4R_ACT_NONE, A4R_ACT_NONE>(s, g);
    if (g.act == A4R_ACT_GELU && g.dact == A4R_ACT_NONE) return launch256<T, T, A4R_ACT_GELU, A4R_ACT_NONE>(s, g);
    if (g.act == A4R_ACT_RELU && g.dact == A4R_ACT_NONE) return launch256<T, T, A4R_ACT_RELU, A4R_ACT_NONE>(s, g);
    if (g.act == A4R_ACT_NONE && g.dact == A4R_DACT_MUL_) return launch256<T, T, A4R_ACT_NONE, A4R_DACT_MUL_>(s, g);
    if constexpr (sizeof(T) == 2) {
        if (g.act == A4R_ACT_NONE && g.dact == A4R_DACT_MULQ8_) return launch256<T, T, A4R_ACT_NONE, A4R_DACT_MULQ8_>(s, g);
    }
    if (g.act == A4R_ACT_NONE && g.dact == A4R_ACT_GELU) return launch256<T, T, A4R_ACT_NONE, A4R_ACT_GELU>(s, g);
    if (g.act == A4R_ACT_NONE && g.dact == A4R_ACT_RELU) return launch256<T, T, A4R_ACT_NONE, A4R_ACT_RELU>(s, g);
    return 1;
}

}  // namespace

// called by a4r_gemm_nt (a4r_gemm.hip) after argument validation; returns 1 when the combination is not instantiated
int a4r_gemm_nt_256(hipStream_t s, const a4r_gemm_t& g) {
    if (g.bias && (reinterpret_cast<uintptr_t>(g.bias) & 3u)) return 1;
    if ((uint64_t)g.M * (uint64_t)g.ldc * (g.out_dtype == A4R_F32 ? 4u : 2u) >= (1ull << 32)) return 1;      // 32-bit per-lane output offsets
    if (g.in_dtype == A4R_FP8) {                      // e4m3 operands (frozen-backbone GEMMs): plain and GELU (+ derivative) epilogues, and the
        if (g.out_dtype != A4R_BF16 || !g.scale_a || !g.scale_b) return 1;          // two forms whose OUTPUT is the next fp8 GEMM's A operand
        const int m = epi_mask(g);
        if (g.c_fp8) {
            if ((g.c_fp8 != 1 && g.c_fp8 != 2) || !(g.c_scale > 0.f) || (g.c_fp8 == 2 && !g.c_scale_out)) return 1;
            if (g.act == A4R_ACT_GELU && g.dact == A4R_ACT_NONE && m == 8 && g.c2_mode == 2)              // FFN-up: u as e4m3 + gelu' as 8 bits
                return launch256<fp8_t, bf16_t, A4R_ACT_GELU, A4R_ACT_NONE, 24 | 64 * A4R_C2Q8_CT>(s, g);
            if (g.act == A4R_ACT_NONE && g.dact == A4R_DACT_MULQ8_ && m == 0)                             // d FFN-down: du = (dy W) * gelu' as e4m3
                return launch256<fp8_t, bf16_t, A4R_ACT_NONE, A4R_DACT_MULQ8_, 16>(s, g);
            return 1;
        }
        if (g.dact != A4R_ACT_NONE) return 1;
        if (g.act == A4R_ACT_NONE) {                  // (round 5: the dropout / residual forms without their run-time tests too -- 33 launches of the fp8 step ran the all-purpose one)
            if (m == 0) return launch256<fp8_t, bf16_t, A4R_ACT_NONE, A4R_ACT_NONE, 0>(s, g);
            if (m == 1) return launch256<fp8_t, bf16_t, A4R_ACT_NONE, A4R_ACT_NONE, 1>(s, g);
            if (m == 2) return launch256<fp8_t, bf16_t, A4R_ACT_NONE, A4R_ACT_NONE, 2>(s, g);
            if (m == 3) return launch256<fp8_t, bf16_t, A4R_ACT_NONE, A4R_ACT_NONE, 3>(s, g);
            return launch256<fp8_t, bf16_t, A4R_ACT_NONE, A4R_ACT_NONE>(s, g);
        }
        if (g.act == A4R_ACT_GELU) return m == 8 ? launch256<fp8_t, bf16_t, A4R_ACT_GELU, A4R_ACT_NONE, 8>(s, g) : launch256<fp8_t, bf16_t, A4R_ACT_GELU, A4R_ACT_NONE>(s, g);
        return 1;
    }
    if (g.in_dtype == A4R_BF16 && g.out_dtype == A4R_BF16) return dispatch_same<bf16_t>(s, g);
    if (g.in_dtype == A4R_F32 && g.out_dtype == A4R_F32) return dispatch_same<float>(s, g);
    if (g.act != A4R_ACT_NONE || g.dact != A4R_ACT_NONE) return 1;
    if (g.in_dtype == A4R_BF16) return launch256<bf16_t, float, A4R_ACT_NONE, A4R_ACT_NONE>(s, g);
    return launch256<float, bf16_t, A4R_ACT_NONE, A4R_ACT_NONE>(s, g);
}
