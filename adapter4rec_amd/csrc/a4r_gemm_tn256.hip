// gemm_tn_256_kernel: the large-output path of a4r_gemm_tn -- C[P,Q] (fp32, +=) = X[M,P]^T . Y[M,Q] with P % 256 == 0, Q % 256 == 0, bf16:
// the weight gradients of TRAINABLE BACKBONE Linears (dW = dy^T x, [768 | 3072] x [768 | 3072] outputs over M = 40 448 token rows), i.e. the
// Pretraining/ half of the reference (`--fine_tune_to all`, Pretraining/Text/run.py:319-324 -> autograd of every nn.Linear of HF BertLayer) and
// Downstream/*'s end-to-end fine-tuning baseline.  The 64 x 64-tile kernel of a4r_gemm_tn.hip is built for the adapters' [*, 64] outputs; on
// these shapes it moves 32 flops per operand byte through the CU's load path and ran at 0.17 of the MFMA peak (16.5 of a 38 ms step).
//
// 256 x 256 output tile per workgroup, 512 threads = 8 waves laid out 2 (P) x 4 (Q); a wave owns 128 x 64 = 8 x 4 MFMA 16x16 tiles (128
// accumulator registers) for its whole token range; two waves per SIMD, one workgroup per CU.  The contraction runs over TOKENS, i.e. down the
// rows of both row-major operands: a stage is 32 tokens of the tile's 256 X columns and 256 Y columns, stored as eight [32 tokens][128 B]
// images (four 64-column groups per operand, 4 KiB each, XOR-swizzled by c ^ ((r >> 1) & 7) on the DMA SOURCE side: the image is
// lane-linear), written by global_load_lds_dwordx4 (one instruction = 8 token rows x one 128-byte line) into a FOUR-deep ring (4 x 32 KiB) that
// stays in flight across the barriers behind counted vmcnt waits (up to three stages = 96 KiB requested ahead per CU), and read
// with ds_read_b64_tr_b16 -- a lane gets tokens {4kg .. 4kg+3} and {16 + 4kg .. +3} of its column, the same permutation of the contraction
// index for both operands (the fragment form of gemm_tn_glds_kernel).  Per stage and wave: 24 transposed reads, 32 MFMAs.
// M is split over the grid (tiles x splits ~ one workgroup per CU); a workgroup flushes its tile ONCE with fp32 atomics.  Workgroups are dealt to
// the XCDs so that one XCD owns whole token ranges: the tiles that share an X slab (same P panel) or a Y slab (same Q panel) of a token range
// read it from that XCD's L2.
// A four-wave form (one wave per SIMD, 128 x 128 wave tiles, 256 AGPR accumulators, fragments double-buffered in registers: 2/3 of the LDS reads per
// MFMA, half the waves per barrier) was built and measured 8 - 11 % slower (commit ec8ff10, profiles/r04_y_tn256_four_wave_ab.txt, LOG.md part D).
// xsum (optional): xsum[p] += the column sums of X -- the bias gradient db = colsum(dy) next to dW = dy^T x -- as MFMAs against an all-ones
// operand in the workgroups of the first Q panel, spread over the four Q-side waves (2 extra MFMAs per 32).
#include <cstdlib>
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

typedef short tnb_v4s_t __attribute__((ext_vector_type(4)));
constexpr int TNB_TOK = 32;                  // tokens per stage
constexpr int TNB_IMG = TNB_TOK * 128;       // one 64-column image of a stage: 4 KiB
constexpr int TNB_STAGE = 8 * TNB_IMG;       // X: images 0-3, Y: images 4-7
constexpr int TNB_NST = 4;

A4R_DEV void tnb_glds16(const void* base, uint32_t voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
A4R_DEV uint4 tnb_frag(const char* img, uint32_t off0, uint32_t off1) {
    const tnb_v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tnb_v4s_t*)(img + off0));
    const tnb_v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tnb_v4s_t*)(img + off1));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// up to four products over the same token rows in one launch (the three slices of a fused qkv gradient + the attention output's: one 64 MB
// flush for 36 tiles instead of four for 9 each); tile0 = index of a product's first tile in the launch's tile list
struct Tn256P { const bf16_t* X; const bf16_t* Y; float* C; float* xsum; int ldx, ldy, ldc, ntq, tile0, pad; };
struct Tn256 { Tn256P pr[4]; int nprob, tiles, M, splits, rows_per_split, flags; };

__global__ void __launch_bounds__(512, 1) gemm_tn_256_kernel(const Tn256 p) {
    __shared__ __attribute__((aligned(16))) char lds[TNB_NST * TNB_STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 2, wq = wave & 3;

    // XCD x (= blockIdx.x & 7 under round-robin placement: speed only) owns a contiguous range of v = split * tiles + tile
    const int tiles = p.tiles, total = tiles * p.splits;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, q8 = total >> 3, r8 = total & 7;
    const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    const int split = v / tiles, tile_l = v - split * tiles;
    int k = 0;
    for (int i = 1; i < p.nprob; ++i) k = tile_l >= p.pr[i].tile0 ? i : k;
    const Tn256P& pk = p.pr[k];
    const int tile = tile_l - pk.tile0;
    const int tp = tile / pk.ntq, tq = tile - tp * pk.ntq;
    const int m_begin = split * p.rows_per_split;
    const int m_end = min(p.M, m_begin + p.rows_per_split);
    const int ns = (m_end - m_begin) / TNB_TOK;
    const int ldx = pk.ldx, ldy = pk.ldy;
    const char* Xb = reinterpret_cast<const char*>(pk.X + (size_t)m_begin * ldx + tp * 256);
    const char* Yb = reinterpret_cast<const char*>(pk.Y + (size_t)m_begin * ldy + tq * 256);
    float* const xsum_k = pk.xsum;
    float* const C_k = pk.C;
    const int ldc = pk.ldc;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // DMA: a stage is 32 one-KiB pieces (image i = piece >> 2, row block rb = piece & 3: token rows 8 rb .. 8 rb + 7); wave w issues X pieces
    // 2w, 2w + 1 and the Y pieces of the same numbers.  Lane l writes LDS slot (row 8 rb + (l >> 3), chunk l & 7) and fetches source chunk
    // (l & 7) ^ ((row >> 1) & 7) of that row's 128-byte line.
    uint32_t voffX[2], voffY[2], dstX[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = 2 * wave + i, img = piece >> 2, rb = piece & 3;
        const int r = 8 * rb + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        voffX[i] = (uint32_t)(r * ldx * 2 + img * 128 + c * 16);
        voffY[i] = (uint32_t)(r * ldy * 2 + img * 128 + c * 16);
        dstX[i] = lds0 + (uint32_t)(img * TNB_IMG + rb * 1024);
    }
    auto issue = [&](int t) {
        if (t < ns) {
            const uint32_t sb = (uint32_t)((t & (TNB_NST - 1)) * TNB_STAGE);
            const char* x = Xb + (size_t)t * TNB_TOK * ldx * 2;
            const char* y = Yb + (size_t)t * TNB_TOK * ldy * 2;
            tnb_glds16(x, voffX[0], dstX[0] + sb);
            tnb_glds16(x, voffX[1], dstX[1] + sb);
            tnb_glds16(y, voffY[0], dstX[0] + sb + 4 * TNB_IMG);
            tnb_glds16(y, voffY[1], dstX[1] + sb + 4 * TNB_IMG);
        }
    };
    issue(0);
    issue(1);
    issue(2);

    // fragment read offsets inside an image for the four 16-column groups d = 0..3 (columns 16 d .. 16 d + 15): tokens r0 = 4 kg + q and r0 + 16
    uint32_t fo0[4], fo1[4];
    {
        const int kg = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int r0 = 4 * kg + q, r1 = r0 + 16;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int chunk = 2 * d + (pp >> 1);
            fo0[d] = (uint32_t)(r0 * 128 + ((chunk ^ ((r0 >> 1) & 7)) << 4) + 8 * (pp & 1));
            fo1[d] = (uint32_t)(r1 * 128 + ((chunk ^ ((r1 >> 1) & 7)) << 4) + 8 * (pp & 1));
        }
    }

    f32x4_t acc[8][4];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[f][g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bool xs = xsum_k != nullptr && tq == 0;                     // (workgroup-uniform)
    f32x4_t xs0 = {0.f, 0.f, 0.f, 0.f}, xs1 = xs0;
    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
    // column sums: wave wq takes X fragments f = 2 wq, 2 wq + 1 of its P half (image f >> 2, 16-column groups f & 3)
    uint32_t xo0[2], xo1[2];
    const int xs_img = (wq >> 1) * TNB_IMG;
    {
        const int kg = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const int r0 = 4 * kg + q, r1 = r0 + 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int chunk = 2 * (((2 * wq) & 3) + i) + (pp >> 1);
            xo0[i] = (uint32_t)(r0 * 128 + ((chunk ^ ((r0 >> 1) & 7)) << 4) + 8 * (pp & 1));
            xo1[i] = (uint32_t)(r1 * 128 + ((chunk ^ ((r1 >> 1) & 7)) << 4) + 8 * (pp & 1));
        }
    }
    const char* xa = lds + wp * 2 * TNB_IMG;                          // this wave's two X images, its Y image
    const char* yb = lds + (4 + wq) * TNB_IMG;

    // Schedule: a PING-PONG between the two waves of every SIMD (waves w and w + 4, i.e. the two P halves wp = 0 / 1), as in a4r_gemm256.hip.  A stage
    // is, for every wave, LOAD segment (issue stage t + 3's DMA, 24 transposed reads of stage t, lgkmcnt(0), counted vmcnt for stage t + 1) |
    // s_barrier | MFMA segment (32 MFMAs) | s_barrier; waves 4-7 execute ONE extra barrier before the loop (waves 0-3 one after it), so one half is in
    // its MFMA segment while the other loads: each SIMD's matrix pipe has one wave issuing MFMAs back to back and the partner's LDS latency hides
    // behind them.  Hazards (half A = waves 0-3, half B one barrier later):
    //   RAW  stage t + 1 is read first in A's LOAD_(t+1); every wave's counted wait for it sits in ITS LOAD_t, which both halves finish before the
    //        barrier in front of A's LOAD_(t+1).  Stage 0: waited for in the prologue, in front of a barrier of all waves.
    //   WAR  LOAD_t refills the slot of stage t - 1; both halves' reads of stage t - 1 were COMPLETE (lgkmcnt(0) inside the LOAD segment) before
    //        a barrier that precedes either half's LOAD_t.
    if (ns > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // stage 0 (stages 1, 2 stay in flight)
    else if (ns > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wp == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < ns; ++t) {
        issue(t + 3);                                                 // into the slot of stage t - 1
        const int sb = (t & (TNB_NST - 1)) * TNB_STAGE;
        uint4 a[8], b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = tnb_frag(yb + sb, fo0[g], fo1[g]);
#pragma unroll
        for (int f = 0; f < 8; ++f) a[f] = tnb_frag(xa + sb + (f >> 2) * TNB_IMG, fo0[f & 3], fo1[f & 3]);
        uint4 sx0 = ones, sx1 = ones;
        if (xs) {        // the wave's two X fragments of the column sums, read once more (an index into a[] by wq would demote the array to scratch)
            sx0 = tnb_frag(xa + sb + xs_img, xo0[0], xo1[0]);
            sx1 = tnb_frag(xa + sb + xs_img, xo0[1], xo1[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // stage t + 1 has landed for this wave when at most the stages issued after it (t + 2, t + 3: 4 instructions each) are outstanding
        if (t + 3 < ns) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (t + 2 < ns) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int g = 0; g < 4; ++g) Mma<bf16_t>::mma(a[f], b[g], acc[f][g]);
        if (xs) {
            Mma<bf16_t>::mma(sx0, ones, xs0);
            Mma<bf16_t>::mma(sx1, ones, xs1);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    if (wp == 0) __builtin_amdgcn_s_barrier();
#ifdef A4R_DEBUG_KNOBS                                                 // tools-only build (make DEBUG_KNOBS=1): never in the shipped library
    if (p.flags & 1) return;                                          // (timing-only diagnostic: A4R_TN256_DBG=1, no flush -- results wrong)
#endif

    const int prow = tp * 256 + wp * 128 + (lane >> 4) * 4, qcol = tq * 256 + wq * 64 + (lane & 15);
    if (xs && (lane & 15) == 0) {            // every column of the ones product holds the row sum: one lane per row writes it
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            atomicAdd(xsum_k + prow + (2 * wq) * 16 + rr, xs0[rr]);
            atomicAdd(xsum_k + prow + (2 * wq + 1) * 16 + rr, xs1[rr]);
        }
    }
    float* __restrict__ C = C_k;
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int g = 0; g < 4; ++g) atomicAdd(C + (size_t)(prow + f * 16 + rr) * ldc + qcol + g * 16, acc[f][g][rr]);
}

}  // namespace

// 1 = the shape takes the 256-tile kernel (a4r_gemm_tn / a4r_gemm_tn_bias dispatch on it; A4R_TN256=0: never, A/B runs)
int g_tn256_on = 1;            // a4r_gemm_variant(6 / 7)
int a4r_tn256_takes(int M, int P, int Q, int dtype) {
    static const int on = getenv("A4R_TN256") ? atoi(getenv("A4R_TN256")) != 0 : 1;
    return on && g_tn256_on && dtype == A4R_BF16 && P % 256 == 0 && Q % 256 == 0 && M % 64 == 0 && M >= 4096;
}

// n <= 4 products over the same M token rows; every product must pass a4r_tn256_takes
int a4r_tn256_launch_multi(void* stream, int n, const void* const* X, const int* ldx, const void* const* Y, const int* ldy, float* const* C, const int* ldc,
                           const int* P, const int* Q, float* const* xsum, int M) {
    Tn256 p;
    int tiles = 0;
    for (int i = 0; i < 4; ++i) {
        const int s = i < n ? i : 0;
        p.pr[i].X = (const bf16_t*)X[s]; p.pr[i].Y = (const bf16_t*)Y[s]; p.pr[i].C = C[s]; p.pr[i].xsum = xsum ? xsum[s] : nullptr;
        p.pr[i].ldx = ldx[s]; p.pr[i].ldy = ldy[s]; p.pr[i].ldc = ldc[s]; p.pr[i].ntq = Q[s] / 256; p.pr[i].tile0 = tiles; p.pr[i].pad = 0;
        if (i < n) tiles += (P[s] / 256) * (Q[s] / 256);
    }
    p.nprob = n; p.tiles = tiles; p.M = M;
#ifdef A4R_DEBUG_KNOBS
    static const int dbg = getenv("A4R_TN256_DBG") ? atoi(getenv("A4R_TN256_DBG")) : 0;          // bit 0: no flush (timing only, results wrong)
    p.flags = dbg;
#else
    p.flags = 0;
#endif
    const int stages = M / 64;
    static const int wgs_env = getenv("A4R_TN256_WGS") ? atoi(getenv("A4R_TN256_WGS")) : 256;          // (A/B runs)
    const int wgs = wgs_env > 0 ? wgs_env : 256;
    int splits = wgs / tiles;                               // one workgroup per CU (128 KiB of LDS), no second round
    if (splits < 1) splits = 1;
    if (splits > stages) splits = stages;
    p.rows_per_split = ((stages + splits - 1) / splits) * 64;
    p.splits = (M + p.rows_per_split - 1) / p.rows_per_split;
    hipLaunchKernelGGL(gemm_tn_256_kernel, dim3(tiles * p.splits), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), p);
    return a4r_launch_status();
}

int a4r_tn256_launch(void* stream, const void* X, int ldx, const void* Y, int ldy, float* C, int ldc, int M, int P, int Q, float* xsum) {
    return a4r_tn256_launch_multi(stream, 1, &X, &ldx, &Y, &ldy, &C, &ldc, &P, &Q, &xsum, M);
}
