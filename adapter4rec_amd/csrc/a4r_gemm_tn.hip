// a4r_gemm_tn: C[P,Q] (fp32, +=) = X[M,P]^T . Y[M,Q]  -- the weight gradients of the trainable
// adapter matrices.  The contraction runs over ROWS (tokens) of both operands, so both MFMA
// operands are gathered down the columns of row-major LDS tiles (gather_chunk).  M is split over
// the grid; each workgroup keeps a 64x64 fp32 partial in registers for its whole token range and
// flushes it once with fp32 atomics (4 waves x 2x2 tiles of 16x16).
// a4r_colsum: out[N] += sum_m X[m, :]  (bias gradients).
#include <cstdlib>
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

template <typename T> struct TnCfg;
template <> struct TnCfg<bf16_t> { static constexpr int STRIDE = 136; };  // 128 B of data + 8: kg rows land 16 banks apart
template <> struct TnCfg<float> { static constexpr int STRIDE = 272; };   // 256 B of data + 16: 4 rows = 272 dwords = 16 banks apart

template <typename T>
__global__ void __launch_bounds__(256) gemm_tn_kernel(const T* __restrict__ X, int ldx, const T* __restrict__ Y, int ldy,
                                                      float* __restrict__ C, int ldc, int M, int ntq, int rows_per_split) {
    constexpr int STRIDE = TnCfg<T>::STRIDE;
    constexpr int PER = Elem<T>::PER16;
    constexpr int CPR = 64 / PER;                  // 16-byte chunks per 64-element tile row
    constexpr int NCH = 64 * CPR / 256;            // chunks each thread stages per operand (2 bf16 / 4 fp32)
    constexpr int KS = 64 / Mma<T>::KSTEP;         // chunk steps per 64-row stage
    __shared__ __attribute__((aligned(16))) char Xs[64 * STRIDE];
    __shared__ __attribute__((aligned(16))) char Ys[64 * STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave >> 1, wq = wave & 1;
    const int tile = blockIdx.x, split = blockIdx.y;
    const int p0 = (tile / ntq) * 64, q0 = (tile % ntq) * 64;
    const int m_begin = split * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);

    f32x4_t acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc10 = acc00, acc11 = acc00;
    for (int m0 = m_begin; m0 < m_end; m0 += 64) {
        // explicit scalars (arrays held across the barrier were demoted to scratch in the fp32 build)
        const int srow = tid / CPR, sch = tid % CPR;              // chunk id = tid + 256*i -> row = srow + (256/CPR)*i
        constexpr int RSTEP = 256 / CPR;
        const T* xg = X + (size_t)(m0 + srow) * ldx + p0 + sch * PER;
        const T* yg = Y + (size_t)(m0 + srow) * ldy + q0 + sch * PER;
        uint4 rx0, rx1, rx2, rx3, ry0, ry1, ry2, ry3;
        rx0 = *reinterpret_cast<const uint4*>(xg);
        rx1 = *reinterpret_cast<const uint4*>(xg + (size_t)RSTEP * ldx);
        ry0 = *reinterpret_cast<const uint4*>(yg);
        ry1 = *reinterpret_cast<const uint4*>(yg + (size_t)RSTEP * ldy);
        if constexpr (NCH == 4) {
            rx2 = *reinterpret_cast<const uint4*>(xg + (size_t)2 * RSTEP * ldx);
            rx3 = *reinterpret_cast<const uint4*>(xg + (size_t)3 * RSTEP * ldx);
            ry2 = *reinterpret_cast<const uint4*>(yg + (size_t)2 * RSTEP * ldy);
            ry3 = *reinterpret_cast<const uint4*>(yg + (size_t)3 * RSTEP * ldy);
        }
        __syncthreads();   // previous stage fully consumed
        char* xs = Xs + srow * STRIDE + sch * 16;
        char* ys = Ys + srow * STRIDE + sch * 16;
        if constexpr (sizeof(T) == 2) {   // 136-byte rows are only 8-byte aligned
            reinterpret_cast<uint2*>(xs)[0] = make_uint2(rx0.x, rx0.y); reinterpret_cast<uint2*>(xs)[1] = make_uint2(rx0.z, rx0.w);
            reinterpret_cast<uint2*>(xs + RSTEP * STRIDE)[0] = make_uint2(rx1.x, rx1.y);
            reinterpret_cast<uint2*>(xs + RSTEP * STRIDE)[1] = make_uint2(rx1.z, rx1.w);
            reinterpret_cast<uint2*>(ys)[0] = make_uint2(ry0.x, ry0.y); reinterpret_cast<uint2*>(ys)[1] = make_uint2(ry0.z, ry0.w);
            reinterpret_cast<uint2*>(ys + RSTEP * STRIDE)[0] = make_uint2(ry1.x, ry1.y);
            reinterpret_cast<uint2*>(ys + RSTEP * STRIDE)[1] = make_uint2(ry1.z, ry1.w);
        } else {
            *reinterpret_cast<uint4*>(xs) = rx0;
            *reinterpret_cast<uint4*>(xs + RSTEP * STRIDE) = rx1;
            *reinterpret_cast<uint4*>(xs + 2 * RSTEP * STRIDE) = rx2;
            *reinterpret_cast<uint4*>(xs + 3 * RSTEP * STRIDE) = rx3;
            *reinterpret_cast<uint4*>(ys) = ry0;
            *reinterpret_cast<uint4*>(ys + RSTEP * STRIDE) = ry1;
            *reinterpret_cast<uint4*>(ys + 2 * RSTEP * STRIDE) = ry2;
            *reinterpret_cast<uint4*>(ys + 3 * RSTEP * STRIDE) = ry3;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int k0 = ks * Mma<T>::KSTEP;
            const uint4 a0 = gather_chunk<T>(Xs, STRIDE, k0, wp * 32, lane);
            const uint4 a1 = gather_chunk<T>(Xs, STRIDE, k0, wp * 32 + 16, lane);
            const uint4 b0 = gather_chunk<T>(Ys, STRIDE, k0, wq * 32, lane);
            const uint4 b1 = gather_chunk<T>(Ys, STRIDE, k0, wq * 32 + 16, lane);
            Mma<T>::mma(a0, b0, acc00);
            Mma<T>::mma(a0, b1, acc01);
            Mma<T>::mma(a1, b0, acc10);
            Mma<T>::mma(a1, b1, acc11);
        }
    }
    const int prow = p0 + wp * 32 + (lane >> 4) * 4, qcol = q0 + wq * 32 + (lane & 15);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        atomicAdd(C + (size_t)(prow + rr) * ldc + qcol, acc00[rr]);
        atomicAdd(C + (size_t)(prow + rr) * ldc + qcol + 16, acc01[rr]);
        atomicAdd(C + (size_t)(prow + 16 + rr) * ldc + qcol, acc10[rr]);
        atomicAdd(C + (size_t)(prow + 16 + rr) * ldc + qcol + 16, acc11[rr]);
    }
}

// ---- bf16 form: LDS-DMA ring + hardware-transposed fragment reads.
// The kernel above stages through registers one 64-token stage at a time (load -> barrier -> ds_write -> barrier -> gather with 64
// ds_read_u16 per wave -> 8 MFMA): nothing is in flight while it computes, 24.5 us for the 67 MB of a 40448 x (768 | 64) pair.
// Here a stage (64 tokens x 64 columns of X and of Y, 8 KiB each, 128-byte rows with the chunk swizzle c ^ ((row >> 1) & 7)
// applied on the DMA source address) is written by global_load_lds_dwordx4 into a 3-deep ring that stays in flight across the
// ONE barrier per stage, and the operands -- contraction index running DOWN the rows -- come from ds_read_b64_tr_b16: a lane
// gets tokens {4kg .. 4kg+3} and {16 + 4kg .. +3} of its column, the same permutation of the contraction index for both
// operands.  48 KiB of LDS: three workgroups per CU, ~96 KiB in flight per CU.
typedef short tn_v4s_t __attribute__((ext_vector_type(4)));
A4R_DEV uint4 tn_frag_tr(const char* img, int d0, int st, int lane) {
    const int kg = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int chunk = (d0 >> 3) + (pp >> 1);
    const int r0 = 32 * st + 4 * kg + q, r1 = r0 + 16;
    const char* a0 = img + r0 * 128 + ((chunk ^ ((r0 >> 1) & 7)) << 4) + 8 * (pp & 1);
    const char* a1 = img + r1 * 128 + ((chunk ^ ((r1 >> 1) & 7)) << 4) + 8 * (pp & 1);
    const tn_v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_v4s_t*)(a0));
    const tn_v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_v4s_t*)(a1));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}
A4R_DEV void tn_glds16(const void* base, uint32_t voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

// xsum (optional): xsum[p] += sum over the rows of X[:, p] -- the bias gradient that belongs to a weight gradient (db_up = colsum(dv) next to
// dW_up = dv^T z, db_down = colsum(dzp) next to dW_down = dzp^T h) as two more MFMAs per stage against an all-ones operand in the workgroups
// of the first Q-tile: 64 atomics per workgroup on ~32-way contended addresses instead of the 832 the fused adapter backward issued from
// each of its 256 workgroups onto the same 832 addresses (6 - 9 us at the end of that launch).
struct TnProb { const bf16_t* X; const bf16_t* Y; float* C; int ldx, ldy, ldc, ntq; float* xsum; };
__global__ void __launch_bounds__(256) gemm_tn_glds_kernel(const TnProb pa, const TnProb pb, int M, int rows_per_split, int xcd_groups) {
    // blockIdx.z picks one of two products of one launch (the two weight gradients of an adapter: different operands, same token range)
    // xcd_groups (round 4): the T = gridDim.x output tiles of one (token range, product) share the NARROW operand's slab ([rows, 64]: z for
    // dW_up, dzp for dW_down) -- 12 tiles at H = 768 -- and cut every row of the wide operand into 128-byte pieces.  Dealt to the XCDs in launch
    // order (x fastest) the 12 land on all 8 XCDs and the slab is fetched 8 times; remapped so that a group sits on ONE XCD (workgroups b
    // and b + 8 share an XCD under round-robin placement: speed only) it is fetched once and re-read from that XCD's L2.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (xcd_groups) {
        const int T = gridDim.x, lin = blockIdx.x + T * (blockIdx.y + gridDim.y * blockIdx.z);
        const int slot = lin >> 3, grp = (slot / T) * 8 + (lin & 7);
        bx = slot % T;
        by = grp % (int)gridDim.y;
        bz = grp / (int)gridDim.y;
    }
    const TnProb& pr = bz == 0 ? pa : pb;
    const bf16_t* __restrict__ X = pr.X;
    const bf16_t* __restrict__ Y = pr.Y;
    float* __restrict__ C = pr.C;
    const int ldx = pr.ldx, ldy = pr.ldy, ldc = pr.ldc, ntq = pr.ntq;
    constexpr int NST = 3, STAGE = 16384;
    __shared__ __attribute__((aligned(16))) char lds[NST * STAGE];        // [stage][X 64 tokens | Y 64 tokens][128 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 1, wq = wave & 1;
    const int tile = bx, split = by;
    const int p0 = (tile / ntq) * 64, q0 = (tile % ntq) * 64;
    const int m_begin = split * rows_per_split;
    const int m_end = min(M, m_begin + rows_per_split);
    const int ns = (m_end - m_begin) / 64;
    const char* Xb = reinterpret_cast<const char*>(X + (size_t)m_begin * ldx + p0);
    const char* Yb = reinterpret_cast<const char*>(Y + (size_t)m_begin * ldy + q0);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    uint32_t voffX[2], voffY[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                          // wave w stages tokens 16w .. 16w+15 of both operands
        const int ul = 8 * (2 * wave + i) + (lane >> 3);
        const int c = (lane & 7) ^ ((ul >> 1) & 7);
        voffX[i] = (uint32_t)(ul * ldx * 2 + c * 16);
        voffY[i] = (uint32_t)(ul * ldy * 2 + c * 16);
    }
    const uint32_t dst0 = lds0 + (uint32_t)(2 * wave) * 1024u;
    auto issue = [&](int t) {
        if (t < ns) {
            const uint32_t d = dst0 + (uint32_t)(t % NST) * STAGE;
            const char* x = Xb + (size_t)t * 64 * ldx * 2;
            const char* y = Yb + (size_t)t * 64 * ldy * 2;
            tn_glds16(x, voffX[0], d);
            tn_glds16(x, voffX[1], d + 1024u);
            tn_glds16(y, voffY[0], d + 8192u);
            tn_glds16(y, voffY[1], d + 8192u + 1024u);
        }
    };
    issue(0);
    issue(1);
    f32x4_t acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc10 = acc00, acc11 = acc00;
    const bool xs = pr.xsum != nullptr && (tile % ntq) == 0 && wq == 0;          // (wave-uniform)
    f32x4_t xs0 = acc00, xs1 = acc00;
    const uint4 ones = make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
    for (int t = 0; t < ns; ++t) {
        if (t + 1 < ns) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(t + 2);
        const char* xs = lds + (t % NST) * STAGE;
        const char* ys = xs + 8192;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const uint4 a0 = tn_frag_tr(xs, wp * 32, st, lane), a1 = tn_frag_tr(xs, wp * 32 + 16, st, lane);
            const uint4 b0 = tn_frag_tr(ys, wq * 32, st, lane), b1 = tn_frag_tr(ys, wq * 32 + 16, st, lane);
            Mma<bf16_t>::mma(a0, b0, acc00);
            Mma<bf16_t>::mma(a0, b1, acc01);
            Mma<bf16_t>::mma(a1, b0, acc10);
            Mma<bf16_t>::mma(a1, b1, acc11);
            if (xs) {
                Mma<bf16_t>::mma(a0, ones, xs0);
                Mma<bf16_t>::mma(a1, ones, xs1);
            }
        }
    }
    const int prow = p0 + wp * 32 + (lane >> 4) * 4, qcol = q0 + wq * 32 + (lane & 15);
    if (xs && (lane & 15) == 0) {            // every column of the ones product holds the row sum: one lane per row writes it
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            atomicAdd(pr.xsum + prow + rr, xs0[rr]);
            atomicAdd(pr.xsum + prow + 16 + rr, xs1[rr]);
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        atomicAdd(C + (size_t)(prow + rr) * ldc + qcol, acc00[rr]);
        atomicAdd(C + (size_t)(prow + rr) * ldc + qcol + 16, acc01[rr]);
        atomicAdd(C + (size_t)(prow + 16 + rr) * ldc + qcol, acc10[rr]);
        atomicAdd(C + (size_t)(prow + 16 + rr) * ldc + qcol + 16, acc11[rr]);
    }
}

// block = NCG column groups (8 columns each, NCG = min(N/8, 32)) x 256/NCG row lanes; grid.y strides the rows.
// Row lanes are reduced through LDS so that a block issues ONE atomic per column (all blocks hit the same
// N addresses: per-thread atomics there ran at the contended-atomic rate, 0.8 ms for a 40k x 64 input).
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ X, int ldx, float* __restrict__ out, int M, int N, int ncg) {
    __shared__ float red[256][9];
    const int nrl = 256 / ncg;
    const int cgl = threadIdx.x % ncg, rlane = threadIdx.x / ncg;
    const int cg = blockIdx.x * ncg + cgl;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cg * 8 < N && rlane < nrl) {
        // UN rows per trip, all requested before the first is summed: with one load in flight per thread the launch ran at the
        // latency-bound rate (1.9 TB/s on the ViT tower's [66192, 768] gradients, 53 us; one workgroup of 4 waves per CU)
        constexpr int UN = 8;
        const int step = gridDim.y * nrl;
        for (int m0 = blockIdx.y * nrl + rlane; m0 < M; m0 += UN * step) {
            float v[UN][8];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int m = m0 + u * step;
                if (m < M) load_vec<T, 8>(X + (size_t)m * ldx + cg * 8, v[u]);
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[u][e] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += v[u][e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    // thread t < ncg*8 sums column (t / 8 group, t % 8 element) over the row lanes
    const int t = threadIdx.x;
    if (t < ncg * 8) {
        const int g = t >> 3, e = t & 7;
        float acc = 0.f;
        for (int r = 0; r < nrl; ++r) acc += red[r * ncg + g][e];
        const int col = (blockIdx.x * ncg + g) * 8 + e;
        if (col < N) atomicAdd(out + col, acc);
    }
}

}  // namespace

int g_tn_variant = 1;          // 0: register-staged kernel for bf16 too (tests / A-B via a4r_gemm_variant(0))

// 1 when the launch's (token range, product) groups can be dealt whole to the 8 XCDs (A4R_TN_XCD=0: launch order, A/B runs)
static int tn_xcd_groups(int groups) {
    static const int on = getenv("A4R_TN_XCD") ? atoi(getenv("A4R_TN_XCD")) != 0 : 1;
    return on && groups % 8 == 0;
}

// a4r_gemm_tn256.hip: the 256 x 256-tile kernel for large outputs (weight gradients of trainable backbone Linears)
int a4r_tn256_takes(int M, int P, int Q, int dtype);
int a4r_tn256_launch(void* stream, const void* X, int ldx, const void* Y, int ldy, float* C, int ldc, int M, int P, int Q, float* xsum);

extern "C" int a4r_gemm_tn(void* stream, const void* X, int ldx, const void* Y, int ldy, float* C, int ldc,
                           int M, int P, int Q, int dtype) {
    if (!X || !Y || !C || M <= 0 || P <= 0 || Q <= 0 || M % 64 || P % 64 || Q % 64) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    if ((dtype != A4R_F32 && dtype != A4R_BF16) || (ldx * esz) % 16 || (ldy * esz) % 16 || ldx < P || ldy < Q || ldc < Q) return A4R_EINVAL;
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15u) return A4R_EINVAL;
    if (g_tn_variant != 0 && a4r_tn256_takes(M, P, Q, dtype)) return a4r_tn256_launch(stream, X, ldx, Y, ldy, C, ldc, M, P, Q, nullptr);
    const int ntp = P / 64, ntq = Q / 64, tiles = ntp * ntq;
    const bool glds = dtype == A4R_BF16 && g_tn_variant != 0;
    int splits = ((glds ? 768 : 512) + tiles - 1) / tiles;  // one round of three (two) workgroups per CU
    const int stages = M / 64;
    if (splits > stages) splits = stages;
    const int rows_per_split = ((stages + splits - 1) / splits) * 64;
    splits = (M + rows_per_split - 1) / rows_per_split;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (glds) {
        const TnProb pa{(const bf16_t*)X, (const bf16_t*)Y, C, ldx, ldy, ldc, ntq, nullptr};
        hipLaunchKernelGGL(gemm_tn_glds_kernel, dim3(tiles, splits, 1), dim3(256), 0, s, pa, pa, M, rows_per_split, tn_xcd_groups(splits));
    }
    else if (dtype == A4R_BF16)
        hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, dim3(tiles, splits), dim3(256), 0, s, (const bf16_t*)X, ldx, (const bf16_t*)Y, ldy,
                           C, ldc, M, ntq, rows_per_split);
    else
        hipLaunchKernelGGL(gemm_tn_kernel<float>, dim3(tiles, splits), dim3(256), 0, s, (const float*)X, ldx, (const float*)Y, ldy,
                           C, ldc, M, ntq, rows_per_split);
    return a4r_launch_status();
}

// Two products over the same token range in one launch (an adapter's dW_up = dv^T z and dW_down = dzp^T h): the ramp-up, tail and
// atomic flush of one overlap the streaming of the other.  Same tile count required (P1 Q1 == P2 Q2), bf16 only.
extern "C" int a4r_gemm_tn2(void* stream, const void* X1, int ldx1, const void* Y1, int ldy1, float* C1, int ldc1, int P1, int Q1,
                            const void* X2, int ldx2, const void* Y2, int ldy2, float* C2, int ldc2, int P2, int Q2, int M, int dtype,
                            float* xsum1, float* xsum2) {
    if (!X1 || !Y1 || !C1 || !X2 || !Y2 || !C2 || M <= 0 || M % 64 || dtype != A4R_BF16) return A4R_EINVAL;
    if (P1 <= 0 || Q1 <= 0 || P2 <= 0 || Q2 <= 0 || P1 % 64 || Q1 % 64 || P2 % 64 || Q2 % 64 || (P1 / 64) * (Q1 / 64) != (P2 / 64) * (Q2 / 64)) return A4R_EINVAL;
    if ((ldx1 * 2) % 16 || (ldy1 * 2) % 16 || (ldx2 * 2) % 16 || (ldy2 * 2) % 16 || ldx1 < P1 || ldy1 < Q1 || ldc1 < Q1 || ldx2 < P2 || ldy2 < Q2 || ldc2 < Q2)
        return A4R_EINVAL;
    if ((reinterpret_cast<uintptr_t>(X1) | reinterpret_cast<uintptr_t>(Y1) | reinterpret_cast<uintptr_t>(X2) | reinterpret_cast<uintptr_t>(Y2)) & 15u) return A4R_EINVAL;
    const int tiles = (P1 / 64) * (Q1 / 64);
    static const int wgs_env = getenv("A4R_TN2_WGS") ? atoi(getenv("A4R_TN2_WGS")) : 384;                 // (A/B runs)
    const int wgs_per_product = wgs_env > 0 ? wgs_env : 384;                                                // (0 / not a number / negative: the default)
    int splits = (wgs_per_product + tiles - 1) / tiles;     // two products: half the splits of the single-product launch each
    const int stages = M / 64;
    if (splits > stages) splits = stages;
    const int rows_per_split = ((stages + splits - 1) / splits) * 64;
    splits = (M + rows_per_split - 1) / rows_per_split;
    const TnProb pa{(const bf16_t*)X1, (const bf16_t*)Y1, C1, ldx1, ldy1, ldc1, Q1 / 64, xsum1};
    const TnProb pb{(const bf16_t*)X2, (const bf16_t*)Y2, C2, ldx2, ldy2, ldc2, Q2 / 64, xsum2};
    hipLaunchKernelGGL(gemm_tn_glds_kernel, dim3(tiles, splits, 2), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), pa, pb, M, rows_per_split,
                       tn_xcd_groups(2 * splits));
    return a4r_launch_status();
}

extern "C" int a4r_colsum(void* stream, const void* X, int ldx, float* out, int M, int N, int dtype);

// a4r_gemm_tn plus xsum[p] += column sums of X (a Linear's weight AND bias gradient from one pass over dy: dW = dy^T x, db = colsum(dy)).
// Large bf16 outputs: both from the 256-tile launch; every other shape: a4r_gemm_tn followed by a4r_colsum.
extern "C" int a4r_gemm_tn_bias(void* stream, const void* X, int ldx, const void* Y, int ldy, float* C, int ldc,
                                int M, int P, int Q, int dtype, float* xsum) {
    if (!xsum) return a4r_gemm_tn(stream, X, ldx, Y, ldy, C, ldc, M, P, Q, dtype);
    if (!X || !Y || !C || M <= 0 || P <= 0 || Q <= 0 || M % 64 || P % 64 || Q % 64) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    if ((dtype != A4R_F32 && dtype != A4R_BF16) || (ldx * esz) % 16 || (ldy * esz) % 16 || ldx < P || ldy < Q || ldc < Q) return A4R_EINVAL;
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15u) return A4R_EINVAL;
    if (g_tn_variant != 0 && a4r_tn256_takes(M, P, Q, dtype)) return a4r_tn256_launch(stream, X, ldx, Y, ldy, C, ldc, M, P, Q, xsum);
    const int rc = a4r_gemm_tn(stream, X, ldx, Y, ldy, C, ldc, M, P, Q, dtype);
    return rc != A4R_OK ? rc : a4r_colsum(stream, X, ldx, xsum, M, P, dtype);
}

int a4r_tn256_launch_multi(void* stream, int n, const void* const* X, const int* ldx, const void* const* Y, const int* ldy, float* const* C, const int* ldc,
                           const int* P, const int* Q, float* const* xsum, int M);

// 1 <= n <= 4 weight (+ bias) gradients over the same M token rows: ONE launch of the 256-tile kernel when every product qualifies for it (the
// workgroups' single atomic flush is shared: 36 tiles x 7 token splits for q, k, v and the attention output together instead of 9 x 28 four times),
// otherwise a4r_gemm_tn_bias per product.
extern "C" int a4r_gemm_tn_multi(void* stream, const a4r_tn_prob_t* pr, int n, int M, int dtype) {
    if (!pr || n < 1 || n > 4) return A4R_EINVAL;
    bool big = g_tn_variant != 0;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    for (int i = 0; i < n; ++i) {
        const a4r_tn_prob_t& q = pr[i];
        if (!q.X || !q.Y || !q.C || M <= 0 || q.P <= 0 || q.Q <= 0 || M % 64 || q.P % 64 || q.Q % 64) return A4R_EINVAL;
        if ((dtype != A4R_F32 && dtype != A4R_BF16) || (q.ldx * esz) % 16 || (q.ldy * esz) % 16 || q.ldx < q.P || q.ldy < q.Q || q.ldc < q.Q) return A4R_EINVAL;
        if ((reinterpret_cast<uintptr_t>(q.X) | reinterpret_cast<uintptr_t>(q.Y)) & 15u) return A4R_EINVAL;
        big = big && a4r_tn256_takes(M, q.P, q.Q, dtype);
    }
    if (big) {
        const void* X[4]; const void* Y[4]; float* Cc[4]; float* xs[4]; int ldx[4], ldy[4], ldc[4], P[4], Q[4];
        for (int i = 0; i < n; ++i) {
            X[i] = pr[i].X; Y[i] = pr[i].Y; Cc[i] = pr[i].C; xs[i] = pr[i].xsum;
            ldx[i] = pr[i].ldx; ldy[i] = pr[i].ldy; ldc[i] = pr[i].ldc; P[i] = pr[i].P; Q[i] = pr[i].Q;
        }
        return a4r_tn256_launch_multi(stream, n, X, ldx, Y, ldy, Cc, ldc, P, Q, xs, M);
    }
    for (int i = 0; i < n; ++i) {
        const int rc = a4r_gemm_tn_bias(stream, pr[i].X, pr[i].ldx, pr[i].Y, pr[i].ldy, pr[i].C, pr[i].ldc, M, pr[i].P, pr[i].Q, dtype, pr[i].xsum);
        if (rc != A4R_OK) return rc;
    }
    return A4R_OK;
}

extern "C" int a4r_colsum(void* stream, const void* X, int ldx, float* out, int M, int N, int dtype) {
    if (!X || !out || M <= 0 || N <= 0 || N % 8) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    if ((dtype != A4R_F32 && dtype != A4R_BF16) || (ldx * esz) % 16 || (reinterpret_cast<uintptr_t>(X) & 15u)) return A4R_EINVAL;
    int ncg = N / 8;                          // column groups per block: a power of two <= 32
    if (ncg > 32) ncg = 32;
    while (ncg & (ncg - 1)) ncg &= ncg - 1;
    const int gx = (N / 8 + ncg - 1) / ncg;
    const int nrl = 256 / ncg;
    // two workgroups per CU for wide inputs (4.5 against 2.9 TB/s at [66304, 768]); narrow ones keep one: every block ends in one
    // atomic per column on the same N addresses, and at N = 64 twice the blocks took 15 us against 9
    int gy = 256 * (N >= 512 ? 2 : 1) / gx; if (gy < 1) gy = 1;
    if (gy > (M + nrl - 1) / nrl) gy = (M + nrl - 1) / nrl;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == A4R_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(gx, gy), dim3(256), 0, s, (const bf16_t*)X, ldx, out, M, N, ncg);
    else hipLaunchKernelGGL(colsum_kernel<float>, dim3(gx, gy), dim3(256), 0, s, (const float*)X, ldx, out, M, N, ncg);
    return a4r_launch_status();
}
