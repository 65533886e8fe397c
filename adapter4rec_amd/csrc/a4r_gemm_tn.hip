// a4r_gemm_tn: C[P,Q] (fp32, +=) = X[M,P]^T . Y[M,Q]  -- the weight gradients of the trainable
// adapter matrices.  The contraction runs over ROWS (tokens) of both operands, so both MFMA
// operands are gathered down the columns of row-major LDS tiles (gather_chunk).  M is split over
// the grid; each workgroup keeps a 64x64 fp32 partial in registers for its whole token range and
// flushes it once with fp32 atomics (4 waves x 2x2 tiles of 16x16).
// a4r_colsum: out[N] += sum_m X[m, :]  (bias gradients).
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
        for (int m = blockIdx.y * nrl + rlane; m < M; m += gridDim.y * nrl) {
            float v[8];
            load_vec<T, 8>(X + (size_t)m * ldx + cg * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += v[e];
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

extern "C" int a4r_gemm_tn(void* stream, const void* X, int ldx, const void* Y, int ldy, float* C, int ldc,
                           int M, int P, int Q, int dtype) {
    if (!X || !Y || !C || M <= 0 || P <= 0 || Q <= 0 || M % 64 || P % 64 || Q % 64) return A4R_EINVAL;
    const int esz = dtype == A4R_F32 ? 4 : 2;
    if ((dtype != A4R_F32 && dtype != A4R_BF16) || (ldx * esz) % 16 || (ldy * esz) % 16 || ldx < P || ldy < Q || ldc < Q) return A4R_EINVAL;
    if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y)) & 15u) return A4R_EINVAL;
    const int ntp = P / 64, ntq = Q / 64, tiles = ntp * ntq;
    int splits = (512 + tiles - 1) / tiles;                 // ~2 workgroups per CU
    const int stages = M / 64;
    if (splits > stages) splits = stages;
    const int rows_per_split = ((stages + splits - 1) / splits) * 64;
    splits = (M + rows_per_split - 1) / rows_per_split;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == A4R_BF16)
        hipLaunchKernelGGL(gemm_tn_kernel<bf16_t>, dim3(tiles, splits), dim3(256), 0, s, (const bf16_t*)X, ldx, (const bf16_t*)Y, ldy,
                           C, ldc, M, ntq, rows_per_split);
    else
        hipLaunchKernelGGL(gemm_tn_kernel<float>, dim3(tiles, splits), dim3(256), 0, s, (const float*)X, ldx, (const float*)Y, ldy,
                           C, ldc, M, ntq, rows_per_split);
    return a4r_launch_status();
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
    int gy = 256 / gx; if (gy < 1) gy = 1;
    if (gy > (M + nrl - 1) / nrl) gy = (M + nrl - 1) / nrl;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == A4R_BF16) hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3(gx, gy), dim3(256), 0, s, (const bf16_t*)X, ldx, out, M, N, ncg);
    else hipLaunchKernelGGL(colsum_kernel<float>, dim3(gx, gy), dim3(256), 0, s, (const float*)X, ldx, out, M, N, ncg);
    return a4r_launch_status();
}
