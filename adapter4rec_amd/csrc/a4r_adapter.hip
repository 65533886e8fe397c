// a4r_adapter_fwd: the Houlsby / Compacter bottleneck fused with residual + LayerNorm in ONE pass over the tokens
//   zp = h Wd^T + bd ; z = act(zp) ; v = z Wu^T + bu (+ h) + x ; y = LN(v) * gamma + beta
// (reference: BertAdaptedSelfOutput.forward model/model.py:292-297 + AdapterBlock modules.py:130-134, and the Compacter
// form model.py:715-720).  Un-fused this is two skinny GEMMs + a LayerNorm kernel = 4 reads + 2 writes of an [M, H]
// activation; here it is 2 reads (h, x; h is re-read for the residual, mostly from L2 / Infinity Cache) + 2 writes (v, y).
//
// Work decomposition: one wave owns 16 token rows, a workgroup = 8 waves = 128 rows, Wd then Wu are staged (swizzled) in
// one LDS region shared by the 8 waves.  All MFMA products are issued with the WEIGHT fragment as the first operand, so
// every result tile is "transposed": lane (fr = l & 15, kg = l >> 4) holds 4 consecutive columns of token row fr --
//   * the down-projection result z[fr][nt*16 + kg*4 + 0..3] is fed back as the B operand of the up-projection WITHOUT any
//     layout change, by permuting the contraction index: k-slot (kg, j) of step s <-> bottleneck index
//     32 s + (j >> 2) * 16 + kg * 4 + (j & 3); the weight fragment is read with the same permutation (two ds_read_b64);
//   * after the up-projection a lane holds a quarter of its token's whole hidden row, so the LayerNorm statistics are two
//     register reductions + two cross-lane adds (lanes l, l^16, l^32, l^48), and v_permlane16_swap pairs neighbouring
//     tiles into 16-byte stores (same trick as the GEMM epilogue).
// HBM-bound: algorithmic bytes per token = 4 * H * sizeof(T) (+ 2 * 64 * sizeof(T) for zp, z).
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

template <typename T> A4R_DEV uint4 ldg16(const T* p) { return *reinterpret_cast<const uint4*>(p); }

// 4 consecutive elements (8 bytes of bf16) <-> fp32
A4R_DEV void unpack4(const uint2& v, float* o) {
    o[0] = bf16_bits_to_f32(v.x & 0xffffu); o[1] = bf16_bits_to_f32(v.x >> 16);
    o[2] = bf16_bits_to_f32(v.y & 0xffffu); o[3] = bf16_bits_to_f32(v.y >> 16);
}
A4R_DEV uint2 pack4(const float* o) {
    return make_uint2(f32_to_bf16_bits(o[0]) | (f32_to_bf16_bits(o[1]) << 16), f32_to_bf16_bits(o[2]) | (f32_to_bf16_bits(o[3]) << 16));
}

// stage a row-major [rows][cols] bf16 matrix into LDS with 16-byte chunks XOR-swizzled by `swz(row)`
template <int COLS, typename F>
A4R_DEV void stage_matrix(char* lds, const bf16_t* __restrict__ src, int rows, int tid, int nthreads, F swz) {
    constexpr int CPR = COLS / 8;
    for (int id = tid; id < rows * CPR; id += nthreads) {
        const int r = id / CPR, c = id % CPR;
        *reinterpret_cast<uint4*>(lds + (size_t)r * COLS * 2 + ((c ^ swz(r)) << 4)) = ldg16(src + (size_t)r * COLS + c * 8);
    }
}

template <int H>
__global__ void __launch_bounds__(512, 2) adapter_fwd_kernel(
    const bf16_t* __restrict__ h, int ldh, const bf16_t* __restrict__ x, int ldx,
    const bf16_t* __restrict__ Wd, const float* __restrict__ bd, const bf16_t* __restrict__ Wu, const float* __restrict__ bu,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act, int inner_res,
    bf16_t* __restrict__ zp, bf16_t* __restrict__ z, bf16_t* __restrict__ v, int ldv, bf16_t* __restrict__ y, int ldy,
    float* __restrict__ stats, int M) {
    constexpr int DP = 64;                 // bottleneck width (padded)
    constexpr int NT = H / 16;             // 16-column tiles of the hidden row
    constexpr int NP = NT / 2;             // tile pairs (32 columns)
    __shared__ __attribute__((aligned(16))) char lds[DP * H * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, kg = lane >> 4;

    for (int blk = blockIdx.x; blk * 128 < M; blk += gridDim.x) {
        const size_t row = (size_t)blk * 128 + wave * 16 + fr;
        // ---- Wd [64][H] -> LDS, chunk ^ (row & 15): 16 fragment rows of one chunk column hit 16 distinct 16-byte slots
        __syncthreads();
        stage_matrix<H>(lds, Wd, DP, tid, 512, [](int r) { return r & 15; });
        __syncthreads();
        // ---- down projection (transposed tiles): accd[nt][r] = zp[row][nt*16 + kg*4 + r]
        f32x4_t accd[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) accd[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const bf16_t* hrow = h + row * ldh;
#pragma unroll 4
        for (int ks = 0; ks < H / 32; ++ks) {
            const int ch = ks * 4 + kg;
            const uint4 hf = ldg16(hrow + ch * 8);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const uint4 wf = *reinterpret_cast<const uint4*>(lds + (size_t)(nt * 16 + fr) * H * 2 + ((ch ^ fr) << 4));
                Mma<bf16_t>::mma(wf, hf, accd[nt]);
            }
        }
        uint2 zb[4];                       // z as packed bf16: the B operand of the up projection
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int c = nt * 16 + kg * 4;
            const float4 b4 = *reinterpret_cast<const float4*>(bd + c);
            float p4[4] = {accd[nt][0] + b4.x, accd[nt][1] + b4.y, accd[nt][2] + b4.z, accd[nt][3] + b4.w};
            *reinterpret_cast<uint2*>(zp + row * DP + c) = pack4(p4);
#pragma unroll
            for (int i = 0; i < 4; ++i) p4[i] = act_fwd(p4[i], act);
            zb[nt] = pack4(p4);
            *reinterpret_cast<uint2*>(z + row * DP + c) = zb[nt];
        }
        // ---- Wu [H][64] -> LDS (128-byte rows), chunk ^ ((row >> 1) & 7)
        __syncthreads();
        stage_matrix<DP>(lds, Wu, H, tid, 512, [](int r) { return (r >> 1) & 7; });
        __syncthreads();
        // ---- up projection with the permuted contraction index (see header), one PAIR of 16-column tiles at a time:
        // tile result c[r] = up[row][nt*16 + kg*4 + r]; v_permlane16_swap pairs the two tiles into 8 consecutive columns per
        // lane; bias + residuals are added, v is stored (bf16) and the row statistics are accumulated from the STORED values
        // (what a separate LayerNorm kernel would read).  Holding the whole row (192 fp32) in registers instead spilled.
        const uint4 zf0 = make_uint4(zb[0].x, zb[0].y, zb[1].x, zb[1].y);     // step 0: bottleneck 0..31
        const uint4 zf1 = make_uint4(zb[2].x, zb[2].y, zb[3].x, zb[3].y);     // step 1: bottleneck 32..63
        const int cbase = (kg & 1) * 16 + (kg >> 1) * 8;                      // + 32 * pair
        const int hb = (kg & 1) * 8, cq = kg >> 1;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 2
        for (int p = 0; p < NP; ++p) {
            f32x4_t c2[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int n = (2 * p + t) * 16 + fr;
                const char* wrow = lds + n * 128;
                const int sw = (n >> 1) & 7;
                // bytes of Wu[n][32 s + kg*4 .. +3]: chunk 4 s + (kg >> 1), half (kg & 1); the same + 16 columns: chunk + 2
                const uint2 a0 = *reinterpret_cast<const uint2*>(wrow + ((cq) ^ sw) * 16 + hb);
                const uint2 a1 = *reinterpret_cast<const uint2*>(wrow + ((cq + 2) ^ sw) * 16 + hb);
                const uint2 a2 = *reinterpret_cast<const uint2*>(wrow + ((cq + 4) ^ sw) * 16 + hb);
                const uint2 a3 = *reinterpret_cast<const uint2*>(wrow + ((cq + 6) ^ sw) * 16 + hb);
                f32x4_t c = {0.f, 0.f, 0.f, 0.f};
                Mma<bf16_t>::mma(make_uint4(a0.x, a0.y, a1.x, a1.y), zf0, c);
                Mma<bf16_t>::mma(make_uint4(a2.x, a2.y, a3.x, a3.y), zf1, c);
                c2[t] = c;
            }
            float vv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(c2[0][r]), __float_as_uint(c2[1][r]), false, false);
                vv[r] = __uint_as_float(sw[0]);
                vv[4 + r] = __uint_as_float(sw[1]);
            }
            const int c = p * 32 + cbase;
            float t8[8];
            load_vec<float, 8>(bu + c, t8);
#pragma unroll
            for (int i = 0; i < 8; ++i) vv[i] += t8[i];
            load_vec<bf16_t, 8>(x + row * ldx + c, t8);
#pragma unroll
            for (int i = 0; i < 8; ++i) vv[i] += t8[i];
            if (inner_res) {
                load_vec<bf16_t, 8>(hrow + c, t8);
#pragma unroll
                for (int i = 0; i < 8; ++i) vv[i] += t8[i];
            }
            const uint4 packed = Elem<bf16_t>::pack(vv);
            *reinterpret_cast<uint4*>(v + row * ldv + c) = packed;
            Elem<bf16_t>::unpack(packed, vv);
#pragma unroll
            for (int i = 0; i < 8; ++i) { s1 += vv[i]; s2 += vv[i] * vv[i]; }
        }
        s1 += __shfl_xor(s1, 16, 64);
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float mean = s1 * (1.f / H);
        const float var = fmaxf(s2 * (1.f / H) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        if (kg == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
        // ---- second sweep: the lane re-reads its own 16-byte pieces of v (L1/L2-resident) and writes y
#pragma unroll 4
        for (int p = 0; p < NP; ++p) {
            const int c = p * 32 + cbase;
            float vv[8], g8[8], b8[8];
            load_vec<bf16_t, 8>(v + row * ldv + c, vv);
            load_vec<float, 8>(gamma + c, g8);
            load_vec<float, 8>(beta + c, b8);
#pragma unroll
            for (int i = 0; i < 8; ++i) vv[i] = (vv[i] - mean) * rstd * g8[i] + b8[i];
            store_vec<bf16_t, 8>(y + row * ldy + c, vv);
        }
    }
}

template <int H>
int launch_fwd(hipStream_t s, const bf16_t* h, int ldh, const bf16_t* x, int ldx, const bf16_t* Wd, const float* bd, const bf16_t* Wu,
               const float* bu, const float* gamma, const float* beta, float eps, int act, int inner_res, bf16_t* zp, bf16_t* z,
               bf16_t* v, int ldv, bf16_t* y, int ldy, float* stats, int M) {
    int grid = M / 128;
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(adapter_fwd_kernel<H>, dim3(grid), dim3(512), 0, s, h, ldh, x, ldx, Wd, bd, Wu, bu, gamma, beta, eps, act, inner_res,
                       zp, z, v, ldv, y, ldy, stats, M);
    return a4r_launch_status();
}

}  // namespace

extern "C" int a4r_adapter_fwd(void* stream, const void* h, int ldh, const void* x, int ldx, const void* Wd, const float* bd,
                               const void* Wu, const float* bu, const float* gamma, const float* beta, float eps, int act,
                               int inner_residual, void* zp, void* z, void* v, int ldv, void* y, int ldy, float* stats,
                               int M, int H, int dp, int dtype) {
    if (!h || !x || !Wd || !bd || !Wu || !bu || !gamma || !beta || !zp || !z || !v || !y || !stats) return A4R_EINVAL;
    if (dtype != A4R_BF16 || dp != 64 || M <= 0 || M % 128) return A4R_EINVAL;
    if ((ldh * 2) % 16 || (ldx * 2) % 16 || (ldv * 2) % 16 || (ldy * 2) % 16 || ldh < H || ldx < H || ldv < H || ldy < H) return A4R_EINVAL;
    const uintptr_t al = reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(Wd) |
                         reinterpret_cast<uintptr_t>(Wu) | reinterpret_cast<uintptr_t>(bd) | reinterpret_cast<uintptr_t>(bu) |
                         reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta) | reinterpret_cast<uintptr_t>(zp) |
                         reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(y);
    if (al & 15u) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define A4R_CASE(H_)                                                                                                              \
    case H_: return launch_fwd<H_>(s, (const bf16_t*)h, ldh, (const bf16_t*)x, ldx, (const bf16_t*)Wd, bd, (const bf16_t*)Wu, bu, \
                                   gamma, beta, eps, act, inner_residual, (bf16_t*)zp, (bf16_t*)z, (bf16_t*)v, ldv, (bf16_t*)y,  \
                                   ldy, stats, M);
    switch (H) {
        A4R_CASE(128) A4R_CASE(256) A4R_CASE(512) A4R_CASE(768) A4R_CASE(1024)
        default: return A4R_EINVAL;
    }
#undef A4R_CASE
}
