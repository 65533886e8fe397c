// a4r_adapter_fwd: the Houlsby / Compacter bottleneck fused with residual + LayerNorm in ONE pass over the tokens
//   zp = h Wd^T + bd ; z = act(zp) ; v = z Wu^T + bu (+ h) + x ; y = LN(v) * gamma + beta
// (reference: BertAdaptedSelfOutput.forward model/model.py:292-297 + AdapterBlock modules.py:130-134, and the Compacter
// form model.py:715-720).  Un-fused this is two skinny GEMMs + a LayerNorm kernel = 4 reads + 2 writes of an [M, H]
// activation; here it is 2 reads (h, x; h is re-read for the residual, mostly from L2 / Infinity Cache) + 2 writes (v, y).
//
// Work decomposition: one wave owns 16 token rows (a workgroup = 4 independent waves, no LDS, no barriers -- M / 16 waves
// keep enough loads in flight to cover HBM latency; staging the weights in LDS capped a CU at one 8-wave workgroup and
// measured 2x slower).  The weight fragments (Wd, Wu: 96 KB each at H = 768) are read straight from L1/L2, which every
// wave shares.  All MFMA products are issued with the WEIGHT fragment as the first operand, so every result tile is
// "transposed": lane (fr = l & 15, kg = l >> 4) holds 4 consecutive columns of token row fr --
//   * v_permlane16_swap pairs neighbouring tiles so a lane holds 8 consecutive columns (16-byte stores, same trick as the
//     GEMM epilogue); the down-projection result in that form is fed back as the B operand of the up-projection WITHOUT
//     any further layout change by permuting the contraction index: k-slot (kg, j) of step s <-> bottleneck index
//     32 s + (kg & 1) * 16 + (kg >> 1) * 8 + j, the weight fragment being read at the same offset;
//   * the LayerNorm statistics are register reductions + two cross-lane adds (lanes l, l^16, l^32, l^48).
// MEASURED (MI355X, M = 40448, H = 768): 161 us against 93 us for the un-fused sequence, with LDS-staged weights (one
// 8-wave workgroup per CU) and without alike; switching phases off one at a time removes time in proportion to the
// vector-memory instructions removed (~70 cycles per 1-KiB wave instruction per CU, ~30 GB/s per CU), i.e. the kernel is
// bound by the CU's vector-memory path, not by HBM, once the weight fragments (2 x 96 KB per 16 rows) go through it.
// The engine therefore keeps the un-fused path (TransRecEngine.fuse_adapters = False); this entry point stays for
// bottlenecks where M is small and launch count matters.
// Algorithmic bytes per token = 4 * H * sizeof(T) (+ 2 * 64 * sizeof(T) for zp, z).
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

template <int H>
__global__ void __launch_bounds__(256) adapter_fwd_kernel(
    const bf16_t* __restrict__ h, int ldh, const bf16_t* __restrict__ x, int ldx,
    const bf16_t* __restrict__ Wd, const float* __restrict__ bd, const bf16_t* __restrict__ Wu, const float* __restrict__ bu,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act, int inner_res,
    bf16_t* __restrict__ zp, bf16_t* __restrict__ z, bf16_t* __restrict__ v, int ldv, bf16_t* __restrict__ y, int ldy,
    float* __restrict__ stats, int M) {
    constexpr int DP = 64;                 // bottleneck width (padded)
    constexpr int NP = H / 32;             // pairs of 16-column tiles of the hidden row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, kg = lane >> 4;
    const size_t row = (size_t)blockIdx.x * 64 + wave * 16 + fr;
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
            const uint4 wf = ldg16(Wd + (size_t)(nt * 16 + fr) * H + ch * 8);
            Mma<bf16_t>::mma(wf, hf, accd[nt]);
        }
    }
    // bias, activation; v_permlane16_swap pairs tiles (2s, 2s+1) so a lane holds 8 CONSECUTIVE bottleneck columns
    // 32 s + cbase .. + 7 of its token: 16-byte stores of zp / z, and the B operand of the up projection as is.
    const int cbase = (kg & 1) * 16 + (kg >> 1) * 8;
    uint4 zf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        uint2 pb[2], ab[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int nt = 2 * s2 + t;
            const float4 b4 = *reinterpret_cast<const float4*>(bd + nt * 16 + kg * 4);
            float p4[4] = {accd[nt][0] + b4.x, accd[nt][1] + b4.y, accd[nt][2] + b4.z, accd[nt][3] + b4.w};
            pb[t] = pack4(p4);
#pragma unroll
            for (int i = 0; i < 4; ++i) p4[i] = act_fwd(p4[i], act);
            ab[t] = pack4(p4);
        }
        const auto px = __builtin_amdgcn_permlane16_swap(pb[0].x, pb[1].x, false, false);
        const auto py = __builtin_amdgcn_permlane16_swap(pb[0].y, pb[1].y, false, false);
        const auto ax = __builtin_amdgcn_permlane16_swap(ab[0].x, ab[1].x, false, false);
        const auto ay = __builtin_amdgcn_permlane16_swap(ab[0].y, ab[1].y, false, false);
        *reinterpret_cast<uint4*>(zp + row * DP + 32 * s2 + cbase) = make_uint4(px[0], py[0], px[1], py[1]);
        zf[s2] = make_uint4(ax[0], ay[0], ax[1], ay[1]);
        *reinterpret_cast<uint4*>(z + row * DP + 32 * s2 + cbase) = zf[s2];
    }
    // ---- up projection, one PAIR of 16-column tiles at a time, contraction index permuted to match zf (k-slot (kg, j) of
    // step s <-> bottleneck 32 s + cbase + j, the weight fragment read with the same offset).  Tile result
    // c[r] = up[row][nt*16 + kg*4 + r]; the same permlane pairing gives 8 consecutive hidden columns per lane; bias +
    // residuals are added, v is stored (bf16) and the row statistics are accumulated from the STORED values (what a
    // separate LayerNorm kernel would read).  Holding the whole row (H/4 fp32 per lane) in registers instead spilled.
    float s1 = 0.f, sq = 0.f;
#pragma unroll 2
    for (int p = 0; p < NP; ++p) {
        f32x4_t c2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16_t* wrow = Wu + (size_t)((2 * p + t) * 16 + fr) * DP + cbase;
            f32x4_t c = {0.f, 0.f, 0.f, 0.f};
            Mma<bf16_t>::mma(ldg16(wrow), zf[0], c);
            Mma<bf16_t>::mma(ldg16(wrow + 32), zf[1], c);
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
        for (int i = 0; i < 8; ++i) { s1 += vv[i]; sq += vv[i] * vv[i]; }
    }
    s1 += __shfl_xor(s1, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    sq += __shfl_xor(sq, 16, 64);
    sq += __shfl_xor(sq, 32, 64);
    const float mean = s1 * (1.f / H);
    const float var = fmaxf(sq * (1.f / H) - mean * mean, 0.f);
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

template <int H>
int launch_fwd(hipStream_t s, const bf16_t* h, int ldh, const bf16_t* x, int ldx, const bf16_t* Wd, const float* bd, const bf16_t* Wu,
               const float* bu, const float* gamma, const float* beta, float eps, int act, int inner_res, bf16_t* zp, bf16_t* z,
               bf16_t* v, int ldv, bf16_t* y, int ldy, float* stats, int M) {
    hipLaunchKernelGGL(adapter_fwd_kernel<H>, dim3(M / 64), dim3(256), 0, s, h, ldh, x, ldx, Wd, bd, Wu, bu, gamma, beta, eps, act, inner_res,
                       zp, z, v, ldv, y, ldy, stats, M);
    return a4r_launch_status();
}

}  // namespace

extern "C" int a4r_adapter_fwd(void* stream, const void* h, int ldh, const void* x, int ldx, const void* Wd, const float* bd,
                               const void* Wu, const float* bu, const float* gamma, const float* beta, float eps, int act,
                               int inner_residual, void* zp, void* z, void* v, int ldv, void* y, int ldy, float* stats,
                               int M, int H, int dp, int dtype) {
    if (!h || !x || !Wd || !bd || !Wu || !bu || !gamma || !beta || !zp || !z || !v || !y || !stats) return A4R_EINVAL;
    if (dtype != A4R_BF16 || dp != 64 || M <= 0 || M % 64) return A4R_EINVAL;
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
