// ViT / ViT-MAE input side (HF ViTEmbeddings / ViTMAEEmbeddings as used by Downstream/CV/model/encoders.py:21-32):
//   a4r_patchify      image -> rows of flattened 16x16x3 patches (the im2col of the stride-16 patch convolution, so that
//                     the projection is one call of the large-tile GEMM); also the on-GPU half of the image pipeline of
//                     Downstream/CV/data_utils/dataset.py:77-81: uint8 HWC pixels -> ToTensor -> Normalize(0.5, 0.5),
//                     i.e. (x / 255 - 0.5) / 0.5; with keep_idx only the patches ViT-MAE keeps are produced at all.
//   a4r_vit_assemble  [cls + pos[0]] ++ [patch_j + pos[1 + idx_j]] -> token rows of the encoder input.
// Both are pure data movement (HBM-bound): one thread per 16 bytes of output.
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

template <typename T, bool U8>
__global__ void __launch_bounds__(256) patchify_kernel(const void* __restrict__ img, T* __restrict__ out, int ldo,
                                                       const int32_t* __restrict__ keep, int n_keep, int n_items, int C, int Hi, int Wi, int P) {
    constexpr int PER = Elem<T>::PER16;
    const int cols = C * P * P, cpr = cols / PER;                     // P % PER == 0: a chunk never leaves one patch row
    const long total = (long)n_items * n_keep * cpr;
    const int npw = Wi / P;
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
        const int ch = (int)(id % cpr);
        const long row = id / cpr;
        const int item = (int)(row / n_keep), j = (int)(row % n_keep);
        const int patch = keep ? keep[(long)item * n_keep + j] : j;
        const int py = patch / npw, px = patch % npw;
        const int col = ch * PER;                                     // = c * P * P + ky * P + kx (Conv2d weight order)
        const int c = col / (P * P), ky = (col / P) % P, kx = col % P;
        const int y = py * P + ky, x = px * P + kx;
        float v[PER];
        if constexpr (U8) {                                           // [item][y][x][c] bytes
            const unsigned char* s = reinterpret_cast<const unsigned char*>(img) + (((long)item * Hi + y) * Wi + x) * C + c;
#pragma unroll
            for (int e = 0; e < PER; ++e) v[e] = ((float)s[e * C] / 255.f - 0.5f) / 0.5f;   // ToTensor (div 255) then Normalize (sub, div): same roundings
        } else {                                                      // [item][c][y][x] fp32, already normalised
            const float* s = reinterpret_cast<const float*>(img) + (((long)item * C + c) * Hi + y) * Wi + x;
#pragma unroll
            for (int e = 0; e < PER; ++e) v[e] = s[e];
        }
        *reinterpret_cast<uint4*>(out + row * ldo + col) = Elem<T>::pack(v);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) vit_assemble_kernel(const T* __restrict__ patches, int ldp, const float* __restrict__ cls,
                                                           const float* __restrict__ pos, const int32_t* __restrict__ keep,
                                                           T* __restrict__ out, int ldo, int n_items, int n_keep, int H, int S_out) {
    constexpr int PER = Elem<T>::PER16;
    const int cpr = H / PER, S = n_keep + 1;
    const long total = (long)n_items * S * cpr;
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
        const int ch = (int)(id % cpr);
        const long irow = id / cpr;
        const int item = (int)(irow / S), t = (int)(irow % S);
        const long row = (long)item * S_out + t;                     // S_out > S leaves room for appended prompt tokens
        float v[PER], p[PER];
        if (t == 0) {
            load_vec<float, PER>(cls + ch * PER, v);
            load_vec<float, PER>(pos + ch * PER, p);
        } else {
            const int patch = keep ? keep[(long)item * n_keep + t - 1] : t - 1;
            load_vec<T, PER>(patches + ((long)item * n_keep + t - 1) * ldp + ch * PER, v);
            load_vec<float, PER>(pos + (long)(1 + patch) * H + ch * PER, p);
        }
#pragma unroll
        for (int e = 0; e < PER; ++e) v[e] += p[e];
        store_vec<T, PER>(out + row * ldo + ch * PER, v);
    }
}

int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : g > 65536 ? 65536 : g);
}

}  // namespace

extern "C" int a4r_patchify(void* stream, const void* img, int src_kind, void* out, int ldo, const int32_t* keep_idx, int n_keep,
                            int n_items, int C, int Himg, int Wimg, int patch, int dtype) {
    if (!img || !out || n_items <= 0 || C <= 0 || patch <= 0 || Himg % patch || Wimg % patch || patch % 8) return A4R_EINVAL;
    if (src_kind != 0 && src_kind != 1) return A4R_EINVAL;
    const int np = (Himg / patch) * (Wimg / patch);
    if (!keep_idx) n_keep = np;
    if (n_keep <= 0 || n_keep > np || ldo < C * patch * patch) return A4R_EINVAL;
    const int es = dtype == A4R_BF16 ? 2 : 4;
    if ((dtype != A4R_BF16 && dtype != A4R_F32) || (ldo * es) % 16 || (reinterpret_cast<uintptr_t>(out) & 15u)) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long total = (long)n_items * n_keep * (C * patch * patch / (16 / es));
#define A4R_GO(T_, U8_) hipLaunchKernelGGL((patchify_kernel<T_, U8_>), dim3(grid_for(total)), dim3(256), 0, s, img, (T_*)out, ldo, keep_idx, \
                                           n_keep, n_items, C, Himg, Wimg, patch)
    if (dtype == A4R_BF16) { if (src_kind) A4R_GO(bf16_t, true); else A4R_GO(bf16_t, false); }
    else { if (src_kind) A4R_GO(float, true); else A4R_GO(float, false); }
#undef A4R_GO
    return a4r_launch_status();
}

extern "C" int a4r_vit_assemble(void* stream, const void* patches, int ldp, const float* cls, const float* pos, const int32_t* keep_idx,
                                void* out, int ldo, int n_items, int n_keep, int H, int dtype, int tokens_out) {
    if (!patches || !cls || !pos || !out || n_items <= 0 || n_keep <= 0 || H % 8) return A4R_EINVAL;
    if (tokens_out == 0) tokens_out = n_keep + 1;
    if (tokens_out < n_keep + 1) return A4R_EINVAL;
    const int es = dtype == A4R_BF16 ? 2 : 4;
    if ((dtype != A4R_BF16 && dtype != A4R_F32) || (ldp * es) % 16 || (ldo * es) % 16 || ldp < H || ldo < H) return A4R_EINVAL;
    if ((reinterpret_cast<uintptr_t>(patches) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(cls) | reinterpret_cast<uintptr_t>(pos)) & 15u)
        return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long total = (long)n_items * (n_keep + 1) * (H / (16 / es));
    if (dtype == A4R_BF16)
        hipLaunchKernelGGL(vit_assemble_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, s, (const bf16_t*)patches, ldp, cls, pos, keep_idx,
                           (bf16_t*)out, ldo, n_items, n_keep, H, tokens_out);
    else
        hipLaunchKernelGGL(vit_assemble_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, (const float*)patches, ldp, cls, pos, keep_idx,
                           (float*)out, ldo, n_items, n_keep, H, tokens_out);
    return a4r_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// a4r_mae_keep_indices: ViT-MAE random masking (HF ViTMAEEmbeddings.random_masking, reached from Downstream/CV/model/encoders.py:8-22):
// ids_keep = argsort(noise, dim = 1)[:, :n_keep].  One workgroup per item; the row of noise sits in LDS and every patch COUNTS the
// patches that sort in front of it (smaller noise, or equal noise and smaller index = the stable order), which IS its position in the
// argsort -- n_patches^2 / 256 compares per thread (196 patches: 150), no sort network, no scratch.  noise NULL: the noise is drawn here,
// uniform in [0, 1) from the counter hash (seed, site, item * n_patches + patch) -- the training path; an explicit noise tensor (parity
// runs, explicit-noise fixtures) is ranked as given and reproduces a stable argsort bit for bit.
namespace {
constexpr int MAE_MAX_PATCHES = 4096;
__global__ void __launch_bounds__(256) mae_keep_kernel(const float* __restrict__ noise, int32_t* __restrict__ keep, int n_patches, int n_keep,
                                                       uint64_t seed, uint32_t site) {
    __shared__ float v[MAE_MAX_PATCHES];
    const int item = blockIdx.x;
    for (int i = threadIdx.x; i < n_patches; i += 256) {
        if (noise) v[i] = noise[(long)item * n_patches + i];
        else v[i] = (float)(uint32_t)(a4r_hash64(seed, site, (uint64_t)item * (uint64_t)n_patches + (uint64_t)i) >> 40) * (1.f / 16777216.f);   // 24 bits: exact in fp32
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_patches; i += 256) {
        const float x = v[i];
        int rank = 0;
        for (int j = 0; j < n_patches; ++j) {
            const float y = v[j];                                      // (same address for the whole wave: an LDS broadcast)
            rank += (y < x || (y == x && j < i)) ? 1 : 0;
        }
        if (rank < n_keep) keep[(long)item * n_keep + rank] = i;
    }
}
}  // namespace

extern "C" int a4r_mae_keep_indices(void* stream, const float* noise, int32_t* keep, int n_items, int n_patches, int n_keep,
                                    uint64_t seed, uint32_t site) {
    if (!keep || n_items <= 0 || n_patches <= 0 || n_patches > MAE_MAX_PATCHES || n_keep <= 0 || n_keep > n_patches) return A4R_EINVAL;
    hipLaunchKernelGGL(mae_keep_kernel, dim3(n_items), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), noise, keep, n_patches, n_keep, seed, site);
    return a4r_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------
// a4r_resample_u8: one pass of Pillow's fixed-point separable resampler (third party: Pillow src/libImaging/Resample.c,
// ImagingResampleHorizontal_8bpc / Vertical_8bpc), which is what torchvision's Resize((R, R)) runs on the PIL image at
// Downstream/CV/data_utils/dataset.py:77-81.  The coefficient tables come from the host (adapter4rec_amd/cv/image_io.py:
// double-precision triangle weights -> 22-bit fixed point exactly as precompute_coeffs / normalize_coeffs_8bpc).
// out = clip8((2^21 + sum_x in[x0 + x] * kk[x]) >> 22); two passes (horizontal, then vertical on the 8-bit intermediate)
// reproduce Image.resize(..., BILINEAR) bit for bit.  One thread per output byte; HBM-bound.
namespace {
__global__ void __launch_bounds__(256) resample_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                          const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk, int ksize,
                                                          long n_outer, int in_len, int out_len, long inner) {
    // src [n_outer][in_len][inner] -> dst [n_outer][out_len][inner]   (horizontal pass: inner = C; vertical: inner = W * C)
    const long total = n_outer * out_len * inner;
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
        const long i = id % inner;
        const int xx = (int)((id / inner) % out_len);
        const long o = id / (inner * out_len);
        const int x0 = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const unsigned char* s = src + (o * in_len + x0) * inner + i;
        const int32_t* k = kk + (long)xx * ksize;
        int acc = 1 << 21;
        for (int x = 0; x < cnt; ++x) acc += (int)s[(long)x * inner] * k[x];
        acc >>= 22;
        dst[id] = (unsigned char)(acc < 0 ? 0 : acc > 255 ? 255 : acc);
    }
}
}  // namespace

extern "C" int a4r_resample_u8(void* stream, const void* src, void* dst, const int32_t* bounds, const int32_t* kk, int ksize,
                               long n_outer, int in_len, int out_len, long inner) {
    if (!src || !dst || !bounds || !kk || ksize <= 0 || n_outer <= 0 || in_len <= 0 || out_len <= 0 || inner <= 0) return A4R_EINVAL;
    const long total = n_outer * out_len * inner;
    hipLaunchKernelGGL(resample_u8_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       (const unsigned char*)src, (unsigned char*)dst, bounds, kk, ksize, n_outer, in_len, out_len, inner);
    return a4r_launch_status();
}
