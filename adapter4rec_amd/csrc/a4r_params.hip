// Parameter-side kernels of the training step (tiny, launch-latency-bound) that used to be torch eager ops between the HIP
// kernels: the LoRA merge W + B A / r into the packed qkv operand, the Compacter / PHM effective matrices and their three
// gradients, the scatter of zero-padded gradient scratch into the flat gradient buffer, and a zero fill.
//   LoRA      : loralib==0.1.1 lora.Linear as injected at Downstream/Text/run.py:414-428, Downstream/CV/run_adapter.py:384-395
//   Compacter : PHMLinear, Downstream/Text/model/layers.py:25-166 (matvec_product :10-22, kronecker_product_einsum_batched
//               kronecker.py:23-34): E[out, in] = (sum_k kron(rule[k], W_left[k] W_right[k]))^T
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

// ------------------------------------------------------------------ LoRA merge
// dst[o, i] = W[o, i] + s * sum_r B[o, r] A[r, i]   and   dstT[i, o] = the same value (the dgrad operand), compute dtype
template <typename T>
__global__ void __launch_bounds__(256) lora_merge_kernel(const float* __restrict__ W, const float* __restrict__ A, const float* __restrict__ B,
                                                         float s, T* __restrict__ dst, int ld, T* __restrict__ dstT, int ldT,
                                                         int out_f, int in_f, int r) {
    const int total = out_f * in_f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int o = e / in_f, i = e % in_f;
        float acc = 0.f;
        for (int k = 0; k < r; ++k) acc += B[o * r + k] * A[k * in_f + i];
        const float v = W[e] + s * acc;
        Elem<T>::st(dst + (size_t)o * ld + i, v);
        Elem<T>::st(dstT + (size_t)i * ldT + o, v);
    }
}

// every LoRA-carrying projection of the model in ONE launch (blockIdx.y = descriptor): 24 merges of 12 us each were launch latency
struct LoraDesc {     // mirrors a4r_lora_desc_t
    const float* W; const float* A; const float* B; void* dst; void* dstT;
    float s; int32_t ld, ldT, out_f, in_f, r;
};
// 4 consecutive elements as one 8-byte (bf16) / 16-byte (fp32) store when the address allows it
template <typename T> A4R_DEV void store4(T* p, const float (&v)[4]) {
    if constexpr (sizeof(T) == 2) {
        if ((reinterpret_cast<uintptr_t>(p) & 7u) == 0) { *reinterpret_cast<uint2*>(p) = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])); return; }
    } else {
        if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); return; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) Elem<T>::st(p + e, v[e]);
}
template <typename T>
__global__ void __launch_bounds__(256) lora_merge_batch_kernel(const LoraDesc* __restrict__ desc) {
    const LoraDesc d = desc[blockIdx.y];
    T* dst = reinterpret_cast<T*>(d.dst);
    T* dstT = reinterpret_cast<T*>(d.dstT);
    if (d.out_f % 64 == 0 && d.in_f % 64 == 0) {
        // 64 x 64 tiles through LDS (round 4): every global access is a whole 8 / 16-byte piece of a row -- the element-wise form below wrote the
        // transposed copy as 2-byte stores a row apart (590 k of them per 768 x 768 projection: 192 us for the image tower's 24 merges)
        __shared__ float tile[64][65];
        const int tn = d.in_f / 64, ntile = (d.out_f / 64) * tn;
        const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
        for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
            const int o0 = (t / tn) * 64, i0 = (t % tn) * 64;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int o = o0 + ty * 4 + rr, i = i0 + tx * 4;
                const float4 w = *reinterpret_cast<const float4*>(d.W + (size_t)o * d.in_f + i);
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                for (int k = 0; k < d.r; ++k) {
                    const float b = d.B[o * d.r + k];
                    const float4 av = *reinterpret_cast<const float4*>(d.A + (size_t)k * d.in_f + i);
                    a0 += b * av.x; a1 += b * av.y; a2 += b * av.z; a3 += b * av.w;
                }
                const float v[4] = {w.x + d.s * a0, w.y + d.s * a1, w.z + d.s * a2, w.w + d.s * a3};
                store4(dst + (size_t)o * d.ld + i, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) tile[ty * 4 + rr][tx * 4 + e] = v[e];
            }
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = ty * 4 + rr;
                const float v[4] = {tile[tx * 4][i], tile[tx * 4 + 1][i], tile[tx * 4 + 2][i], tile[tx * 4 + 3][i]};
                store4(dstT + (size_t)(i0 + i) * d.ldT + o0 + tx * 4, v);
            }
            __syncthreads();
        }
        return;
    }
    const int total = d.out_f * d.in_f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int o = e / d.in_f, i = e % d.in_f;
        float acc = 0.f;
        for (int k = 0; k < d.r; ++k) acc += d.B[o * d.r + k] * d.A[k * d.in_f + i];
        const float v = d.W[e] + d.s * acc;
        Elem<T>::st(dst + (size_t)o * d.ld + i, v);
        Elem<T>::st(dstT + (size_t)i * d.ldT + o, v);
    }
}

// ------------------------------------------------------------------ PHM (Compacter)
struct PhmDesc {      // mirrors a4r_phm_desc_t
    int64_t rule_off, wl_off, wr_off;     // fp32 offsets into `params`: rule [n, n, n], W_left [n, in/n], W_right [n, out/n]
    int64_t out_off;                      // build: offset of E [out, in] in `eff`;  backward: unused
    const float* G; int32_t ldg;          // backward: dL/dE, row stride ldg (zero-padded scratch of the weight-gradient GEMM)
    int32_t in_f, out_f, n, pad_;
};

// E[b * oq + q][a * ip + p] = sum_k rule[k][a][b] Wl[k][p] Wr[k][q]            gridDim.y workgroups per PHMLinear
__global__ void __launch_bounds__(256) phm_build_kernel(const float* __restrict__ params, const PhmDesc* __restrict__ desc, float* __restrict__ eff) {
    const PhmDesc d = desc[blockIdx.x];
    const int n = d.n, ip = d.in_f / n, oq = d.out_f / n;
    const float* rule = params + d.rule_off;
    const float* wl = params + d.wl_off;
    const float* wr = params + d.wr_off;
    float* E = eff + d.out_off;
    const int total = d.in_f * d.out_f;
    for (int e = blockIdx.y * 256 + threadIdx.x; e < total; e += 256 * gridDim.y) {      // (gridDim.y workgroups share a PHMLinear: one alone took 103 us for 52 of them)
        const int o = e / d.in_f, i = e % d.in_f;
        const int b = o / oq, q = o % oq, a = i / ip, p = i % ip;
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc += rule[(k * n + a) * n + b] * wl[k * ip + p] * wr[k * oq + q];
        E[e] = acc;
    }
}

// d rule[k][a][b] += sum_{p,q} G[b oq + q][a ip + p] Wl[k][p] Wr[k][q]
// d Wl[k][p]      += sum_{a,b,q} G[.][.] rule[k][a][b] Wr[k][q]        d Wr[k][q] += sum_{a,b,p} G[.][.] rule[k][a][b] Wl[k][p]
// through T[a][b][p][k'] := sum_q G[b oq + q][a ip + p] Wr[k'][q]  (LDS-free: recomputed per output; the matrices are <= 768 x 64)
__global__ void __launch_bounds__(256) phm_bwd_kernel(const float* __restrict__ params, const PhmDesc* __restrict__ desc, float* __restrict__ grads) {
    const PhmDesc d = desc[blockIdx.x];
    const int n = d.n, ip = d.in_f / n, oq = d.out_f / n;
    const float* rule = params + d.rule_off;
    const float* wl = params + d.wl_off;
    const float* wr = params + d.wr_off;
    const float* G = d.G;
    const int n_rule = n * n * n, n_wl = n * ip, n_wr = n * oq;
    // ONE WAVE per output, its lanes over the reduction index (gridDim.y workgroups share a PHMLinear's outputs).  One lane per
    // output left the 64 + 64 long reductions (3072 strided terms each) to 128 lanes of the whole chip: 425 us per step on the
    // ViT-MAE + Compacter workload.
    const int lane = threadIdx.x & 63, wave = blockIdx.y * 4 + (threadIdx.x >> 6), nwave = gridDim.y * 4;
    for (int w = wave; w < n_rule + n_wl + n_wr; w += nwave) {
        float acc = 0.f;
        if (w < n_rule) {                                   // sum over (q, p)
            const int k = w / (n * n), a = (w / n) % n, b = w % n;
            for (int t = lane; t < oq * ip; t += 64) {
                const int q = t / ip, p = t % ip;
                acc += G[(size_t)(b * oq + q) * d.ldg + a * ip + p] * wl[k * ip + p] * wr[k * oq + q];
            }
            acc = wave_sum(acc);
            if (lane == 0) atomicAdd(grads + d.rule_off + w, acc);
        } else if (w < n_rule + n_wl) {                     // sum over (a, b, q)
            const int k = (w - n_rule) / ip, p = (w - n_rule) % ip;
            for (int t = lane; t < n * n * oq; t += 64) {
                const int a = t / (n * oq), b = (t / oq) % n, q = t % oq;
                acc += G[(size_t)(b * oq + q) * d.ldg + a * ip + p] * wr[k * oq + q] * rule[(k * n + a) * n + b];
            }
            acc = wave_sum(acc);
            if (lane == 0) atomicAdd(grads + d.wl_off + (w - n_rule), acc);
        } else {                                            // sum over (a, b, p)
            const int k = (w - n_rule - n_wl) / oq, q = (w - n_rule - n_wl) % oq;
            for (int t = lane; t < n * n * ip; t += 64) {
                const int a = t / (n * ip), b = (t / ip) % n, p = t % ip;
                acc += G[(size_t)(b * oq + q) * d.ldg + a * ip + p] * wl[k * ip + p] * rule[(k * n + a) * n + b];
            }
            acc = wave_sum(acc);
            if (lane == 0) atomicAdd(grads + d.wr_off + (w - n_rule - n_wl), acc);
        }
    }
}

// ------------------------------------------------------------------ scratch corner -> flat gradient
struct AddDesc {      // mirrors a4r_add_desc_t
    const float* src; int64_t dst_off; int32_t rows, cols, ld; float alpha;
};
__global__ void __launch_bounds__(256) unpack_add_kernel(float* __restrict__ target, const AddDesc* __restrict__ desc) {
    const AddDesc d = desc[blockIdx.y];
    const int total = d.rows * d.cols;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int r = e / d.cols, c = e % d.cols;
        target[d.dst_off + e] += d.alpha * d.src[(size_t)r * d.ld + c];
    }
}

}  // namespace

extern "C" int a4r_lora_merge(void* stream, const float* W, const float* A, const float* B, float scaling,
                              void* dst, int ld, void* dstT, int ldT, int out_f, int in_f, int r, int dtype) {
    if (!W || !dst || !dstT || out_f <= 0 || in_f <= 0 || r < 0 || (r > 0 && (!A || !B)) || ld < in_f || ldT < out_f) return A4R_EINVAL;
    if (dtype != A4R_BF16 && dtype != A4R_F32) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int grid = (out_f * in_f + 255) / 256; if (grid > 1024) grid = 1024;
    if (dtype == A4R_BF16)
        hipLaunchKernelGGL(lora_merge_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, W, A, B, scaling, (bf16_t*)dst, ld, (bf16_t*)dstT, ldT, out_f, in_f, r);
    else
        hipLaunchKernelGGL(lora_merge_kernel<float>, dim3(grid), dim3(256), 0, s, W, A, B, scaling, (float*)dst, ld, (float*)dstT, ldT, out_f, in_f, r);
    return a4r_launch_status();
}

extern "C" int a4r_lora_merge_batch(void* stream, const a4r_lora_desc_t* desc_dev, int n_desc, int max_elems, int dtype) {
    if (!desc_dev || n_desc <= 0 || max_elems <= 0 || (dtype != A4R_BF16 && dtype != A4R_F32)) return A4R_EINVAL;
    int gx = (max_elems + 255) / 256; if (gx > 256) gx = 256;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const LoraDesc* d = reinterpret_cast<const LoraDesc*>(desc_dev);
    if (dtype == A4R_BF16) hipLaunchKernelGGL(lora_merge_batch_kernel<bf16_t>, dim3(gx, n_desc), dim3(256), 0, s, d);
    else hipLaunchKernelGGL(lora_merge_batch_kernel<float>, dim3(gx, n_desc), dim3(256), 0, s, d);
    return a4r_launch_status();
}

extern "C" int a4r_phm_build(void* stream, const float* params, const a4r_phm_desc_t* desc_dev, int n_desc, float* eff) {
    if (!params || !desc_dev || n_desc <= 0 || !eff) return A4R_EINVAL;
    hipLaunchKernelGGL(phm_build_kernel, dim3(n_desc, 8), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), params,
                       reinterpret_cast<const PhmDesc*>(desc_dev), eff);
    return a4r_launch_status();
}

extern "C" int a4r_phm_bwd(void* stream, const float* params, const a4r_phm_desc_t* desc_dev, int n_desc, float* grads) {
    if (!params || !desc_dev || n_desc <= 0 || !grads) return A4R_EINVAL;
    hipLaunchKernelGGL(phm_bwd_kernel, dim3(n_desc, 56), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), params,      // 896 outputs of a [64, 768] PHMLinear = 4 per wave
                       reinterpret_cast<const PhmDesc*>(desc_dev), grads);
    return a4r_launch_status();
}

extern "C" int a4r_unpack_add(void* stream, float* target, const a4r_add_desc_t* desc_dev, int n_desc, int max_elems) {
    if (!target || !desc_dev || n_desc <= 0 || max_elems <= 0) return A4R_EINVAL;
    int gx = (max_elems + 255) / 256; if (gx > 64) gx = 64;
    hipLaunchKernelGGL(unpack_add_kernel, dim3(gx, n_desc), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), target,
                       reinterpret_cast<const AddDesc*>(desc_dev));
    return a4r_launch_status();
}

extern "C" int a4r_memset_zero(void* stream, void* p, int64_t bytes) {
    if (!p || bytes <= 0) return A4R_EINVAL;
    return hipMemsetAsync(p, 0, (size_t)bytes, reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? A4R_OK : A4R_ELAUNCH;
}
