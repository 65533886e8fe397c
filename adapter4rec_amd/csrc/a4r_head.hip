// Scoring head + BCE loss (model/model.py:53-68, ModelCPC :118-133), Adam, parameter re-packing.
// All fp32; these are tiny, launch-latency-bound kernels.
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

A4R_DEV float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
A4R_DEV float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// one wave per (b, t), t in [0, L-1): pos = prec[b,t] . emb[b,t+1,0], neg = prec[b,t] . emb[b,t,1]
__global__ void __launch_bounds__(256) score_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ prec,
                                                        const float* __restrict__ log_mask, float* __restrict__ pos,
                                                        float* __restrict__ neg, float* __restrict__ ws, int B, int L, int E, int cpc) {
    const int lane = threadIdx.x & 63;
    const int T = L - 1;
    float lsum = 0.f, lcnt = 0.f;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < B * T; r += gridDim.x * 4) {
        const int b = r / T, t = r % T;
        const float* pv = prec + (size_t)r * E;
        const float* tp = emb + (((size_t)b * L + t + 1) * 2 + 0) * E;
        const float* tn = emb + (((size_t)b * L + t) * 2 + 1) * E;
        float sp = 0.f, sn = 0.f;
        for (int e = lane; e < E; e += 64) { sp += pv[e] * tp[e]; sn += pv[e] * tn[e]; }
        sp = wave_sum(sp);
        sn = wave_sum(sn);
        if (lane == 0) {
            pos[r] = sp;
            neg[r] = sn;
            const bool valid = cpc ? (t == T - 1) : (log_mask[r] != 0.f);
            if (valid) { lsum += softplusf(-sp) + softplusf(sn); lcnt += 1.f; }
        }
    }
    if (lane == 0 && lcnt > 0.f) { atomicAdd(ws + 1, lsum); atomicAdd(ws + 2, lcnt); }
}
__global__ void loss_finalize_kernel(float* ws) { ws[0] = ws[1] / ws[2]; }

// one wave per (b, l): d_emb[b,l,0] = dpos(t=l-1) * prec[b,l-1];  d_emb[b,l,1] = dneg(t=l) * prec[b,l];
// and per (b, t): d_prec[b,t] = dpos * emb[b,t+1,0] + dneg * emb[b,t,1]
__global__ void __launch_bounds__(256) score_bwd_kernel(const float* __restrict__ emb, const float* __restrict__ prec,
                                                        const float* __restrict__ log_mask, const float* __restrict__ pos,
                                                        const float* __restrict__ neg, const float* __restrict__ ws, float loss_scale,
                                                        const float* __restrict__ loss_scale_dev, float* __restrict__ d_prec, float* __restrict__ d_emb, int B, int L, int E, int cpc) {
    const int lane = threadIdx.x & 63;
    const int T = L - 1;
    const float g = loss_scale * (loss_scale_dev ? loss_scale_dev[0] : 1.f) / ws[2];
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < B * L; r += gridDim.x * 4) {
        const int b = r / L, l = r % L;
        float dpos_prev = 0.f, dneg_here = 0.f, dpos_here = 0.f;
        if (l >= 1) {
            const int t = l - 1, i = b * T + t;
            const bool valid = cpc ? (t == T - 1) : (log_mask[i] != 0.f);
            if (valid) dpos_prev = (sigmoidf(pos[i]) - 1.f) * g;
        }
        if (l < T) {
            const int i = b * T + l;
            const bool valid = cpc ? (l == T - 1) : (log_mask[i] != 0.f);
            if (valid) { dneg_here = sigmoidf(neg[i]) * g; dpos_here = (sigmoidf(pos[i]) - 1.f) * g; }
        }
        float* de0 = d_emb + (((size_t)b * L + l) * 2 + 0) * E;
        float* de1 = d_emb + (((size_t)b * L + l) * 2 + 1) * E;
        for (int e = lane; e < E; e += 64) {
            de0[e] = (l >= 1) ? dpos_prev * prec[((size_t)b * T + l - 1) * E + e] : 0.f;
            de1[e] = (l < T) ? dneg_here * prec[((size_t)b * T + l) * E + e] : 0.f;
            if (l < T)
                d_prec[((size_t)b * T + l) * E + e] = dpos_here * emb[(((size_t)b * L + l + 1) * 2 + 0) * E + e] +
                                                      dneg_here * emb[(((size_t)b * L + l) * 2 + 1) * E + e];
        }
    }
}

__global__ void __launch_bounds__(256) emb_grad_add_inputs_kernel(const float* __restrict__ d_in, int ldi, float* __restrict__ d_emb,
                                                                  int B, int L, int E) {
    const int T = L - 1;
    const size_t total = (size_t)B * T * E;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int e = (int)(i % E);
        const size_t r = i / E;
        const int b = (int)(r / T), l = (int)(r % T);
        d_emb[(((size_t)b * L + l) * 2 + 0) * E + e] += d_in[r * ldi + e];
    }
}
__global__ void __launch_bounds__(256) take_inputs_kernel(const float* __restrict__ emb, float* __restrict__ out, int ldo, int B, int L, int E) {
    const int T = L - 1;
    const size_t total = (size_t)B * T * E;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int e = (int)(i % E);
        const size_t r = i / E;
        const int b = (int)(r / T), l = (int)(r % T);
        out[r * ldo + e] = emb[(((size_t)b * L + l) * 2 + 0) * E + e];
    }
}

// ------------------------------------------------------------------ Adam over a flat buffer
// One Adam update (torch.optim.Adam: p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)), shared by the scalar and the four-per-lane kernel; no
// fused multiply-adds, so that both kernels round every product the same way (they must agree bit for bit: tests/test_kernels_gpu.py).
A4R_DEV void adam_one(float& p, float g, float& m, float& v, float lr, float bc1, float bc2_sqrt, float beta1, float beta2, float eps, float grad_scale) {
#pragma clang fp contract(off)
    const float gi = g * grad_scale;
    const float mi = beta1 * m + (1.f - beta1) * gi;
    const float vi = beta2 * v + (1.f - beta2) * gi * gi;
    m = mi;
    v = vi;
    p -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, const int32_t* __restrict__ seg_end,
                                                   const int32_t* __restrict__ seg_group, int n_seg, const float* __restrict__ group_lr,
                                                   float bc1, float bc2_sqrt, float beta1, float beta2, float eps, float grad_scale, int64_t base) {
    for (int64_t i = base + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int lo = 0, hi = n_seg - 1;            // first segment with seg_end > i
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (seg_end[mid] > i) hi = mid; else lo = mid + 1;
        }
        const float lr = group_lr[seg_group[lo]];
        float pi = p[i], mi = m[i], vi = v[i];
        adam_one(pi, g[i], mi, vi, lr, bc1, bc2_sqrt, beta1, beta2, eps, grad_scale);
        m[i] = mi;
        v[i] = vi;
        p[i] = pi;
    }
}

// Four parameters per lane and trip (16-byte accesses): full fine-tuning steps ~110 M parameters -- 3.1 GB of state per step; the scalar form
// above moved it at 2.5 TB/s (1.2 ms).  Same arithmetic per element, so the results are bit-identical.  The segment (-> lr group) is looked up once
// per quad and per element only where a quad straddles a segment boundary.
__global__ void __launch_bounds__(256) adam4_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n4, const int32_t* __restrict__ seg_end,
                                                    const int32_t* __restrict__ seg_group, int n_seg, const float* __restrict__ group_lr,
                                                    float bc1, float bc2_sqrt, float beta1, float beta2, float eps, float grad_scale) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += (int64_t)gridDim.x * 256) {
        const int64_t i = q * 4;
        int lo = 0, hi = n_seg - 1;            // first segment with seg_end > i
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (seg_end[mid] > i) hi = mid; else lo = mid + 1;
        }
        float lr[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            while (lo < n_seg - 1 && seg_end[lo] <= i + e) ++lo;
            lr[e] = group_lr[seg_group[lo]];
        }
        const float4 g4 = reinterpret_cast<const float4*>(g)[q];
        float4 m4 = reinterpret_cast<float4*>(m)[q], v4 = reinterpret_cast<float4*>(v)[q], p4 = reinterpret_cast<float4*>(p)[q];
        const float gs[4] = {g4.x, g4.y, g4.z, g4.w};
        float ms[4] = {m4.x, m4.y, m4.z, m4.w}, vs[4] = {v4.x, v4.y, v4.z, v4.w}, ps[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) adam_one(ps[e], gs[e], ms[e], vs[e], lr[e], bc1, bc2_sqrt, beta1, beta2, eps, grad_scale);
        reinterpret_cast<float4*>(m)[q] = make_float4(ms[0], ms[1], ms[2], ms[3]);
        reinterpret_cast<float4*>(v)[q] = make_float4(vs[0], vs[1], vs[2], vs[3]);
        reinterpret_cast<float4*>(p)[q] = make_float4(ps[0], ps[1], ps[2], ps[3]);
    }
}

struct PackDesc {   // mirrors a4r_pack_desc_t
    int64_t src_off; void* dst; int32_t rows, cols, rows_pad, cols_pad, transpose, dst_ld;
};
// Where element (r, c) of the [rows_pad, cols_pad] destination goes.  layout 0: row-major with stride ld.  Layouts 1 / 2 (a4r_pack_desc_t.transpose bits
// 1-2): the MFMA FRAGMENT order of the one-launch adapter kernels (a4r_adapter_fused.hip), H = NW x CW columns per row of activations, KS = CW / 32:
//   1: a [64, H] matrix (fc_down.weight, fc_up.weight^T): 16 bytes of lane (kg, fr) in fragment (wave w, step s, row tile nt)
//      = row 16 nt + fr, columns w CW + 32 s + 8 kg + [0, 8)                                  -> ((((w KS + s) 4 + nt) 64 + 16 kg + fr) 8 + j
//   2: an [H, 64] matrix (fc_up.weight, fc_down.weight^T): fragment (w, s, half h, step ks) = row w CW + 32 s + 8 (fr >> 2) + 4 h + (fr & 3),
//      columns 32 ks + 8 kg + [0, 8)                                                          -> (((((w KS + s) 2 + h) 2 + ks) 64 + 16 kg + fr) 8 + j
// so that every wave instruction of those kernels' prologues reads 1 KiB contiguous.
A4R_DEV size_t pack_dst_index(int layout, int r, int c, int rows_pad, int cols_pad, int ld) {
    if (layout == 0) return (size_t)r * ld + c;
    const int Hh = layout == 1 ? cols_pad : rows_pad, NW = Hh == 128 ? 4 : 8, CW = Hh / NW, KS = CW / 32;
    if (layout == 1) {
        const int w = c / CW, cc = c % CW, s_ = cc >> 5, kg = (cc & 31) >> 3, j = cc & 7, nt = r >> 4, fr = r & 15;
        return (size_t)((((w * KS + s_) * 4 + nt) * 64 + kg * 16 + fr) * 8 + j);
    }
    const int w = r / CW, rr = r % CW, s_ = rr >> 5, q = rr & 31, fr = (q >> 3) * 4 + (q & 3), h = (q >> 2) & 1, ks = c >> 5, kg = (c & 31) >> 3, j = c & 7;
    return (size_t)(((((w * KS + s_) * 2 + h) * 2 + ks) * 64 + kg * 16 + fr) * 8 + j);
}
template <typename T>
__global__ void __launch_bounds__(256) pack_kernel(const float* __restrict__ flat, const PackDesc* __restrict__ desc) {
    const PackDesc d = desc[blockIdx.y];
    T* dst = reinterpret_cast<T*>(d.dst);
    const int total = d.rows_pad * d.cols_pad;
    const int ld = d.dst_ld ? d.dst_ld : d.cols_pad;          // dst may be a column block of a wider matrix (fused q|k|v operand)
    const int tr = d.transpose & 1, layout = d.transpose >> 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int r = i / d.cols_pad, c = i % d.cols_pad;     // destination coordinates
        const int sr = tr ? c : r, sc = tr ? r : c;
        float val = 0.f;
        if (sr < d.rows && sc < d.cols) val = flat[d.src_off + (int64_t)sr * d.cols + sc];
        Elem<T>::st(dst + pack_dst_index(layout, r, c, d.rows_pad, d.cols_pad, ld), val);
    }
}

// Large matrices (full fine-tuning re-packs every backbone weight and its transpose each step: 440 MB read, 2 x 220 MB written): 64 x 64 destination
// tiles through LDS so that BOTH sides move whole 128 / 256-byte row pieces -- the element-per-lane form above reads a transposed source a row apart
// per lane (1.2 TB/s over the whole pack).  Same values (a copy with one rounding), so the results are bit-identical.
template <typename T>
__global__ void __launch_bounds__(256) pack_tiled_kernel(const float* __restrict__ flat, const PackDesc* __restrict__ desc) {
    __shared__ float tile[64][65];
    const PackDesc d = desc[blockIdx.y];
    T* dst = reinterpret_cast<T*>(d.dst);
    const int ld = d.dst_ld ? d.dst_ld : d.cols_pad;
    const int tr = (d.rows_pad + 63) >> 6, tc = (d.cols_pad + 63) >> 6;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 columns x 4 row lanes
    const int layout = d.transpose >> 1;                             // (fragment-ordered adapter copies in a list that also holds large matrices)
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
        const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;            // destination tile
        if (d.transpose & 1) {                                           // dst[r, c] = src[c, r]: read source rows c0 .. c0 + 63, columns r0 .. r0 + 63
            __syncthreads();
#pragma unroll 4
            for (int k = ty; k < 64; k += 4) {
                const int sr = c0 + k, sc = r0 + tx;
                tile[k][tx] = (sr < d.rows && sc < d.cols) ? flat[d.src_off + (int64_t)sr * d.cols + sc] : 0.f;
            }
            __syncthreads();
#pragma unroll 4
            for (int k = ty; k < 64; k += 4) {
                const int r = r0 + k, c = c0 + tx;
                if (r < d.rows_pad && c < d.cols_pad) Elem<T>::st(dst + pack_dst_index(layout, r, c, d.rows_pad, d.cols_pad, ld), tile[tx][k]);
            }
        } else {
#pragma unroll 4
            for (int k = ty; k < 64; k += 4) {
                const int r = r0 + k, c = c0 + tx;
                if (r < d.rows_pad && c < d.cols_pad)
                    Elem<T>::st(dst + pack_dst_index(layout, r, c, d.rows_pad, d.cols_pad, ld), (r < d.rows && c < d.cols) ? flat[d.src_off + (int64_t)r * d.cols + c] : 0.f);
            }
        }
    }
}

}  // namespace

extern "C" int a4r_score_bce_fwd(void* stream, const float* emb, const float* prec, const float* log_mask,
                                 float* pos, float* neg, float* loss_ws, int B, int L, int E, int cpc) {
    if (!emb || !prec || !log_mask || !pos || !neg || !loss_ws || B <= 0 || L < 2 || E <= 0) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int grid = (B * (L - 1) + 3) / 4; if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(score_fwd_kernel, dim3(grid), dim3(256), 0, s, emb, prec, log_mask, pos, neg, loss_ws, B, L, E, cpc);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, s, loss_ws);
    return a4r_launch_status();
}

extern "C" int a4r_score_bce_bwd(void* stream, const float* emb, const float* prec, const float* log_mask,
                                 const float* pos, const float* neg, const float* loss_ws, float loss_scale, const float* loss_scale_dev,
                                 float* d_prec, float* d_emb, int B, int L, int E, int cpc) {
    if (!emb || !prec || !log_mask || !pos || !neg || !loss_ws || !d_prec || !d_emb || B <= 0 || L < 2 || E <= 0) return A4R_EINVAL;
    int grid = (B * L + 3) / 4; if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(score_bwd_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), emb, prec, log_mask, pos, neg,
                       loss_ws, loss_scale, loss_scale_dev, d_prec, d_emb, B, L, E, cpc);
    return a4r_launch_status();
}

extern "C" int a4r_emb_grad_add_inputs(void* stream, const float* d_in, int ldi, float* d_emb, int B, int L, int E) {
    if (!d_in || !d_emb || B <= 0 || L < 2 || E <= 0 || ldi < E) return A4R_EINVAL;
    const size_t total = (size_t)B * (L - 1) * E;
    int grid = (int)((total + 255) / 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(emb_grad_add_inputs_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d_in, ldi, d_emb, B, L, E);
    return a4r_launch_status();
}

extern "C" int a4r_take_inputs(void* stream, const float* emb, float* out, int ldo, int B, int L, int E) {
    if (!emb || !out || B <= 0 || L < 2 || E <= 0 || ldo < E) return A4R_EINVAL;
    const size_t total = (size_t)B * (L - 1) * E;
    int grid = (int)((total + 255) / 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(take_inputs_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), emb, out, ldo, B, L, E);
    return a4r_launch_status();
}

extern "C" int a4r_adam_step(void* stream, float* p, const float* g, float* m, float* v, int64_t n,
                             const int32_t* seg_end, const int32_t* seg_group, int n_seg,
                             const float* group_lr, int step, float beta1, float beta2, float eps, float grad_scale) {
    if (!p || !g || !m || !v || !seg_end || !seg_group || !group_lr || n <= 0 || n_seg <= 0 || step < 1) return A4R_EINVAL;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int64_t n4 = 0;
    if (n >= (1 << 20) && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15u) == 0) {
        n4 = n / 4;                                            // quads through adam4_kernel, the last n % 4 elements through the scalar kernel
        int grid4 = (int)((n4 + 255) / 256); if (grid4 > 2048) grid4 = 2048;
        hipLaunchKernelGGL(adam4_kernel, dim3(grid4), dim3(256), 0, s, p, g, m, v, n4, seg_end, seg_group, n_seg, group_lr, bc1, sqrtf(bc2), beta1, beta2,
                           eps, grad_scale);
        if (n4 * 4 == n) return a4r_launch_status();
    }
    // (the scalar kernel indexes from 0: hand it the tail through offset pointers and an offset-free segment search -- seg_end is absolute, so the
    // tail keeps absolute indices by starting the grid-stride loop at n4 * 4: done with a base argument)
    const int64_t base = n4 * 4;
    int grid = (int)((n - base + 255) / 256); if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, s, p, g, m, v, n, seg_end, seg_group,
                       n_seg, group_lr, bc1, sqrtf(bc2), beta1, beta2, eps, grad_scale, base);
    return a4r_launch_status();
}

extern "C" int a4r_pack_matrices(void* stream, const float* flat, const a4r_pack_desc_t* desc_dev, int n_desc, int max_elems, int dtype) {
    if (!flat || !desc_dev || n_desc <= 0 || max_elems <= 0 || (dtype != A4R_BF16 && dtype != A4R_F32)) return A4R_EINVAL;
    int gx = (max_elems + 255) / 256; if (gx > 256) gx = 256;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const PackDesc* d = reinterpret_cast<const PackDesc*>(desc_dev);
    if (max_elems >= 256 * 256) {                              // large matrices in the list: 64 x 64 tiles through LDS (small ones cost a few idle tiles)
        int gt = (max_elems + 4095) / 4096; if (gt > 64) gt = 64;
        if (dtype == A4R_BF16) hipLaunchKernelGGL(pack_tiled_kernel<bf16_t>, dim3(gt, n_desc), dim3(256), 0, s, flat, d);
        else hipLaunchKernelGGL(pack_tiled_kernel<float>, dim3(gt, n_desc), dim3(256), 0, s, flat, d);
        return a4r_launch_status();
    }
    if (dtype == A4R_BF16) hipLaunchKernelGGL(pack_kernel<bf16_t>, dim3(gx, n_desc), dim3(256), 0, s, flat, d);
    else hipLaunchKernelGGL(pack_kernel<float>, dim3(gx, n_desc), dim3(256), 0, s, flat, d);
    return a4r_launch_status();
}

extern "C" int a4r_version(void) { return A4R_ABI_VERSION; }
