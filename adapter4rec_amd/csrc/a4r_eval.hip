// Eval: rank of the held-out target among all items (data_utils/metrics.py:82-116) without
// materialising the [users, items] score matrix and without the per-user Python loop.
//
// rank[u] = 1 + #{ i in [1, N1) : i not in hist(u), score(u, i) > score(u, target[u]) }.
// A workgroup owns 16 users; their vectors are the A operand (kept in registers) of fp32 MFMA
// 16x16x4 tiles whose B operand streams 16 items at a time straight from global memory (the item
// table is read once per 16 users; it is L2/Infinity-Cache resident).  The target's score is
// produced by the very same instruction sequence (B rows = the 16 users' target items, diagonal
// taken), so the comparison is bit-consistent.  History items are excluded by counting every item first and taking back the user's
// (<= A4R_EVAL_MAX_HISTORY, distinct) history ids that beat the target -- their scores from the same instruction sequence again.  fp32 throughout: ranks are integers (bit-exact
// against the oracle up to fp32 summation order of the 64-term dot products).
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

constexpr int MAXH = A4R_EVAL_MAX_HISTORY;   // history ids per user kept in LDS (reference keeps <= max_seq_len + 2 = 22; the engine allows max_seq_len <= 255)

template <int E>
__global__ void __launch_bounds__(256) eval_rank_kernel(const float* __restrict__ prec, const float* __restrict__ item_emb,
                                                        const int32_t* __restrict__ target, const int32_t* __restrict__ hist_ptr,
                                                        const int32_t* __restrict__ hist_idx, int32_t* __restrict__ rank, int U, int N1) {
    constexpr int KS = E / 16;                       // chunk steps (fp32: 16 k per step)
    __shared__ int32_t hist[16][MAXH];
    __shared__ int32_t nhist[16];
    __shared__ float tscore[16];
    __shared__ int32_t cnt[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int u0 = blockIdx.x * 16;
    {   // 16 threads per user copy its history ids
        const int ul = tid >> 4, u = min(u0 + ul, U - 1);
        const int b = hist_ptr[u], n = min(hist_ptr[u + 1] - b, MAXH);
        for (int j = tid & 15; j < n; j += 16) hist[ul][j] = hist_idx[b + j];
        if ((tid & 15) == 0) { nhist[ul] = n; cnt[ul] = 0; }
    }
    // A operand: 16 users x E, lane (user r16, kg) holds chunk (ks*4 + kg)
    uint4 ua[KS];
    const int urow = min(u0 + r16, U - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ua[ks] = *reinterpret_cast<const uint4*>(prec + (size_t)urow * E + (ks * 4 + kg) * 4);
    // target scores: B rows = target items of the 16 users, keep the diagonal
    {
        const int trow = target[urow];
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uint4 b = *reinterpret_cast<const uint4*>(item_emb + (size_t)trow * E + (ks * 4 + kg) * 4);
            Mma<float>::mma(ua[ks], b, acc);
        }
        // element (row = kg*4 + rr, col = r16): diagonal where kg*4 + rr == r16
        if (wave == 0) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (kg * 4 + rr == r16) tscore[r16] = acc[rr];
        }
    }
    __syncthreads();
    float ts[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) ts[rr] = tscore[kg * 4 + rr];
    int local[4] = {0, 0, 0, 0};
    // items 1 .. N1-1 in tiles of 16; the 4 waves of the block take every 4th tile
    // The item fragments of the NEXT tile are requested before this tile's products (round 6): the table comes from L2 / the Infinity Cache at 500+ cycles a
    // request, and with one tile in flight per wave the launch ran at 0.23 of the fp32 MFMA rate -- waiting, not streaming (profiles/r06_e_eval_host.txt).
    const int ntiles = (N1 - 1 + 15) / 16, tstep = gridDim.y * 4;
    int t = blockIdx.y * 4 + wave;
    uint4 bn[KS];
    {
        const int irow = min(1 + min(t, ntiles - 1) * 16 + r16, N1 - 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) bn[ks] = *reinterpret_cast<const uint4*>(item_emb + (size_t)irow * E + (ks * 4 + kg) * 4);
    }
    for (; t < ntiles; t += tstep) {
        const int i = 1 + t * 16 + r16;                 // this lane's item column
        uint4 b[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b[ks] = bn[ks];
        {
            const int irow = min(1 + min(t + tstep, ntiles - 1) * 16 + r16, N1 - 1);      // (past the end: the last tile again, never used)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) bn[ks] = *reinterpret_cast<const uint4*>(item_emb + (size_t)irow * E + (ks * 4 + kg) * 4);
        }
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) Mma<float>::mma(ua[ks], b[ks], acc);
        // EVERY item that beats the target is counted here; the user's history items among them are taken back below -- a scan of the (up to 264) history ids
        // per beating item was most of this loop (on random scores half the items beat the target: 36 TF/s = 0.23 of the fp32 MFMA rate, round 5)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) local[rr] += (i < N1 && acc[rr] > ts[rr]) ? 1 : 0;
    }
    // The history items: score(u, h) for the j-th history id of each of the 16 users by the SAME instruction sequence (B rows = those ids, the diagonal kept, as for
    // the target: an element's bits do not depend on the column it sits in), once per id -- a repeated id, the pad item 0 or an id outside the table counts nothing,
    // as a mask would.  Only the workgroup that owns the first item range does this (the others share its users).
    if (blockIdx.y == 0) {
        int mx = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) mx = max(mx, nhist[u]);
        for (int j = wave; j < mx; j += 4) {
            const int h = j < nhist[r16] ? hist[r16][j] : 0;
            bool ok = h >= 1 && h < N1 && u0 + r16 < U;
            for (int k = 0; k < j && ok; ++k) ok = hist[r16][k] != h;          // first occurrence only
            const int hrow = ok ? h : 0;
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint4 b = *reinterpret_cast<const uint4*>(item_emb + (size_t)hrow * E + (ks * 4 + kg) * 4);
                Mma<float>::mma(ua[ks], b, acc);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if (kg * 4 + rr == r16 && ok && acc[rr] > ts[rr]) atomicSub(&cnt[r16], 1);
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        int c = local[rr];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);     // over the 16 item columns
        if (r16 == 0 && c) atomicAdd(&cnt[kg * 4 + rr], c);
    }
    __syncthreads();
    if (tid < 16 && u0 + tid < U && cnt[tid]) atomicAdd(rank + u0 + tid, cnt[tid]);
}

__global__ void fill_i32_kernel(int32_t* p, int n, int v) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = v;
}

}  // namespace

extern "C" int a4r_eval_rank(void* stream, const float* prec, const float* item_emb, const int32_t* target,
                             const int32_t* hist_ptr, const int32_t* hist_idx, int32_t* rank, int U, int N1, int E) {
    if (!prec || !item_emb || !target || !hist_ptr || !hist_idx || !rank || U <= 0 || N1 < 2) return A4R_EINVAL;
    if (E != 64 && E != 128 && E != 256 && E != 512) return A4R_EINVAL;
    if ((reinterpret_cast<uintptr_t>(prec) | reinterpret_cast<uintptr_t>(item_emb)) & 15u) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(fill_i32_kernel, dim3((U + 255) / 256), dim3(256), 0, s, rank, U, 1);
    const int gx = (U + 15) / 16;
    int gy = 2048 / gx; if (gy < 1) gy = 1;
    const int ntiles = (N1 - 1 + 15) / 16;
    if (gy > (ntiles + 3) / 4) gy = (ntiles + 3) / 4;
    if (E == 64) hipLaunchKernelGGL(eval_rank_kernel<64>, dim3(gx, gy), dim3(256), 0, s, prec, item_emb, target, hist_ptr, hist_idx, rank, U, N1);
    else if (E == 128) hipLaunchKernelGGL(eval_rank_kernel<128>, dim3(gx, gy), dim3(256), 0, s, prec, item_emb, target, hist_ptr, hist_idx, rank, U, N1);
    else if (E == 256) hipLaunchKernelGGL(eval_rank_kernel<256>, dim3(gx, gy), dim3(256), 0, s, prec, item_emb, target, hist_ptr, hist_idx, rank, U, N1);
    else hipLaunchKernelGGL(eval_rank_kernel<512>, dim3(gx, gy), dim3(256), 0, s, prec, item_emb, target, hist_ptr, hist_idx, rank, U, N1);
    return a4r_launch_status();
}
