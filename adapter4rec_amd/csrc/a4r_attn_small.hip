// a4r_attn_fwd / a4r_attn_bwd for NARROW heads (dh <= 16, e.g. 8 and 16; S <= 32): the two-block transformers inside a
// K-Adapter (Downstream/Text/model/modules.py:161-206 KAdapterBlock; heads of width 192/12 = 16 on the BERT side and
// 16/2 = 8 on the SASRec side with the reference's launcher values) -- shapes the MFMA kernels of a4r_attn.hip (dh 32 / 64)
// do not cover and that are far too small to matter for time (M x 192 activations next to the M x 768 backbone).
// Plain fp32 VALU arithmetic: 32 lanes own one (item, head) pair, lane = query row; K, V (and Q, dO in backward) of the pair
// sit in LDS and are read as broadcasts; the backward re-maps lane = key row after exchanging P and dS through LDS.
// Same semantics as a4r_attn.hip (key mask, causal, mask_neg, counter-based dropout regenerated in backward); the dropout
// lot of probability (q, k) is hash(seed, site, ((item * heads + h) * S + q) * S + k) -- private to this kernel pair.
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

constexpr int PAIRS = 2;              // (item, head) pairs per 64-thread workgroup (backward holds 33 KB of LDS)
constexpr int SMAX = 32, DMAX = 16;   // dh <= 16

struct SmallArgs {
    const void* qkv; int ld, q_off, k_off, v_off;
    void* out; int ldo; const void* dout; void* dqkv;
    const float* key_mask; int n_items, S, nh, dh, causal;
    float scale, mask_neg, keep_scale; uint32_t thr16, site; uint64_t seed;
};

template <typename T> A4R_DEV float ldf(const T* p) { return Elem<T>::ld(p); }

// probabilities of query row q of one pair: p[k] (after softmax, BEFORE dropout); returns nothing S x S outside registers
template <typename T>
A4R_DEV void row_probs(const SmallArgs& a, const float* Ks, const float* qv, int item, int q, float (&p)[SMAX]) {
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < SMAX; ++k) {
        float s = -INFINITY;
        if (k < a.S) {
            s = 0.f;
            for (int d = 0; d < a.dh; ++d) s += qv[d] * Ks[k * DMAX + d];
            s *= a.scale;
            const bool allowed = (!a.key_mask || a.key_mask[(size_t)item * a.S + k] != 0.f) && (!a.causal || k <= q);
            if (!allowed) s += a.mask_neg;
        }
        p[k] = s;
        m = fmaxf(m, s);
    }
    float l = 0.f;
#pragma unroll
    for (int k = 0; k < SMAX; ++k) { p[k] = k < a.S ? expf(p[k] - m) : 0.f; l += p[k]; }
    const float inv = 1.f / l;
#pragma unroll
    for (int k = 0; k < SMAX; ++k) p[k] *= inv;
}

A4R_DEV float keep_of(const SmallArgs& a, int item, int h, int q, int k) {
    if (!a.thr16) return 1.f;
    const uint64_t e = (((uint64_t)item * a.nh + h) * a.S + q) * a.S + k;
    const uint64_t hsh = a4r_hash64(a.seed, a.site, e >> 2);
    return (((uint32_t)(hsh >> (16 * (e & 3))) & 0xffffu) >= a.thr16) ? a.keep_scale : 0.f;
}

template <typename T>
__global__ void __launch_bounds__(32 * PAIRS) attn_small_fwd_kernel(const SmallArgs a) {
    __shared__ float Ks[PAIRS][SMAX * DMAX], Vs[PAIRS][SMAX * DMAX];
    const int pl = threadIdx.x >> 5, q = threadIdx.x & 31;
    const long pair = (long)blockIdx.x * PAIRS + pl;
    const bool live = pair < (long)a.n_items * a.nh;
    const int item = live ? (int)(pair / a.nh) : 0, h = live ? (int)(pair % a.nh) : 0;
    const T* base = reinterpret_cast<const T*>(a.qkv) + (size_t)item * a.S * a.ld + h * a.dh;
    if (live && q < a.S)
        for (int d = 0; d < a.dh; ++d) {
            Ks[pl][q * DMAX + d] = ldf(base + (size_t)q * a.ld + a.k_off + d);
            Vs[pl][q * DMAX + d] = ldf(base + (size_t)q * a.ld + a.v_off + d);
        }
    __syncthreads();
    if (!live || q >= a.S) return;
    float qv[DMAX], p[SMAX], o[DMAX];
    for (int d = 0; d < a.dh; ++d) { qv[d] = ldf(base + (size_t)q * a.ld + a.q_off + d); o[d] = 0.f; }
    row_probs<T>(a, Ks[pl], qv, item, q, p);
#pragma unroll
    for (int k = 0; k < SMAX; ++k) {
        if (k < a.S) {
            const float pk = p[k] * keep_of(a, item, h, q, k);
            for (int d = 0; d < a.dh; ++d) o[d] += pk * Vs[pl][k * DMAX + d];
        }
    }
    T* orow = reinterpret_cast<T*>(a.out) + ((size_t)item * a.S + q) * a.ldo + h * a.dh;
    for (int d = 0; d < a.dh; ++d) Elem<T>::st(orow + d, o[d]);
}

template <typename T>
__global__ void __launch_bounds__(32 * PAIRS) attn_small_bwd_kernel(const SmallArgs a) {
    __shared__ float Ks[PAIRS][SMAX * DMAX], Vs[PAIRS][SMAX * DMAX], Qs[PAIRS][SMAX * DMAX], Os[PAIRS][SMAX * DMAX];
    __shared__ float Ps[PAIRS][SMAX * (SMAX + 1)], Ss[PAIRS][SMAX * (SMAX + 1)];
    const int pl = threadIdx.x >> 5, q = threadIdx.x & 31;
    const long pair = (long)blockIdx.x * PAIRS + pl;
    const bool live = pair < (long)a.n_items * a.nh;
    const int item = live ? (int)(pair / a.nh) : 0, h = live ? (int)(pair % a.nh) : 0;
    const T* base = reinterpret_cast<const T*>(a.qkv) + (size_t)item * a.S * a.ld + h * a.dh;
    const T* dob = reinterpret_cast<const T*>(a.dout) + (size_t)item * a.S * a.ldo + h * a.dh;
    const bool row = live && q < a.S;
    if (row)
        for (int d = 0; d < a.dh; ++d) {
            Ks[pl][q * DMAX + d] = ldf(base + (size_t)q * a.ld + a.k_off + d);
            Vs[pl][q * DMAX + d] = ldf(base + (size_t)q * a.ld + a.v_off + d);
            Qs[pl][q * DMAX + d] = ldf(base + (size_t)q * a.ld + a.q_off + d);
            Os[pl][q * DMAX + d] = ldf(dob + (size_t)q * a.ldo + d);
        }
    __syncthreads();
    T* drow = reinterpret_cast<T*>(a.dqkv) + ((size_t)item * a.S + q) * a.ld + h * a.dh;
    if (row) {      // lane = query: P~ (dropped probabilities), dS; dQ = dS K
        float qv[DMAX], p[SMAX];
        for (int d = 0; d < a.dh; ++d) qv[d] = Qs[pl][q * DMAX + d];
        row_probs<T>(a, Ks[pl], qv, item, q, p);
        float dp[SMAX], delta = 0.f;
#pragma unroll
        for (int k = 0; k < SMAX; ++k) {
            dp[k] = 0.f;
            if (k < a.S) {
                float t = 0.f;
                for (int d = 0; d < a.dh; ++d) t += Os[pl][q * DMAX + d] * Vs[pl][k * DMAX + d];
                const float kp = keep_of(a, item, h, q, k);
                dp[k] = t * kp;                                   // gradient wrt the un-dropped probability
                Ps[pl][q * (SMAX + 1) + k] = p[k] * kp;
                delta += p[k] * dp[k];
            }
        }
        float dq[DMAX];
        for (int d = 0; d < a.dh; ++d) dq[d] = 0.f;
#pragma unroll
        for (int k = 0; k < SMAX; ++k) {
            if (k < a.S) {
                const float ds = p[k] * (dp[k] - delta) * a.scale;
                Ss[pl][q * (SMAX + 1) + k] = ds;
                for (int d = 0; d < a.dh; ++d) dq[d] += ds * Ks[pl][k * DMAX + d];
            }
        }
        for (int d = 0; d < a.dh; ++d) Elem<T>::st(drow + a.q_off + d, dq[d]);
    }
    __syncthreads();
    if (row) {      // lane = key: dK = dS^T Q, dV = P~^T dO
        const int k = q;
        float dk[DMAX], dv[DMAX];
        for (int d = 0; d < a.dh; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        for (int qq = 0; qq < a.S; ++qq) {
            const float ds = Ss[pl][qq * (SMAX + 1) + k], pp = Ps[pl][qq * (SMAX + 1) + k];
            for (int d = 0; d < a.dh; ++d) { dk[d] += ds * Qs[pl][qq * DMAX + d]; dv[d] += pp * Os[pl][qq * DMAX + d]; }
        }
        for (int d = 0; d < a.dh; ++d) { Elem<T>::st(drow + a.k_off + d, dk[d]); Elem<T>::st(drow + a.v_off + d, dv[d]); }
    }
}

}  // namespace

// called by a4r_attn_fwd / a4r_attn_bwd (a4r_attn.hip) for dh < 32 after their argument validation
int a4r_attn_small(hipStream_t s, const a4r_attn_t* t, bool bwd) {
    if (t->S > SMAX || t->dh > DMAX || t->dh <= 0) return A4R_EINVAL;
    SmallArgs a;
    a.qkv = t->qkv; a.ld = t->ld; a.q_off = t->q_off; a.k_off = t->k_off; a.v_off = t->v_off;
    a.out = t->out; a.ldo = t->ldo; a.dout = t->dout; a.dqkv = t->dqkv; a.key_mask = t->key_mask;
    a.n_items = t->n_items; a.S = t->S; a.nh = t->n_heads; a.dh = t->dh; a.causal = t->causal;
    a.scale = t->scale; a.mask_neg = t->mask_neg; a.keep_scale = a4r_keep_scale(t->drop_p); a.thr16 = a4r_thr16(t->drop_p);
    a.site = t->drop_site; a.seed = t->drop_seed;
    const long pairs = (long)t->n_items * t->n_heads;
    const dim3 grid((unsigned)((pairs + PAIRS - 1) / PAIRS)), block(32 * PAIRS);
    if (t->dtype == A4R_BF16) {
        if (bwd) hipLaunchKernelGGL(attn_small_bwd_kernel<bf16_t>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(attn_small_fwd_kernel<bf16_t>, grid, block, 0, s, a);
    } else {
        if (bwd) hipLaunchKernelGGL(attn_small_bwd_kernel<float>, grid, block, 0, s, a);
        else hipLaunchKernelGGL(attn_small_fwd_kernel<float>, grid, block, 0, s, a);
    }
    return a4r_launch_status();
}
