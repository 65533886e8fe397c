// a4r_attn_long_fwd / a4r_attn_long_bwd: un-masked multi-head attention for 32 < S <= 256 tokens per item, head width 64 or 32
// -- the ViT / MAE item tower (S = 197 / 50, dh 64; HF ViTSelfAttention as called from Downstream/CV/model/encoders.py:21-32)
// and the two TransformerBlocks inside each KAdapterBlock of VITKAdaptedCVModel (Downstream/CV/model/model.py:374-404,
// modules.py:24-36,148-187: width 384, 12 heads of 32, all-ones mask, dropout on the probabilities).
// The S <= 32 kernels of a4r_attn.hip keep a whole score matrix in one wave; here one workgroup (8 waves) owns one
// (item, head) pair, stages the key-side matrices of that pair in LDS and never writes anything S x S to HBM.
//
// Everything is the 16x16 "chunk" primitive of a4r_common.h (so the bf16 and exact-fp32 instantiations share all
// addressing).  A score tile is produced TRANSPOSED (key-side fragment first): lane (c = l & 15, kg = l >> 4) then holds,
// for ONE query column c, the 4 consecutive keys 16*kt + 4*kg + r.  The whole key range of a 16-query block lives in
// registers (<= 16 tiles), so the softmax is plain register arithmetic + two cross-lane steps (l ^ 16, l ^ 32), and the
// probabilities feed the next product as an MFMA operand WITHOUT any re-layout by permuting the contraction index:
//     k-slot (kg, j) of chunk step st  <->  key  KSTEP*st + (j >> 2)*16 + 4*kg + (j & 3)
// (two score tiles per step for bf16, one for fp32).  The other operand of such a product needs, per lane, 4 + 4
// consecutive KEYS at one head column -- i.e. V (or K, Q, dO) read TRANSPOSED from its row-major LDS image: ds_read_b64_tr_b16 for
// bf16, four 4-byte gathers for fp32 (frag_T; no second, transposed image of anything).
//
// forward  (per 16-query block): S^T = K Q^T | softmax | O^T = V^T P^T         + lse = max + log(sum) per query (fp32)
// backward, two kernels (FlashAttention-2 split, no atomics) -- as two launches above 64 tokens, as ONE launch (attn_long_bwd_kernel: dq half, barrier, dkdv half in the
// same workgroup and LDS) up to 64, where the second half then finds the first half's reads in the caches:
//   dq   (per 16-query block): delta = dO . O (forward output), then per chunk step of keys S^T, dP^T = V dO^T,
//        dS = P (dP - delta) scale, dQ^T += K^T dS^T  (nothing held for the whole key range)
//   dkdv (per 16-key tile, a wave owns its key tiles): S = Q K^T, dP = dO V^T (tiles with the KEY on the lane),
//        dV^T += dO^T P, dK^T += Q^T dS  over all query groups; lse and delta come from LDS.
// HBM-bound in principle (fwd: reads 3 M H, writes M H); measured numbers are in DESIGN.md.
#include "a4r_attn_long.h"

int a4r_attn_long_bwd1_launch(hipStream_t s, const a4r_attn_t* a, const float* lse);      // a4r_attn_long1.hip

namespace {

// ------------------------------------------------------------------------------------------------ forward
template <typename T, int DH, int NKT, bool KM = false>      // KM: the launch carries a key mask (text towers; instantiated for head width 64)
__global__ void __launch_bounds__(WG<NKT>::NTHR, DH > 64 ? 2 : 4) attn_long_fwd_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                            T* __restrict__ ctx, int ldo, float* __restrict__ lse,
                                                            int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
    using G = Geo<T, DH>;
    constexpr int NTHR = WG<NKT>::NTHR, NWAVE = WG<NKT>::NWAVE;
    constexpr int SP = NKT * 16, SPT = SP + 8, NST = SP / G::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kr = smem;                                                  // [SP][DH] row-major
    char* Vimg = smem + SP * G::ROWB;                                 // [SP][DH] row-major (read transposed: ds_read_b64_tr_b16 / 4-byte gathers)
    const int item = blockIdx.x / nh, h = blockIdx.x % nh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kg = lane >> 4;
    char* stg = smem + lds_main_fwd<T, DH, NKT>() + wave * STG<T, DH>::BYTES;      // this wave's output staging block (bf16)
    const T* base = qkv + (size_t)item * S * ld + h * DH;
    const int nqb = (S + 15) >> 4;
    // the query rows of a block are requested one block ahead: the first before the key side is staged (its latency hides behind the staging's),
    // the next at the top of the current block's arithmetic
    uint4 qn[G::KS];
    const RowsView qview = RowsView::make(base + q_off, ld, S, DH), cview = RowsView::make(ctx + (size_t)item * S * ldo + h * DH, ldo, S, DH);
    auto request_q = [&](int qb) {
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) qn[ks] = qview.row_chunk(qb * 16 + fr, ks * 4 + kg);
    };
    if (wave < nqb) request_q(wave);
    {
        Stager<T, DH, SP, NTHR> sk, sv;
        sk.request(RowsView::make(base + k_off, ld, S, DH), tid);
        sv.request(RowsView::make(base + v_off, ld, S, DH), tid);
        sk.commit(Kr, tid);
        sv.commit(Vimg, tid);
    }
    [[maybe_unused]] float* km = reinterpret_cast<float*>(smem + lds_main_fwd<T, DH, NKT>() - lds_km<NKT>());
    if constexpr (KM) stage_key_mask<SP, NTHR>(km, kmask, item, S, tid);
    __syncthreads();
    int first = 0;
    if constexpr (KM) first = first_attended<SP>(km, lane);
    for (int qb = wave; qb < nqb; qb += NWAVE) {
        const int rq = qb * 16 + fr;
        const bool valid = rq < S;
        uint4 qf[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) qf[ks] = qn[ks];
        if (qb + NWAVE < nqb) request_q(qb + NWAVE);
        f32x4_t s[NKT];
        scores_t<T, DH, NKT>(Kr, qf, s, S, fr, kg);
        if constexpr (KM) {
            const bool empty = causal ? rq < first : first >= S;          // this query has no allowed key
            const int kmax = causal ? rq : SP;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const f32x4_t m4 = *reinterpret_cast<const f32x4_t*>(km + kt * 16 + kg * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + kg * 4 + r;
                    s[kt][r] = empty ? (key < S ? 0.f : -INFINITY) : ((m4[r] != 0.f && key <= kmax) ? s[kt][r] : -INFINITY);
                }
            }
        }
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kt][r]);
        m = red4(m, true);                                   // max of the RAW scores (scale > 0)
        // p = exp(scale (s - m)) = exp2(s c - m c), c = scale log2(e): one (packed) fma + v_exp_f32 per element; the 1 / row-sum factor is
        // applied to the 16 output values of the lane instead of its 4 NKT probabilities
        const float c2 = scale * 1.44269504088896f, mc = m * c2;
        const f32x4_t c2v = {c2, c2, c2, c2}, mcv = {-mc, -mc, -mc, -mc};
        f32x4_t lv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4_t t = __builtin_elementwise_fma(s[kt], c2v, mcv);
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = __builtin_amdgcn_exp2f(t[r]);
            s[kt] = t;
            lv += t;
        }
        const float l = red4((lv[0] + lv[1]) + (lv[2] + lv[3]), false);           // >= 1: the row maximum contributes exp2(0)
        float inv = sizeof(T) == 2 ? __builtin_amdgcn_rcpf(l) : 1.f / l;
        if (dr.thr16) {                                       // P' = dropout(P): the 4 keys of a lane's tile column share one hash
            inv *= dr.keep_scale;
            const int ol = opaque_lane(lane), orq = qb * 16 + (ol & 15), okg = ol >> 4;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const uint64_t hsh = a4r_hash64(dr.seed, dr.site, drop_idx(blockIdx.x, orq, kt * 16 + okg * 4) >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] = (((uint32_t)(hsh >> (16 * r)) & 0xffffu) >= dr.thr16) ? s[kt][r] : 0.f;
            }
        }
        if (valid && kg == 0) lse[((size_t)item * nh + h) * S + rq] = m * scale + __builtin_amdgcn_logf(l) * 0.69314718055994531f;   // v_log_f32 = log2; l is a normal number
        f32x4_t o[G::ND];
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // O^T += V^T P^T in regions of HALF head-column tiles: the transposed V fragments of the next region are requested before the products of the
        // current one; probabilities are packed step by step (keeping all NST chunks spilled)
        constexpr int HALF = G::ND >= 4 ? 2 : G::ND, NR = G::ND / HALF, NREG = NST * NR;
        uint4 vf[2][HALF], pf = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < HALF; ++j) vf[0][j] = frag_T<T, DH>(Vimg, SPT, j * 16, 0, lane);
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const int st = i / NR, hr = i % NR;
            if (hr == 0) pf = pack_step<T, NKT>(s, st);
            if (i + 1 < NREG) {
#pragma unroll
                for (int j = 0; j < HALF; ++j) vf[(i + 1) & 1][j] = frag_T<T, DH>(Vimg, SPT, (((i + 1) % NR) * HALF + j) * 16, (i + 1) / NR, lane);
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) Mma<T>::mma(vf[i & 1][j], pf, o[hr * HALF + j]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) o[dt] *= f32x4_t{inv, inv, inv, inv};
        store_block16<T, DH>(stg, o, cview, qb * 16, lane);
    }
}

// ------------------------------------------------------------------------------------------------ backward: dq + delta
template <typename T, int DH, int NKT, bool KM>
A4R_DEV void dq_body(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off, const T* __restrict__ dctx, int ldo, const T* __restrict__ octx,
                     const float* __restrict__ lse, float* delta, T* dqkv, int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
    using G = Geo<T, DH>;
    constexpr int NTHR = WG<NKT>::NTHR, NWAVE = WG<NKT>::NWAVE;
    constexpr int SP = NKT * 16, SPT = SP + 8, NST = SP / G::KSTEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Kr = smem;
    char* Vr = smem + SP * G::ROWB;
    char* Kimg = Kr;                                                  // K's row-major image doubles as the transposed operand
    const int item = blockIdx.x / nh, h = blockIdx.x % nh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, kg = lane >> 4;
    char* stg = smem + lds_main_dq<T, DH, NKT>() + wave * STG<T, DH>::BYTES;
    const T* base = qkv + (size_t)item * S * ld + h * DH;
    const int nqb = (S + 15) >> 4;
    // the q / dO / O rows and the lse of a wave's FIRST block are requested before the key side is staged (their latency runs under the staging's);
    // a second set for the next block does not fit the registers of the head-width-64 instantiations: with it they spilled inside the key loop, and a
    // scratch reload's vmcnt(0) waits for every prefetch in flight
    uint4 qn[G::KS], don[G::KS], on[G::KS];
    float lqn = 0.f;
    const RowsView qview = RowsView::make(base + q_off, ld, S, DH), doview = RowsView::make(dctx + (size_t)item * S * ldo + h * DH, ldo, S, DH),
                   oview = RowsView::make(octx + (size_t)item * S * ldo + h * DH, ldo, S, DH),
                   dqview = RowsView::make(dqkv + (size_t)item * S * ld + q_off + h * DH, ld, S, DH);
    auto request_rows = [&](int qb) {
        const int ol = opaque_lane(lane), rq = qb * 16 + (ol & 15), okg = ol >> 4;      // (offsets rebuilt per call, not held across the loops)
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            qn[ks] = qview.row_chunk(rq, ks * 4 + okg);
            don[ks] = doview.row_chunk(rq, ks * 4 + okg);
            on[ks] = oview.row_chunk(rq, ks * 4 + okg);
        }
        lqn = rq < S ? lse[((size_t)item * nh + h) * S + rq] : 0.f;
    };
    if (wave < nqb) request_rows(wave);
    {
        Stager<T, DH, SP, NTHR> sk, sv;
        sk.request(RowsView::make(base + k_off, ld, S, DH), tid);
        sv.request(RowsView::make(base + v_off, ld, S, DH), tid);
        sk.commit(Kr, tid);
        sv.commit(Vr, tid);
    }
    [[maybe_unused]] float* km = reinterpret_cast<float*>(smem + lds_main_dq<T, DH, NKT>() - lds_km<NKT>());
    if constexpr (KM) stage_key_mask<SP, NTHR>(km, kmask, item, S, tid);
    __syncthreads();
    int first = 0;
    if constexpr (KM) first = first_attended<SP>(km, lane);
    const float c2 = scale * 1.44269504088896f;
    const f32x4_t c2v = {c2, c2, c2, c2};
    for (int qb = wave; qb < nqb; qb += NWAVE) {
        const int rq = qb * 16 + fr;
        const bool valid = rq < S;
        [[maybe_unused]] const bool empty = KM && (causal ? rq < first : first >= S);
        [[maybe_unused]] const int kmax = causal ? rq : SP;
        if (qb != wave) request_rows(qb);
        uint4 qf[G::KS], dof[G::KS], of[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) { qf[ks] = qn[ks]; dof[ks] = don[ks]; of[ks] = on[ks]; }
        const float lq2 = lqn * 1.44269504088896f;
        // delta = sum_k P' dP' = dO . O (the forward output, dropout included): one dot product per query instead of a pass over all
        // key tiles, so P, dP and dS are produced and consumed one chunk step (32 / 16 keys) at a time and never held for the whole row
        float dsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            if constexpr (sizeof(T) == 2) {                   // v_dot2c_f32_bf16: two exact products and the fp32 sums per instruction
                const uint32_t a4[4] = {of[ks].x, of[ks].y, of[ks].z, of[ks].w}, b4[4] = {dof[ks].x, dof[ks].y, dof[ks].z, dof[ks].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) dsum = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_raw_t, a4[e]), __builtin_bit_cast(bf16x2_raw_t, b4[e]), dsum, false);
            } else {
                float a8[G::PER], b8[G::PER];
                Elem<T>::unpack(of[ks], a8);
                Elem<T>::unpack(dof[ks], b8);
#pragma unroll
                for (int e = 0; e < G::PER; ++e) dsum += a8[e] * b8[e];
            }
        }
        dsum = red4(dsum, false);
        if (valid && kg == 0) delta[((size_t)item * nh + h) * S + rq] = dsum;
        const f32x4_t lqv = {-lq2, -lq2, -lq2, -lq2}, ndsv = {-dsum, -dsum, -dsum, -dsum};
        const int lim = S - kg * 4;
        f32x4_t o[G::ND];
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // per key tile: S^T and dP^T (row fragments of K and V), then the element-wise stage; the next tile's fragments are requested between the
        // two (and the transposed K fragments of the step's dQ product before its first tile), so the LDS latency runs under vector work
        uint4 rk[G::KS], rv[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) { rk[ks] = frag_rows<T, DH>(Kr, fr, ks, kg); rv[ks] = frag_rows<T, DH>(Vr, fr, ks, kg); }
#pragma unroll
        for (int st = 0; st < NST; ++st) {
            f32x4_t ds[G::TPS];
            constexpr int HALF = G::ND >= 4 ? 2 : G::ND;
            uint4 tf[2][HALF];                                  // transposed K fragments, two slots of HALF head-column tiles
#pragma unroll
            for (int t = 0; t < G::TPS; ++t) {
                const int kt = st * G::TPS + t;
                f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = ndsv;      // dP - delta straight from the matrix pipe: the accumulator starts at -delta
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    Mma<T>::mma(rk[ks], qf[ks], sc);
                    Mma<T>::mma(rv[ks], dof[ks], dp);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kt + 1 < NKT) {
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) {
                        rk[ks] = frag_rows<T, DH>(Kr, (kt + 1) * 16 + fr, ks, kg);
                        rv[ks] = frag_rows<T, DH>(Vr, (kt + 1) * 16 + fr, ks, kg);
                    }
                }
                if (t == G::TPS - 1) {
#pragma unroll
                    for (int j = 0; j < HALF; ++j) tf[0][j] = frag_T<T, DH>(Kimg, SPT, j * 16, st, lane);
                }
                if constexpr (KM) { if (empty) sc = f32x4_t{0.f, 0.f, 0.f, 0.f}; }       // (a query without allowed keys: the forward's convention)
                f32x4_t pv = __builtin_elementwise_fma(sc, c2v, lqv);                   // P = exp(scale s - lse)
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(pv[r]);
                if constexpr (KM) if (!empty) {
                    const f32x4_t m4 = *reinterpret_cast<const f32x4_t*>(km + kt * 16 + kg * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = (m4[r] != 0.f && kt * 16 + kg * 4 + r <= kmax) ? pv[r] : 0.f;
                }
                if (kt >= kt_partial_lo<NKT>() && kt * 16 + 16 > S) {                    // wave-uniform: the last one or two tiles
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = (kt * 16 + r < lim) ? pv[r] : 0.f;
                }
                if (dr.thr16) {
                    const int ol = opaque_lane(lane);
                    const uint64_t hsh = a4r_hash64(dr.seed, dr.site, drop_idx(blockIdx.x, qb * 16 + (ol & 15), kt * 16 + (ol >> 4) * 4) >> 2);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dp[r] = ((((uint32_t)(hsh >> (16 * r)) & 0xffffu) >= dr.thr16) ? (dp[r] + dsum) * dr.keep_scale : 0.f) - dsum;
                }
                ds[t] = pv * dp;                                                       // (x scale: applied to the 16 outputs of the lane)
                __builtin_amdgcn_sched_barrier(0);
            }
            const uint4 dsf = pack_step<T, G::TPS>(ds, 0);
            if constexpr (G::ND > HALF) {
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[1][j] = frag_T<T, DH>(Kimg, SPT, (HALF + j) * 16, st, lane);
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[0][j], dsf, o[j]);
            if constexpr (G::ND > HALF) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[1][j], dsf, o[HALF + j]);
            }
            if constexpr (G::ND > 2 * HALF) {                       // head width 128 (fp32, the user tower): the column tiles behind the first four, unscheduled
#pragma unroll
                for (int j = 2 * HALF; j < G::ND; ++j) Mma<T>::mma(frag_T<T, DH>(Kimg, SPT, j * 16, st, lane), dsf, o[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) o[dt] *= f32x4_t{scale, scale, scale, scale};
        store_block16<T, DH>(stg, o, dqview, qb * 16, lane);
    }
}

template <typename T, int DH, int NKT, bool KM = false>      // KM: the launch carries a key mask (text towers / the user tower)
__global__ void __launch_bounds__(WG<NKT>::NTHR, DH > 64 ? 2 : 4) attn_long_dq_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                           const T* __restrict__ dctx, int ldo, const T* __restrict__ octx,
                                                           const float* __restrict__ lse, float* __restrict__ delta,
                                                           T* __restrict__ dqkv, int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
    dq_body<T, DH, NKT, KM>(qkv, ld, q_off, k_off, v_off, dctx, ldo, octx, lse, delta, dqkv, S, nh, scale, dr, kmask, causal);
}

// ------------------------------------------------------------------------------------------------ backward: dk, dv
template <typename T, int DH, int NKT, bool KM>
A4R_DEV void dkdv_body(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off, const T* __restrict__ dctx, int ldo, const float* __restrict__ lse,
                       const float* delta, T* dqkv, int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
    using G = Geo<T, DH>;
    constexpr int NTHR = WG<NKT>::NTHR, NWAVE = WG<NKT>::NWAVE;
    constexpr int SP = NKT * 16, SPT = SP + 8, NG = SP / G::KSTEP;          // NG query groups of KSTEP queries
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qr = smem;
    char* Or = smem + SP * G::ROWB;
    char* Qimg = Qr;
    char* Oimg = Or;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * SP * G::ROWB);
    float* del_s = lse_s + SP;
    const int item = blockIdx.x / nh, h = blockIdx.x % nh;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = tid >> 6;
    char* stg = smem + lds_main_dkdv<T, DH, NKT>() + wave * STG<T, DH>::BYTES;
    const T* base = qkv + (size_t)item * S * ld + h * DH;
    const T* dob = dctx + (size_t)item * S * ldo + h * DH;
    const int nkt = (S + 15) >> 4;
    // the K / V rows of a wave's first key tile are requested before the query side is staged
    uint4 kn[G::KS], vn[G::KS];
    const RowsView kview = RowsView::make(base + k_off, ld, S, DH), vview = RowsView::make(base + v_off, ld, S, DH),
                   dkview = RowsView::make(dqkv + (size_t)item * S * ld + k_off + h * DH, ld, S, DH),
                   dvview = RowsView::make(dqkv + (size_t)item * S * ld + v_off + h * DH, ld, S, DH);
    auto request_kv = [&](int kt) {
        const int ol = opaque_lane(lane0), rk = kt * 16 + (ol & 15), okg = ol >> 4;      // (offsets rebuilt per call, not held across the loops)
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            kn[ks] = kview.row_chunk(rk, ks * 4 + okg);
            vn[ks] = vview.row_chunk(rk, ks * 4 + okg);
        }
    };
    if (wave < nkt) request_kv(wave);
    {
        Stager<T, DH, SP, NTHR> sq, so;
        sq.request(RowsView::make(base + q_off, ld, S, DH), tid);
        so.request(RowsView::make(dob, ldo, S, DH), tid);
        static_assert(SP <= NTHR, "one row statistic per thread");
        const bool rv = tid < S;
        const float lv = rv ? lse[((size_t)item * nh + h) * S + tid] * -1.44269504088896f : 0.f;
        const float dv0 = rv ? -delta[((size_t)item * nh + h) * S + tid] : 0.f;
        sq.commit(Qr, tid);
        so.commit(Or, tid);
        if (tid < SP) { lse_s[tid] = lv; del_s[tid] = dv0; }
    }
    [[maybe_unused]] float* km = reinterpret_cast<float*>(smem + lds_main_dkdv<T, DH, NKT>() - lds_km<NKT>());
    if constexpr (KM) stage_key_mask<SP, NTHR>(km, kmask, item, S, tid);
    __syncthreads();
    int first = 0;
    if constexpr (KM) first = first_attended<SP>(km, lane0);
    const float c2 = scale * 1.44269504088896f;
    const f32x4_t c2v = {c2, c2, c2, c2};
    for (int kt = wave; kt < nkt; kt += NWAVE) {
        // (lane-derived addresses are rebuilt per key tile from an opaque copy of the lane index: held across this loop they were spilled and reloaded)
        const int lane = opaque_lane(lane0), fr = lane & 15, kg = lane >> 4;
        if (kt != wave) request_kv(kt);
        uint4 kf[G::KS], vf[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) { kf[ks] = kn[ks]; vf[ks] = vn[ks]; }
        f32x4_t dk[G::ND], dv[G::ND];
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) { dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        // per 16-query tile: S and dP (row fragments of Q and dO; the next tile's are requested right after the products), then the element-wise
        // stage.  Query rows >= S need no select: their Q / dO rows, lse and delta are zero in LDS, so P = 1 meets dO = 0 and dS = 0; key columns >= S
        // (zero K / V rows) only reach output columns that are never stored.
        uint4 rq_[G::KS], ro[G::KS];
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) { rq_[ks] = frag_rows<T, DH>(Qr, fr, ks, kg); ro[ks] = frag_rows<T, DH>(Or, fr, ks, kg); }
#pragma unroll                                          // (round 4: rolled, the loop rebuilt its LDS offsets every group: unrolled -3 % on the backward at S = 197)
        for (int g = 0; g < NG; ++g) {
            constexpr int HALF = G::ND >= 4 ? 2 : G::ND;
            uint4 tf[2][HALF];                                  // transposed fragments, two slots of HALF head-column tiles
            uint32_t pw[4], dw[4];                            // the step's P and dS operand chunks, packed tile by tile
#pragma unroll
            for (int t = 0; t < G::TPS; ++t) {
                const int q0 = g * G::KSTEP + t * 16;         // tile rows = queries q0 + 4 kg + r, column = key rk
                // (del_s holds -delta: dP - delta comes straight from the matrix pipe, the accumulator starts there; lse_s holds -lse log2(e))
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_s + q0 + kg * 4), d4 = *reinterpret_cast<const f32x4_t*>(del_s + q0 + kg * 4);
                f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dpt = d4;
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) {
                    Mma<T>::mma(rq_[ks], kf[ks], sc);
                    Mma<T>::mma(ro[ks], vf[ks], dpt);
                }
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int qn0 = q0 + 16 < SP ? q0 + 16 : 0;       // (the last tile wraps to rows that are simply not used)
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) { rq_[ks] = frag_rows<T, DH>(Qr, qn0 + fr, ks, kg); ro[ks] = frag_rows<T, DH>(Or, qn0 + fr, ks, kg); }
                }
                if (t == G::TPS - 1) {
#pragma unroll
                    for (int j = 0; j < HALF; ++j) tf[0][j] = frag_T<T, DH>(Oimg, SPT, j * 16, g, lane);
                }
                if constexpr (KM) {                            // rows = queries q0 + 4 kg + r, this lane's key = 16 kt + fr
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (causal ? q0 + kg * 4 + r < first : first >= S) sc[r] = 0.f;
                }
                f32x4_t pv = __builtin_elementwise_fma(sc, c2v, l4), dsv;
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(pv[r]);
                if constexpr (KM) {
                    const int key = kt * 16 + fr;
                    const bool kon = km[key] != 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = q0 + kg * 4 + r;
                        const bool empty = causal ? q < first : first >= S;
                        if (!empty && !(kon && (!causal || key <= q))) pv[r] = 0.f;
                    }
                }
                if (dr.thr16) {                               // here the tile's 4 rows are 4 QUERIES at one key: one hash each
                    const int ol = opaque_lane(lane), ork = kt * 16 + (ol & 15), okg = ol >> 4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float keepf = dropout_keep(dr.seed, dr.site, drop_idx(blockIdx.x, q0 + okg * 4 + r, ork), dr.thr16) ? dr.keep_scale : 0.f;
                        dsv[r] = pv[r] * ((dpt[r] - d4[r]) * keepf + d4[r]);
                        pv[r] *= keepf;
                    }
                } else {
                    dsv = pv * dpt;                                                     // (x scale: applied to dK at the end)
                }
                if constexpr (sizeof(T) == 2) {
                    pw[2 * t] = pack2_bf16(pv[0], pv[1]); pw[2 * t + 1] = pack2_bf16(pv[2], pv[3]);
                    dw[2 * t] = pack2_bf16(dsv[0], dsv[1]); dw[2 * t + 1] = pack2_bf16(dsv[2], dsv[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pw[r] = __float_as_uint(pv[r]); dw[r] = __float_as_uint(dsv[r]); }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const uint4 pf = make_uint4(pw[0], pw[1], pw[2], pw[3]), dsf = make_uint4(dw[0], dw[1], dw[2], dw[3]);
            // dV^T += dO^T P, dK^T += Q^T dS, HALF head-column tiles at a time: a slot is refilled as soon as its products are issued
            if constexpr (G::ND > HALF) {
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[1][j] = frag_T<T, DH>(Oimg, SPT, (HALF + j) * 16, g, lane);
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[0][j], pf, dv[j]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < HALF; ++j) tf[0][j] = frag_T<T, DH>(Qimg, SPT, j * 16, g, lane);
            if constexpr (G::ND > HALF) {
#pragma unroll
                for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[1][j], pf, dv[HALF + j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[1][j] = frag_T<T, DH>(Qimg, SPT, (HALF + j) * 16, g, lane);
            }
#pragma unroll
            for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[0][j], dsf, dk[j]);
            if constexpr (G::ND > HALF) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[1][j], dsf, dk[HALF + j]);
            }
            if constexpr (G::ND > 2 * HALF) {                       // head width 128: the column tiles behind the first four, unscheduled
#pragma unroll
                for (int j = 2 * HALF; j < G::ND; ++j) {
                    Mma<T>::mma(frag_T<T, DH>(Oimg, SPT, j * 16, g, lane), pf, dv[j]);
                    Mma<T>::mma(frag_T<T, DH>(Qimg, SPT, j * 16, g, lane), dsf, dk[j]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int dt = 0; dt < G::ND; ++dt) dk[dt] *= f32x4_t{scale, scale, scale, scale};
        store_block16<T, DH>(stg, dk, dkview, kt * 16, lane);
        store_block16<T, DH>(stg, dv, dvview, kt * 16, lane);
    }
}

template <typename T, int DH, int NKT, bool KM = false>
__global__ void __launch_bounds__(WG<NKT>::NTHR, DH > 64 ? 2 : 4) attn_long_dkdv_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                             const T* __restrict__ dctx, int ldo, const float* __restrict__ lse,
                                                             const float* __restrict__ delta, T* __restrict__ dqkv,
                                                             int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
    dkdv_body<T, DH, NKT, KM>(qkv, ld, q_off, k_off, v_off, dctx, ldo, lse, delta, dqkv, S, nh, scale, dr, kmask, causal);
}

// Both halves of the backward in ONE launch per (item, head): dq + delta first, then -- same workgroup, same LDS -- dk and dv.  The second half's reads of
// Q / K / V / dO repeat what this workgroup read tens of microseconds earlier (L2 / Infinity Cache instead of HBM: as two launches over all items the dkdv
// kernel found nothing of the dq kernel's stream left in a 256-MB cache at ViT-B/16's 500 MB), and delta goes through global memory within the workgroup.
template <typename T, int DH, int NKT, bool KM = false>
__global__ void __launch_bounds__(WG<NKT>::NTHR, DH > 64 ? 2 : 4) attn_long_bwd_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                            const T* __restrict__ dctx, int ldo, const T* __restrict__ octx,
                                                            const float* __restrict__ lse, float* delta, T* dqkv,
                                                            int S, int nh, float scale, Drop dr, const float* __restrict__ kmask, int causal) {
    dq_body<T, DH, NKT, KM>(qkv, ld, q_off, k_off, v_off, dctx, ldo, octx, lse, delta, dqkv, S, nh, scale, dr, kmask, causal);
    // delta written by this workgroup's waves is read by others of it below: workgroup-scope visibility (the fence) + the barrier.  Two things this relies on:
    // dq_body has NO early return (every wave of the workgroup reaches the barrier; a bounds exit would have to become a predicate), and the workgroup runs
    // on one CU (CU mode: its waves share the L1 the fence makes coherent).  tests/test_kernels_gpu.py forces this form at S = 197 as well (A4R_ATTN_BWD_FUSED=1).
    __threadfence_block();
    __syncthreads();                                        // ... and the LDS images of the first half are dead
    dkdv_body<T, DH, NKT, KM>(qkv, ld, q_off, k_off, v_off, dctx, ldo, lse, delta, dqkv, S, nh, scale, dr, kmask, causal);
}

int nkt_for(int S) { return S <= 32 ? 2 : S <= 64 ? 4 : S <= 128 ? 8 : S <= 224 ? 14 : 16; }
// kt_partial_lo<NKT>() (which key tiles can reach past S) restates this map: a launch whose S does not fit its instantiation's assumption is refused
template <int NKT> bool s_fits(int S) { return S <= NKT * 16 && S > 16 * kt_partial_lo<NKT>(); }

template <typename T, int DH, int NKT> size_t lds_fwd() { return lds_main_fwd<T, DH, NKT>() + WG<NKT>::NWAVE * STG<T, DH>::BYTES; }
template <typename T, int DH, int NKT> size_t lds_dq() { return lds_main_dq<T, DH, NKT>() + WG<NKT>::NWAVE * STG<T, DH>::BYTES; }
template <typename T, int DH, int NKT> size_t lds_dkdv() { return lds_main_dkdv<T, DH, NKT>() + WG<NKT>::NWAVE * STG<T, DH>::BYTES; }
constexpr size_t LDS_MAX = 160 * 1024;

template <typename K> int set_lds(K kernel, size_t bytes) {
    if (bytes > LDS_MAX) return A4R_EINVAL;
    if (bytes > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return A4R_ELAUNCH;
    return A4R_OK;
}

Drop drop_of(const a4r_attn_t* a) {
    return Drop{a->drop_seed, a->drop_site, a4r_thr16(a->drop_p), a4r_keep_scale(a->drop_p)};
}

template <typename T, int DH, int NKT, bool KM> int run_fwd_km(hipStream_t s, const a4r_attn_t* a, float* lse) {
    const size_t lds = lds_fwd<T, DH, NKT>();
    if (int rc = set_lds(attn_long_fwd_kernel<T, DH, NKT, KM>, lds)) return rc;
    hipLaunchKernelGGL((attn_long_fwd_kernel<T, DH, NKT, KM>), dim3(a->n_items * a->n_heads), dim3(WG<NKT>::NTHR), lds, s, (const T*)a->qkv, a->ld, a->q_off,
                       a->k_off, a->v_off, (T*)a->out, a->ldo, lse, a->S, a->n_heads, a->scale, drop_of(a), (const float*)a->key_mask, a->causal);
    return a4r_launch_status();
}
template <typename T, int DH, int NKT> int run_fwd(hipStream_t s, const a4r_attn_t* a, float* lse) {
    if (!s_fits<NKT>(a->S)) return A4R_EINVAL;
    if (a->key_mask) return run_fwd_km<T, DH, NKT, true>(s, a, lse);
    return run_fwd_km<T, DH, NKT, false>(s, a, lse);
}
template <typename T, int DH, int NKT, bool KM> int run_bwd_km(hipStream_t s, const a4r_attn_t* a, const float* lse, float* delta) {
    const size_t l1 = lds_dq<T, DH, NKT>(), l2 = lds_dkdv<T, DH, NKT>();
    if (int rc = set_lds(attn_long_dq_kernel<T, DH, NKT, KM>, l1)) return rc;
    if (int rc = set_lds(attn_long_dkdv_kernel<T, DH, NKT, KM>, l2)) return rc;
    const dim3 grid(a->n_items * a->n_heads), block(WG<NKT>::NTHR);
    if constexpr (sizeof(T) == 2 && DH == 64 && NKT == 14 && !KM) {
        // ViT-B/16's 197 tokens (129 .. 224): the one-pass backward (round 6, a4r_attn_long1.hip).  A4R_ATTN_BWD_ONEPASS=0 keeps the two launches (A/B runs, tests).
        static const int one = []{ const char* e = getenv("A4R_ATTN_BWD_ONEPASS"); return e ? atoi(e) != 0 : 1; }();
        if (one) return a4r_attn_long_bwd1_launch(s, a, lse);
    }
    // one launch up to 64 tokens (ViT-MAE's 50: 56 -> 48 us, the step -0.6 % bf16 / -0.9 % fp8); above, the two launches stay -- at ViT-B/16's 197 the one-launch
    // form measured the same (351 vs 340 - 354 us, step -0.1 %) with 7 spilled registers instead of 2 (profiles/r05_q_attn_bwd_fused.txt).  A4R_ATTN_BWD_FUSED=0 / 1 forces either.
    static const int knob = []{ const char* e = getenv("A4R_ATTN_BWD_FUSED"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
    const bool fused = knob < 0 ? NKT <= 4 : knob != 0;
    if (fused) {
        const size_t l = l1 > l2 ? l1 : l2;
        if (int rc = set_lds(attn_long_bwd_kernel<T, DH, NKT, KM>, l)) return rc;
        hipLaunchKernelGGL((attn_long_bwd_kernel<T, DH, NKT, KM>), grid, block, l, s, (const T*)a->qkv, a->ld, a->q_off, a->k_off, a->v_off,
                           (const T*)a->dout, a->ldo, (const T*)a->out, lse, delta, (T*)a->dqkv, a->S, a->n_heads, a->scale, drop_of(a), (const float*)a->key_mask, a->causal);
        return a4r_launch_status();
    }
    hipLaunchKernelGGL((attn_long_dq_kernel<T, DH, NKT, KM>), grid, block, l1, s, (const T*)a->qkv, a->ld, a->q_off, a->k_off, a->v_off,
                       (const T*)a->dout, a->ldo, (const T*)a->out, lse, delta, (T*)a->dqkv, a->S, a->n_heads, a->scale, drop_of(a), (const float*)a->key_mask, a->causal);
    hipLaunchKernelGGL((attn_long_dkdv_kernel<T, DH, NKT, KM>), grid, block, l2, s, (const T*)a->qkv, a->ld, a->q_off, a->k_off, a->v_off,
                       (const T*)a->dout, a->ldo, lse, (const float*)delta, (T*)a->dqkv, a->S, a->n_heads, a->scale, drop_of(a), (const float*)a->key_mask, a->causal);
    return a4r_launch_status();
}
template <typename T, int DH, int NKT> int run_bwd(hipStream_t s, const a4r_attn_t* a, const float* lse, float* delta) {
    if (!s_fits<NKT>(a->S)) return A4R_EINVAL;
    if (a->key_mask) return run_bwd_km<T, DH, NKT, true>(s, a, lse, delta);
    return run_bwd_km<T, DH, NKT, false>(s, a, lse, delta);
}

int check(const a4r_attn_t* a, bool bwd) {
    if (!a || !a->qkv || a->n_items <= 0 || a->S <= 0 || a->S > 256 || (a->dh != 64 && a->dh != 32 && a->dh != 128) || a->n_heads <= 0) return A4R_EINVAL;
    if (a->dh == 128 && (a->dtype != A4R_F32 || a->S > 128)) return A4R_EINVAL;     // head width 128: fp32, up to 128 tokens (two [128, 128] fp32 images = 128 KB of LDS)
    if (a->offsets || (a->causal && !a->key_mask)) return A4R_EINVAL;              // no packed items; causal only together with a key mask (the user tower); key_mask (fp32 [n_items, S], optional)
    if (a->key_mask && (reinterpret_cast<uintptr_t>(a->key_mask) & 3u)) return A4R_EINVAL;
    if (a->drop_p < 0.f || a->drop_p >= 1.f) return A4R_EINVAL;
    if ((int64_t)a->n_items * a->n_heads >= (1ll << 40)) return A4R_EINVAL;        // dropout counter: 40 + 8 + 8 bits
    if (a->dtype != A4R_BF16 && a->dtype != A4R_F32) return A4R_EINVAL;
    const int es = a->dtype == A4R_BF16 ? 2 : 4;
    if ((a->ld * es) % 16 || (a->ldo * es) % 16 || (a->q_off * es) % 16 || (a->k_off * es) % 16 || (a->v_off * es) % 16) return A4R_EINVAL;
    if (reinterpret_cast<uintptr_t>(a->qkv) & 15u) return A4R_EINVAL;
    if (!bwd && (!a->out || (reinterpret_cast<uintptr_t>(a->out) & 15u))) return A4R_EINVAL;
    if (bwd && (!a->dout || !a->dqkv || !a->out || ((reinterpret_cast<uintptr_t>(a->dout) | reinterpret_cast<uintptr_t>(a->dqkv) | reinterpret_cast<uintptr_t>(a->out)) & 15u)))
        return A4R_EINVAL;                                                         // bwd reads the forward output too (a->out, same ldo)
    return A4R_OK;
}

#define A4R_NKT_SWITCH(T_, D_, CALL_)                   \
    switch (nkt_for(a->S)) {                            \
        case 2: return CALL_(T_, D_, 2);                \
        case 4: return CALL_(T_, D_, 4);                \
        case 8: return CALL_(T_, D_, 8);                \
        case 14: return CALL_(T_, D_, 14);              \
        default: return CALL_(T_, D_, 16);              \
    }
#define A4R_DH_SWITCH(T_, CALL_)                                            \
    if (a->dh == 64) { A4R_NKT_SWITCH(T_, 64, CALL_) } else { A4R_NKT_SWITCH(T_, 32, CALL_) }
// head width 128 (fp32, S <= 128: the user tower at the parser's default --embedding_dim 256 with two heads and --max_seq_len above 32)
#define A4R_DH128_SWITCH(CALL_)                         \
    if (a->dh == 128) switch (nkt_for(a->S)) {          \
        case 2: return CALL_(float, 128, 2);            \
        case 4: return CALL_(float, 128, 4);            \
        default: return CALL_(float, 128, 8);           \
    }

}  // namespace

extern "C" int a4r_attn_long_fwd(void* stream, const a4r_attn_t* a, float* lse) {
    if (int rc = check(a, false)) return rc;
    if (!lse) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define A4R_F(T_, D_, N_) run_fwd<T_, D_, N_>(s, a, lse)
    if (a->dtype == A4R_BF16) { A4R_DH_SWITCH(bf16_t, A4R_F) }
    A4R_DH128_SWITCH(A4R_F)
    A4R_DH_SWITCH(float, A4R_F)
#undef A4R_F
}

extern "C" int a4r_attn_long_bwd(void* stream, const a4r_attn_t* a, const float* lse, float* delta_ws) {
    if (int rc = check(a, true)) return rc;
    if (!lse || !delta_ws) return A4R_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define A4R_B(T_, D_, N_) run_bwd<T_, D_, N_>(s, a, lse, delta_ws)
    if (a->dtype == A4R_BF16) { A4R_DH_SWITCH(bf16_t, A4R_B) }
    A4R_DH128_SWITCH(A4R_B)
    A4R_DH_SWITCH(float, A4R_B)
#undef A4R_B
}
