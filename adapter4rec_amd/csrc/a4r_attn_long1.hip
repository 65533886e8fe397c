// a4r_attn_long_bwd, ONE pass (round 6; bf16, head width 64, 129 .. 224 tokens: ViT-B/16's 197 under Downstream/CV/model/encoders.py:21-32).
//
// the two-launch backward (a4r_attn_long.hip: dq, then dk/dv) reads qkv + dO twice and forms S, P, dP, dS twice per
// (item, head) -- 1.55 x its algorithmic traffic, 7 products instead of 5, two staging prologues that nothing overlaps.  Here P and dS are formed
// ONCE per (item, head), every byte is read once, and the HBM reads of the next pair run under the arithmetic of the current one:
//
//   * PERSISTENT 8-wave workgroups, one per CU (146 KB of LDS), walk the (item, head) pairs.  Q and dO of a pair live in LDS as row-major swizzled
//     images (the dk/dv kernel's layout) in one of TWO image sets: while a pair computes from one set, LDS-DMA (global_load_lds_dwordx4, no
//     registers) fills the other with the next pair's rows.  Rows >= S of both sets are zeroed once and never written again.
//   * wave w < NKT / 2 owns the 32 keys [32 w, 32 w + 32): their K / V rows are row fragments in registers (requested during the previous pair's
//     last products), dK / dV of those keys stay in 64 accumulator registers for the whole pair.
//   * per step of 32 queries an owner forms S = Q K^T and dP = dO V^T (4 tiles of 16 x 16), P, dS, feeds dV^T += dO^T P and dK^T += Q^T dS
//     from its registers and writes dS -- 8 bytes per lane and tile -- into a [keys][32 queries] LDS buffer.  After ONE workgroup barrier per step
//     (the buffer is double) wave c computes one 16 x 16 tile of dQ^T = K^T dS^T over ALL keys: 7 MFMAs whose A operands (its 16 head columns
//     of K^T, all 224 keys) sit in 28 registers for the whole pair and whose B operands are transposed 8-byte reads (ds_read_b64_tr_b16) of the
//     buffer.  No atomics (ds_add_f32 measured 165 cycles per wave instruction: profiles/r06_b_attn_onepass.txt), no dQ image, no write-out pass.
//   * delta = dO . O is formed per pair by all threads from the O rows they requested a pair ahead and the dO rows in LDS.
// HBM traffic = the algorithmic 5 reads + 3 writes of [S][64] per pair.  Dropout counters = the two-launch form's (bit-compatible masks).
#include "a4r_attn_long.h"

namespace {

template <int NKT> struct OnePass {
    using G = Geo<bf16_t, 64>;
    static constexpr int SP = NKT * 16, NG = SP / 32, NPAIR = NKT / 2, NW = 8, NTHR = NW * 64;
    static constexpr int IMG = SP * 128;                                        // one [SP][64] bf16 image
    static constexpr int OFF_STAT = 4 * IMG, OFF_R = OFF_STAT + 2 * SP * (int)sizeof(float);
    static constexpr int DSB = NPAIR * 2048;                                    // one dS buffer: [NPAIR owners][4 tiles][16 keys][16 queries] bf16
    static constexpr int R_BYTES = NPAIR * 4096;                                // K blocks (pair start) / 2 dS buffers (steps) / store staging (pair end)
    static constexpr int BYTES = OFF_R + R_BYTES;
    static constexpr int NPIECE = 2 * SP / 8;                                   // 1-KiB DMA pieces of a pair's two images (8 rows each)
    static_assert(NKT % 2 == 0 && NPAIR <= NW && 2 * DSB <= R_BYTES && NW * 2048 <= R_BYTES + 2048 && BYTES <= 160 * 1024 && NPIECE % NW == 0, "one-pass backward geometry");
};

#ifdef A4R_OP_STAMP
// diagnostic build only (tools/op_stamps.py): s_memtime of wave 0 of every workgroup at seven points of its THIRD pair
__device__ unsigned long long g_a4r_op_stamps[256 * 8];
#define A4R_OP_ST(k_) if (threadIdx.x == 0 && n_done == 2 && blockIdx.x < 256) g_a4r_op_stamps[blockIdx.x * 8 + (k_)] = __builtin_amdgcn_s_memtime();
#else
#define A4R_OP_ST(k_)
#endif

A4R_DEV void glds16(const void* base, uint32_t voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

template <int NKT, bool DROP>     // DROP: probability dropout (a ViT configured with attention dropout); false: no dropout code in the step loop
__global__ void __launch_bounds__(OnePass<NKT>::NTHR, 2) attn_long_bwd1_kernel(const bf16_t* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                                              const bf16_t* __restrict__ dctx, int ldo, const bf16_t* __restrict__ octx,
                                                                              const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                                              int S, int nh, int n_pairs, float scale, Drop dr) {
    using T = bf16_t;
    using OP = OnePass<NKT>;
    using G = Geo<T, 64>;
    constexpr int DH = 64, SP = OP::SP, SPT = SP + 8, NG = OP::NG, NTHR = OP::NTHR, NPAIR = OP::NPAIR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lse_s = reinterpret_cast<float*>(smem + OP::OFF_STAT);
    float* del_s = lse_s + SP;
    char* R = smem + OP::OFF_R;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool owner = wave < NPAIR;                        // (wave-uniform) this wave owns 32 keys
    const int ct = wave >> 2, cdt = wave & 3;               // consumer role: dQ^T tile (head columns 16 cdt .., queries 16 ct .. of a step)
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    const int G_ = (int)gridDim.x;

    // ---- once per workgroup: both image sets start at zero (rows >= S are never written again)
    for (int i = tid; i < 4 * OP::IMG / 16; i += NTHR) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // LDS-DMA of pair `pr` into image set `set`: piece i (1 KiB = 8 rows x 128 B) of [Q image | dO image]; lane -> row 8 i + (lane >> 3), LDS chunk slot
    // lane & 7 <- global chunk (lane & 7) ^ swz(row).  Wave w issues pieces w, w + 8, ...
    auto dma_pair = [&](int pr, int set) {
        const int item = pr / nh, h = pr % nh;
        const T* qb = qkv + (size_t)item * S * ld + h * DH + q_off;
        const T* ob = dctx + (size_t)item * S * ldo + h * DH;
#pragma unroll
        for (int j = 0; j < OP::NPIECE / OP::NW; ++j) {
            const int i = wave + j * OP::NW;
            const bool second = i >= SP / 8;
            const int piece = second ? i - SP / 8 : i;
            const int row = piece * 8 + (lane0 >> 3), slot = lane0 & 7;
            const uint32_t voff = (uint32_t)row * (uint32_t)((second ? ldo : ld) * 2) + (uint32_t)((slot ^ G::swz(row)) << 4);
            const uint32_t dst = lds_base + (uint32_t)(set * 2 * OP::IMG + (second ? OP::IMG : 0) + piece * 1024);
            if (row < S) glds16(second ? (const void*)ob : (const void*)qb, voff, dst);
        }
    };
    constexpr int NIT = (SP * G::CPR + NTHR - 1) / NTHR;
    uint4 ov[NIT];                                           // O chunks of the NEXT pair (delta = dO . O)
    auto request_o = [&](int pr) {
        const int item = pr / nh, h = pr % nh;
        const RowsView oview = RowsView::make(octx + (size_t)item * S * ldo + h * DH, ldo, S, DH);
        const uint32_t voff = (uint32_t)(tid / G::CPR) * oview.ldb + (uint32_t)(tid % G::CPR) * 16u;
#pragma unroll
        for (int it = 0; it < NIT; ++it) ov[it] = oview.load(voff, (uint32_t)(it * (NTHR / G::CPR)) * oview.ldb);
    };
    uint4 kf[2][G::KS], vf[2][G::KS];
    auto request_kv = [&](int pr) {
        const int item = pr / nh, h = pr % nh;
        const T* base = qkv + (size_t)item * S * ld + h * DH;
        const RowsView kview = RowsView::make(base + k_off, ld, S, DH), vview = RowsView::make(base + v_off, ld, S, DH);
        const int ol = opaque_lane(lane0), fr = ol & 15, kg = ol >> 4;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                kf[kk][ks] = kview.row_chunk((2 * wave + kk) * 16 + fr, ks * 4 + kg);
                vf[kk][ks] = vview.row_chunk((2 * wave + kk) * 16 + fr, ks * 4 + kg);
            }
    };

    int pair = blockIdx.x, cur = 0;
    [[maybe_unused]] int n_done = 0;
    if (pair < n_pairs) {
        dma_pair(pair, 0);
        request_o(pair);
        if (owner) request_kv(pair);
    }
    const float c2 = scale * 1.44269504088896f;
    const f32x4_t c2v = {c2, c2, c2, c2};
    for (; pair < n_pairs; pair += G_, cur ^= 1, ++n_done) {
        const int item = pair / nh, h = pair % nh;
        char* Qr = smem + cur * 2 * OP::IMG;
        char* Or = Qr + OP::IMG;
        const RowsView dqview = RowsView::make(dqkv + (size_t)item * S * ld + q_off + h * DH, ld, S, DH);
        A4R_OP_ST(0)
        // ---- B0: this wave's DMA pieces, O rows and K / V rows have landed; every wave is done with the previous pair (R, the other image set)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        A4R_OP_ST(1)
        {
            // delta[row] = dO[row] . O[row]: a row's 8 chunks sit on 8 consecutive lanes; del_s holds -delta (rows >= S: zero rows, 0); lse_s = -lse log2(e)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int id = tid + it * NTHR, row = id / G::CPR, ch = id % G::CPR;
                float d = 0.f;
                if (id < SP * G::CPR) {
                    const uint4 dov = *reinterpret_cast<const uint4*>(Or + row * G::ROWB + ((ch ^ G::swz(row)) << 4));
                    const uint32_t a4[4] = {ov[it].x, ov[it].y, ov[it].z, ov[it].w}, b4[4] = {dov.x, dov.y, dov.z, dov.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_raw_t, a4[e]), __builtin_bit_cast(bf16x2_raw_t, b4[e]), d, false);
                }
                d += __shfl_xor(d, 1, 64);
                d += __shfl_xor(d, 2, 64);
                d += __shfl_xor(d, 4, 64);
                if ((tid & 7) == 0 && id < SP * G::CPR) del_s[row] = -d;
            }
            static_assert(SP <= NTHR, "one row statistic per thread");
            if (tid < SP) lse_s[tid] = tid < S ? lse[((size_t)item * nh + h) * S + tid] * -1.44269504088896f : 0.f;
            if (owner) {                                     // this wave's 32 K rows, row-major swizzled, for the consumers' transposed reads
                const int fr = lane0 & 15, kg = lane0 >> 4;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) {
                        const int row = kk * 16 + fr;
                        *reinterpret_cast<uint4*>(R + wave * 4096 + row * G::ROWB + (((ks * 4 + kg) ^ G::swz(row)) << 4)) = kf[kk][ks];
                    }
            }
        }
        __syncthreads();                                     // B1
        uint4 ktc[NPAIR];                                    // K^T, head columns 16 cdt .. + 15, of ALL keys: A operands of this wave's dQ^T tile
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) ktc[p] = frag_tr<DH>(R + p * 4096, cdt * 16, 0, lane0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                     // B2: R is free for the dS buffers
        A4R_OP_ST(2)
        const int nxt = pair + G_;
        if (nxt < n_pairs) dma_pair(nxt, cur ^ 1);             // the next pair's rows stream into the other image set under this pair's arithmetic
        f32x4_t dk[2][G::ND], dv[2][G::ND];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int dt = 0; dt < G::ND; ++dt) { dk[kk][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[kk][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
        uint4 rq_[G::KS], ro[G::KS];
        if (owner) {
            const int fr = lane0 & 15, kg = lane0 >> 4;
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) { rq_[ks] = frag_rows<T, DH>(Qr, fr, ks, kg); ro[ks] = frag_rows<T, DH>(Or, fr, ks, kg); }
        }
        const int kt0 = 2 * wave;
#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const int lane = opaque_lane(lane0), fr = lane & 15, kg = lane >> 4;
            char* dsb = R + (g & 1) * OP::DSB;
            uint32_t pw[2][4], dw[2][4];                     // P and dS operand chunks of the step (key on the lane), per key tile
            constexpr int HALF = 2;
            uint4 tf[2][HALF];
            if (owner) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int q0 = g * 32 + t * 16;          // tile rows = queries q0 + 4 kg + r, column = key (kt0 + kk) * 16 + fr
                    const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_s + q0 + kg * 4), d4 = *reinterpret_cast<const f32x4_t*>(del_s + q0 + kg * 4);
                    f32x4_t sc[2], dpt[2];
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) { sc[kk] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dpt[kk] = d4; }
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            Mma<T>::mma(rq_[ks], kf[kk][ks], sc[kk]);
                            Mma<T>::mma(ro[ks], vf[kk][ks], dpt[kk]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        const int qn0 = q0 + 16 < SP ? q0 + 16 : 0;   // (the last tile wraps to rows that are simply not used)
#pragma unroll
                        for (int ks = 0; ks < G::KS; ++ks) { rq_[ks] = frag_rows<T, DH>(Qr, qn0 + fr, ks, kg); ro[ks] = frag_rows<T, DH>(Or, qn0 + fr, ks, kg); }
                    }
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        f32x4_t pv = __builtin_elementwise_fma(sc[kk], c2v, l4), dsv;
#pragma unroll
                        for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(pv[r]);
                        if (DROP && dr.thr16) {               // the tile's 4 rows are 4 QUERIES at one key: one hash each (the dk/dv kernel's counters)
                            const int ol = opaque_lane(lane), ork = (kt0 + kk) * 16 + (ol & 15), okg = ol >> 4;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float keepf = dropout_keep(dr.seed, dr.site, drop_idx(pair, q0 + okg * 4 + r, ork), dr.thr16) ? dr.keep_scale : 0.f;
                                dsv[r] = pv[r] * ((dpt[kk][r] - d4[r]) * keepf + d4[r]);
                                pv[r] *= keepf;
                            }
                        } else {
                            dsv = pv * dpt[kk];                // (x scale: applied to dK / dQ at the end)
                        }
                        pw[kk][2 * t] = pack2_bf16(pv[0], pv[1]); pw[kk][2 * t + 1] = pack2_bf16(pv[2], pv[3]);
                        dw[kk][2 * t] = pack2_bf16(dsv[0], dsv[1]); dw[kk][2 * t + 1] = pack2_bf16(dsv[2], dsv[3]);
                        // dS tile (kk, t) as [16 keys][16 queries] bf16, 32 bytes per key row: this lane's 4 queries of key fr
                        *reinterpret_cast<uint2*>(dsb + wave * 2048 + (kk * 2 + t) * 512 + fr * 32 + kg * 8) = make_uint2(dw[kk][2 * t], dw[kk][2 * t + 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the transposed fragments of the step's dV products (shared by the wave's two key tiles) are requested before the barrier
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[0][j] = frag_T<T, DH>(Or, SPT, j * 16, g, lane);
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[1][j] = frag_T<T, DH>(Or, SPT, (HALF + j) * 16, g, lane);
            }
            __syncthreads();                                 // the step's dS tiles of every owner are in the buffer (its other half is being re-filled next step)
            {
                // dQ^T tile (head columns 16 cdt .., queries g * 32 + 16 ct ..) over all keys; B chunk of owner p: keys (j >> 2) * 16 + 4 kg + (j & 3) of its 32
                const int qd = (lane >> 2) & 3, pp = lane & 3;
                const char* b0 = dsb + (0 * 2 + ct) * 512 + (4 * kg + qd) * 32 + pp * 8;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int p = 0; p < NPAIR; ++p) {
                    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(b0 + p * 2048));
                    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(b0 + p * 2048 + 1024));
                    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
                    Mma<T>::mma(ktc[p], make_uint4(l2.x, l2.y, h2.x, h2.y), acc);
                }
                acc *= f32x4_t{scale, scale, scale, scale};
                // lane (query fr, kg): head columns 16 cdt + 4 kg .. + 3 of row g * 32 + 16 ct + fr (a row >= S is dropped by the view)
                dqview.store8((uint32_t)(g * 32 + ct * 16 + fr) * dqview.ldb + (uint32_t)(cdt * 16 + kg * 4) * 2u,
                              make_uint2(pack2_bf16(acc[0], acc[1]), pack2_bf16(acc[2], acc[3])));
            }
            if (owner) {
                // dV^T += dO^T P for both key tiles
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const uint4 pf = make_uint4(pw[kk][0], pw[kk][1], pw[kk][2], pw[kk][3]);
#pragma unroll
                    for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[0][j], pf, dv[kk][j]);
#pragma unroll
                    for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[1][j], pf, dv[kk][HALF + j]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[0][j] = frag_T<T, DH>(Qr, SPT, j * 16, g, lane);
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[1][j] = frag_T<T, DH>(Qr, SPT, (HALF + j) * 16, g, lane);
                // dK^T += Q^T dS for both key tiles
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const uint4 dsf = make_uint4(dw[kk][0], dw[kk][1], dw[kk][2], dw[kk][3]);
#pragma unroll
                    for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[0][j], dsf, dk[kk][j]);
#pragma unroll
                    for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[1][j], dsf, dk[kk][HALF + j]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        A4R_OP_ST(3)
        // K / V fragments are dead: the next pair's rows are requested under the dk / dv stores.  (NOT inside the step loop: a load whose result is
        // live around the loop's back edge makes hipcc wait vmcnt(0) in EVERY step -- i.e. for this wave's LDS-DMA of the next pair as well)
        if (nxt < n_pairs) {
            request_o(nxt);                                  // (16 registers: requested here, not before the steps, where every register is spoken for)
            if (owner) request_kv(nxt);
        }
        if (owner) {
            // (after the last step's barrier the second dS buffer is dead: owner w stages its blocks in that buffer's own 2 KB)
            const T* base = dqkv + (size_t)item * S * ld + h * DH;
            const RowsView dkview = RowsView::make(base + k_off, ld, S, DH), dvview = RowsView::make(base + v_off, ld, S, DH);
            char* stg = R + OP::DSB + wave * 2048;
            static_assert((OnePass<NKT>::NG & 1) == 1, "the last step uses the FIRST dS buffer");
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int dt = 0; dt < G::ND; ++dt) dk[kk][dt] *= f32x4_t{scale, scale, scale, scale};
                store_block16<T, DH>(stg, dk[kk], dkview, (kt0 + kk) * 16, lane0);
                store_block16<T, DH>(stg, dv[kk], dvview, (kt0 + kk) * 16, lane0);
            }
        }
        A4R_OP_ST(4)
    }
}

}  // namespace

#ifdef A4R_OP_STAMP
extern "C" int a4r_debug_op_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_a4r_op_stamps), sizeof(g_a4r_op_stamps)) == hipSuccess ? 0 : -2;
}
#endif

// called by a4r_attn_long_bwd (a4r_attn_long.hip) for bf16, head width 64, 129 .. 224 tokens, no key mask; the arguments were checked there
int a4r_attn_long_bwd1_launch(hipStream_t s, const a4r_attn_t* a, const float* lse) {
    constexpr int NKT = 14;
    using OP = OnePass<NKT>;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return A4R_ELAUNCH;
        n_cu = prop.multiProcessorCount;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_long_bwd1_kernel<NKT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, OP::BYTES) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(attn_long_bwd1_kernel<NKT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, OP::BYTES) != hipSuccess) return A4R_ELAUNCH;
    }
    const int n_pairs = a->n_items * a->n_heads;
    const int grid = n_pairs < n_cu ? n_pairs : n_cu;
    const Drop dr{a->drop_seed, a->drop_site, a4r_thr16(a->drop_p), a4r_keep_scale(a->drop_p)};
    if (dr.thr16)
        hipLaunchKernelGGL((attn_long_bwd1_kernel<NKT, true>), dim3(grid), dim3(OP::NTHR), OP::BYTES, s, (const bf16_t*)a->qkv, a->ld, a->q_off, a->k_off, a->v_off,
                           (const bf16_t*)a->dout, a->ldo, (const bf16_t*)a->out, lse, (bf16_t*)a->dqkv, a->S, a->n_heads, n_pairs, a->scale, dr);
    else
        hipLaunchKernelGGL((attn_long_bwd1_kernel<NKT, false>), dim3(grid), dim3(OP::NTHR), OP::BYTES, s, (const bf16_t*)a->qkv, a->ld, a->q_off, a->k_off, a->v_off,
                           (const bf16_t*)a->dout, a->ldo, (const bf16_t*)a->out, lse, (bf16_t*)a->dqkv, a->S, a->n_heads, n_pairs, a->scale, dr);
    return a4r_launch_status();
}
