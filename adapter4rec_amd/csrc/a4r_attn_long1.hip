// a4r_attn_long_bwd, ONE pass (round 6; bf16, head width 64, 129 .. 224 tokens: ViT-B/16's 197 under Downstream/CV/model/encoders.py:21-32).
//
// the two-launch backward (a4r_attn_long.hip: dq, then dk/dv) reads qkv + dO twice and forms S, P, dP, dS twice per
// (item, head) -- 1.55 x its algorithmic traffic, 7 products instead of 5, two staging prologues that nothing overlaps.  Here P and dS are formed
// ONCE per (item, head), every byte is read once, and the HBM reads of the next pair run under the arithmetic of the current one:
//
//   * PERSISTENT 16-wave workgroups, one per CU (146 KB of LDS, 4 waves per SIMD), walk the (item, head) pairs.  Q and dO of a pair live in LDS as row-major swizzled
//     images (the dk/dv kernel's layout) in one of TWO image sets: while a pair computes from one set, LDS-DMA (global_load_lds_dwordx4, no
//     registers) fills the other with the next pair's rows.  Rows >= S of both sets are zeroed once and never written again.
//   * PRODUCER wave w < NKT owns the 16 keys [16 w, 16 w + 16): their K / V rows are row fragments in registers (requested under the previous
//     pair's stores), dK / dV of those keys stay in 32 accumulator registers for the whole pair.  Per step of 32 queries it forms S = Q K^T and
//     dP = dO V^T (2 tiles of 16 x 16), P, dS, feeds dV^T += dO^T P and dK^T += Q^T dS from its registers and writes dS -- 8 bytes per lane and
//     tile -- into a [keys][32 queries] LDS buffer.
//   * two CONSUMER waves (the last two of the 16) own dQ: after ONE workgroup barrier per step (the dS buffer is double, so the consumers trail
//     the producers by up to a step) each computes 4 of the step's 8 tiles of dQ^T = K^T dS^T over ALL keys: 7 x 4 MFMAs whose A operands (32 head
//     columns of K^T, all 224 keys) sit in 56 registers for the whole pair and whose B operands are transposed 8-byte reads
//     (ds_read_b64_tr_b16) of the buffer.  No atomics (ds_add_f32 measured 165 cycles per wave instruction: profiles/r06_b_attn_onepass.txt),
//     no dQ image, no write-out pass.  (An 8-wave form with 32 keys per wave and every wave also a consumer ran 2 waves per SIMD and could not
//     hide its LDS / MFMA latencies: 320 us against this form's time, same file.)
//   * delta = dO . O is formed per pair by all threads from the O rows they requested a pair ahead and the dO rows in LDS.
// HBM traffic = the algorithmic 5 reads + 3 writes of [S][64] per pair.  Dropout counters = the two-launch form's (bit-compatible masks).
#include "a4r_attn_long.h"

namespace {

template <int NKT> struct OnePass {
    using G = Geo<bf16_t, 64>;
    static constexpr int SP = NKT * 16, NG = SP / 32, NPROD = NKT, NCONS = 2, NW = 16, NTHR = NW * 64;
    static constexpr int IMG = SP * 128;                                        // one [SP][64] bf16 image
    static constexpr int OFF_STAT = 4 * IMG, OFF_R = OFF_STAT + 2 * SP * (int)sizeof(float);
    static constexpr int DSB = NPROD * 1024;                                    // one dS buffer: [NPROD producers][2 tiles][16 keys][16 queries] bf16
    static constexpr int R_BYTES = NPROD * 2048;                                // K image (pair start) / 2 dS buffers (steps) / the producers' store staging (pair end)
    static constexpr int OFF_CSTG = OFF_R + R_BYTES, BYTES = OFF_CSTG + NCONS * 2048;      // + the consumers' store staging
    static constexpr int NPIECE = 2 * SP / 8;                                   // 1-KiB DMA pieces of a pair's two images (8 rows each)
    static_assert(NKT % 2 == 0 && NPROD + NCONS <= NW && 2 * DSB <= R_BYTES && BYTES <= 160 * 1024 && NPIECE % NPROD == 0 && (NG & 1) == 1, "one-pass backward geometry");
};

#ifdef A4R_OP_STAMP
// diagnostic build only (tools/op_stamps.py): s_memtime of wave 0 of every workgroup at five points of its THIRD pair
__device__ unsigned long long g_a4r_op_stamps[256 * 8];
#define A4R_OP_ST(k_) if (threadIdx.x == 0 && n_done == 2 && blockIdx.x < 256) g_a4r_op_stamps[blockIdx.x * 8 + (k_)] = __builtin_amdgcn_s_memtime();
#define A4R_OP_STC(k_) if (threadIdx.x == 14 * 64 && n_done == 2 && blockIdx.x < 256) g_a4r_op_stamps[blockIdx.x * 8 + (k_)] = __builtin_amdgcn_s_memtime();
#else
#define A4R_OP_ST(k_)
#define A4R_OP_STC(k_)
#endif

A4R_DEV void glds16(const void* base, uint32_t voff, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
// One dword of a 128-byte line, result never used: brings the line into this XCD's L2 ahead of the real request (asynchronous: the destination register
// must stay reserved until the caller's next vmcnt(0))
A4R_DEV void touch_line(const void* p, uint32_t& sink) { asm volatile("global_load_dword %0, %1, off" : "=v"(sink) : "v"(p) : "memory"); }
// LDS accesses by 32-bit LDS ADDRESS (an integer the compiler cannot mistake for a generic pointer: running addresses updated by adds / xors stay
// ds_read / ds_write with one address register)
typedef unsigned int op_u32x2_t __attribute__((ext_vector_type(2)));
A4R_DEV uint4 lds_ld128(uint32_t a) { return __builtin_bit_cast(uint4, *(__attribute__((address_space(3))) const u32x4_raw_t*)(a)); }
A4R_DEV f32x4_t lds_ldf4(uint32_t a) { return *(__attribute__((address_space(3))) const f32x4_t*)(a); }
A4R_DEV void lds_st64(uint32_t a, uint2 v) { *(__attribute__((address_space(3))) op_u32x2_t*)(a) = op_u32x2_t{v.x, v.y}; }
A4R_DEV uint4 lds_tr_frag(uint32_t a) {                       // transposed operand chunk: rows r .. r + 3 and r + 16 .. r + 19 (2048 bytes further)
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a));
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a + 2048u));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}
A4R_DEV uint4 tr_pair(const char* a0, const char* a1) {       // two transposed 8-byte reads -> one operand chunk
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a0));
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(a1));
    const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
    return make_uint4(l2.x, l2.y, h2.x, h2.y);
}

// A producer's [16 keys][64] result block (transposed in accumulators: lane (key c, kg) holds head columns dt * 16 + 4 kg .. + 3) leaves through a 1-KB
// wave-private LDS block, 32 head columns at a time: 16-byte global stores, 4 lanes per 64-byte half row.  (store_block16's whole-row form needs 2 KB per
// wave: 14 producers x 2 KB = both dS buffers, i.e. a workgroup barrier behind the consumers' last step.)
A4R_DEV void store_block16_halves(char* stg, const f32x4_t (&o)[4], const RowsView& dst, int row0, int lane_) {
    const int lane = opaque_lane(lane_), fr = lane & 15, kg = lane >> 4, sw = (fr >> 1) & 3;
    const int rrow = lane >> 2, rch = lane & 3;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int d = 0; d < 2; ++d)
            store4<bf16_t>(reinterpret_cast<bf16_t*>(stg + fr * 64 + (((d * 2 + (kg >> 1)) ^ sw) << 4) + 8 * (kg & 1)), o[hh * 2 + d]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        dst.store((uint32_t)(row0 + rrow) * dst.ldb + (uint32_t)(hh * 4 + rch) * 16u, *reinterpret_cast<const uint4*>(stg + rrow * 64 + ((rch ^ ((rrow >> 1) & 3)) << 4)));
        __builtin_amdgcn_wave_barrier();
    }
}

template <int NKT, bool DROP>     // DROP: probability dropout (a ViT configured with attention dropout); false: no dropout code in the step loop
__global__ void __launch_bounds__(OnePass<NKT>::NTHR, 4) attn_long_bwd1_kernel(const bf16_t* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                                              const bf16_t* __restrict__ dctx, int ldo, const bf16_t* __restrict__ octx,
                                                                              const float* __restrict__ lse, bf16_t* __restrict__ dqkv,
                                                                              int S, int nh, int n_pairs, float scale, Drop dr) {
    using T = bf16_t;
    using OP = OnePass<NKT>;
    using G = Geo<T, 64>;
    constexpr int DH = 64, SP = OP::SP, SPT = SP + 8, NG = OP::NG, NTHR = OP::NTHR, NPROD = OP::NPROD, NST = SP / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lse_s = reinterpret_cast<float*>(smem + OP::OFF_STAT);
    float* del_s = lse_s + SP;
    char* R = smem + OP::OFF_R;
    const int tid = threadIdx.x, lane0 = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool prod = wave < NPROD;                          // (wave-uniform) this wave owns the 16 keys of tile `wave`
    const bool cons = wave >= OP::NW - OP::NCONS;            // ... or half of dQ: head columns 32 cw .. 32 cw + 31
    const int cw = wave - (OP::NW - OP::NCONS);
    const uint32_t lds_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
    const int G_ = (int)gridDim.x;

    // ---- once per workgroup: both image sets start at zero (rows >= S are never written again)
    for (int i = tid; i < 4 * OP::IMG / 16; i += NTHR) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // LDS-DMA of pair `pr` into image set `set`: piece i (1 KiB = 8 rows x 128 B) of [Q image | dO image]; lane -> row 8 i + (lane >> 3), LDS chunk slot
    // lane & 7 <- global chunk (lane & 7) ^ swz(row).  Producer w issues pieces w, w + NPROD, ...
    auto dma_pair = [&](int pr, int set) {
        const int item = pr / nh, h = pr % nh;
        const T* qb = qkv + (size_t)item * S * ld + h * DH + q_off;
        const T* ob = dctx + (size_t)item * S * ldo + h * DH;
#pragma unroll
        for (int j = 0; j < OP::NPIECE / NPROD; ++j) {
            const int i = wave + j * NPROD;
            const bool second = i >= SP / 8;
            const int piece = second ? i - SP / 8 : i;
            const int ol = opaque_lane(lane0);                // (rebuilt per call: hoisted out of the pair loop these offsets were spilled, and a scratch reload waits vmcnt(0) = for the DMA piece issued before it)
            const int row = piece * 8 + (ol >> 3), slot = ol & 7;
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
        const int ot = opaque_lane(tid);                     // (thread-derived offsets are rebuilt where they are used: kept across the pair loop they were spilled)
        const uint32_t voff = (uint32_t)(ot / G::CPR) * oview.ldb + (uint32_t)(ot % G::CPR) * 16u;
#pragma unroll
        for (int it = 0; it < NIT; ++it) ov[it] = oview.load(voff, (uint32_t)(it * (NTHR / G::CPR)) * oview.ldb);
    };
    uint4 kf[G::KS], vf[G::KS];
    auto request_kv = [&](int pr) {
        const int item = pr / nh, h = pr % nh;
        const T* base = qkv + (size_t)item * S * ld + h * DH;
        const RowsView kview = RowsView::make(base + k_off, ld, S, DH), vview = RowsView::make(base + v_off, ld, S, DH);
        const int ol = opaque_lane(lane0), fr = ol & 15, kg = ol >> 4;
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
            kf[ks] = kview.row_chunk(wave * 16 + fr, ks * 4 + kg);
            vf[ks] = vview.row_chunk(wave * 16 + fr, ks * 4 + kg);
        }
    };

    uint32_t sink = 0;                                       // destination of the producers' touch loads (asynchronous: reserved across the pair loop)
    int pair = blockIdx.x, cur = 0;
    [[maybe_unused]] int n_done = 0;
    if (pair < n_pairs) {
        if (prod) dma_pair(pair, 0);
        request_o(pair);
        if (prod) request_kv(pair);
    }
    const float c2 = scale * 1.44269504088896f;
    const f32x4_t c2v = {c2, c2, c2, c2};
    for (; pair < n_pairs; pair += G_, cur ^= 1, ++n_done) {
        const int item = pair / nh, h = pair % nh;
        char* Qr = smem + cur * 2 * OP::IMG;
        char* Or = Qr + OP::IMG;
        A4R_OP_ST(0)
        // ---- B0: this wave's DMA pieces, O rows and K / V rows have landed; every wave is done with the previous pair (R, the other image set)
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(sink) : "memory");      // (sink: the touch loads' destination stays reserved up to here)
        __syncthreads();
        A4R_OP_ST(1)
        {
            // delta[row] = dO[row] . O[row]: a row's 8 chunks sit on 8 consecutive lanes; del_s holds -delta (rows >= S: zero rows, 0); lse_s = -lse log2(e)
            const int ot = opaque_lane(tid);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int id = ot + it * NTHR, row = id / G::CPR, ch = id % G::CPR;
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
                if ((ot & 7) == 0 && id < SP * G::CPR) del_s[row] = -d;
            }
            static_assert(SP <= NTHR, "one row statistic per thread");
            if (ot < SP) lse_s[ot] = ot < S ? lse[((size_t)item * nh + h) * S + ot] * -1.44269504088896f : 0.f;
            if (prod) {                                      // this wave's 16 K rows into the [SP][64] K image (row-major swizzled) the consumers read transposed
                const int ol = opaque_lane(lane0), fr = ol & 15, kg = ol >> 4, row = wave * 16 + fr;
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) *reinterpret_cast<uint4*>(R + row * G::ROWB + (((ks * 4 + kg) ^ G::swz(row)) << 4)) = kf[ks];
            }
        }
        __syncthreads();                                     // B1
        const int nxt = pair + G_;
        // The two roles run their own code from here to the end of the pair (the register allocator then sees each role's live set alone: with both
        // in one loop body every wave carried the consumers' 56 K^T registers beside the producers' accumulators); both execute the same barriers.
        if (cons) {
            uint4 ktc[2][NST];                               // K^T, head columns 32 cw .. + 31, of ALL keys: A operands of this wave's dQ^T tiles
            {
                // (address rebuilt from the lane index HERE: hoisted out of the pair loop the 14 offsets were spilled, and their scratch reloads --
                // vmcnt(0) each -- sat between B1 and B2, where fourteen producers wait for these two waves)
                const int ol = opaque_lane(lane0), kg = ol >> 4, qd = (ol >> 2) & 3, pp = ol & 3, r0 = 4 * kg + qd;
                const uint32_t kb = lds_base + (uint32_t)(OP::OFF_R + r0 * G::ROWB + 8 * (pp & 1));
#pragma unroll
                for (int d2 = 0; d2 < 2; ++d2) {
                    const uint32_t kc = kb + (uint32_t)(((((cw * 2 + d2) * 2) + (pp >> 1)) ^ G::swz(r0)) << 4);
#pragma unroll
                    for (int p = 0; p < NST; ++p) ktc[d2][p] = lds_tr_frag(kc + (uint32_t)(p * 4096));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();                                 // B2: R is free for the dS buffers
            const RowsView dqview = RowsView::make(dqkv + (size_t)item * S * ld + q_off + h * DH, ld, S, DH);
            __builtin_amdgcn_s_setprio(3);                   // two waves carry a dependent chain per step for fourteen: they issue first
#pragma unroll 1
            for (int g = 0; g < NG; ++g) {
                const int lane = opaque_lane(lane0), fr = lane & 15, kg = lane >> 4;
                const char* dsb = R + (g & 1) * OP::DSB;
                __syncthreads();                             // the step's dS tiles of every producer are in the buffer (its other half is re-filled next step)
                // dQ^T tiles (head columns 32 cw + 16 d2 .., queries g * 32 + 16 ct ..) over all keys; B chunk of contraction step p, query tile ct:
                // keys 32 p + (j >> 2) * 16 + 4 kg + (j & 3) = producers 2 p and 2 p + 1
                const int qd = (lane >> 2) & 3, pp = lane & 3;
                const char* b0 = dsb + (4 * kg + qd) * 32 + pp * 8;
                f32x4_t acc[2][2];                            // [ct][d2]
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int d2 = 0; d2 < 2; ++d2) acc[ct][d2] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int p = 0; p < NST; ++p) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        const uint4 b = tr_pair(b0 + (2 * p) * 1024 + ct * 512, b0 + (2 * p + 1) * 1024 + ct * 512);
#pragma unroll
                        for (int d2 = 0; d2 < 2; ++d2) Mma<T>::mma(ktc[d2][p], b, acc[ct][d2]);
                    }
                }
                // lane (query fr, kg): head columns 32 cw + 16 d2 + 4 kg .. + 3 of row g * 32 + 16 ct + fr (a row >= S is dropped by the view)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int d2 = 0; d2 < 2; ++d2) {
                        const f32x4_t o = acc[ct][d2] * f32x4_t{scale, scale, scale, scale};
                        dqview.store8((uint32_t)(g * 32 + ct * 16 + fr) * dqview.ldb + (uint32_t)(cw * 32 + d2 * 16 + kg * 4) * 2u,
                                      make_uint2(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3])));
                    }
            }
            __builtin_amdgcn_s_setprio(0);
            A4R_OP_STC(7)
            if (nxt < n_pairs) request_o(nxt);
            // (no barrier here: the producers stage their stores in the SECOND dS buffer, which the last step -- an even one -- does not use; B0 of
            // the next pair is where this wave says it is done with the first)
        } else {
            __syncthreads();                                 // B2
            A4R_OP_ST(2)
            // (every register the loads of the previous pair's epilogue wrote is touched BEFORE the DMA is issued: hipcc places its own vmcnt wait at a
            // value's first use -- V's would sit in the first step and wait for the DMA pieces issued in between as well)
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) asm volatile("" : "+v"(vf[ks].x), "+v"(vf[ks].y), "+v"(vf[ks].z), "+v"(vf[ks].w));
            if (nxt < n_pairs) dma_pair(nxt, cur ^ 1);         // the next pair's rows stream into the other image set under this pair's arithmetic
            f32x4_t dk[G::ND], dv[G::ND];
#pragma unroll
            for (int dt = 0; dt < G::ND; ++dt) { dk[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
            // Per-lane LDS addresses of the step loop, built ONCE per pair and advanced by constants (rebuilt from the lane index in every step they were
            // ~45 of the step's ~145 instructions, and the steps are bound by the SIMD's issue port: profiles/r06_b_attn_onepass.txt):
            //   aq[ks]  row fragment (row fr, chunk step ks) of the Q image; dO's is IMG further; a step is 32 rows = 4096 bytes, its second tile 2048
            //   tq      transposed fragment (ds_read_b64_tr_b16) of head columns 0 .. 15, rows 4 kg + qd of a step; columns 16 j ..: tq ^ (j << 5); rows + 16: 2048
            //   st      the lane's 4 row statistics of a tile (lse at st, delta SP floats further)
            //   dsw     this lane's 8 bytes of the wave's dS tile in the step's buffer
            uint32_t aq[G::KS], tq, st, dsw;
            {
                const int ol = opaque_lane(lane0), fr = ol & 15, kg = ol >> 4, qd = (ol >> 2) & 3, pp = ol & 3, r0 = 4 * kg + qd;
                const uint32_t qa = lds_base + (uint32_t)(cur * 2 * OP::IMG);
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) aq[ks] = qa + (uint32_t)(fr * G::ROWB + (((ks * 4 + kg) ^ G::swz(fr)) << 4));
                tq = qa + (uint32_t)(r0 * G::ROWB + (((pp >> 1) ^ G::swz(r0)) << 4) + 8 * (pp & 1));
                st = lds_base + (uint32_t)(OP::OFF_STAT + kg * 16);
                dsw = lds_base + (uint32_t)(OP::OFF_R + wave * 1024 + fr * 32 + kg * 8);
            }
            uint4 rq_[G::KS], ro[G::KS];
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) { rq_[ks] = lds_ld128(aq[ks]); ro[ks] = lds_ld128(aq[ks] + OP::IMG); }
#pragma unroll 1
            for (int g = 0; g < NG; ++g) {
                if (g == NG - 4 && nxt < n_pairs) {
                    // The next pair's K / V rows of this producer and 16 of its O rows are requested for real only after the steps (their registers are
                    // busy until then) -- ~3 us of HBM latency nothing covered (profiles/r06_b_attn_onepass.txt).  Lanes 0 .. 47 touch those 48 lines
                    // here, half a pair ahead: the real requests then find them in this XCD's L2.
                    const int ol = opaque_lane(lane0), which = ol >> 4, r = wave * 16 + (ol & 15);
                    const int it2 = nxt / nh, h2 = nxt % nh;
                    const T* pk = qkv + ((size_t)it2 * S + r) * ld + h2 * DH + (which == 1 ? v_off : k_off);
                    const T* po = octx + ((size_t)it2 * S + r) * ldo + h2 * DH;
                    if (which < 3 && r < S) touch_line(which == 2 ? (const void*)po : (const void*)pk, sink);
                }
                uint32_t pw[4], dw[4];                       // P and dS operand chunks of the step (key on the lane)
                constexpr int HALF = 2;
                uint4 tf[HALF];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    // tile rows = queries 32 g + 16 t + 4 kg + r, column = key wave * 16 + fr
                    const f32x4_t l4 = lds_ldf4(st + t * 64), d4 = lds_ldf4(st + SP * 4 + t * 64);
                    f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dpt = d4;    // (del_s holds -delta: dP - delta straight from the matrix pipe)
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) {
                        Mma<T>::mma(rq_[ks], kf[ks], sc);
                        Mma<T>::mma(ro[ks], vf[ks], dpt);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // the next tile's row fragments (behind the last tile: the 16 rows behind the image -- never used, inside the allocation)
#pragma unroll
                    for (int ks = 0; ks < G::KS; ++ks) {
                        rq_[ks] = lds_ld128(aq[ks] + (t + 1) * 2048);
                        ro[ks] = lds_ld128(aq[ks] + OP::IMG + (t + 1) * 2048);
                    }
                    if (t == 1) {
#pragma unroll
                        for (int j = 0; j < HALF; ++j) tf[j] = lds_tr_frag((tq ^ (uint32_t)(j << 5)) + OP::IMG);
                    }
                    f32x4_t pv = __builtin_elementwise_fma(sc, c2v, l4), dsv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = __builtin_amdgcn_exp2f(pv[r]);
                    if (DROP && dr.thr16) {                   // the tile's 4 rows are 4 QUERIES at one key: one hash each (the dk/dv kernel's counters)
                        const int ol = opaque_lane(lane0), ork = wave * 16 + (ol & 15), okg = ol >> 4;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float keepf = dropout_keep(dr.seed, dr.site, drop_idx(pair, g * 32 + t * 16 + okg * 4 + r, ork), dr.thr16) ? dr.keep_scale : 0.f;
                            dsv[r] = pv[r] * ((dpt[r] - d4[r]) * keepf + d4[r]);
                            pv[r] *= keepf;
                        }
                    } else {
                        dsv = pv * dpt;                        // (x scale: applied to dK / dQ at the end)
                    }
                    pw[2 * t] = pack2_bf16(pv[0], pv[1]); pw[2 * t + 1] = pack2_bf16(pv[2], pv[3]);
                    dw[2 * t] = pack2_bf16(dsv[0], dsv[1]); dw[2 * t + 1] = pack2_bf16(dsv[2], dsv[3]);
                    // dS tile t as [16 keys][16 queries] bf16, 32 bytes per key row: this lane's 4 queries of key fr
                    lds_st64(dsw + t * 512, make_uint2(dw[2 * t], dw[2 * t + 1]));
                    __builtin_amdgcn_sched_barrier(0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __syncthreads();                             // (the consumers' barrier of this step)
                const uint4 pf = make_uint4(pw[0], pw[1], pw[2], pw[3]), dsf = make_uint4(dw[0], dw[1], dw[2], dw[3]);
                // dV^T += dO^T P, dK^T += Q^T dS, HALF head-column tiles at a time through ONE fragment slot (a second slot cost the 8 registers
                // whose spills -- any scratch reload waits vmcnt(0), i.e. for the LDS-DMA in flight -- cost more than the exposed LDS latency does
                // at four waves per SIMD)
#pragma unroll
                for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[j], pf, dv[j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[j] = lds_tr_frag((tq ^ (uint32_t)((HALF + j) << 5)) + OP::IMG);
#pragma unroll
                for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[j], pf, dv[HALF + j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[j] = lds_tr_frag(tq ^ (uint32_t)(j << 5));
#pragma unroll
                for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[j], dsf, dk[j]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < HALF; ++j) tf[j] = lds_tr_frag(tq ^ (uint32_t)((HALF + j) << 5));
#pragma unroll
                for (int j = 0; j < HALF; ++j) Mma<T>::mma(tf[j], dsf, dk[HALF + j]);
                __builtin_amdgcn_sched_barrier(0);
                // the next step: 32 rows further in the images and statistics, the other dS buffer
#pragma unroll
                for (int ks = 0; ks < G::KS; ++ks) aq[ks] += 4096;
                tq += 4096;
                st += 128;
                dsw += (g & 1) ? (uint32_t)(-OP::DSB) : (uint32_t)OP::DSB;
            }
            A4R_OP_ST(3)
            // K / V fragments are dead: the next pair's rows are requested under the dk / dv stores.  (NOT inside the step loop: a load whose result is
            // live around the loop's back edge makes hipcc wait vmcnt(0) in EVERY step -- i.e. for this wave's LDS-DMA of the next pair as well)
            if (nxt < n_pairs) {
                request_o(nxt);
                request_kv(nxt);
            }
            A4R_OP_ST(6)
            A4R_OP_ST(5)
            // (the SECOND dS buffer is dead since the last step's barrier -- NG is odd, the last step used the first --: 1 KB of it per producer)
            const T* base = dqkv + (size_t)item * S * ld + h * DH;
            const RowsView dkview = RowsView::make(base + k_off, ld, S, DH), dvview = RowsView::make(base + v_off, ld, S, DH);
            char* stg = R + OP::DSB + wave * 1024;
#pragma unroll
            for (int dt = 0; dt < G::ND; ++dt) dk[dt] *= f32x4_t{scale, scale, scale, scale};
            store_block16_halves(stg, dk, dkview, wave * 16, lane0);
            store_block16_halves(stg, dv, dvview, wave * 16, lane0);
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
