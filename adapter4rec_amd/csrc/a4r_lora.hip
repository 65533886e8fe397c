// a4r_lora_bwd_fused: the low-rank gradients of a block's two small-rank LoRAs (loralib's Linear on the query and value projections, r <= 8:
// Downstream/CV/run_adapter.py:384-395, Downstream/Text/run.py:414-428) in ONE pass over the rows.
//
//   t  = x [A_q ; A_v]^T                       [M, 16]   (ranks 0 - 7: query, 8 - 15: value)
//   dt = (dq B_q) s_q | (dv B_v) s_v           [M, 16]
//   dA_q | dA_v   = dt^T x                     [16, H]
//   dB_q = dq^T t[:, 0:8],  dB_v = dv^T t[:, 8:16]        [H, 8] each      (the caller applies the LoRA scaling when it flushes them)
//   db_q = column sums of dq, db_v = column sums of dv    (a row of ones in the dB products' rank operand)
//
// Round 3 ran these as five launches (two products over x, three over dq / dv: 612 MB per layer at the image tower's 66 304 rows, 160 us); every one
// of them is row-local arithmetic followed by a column reduction, so one kernel that holds a 16-row tile of x, dq and dv reads 306 MB.
//
// One workgroup per CU, 8 waves, wave w owns the 96 columns 96 w .. 96 w + 95 of all three operands.  Per 16-row tile: the wave's partial t and dt
// (9 MFMA 16x16x32 against its column slice of the weights, held in registers) -> LDS -> one barrier -> every wave sums the 8 partials; t and dt then
// become the rank-side operand of 18 MFMA 16x16x16 (contraction over the tile's 16 rows) against the wave's own column slice of dq, dv and x, read
// back transposed from a wave-private LDS image (ds_read_b64_tr_b16).  The 72 accumulator registers hold the wave's [16, 96] slices of dA, dB_q^T and
// dB_v^T for the whole launch and are flushed with atomics at the end.  The next tile's rows are requested before the current tile's arithmetic.
#include "a4r_common.h"
#include "../../include/a4r.h"

namespace {

struct LoraBwdArgs {
    const bf16_t *x, *dqa, *dqb;
    int ldx, lddq;
    const bf16_t *Aa, *Ab, *BTa, *BTb;          // 8 rank rows each, row stride H
    int ldw;
    float sa, sb;
    float *dAa, *dAb;                            // [8, H] each, row stride lda
    int lda;
    float *dBa, *dBb;                            // [H, 8] each, row stride ldb
    int ldb;
    float *dba, *dbb;                            // [H], element stride ldbias (0 pointers: not wanted)
    int ldbias;
    int M;
    float* ws;                                   // [gridDim.x][WS_ROWS][H] per-workgroup column sums (reduced by lora_reduce_kernel)
};
constexpr int WS_ROWS = 34;                      // 0 - 15: dA ranks; 16 - 23: dB_q ranks 0 - 7; 24: db_q; 25 - 32: dB_v ranks 8 - 15; 33: db_v

typedef short v4s_t __attribute__((ext_vector_type(4)));

#define A4R_LORA_BARRIER()                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    __builtin_amdgcn_s_barrier();                            \
    asm volatile("" ::: "memory")

constexpr int SLAB_LD = 208;       // bytes per row of a wave's [16][96] bf16 slab image (192 + 16: rows 4 apart land on different banks)

template <int CW, int NW>
__global__ void __launch_bounds__(NW * 64) lora_bwd_kernel(const LoraBwdArgs p) {
    constexpr int KS = CW / 32, NCT = CW / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];                       // 122 KB: one workgroup per CU
    typedef float Part[NW][2][16][16];                                                 // [wave][t | dt][row][rank]
    typedef char Slab[3][16 * SLAB_LD];                                                // [x | dq | dv]
    typedef unsigned short Timg[2][16][16 + 4];                                        // [t | dt][rank][row] (bf16 bits; 40-byte rows)
    Part* part = reinterpret_cast<Part*>(smem);                                        // [tile parity]
    Slab* slab = reinterpret_cast<Slab*>(smem + 2 * sizeof(Part));                     // [wave]
    Timg* timg = reinterpret_cast<Timg*>(smem + 2 * sizeof(Part) + NW * sizeof(Slab)); // [wave]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kg = lane >> 4;
    const int c0 = wave * CW, cl = c0 + kg * 8;

    // weight fragments of this wave's column slice (the rank side of the 16x16x32 products: lane (rank fr, k chunk kg))
    uint4 wA[KS], wBa[KS], wBb[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const size_t off = (size_t)(fr & 7) * p.ldw + cl + s * 32;
        wA[s] = *reinterpret_cast<const uint4*>((fr < 8 ? p.Aa : p.Ab) + off);
        const uint4 a = *reinterpret_cast<const uint4*>(p.BTa + off), b = *reinterpret_cast<const uint4*>(p.BTb + off);
        wBa[s] = fr < 8 ? a : make_uint4(0, 0, 0, 0);
        wBb[s] = fr < 8 ? make_uint4(0, 0, 0, 0) : b;
    }
    f32x4_t accA[NCT], accBa[NCT], accBb[NCT];      // D[rank][col]: lane (col fr, kg) holds ranks 4 kg .. 4 kg + 3
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) { accA[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accBa[ct] = accA[ct]; accBb[ct] = accA[ct]; }

    const int ntiles = p.M / 16;
    uint4 xc[KS], ac[KS], bc[KS], xn[KS], an[KS], bn[KS];
    auto request = [&](int tile, uint4 (&xd)[KS], uint4 (&ad)[KS], uint4 (&bd)[KS]) {
        const size_t row = (size_t)tile * 16 + fr;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xd[s] = *reinterpret_cast<const uint4*>(p.x + row * p.ldx + cl + s * 32);
            ad[s] = *reinterpret_cast<const uint4*>(p.dqa + row * p.lddq + cl + s * 32);
            bd[s] = *reinterpret_cast<const uint4*>(p.dqb + row * p.lddq + cl + s * 32);
        }
    };
    if ((int)blockIdx.x < ntiles) request(blockIdx.x, xc, ac, bc);
    char* sx = slab[wave][0];
    char* sa_ = slab[wave][1];
    char* sb_ = slab[wave][2];
    int par = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
        {   // the next tile's rows (the last tile re-requests itself: the counted waits stay static)
            const int tn = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;
            request(tn, xn, an, bn);
        }
        // ---- partial t^T and dt^T over this wave's columns: D[rank][row], lane (row fr, kg) holds ranks 4 kg .. + 3
        f32x4_t t4 = {0.f, 0.f, 0.f, 0.f}, d4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Mma<bf16_t>::mma(wA[s], xc[s], t4);
            Mma<bf16_t>::mma(wBa[s], ac[s], d4);
            Mma<bf16_t>::mma(wBb[s], bc[s], d4);
        }
        *reinterpret_cast<f32x4_t*>(&part[par][wave][0][fr][kg * 4]) = t4;
        *reinterpret_cast<f32x4_t*>(&part[par][wave][1][fr][kg * 4]) = d4;
        // the wave's slices of the three operands, row-major, for the transposed reads below (wave-private: no workgroup barrier)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *reinterpret_cast<uint4*>(sx + fr * SLAB_LD + (s * 32 + kg * 8) * 2) = xc[s];
            *reinterpret_cast<uint4*>(sa_ + fr * SLAB_LD + (s * 32 + kg * 8) * 2) = ac[s];
            *reinterpret_cast<uint4*>(sb_ + fr * SLAB_LD + (s * 32 + kg * 8) * 2) = bc[s];
        }
        A4R_LORA_BARRIER();
        t4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
        d4 = t4;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            t4 += *reinterpret_cast<const f32x4_t*>(&part[par][w][0][fr][kg * 4]);
            d4 += *reinterpret_cast<const f32x4_t*>(&part[par][w][1][fr][kg * 4]);
        }
        d4 *= (kg < 2 ? p.sa : p.sb);
        // t, dt rounded to the element type (what the five-launch form stored), laid out [rank][row] for the rank-side operand of the row contraction
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            timg[wave][0][kg * 4 + r][fr] = (unsigned short)f32_to_bf16_bits(t4[r]);
            timg[wave][1][kg * 4 + r][fr] = (unsigned short)f32_to_bf16_bits(d4[r]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        // rank-side operands: lane (rank fr, kg) holds rows 4 kg .. + 3.  dB_q uses ranks 0 - 7 of t with rank 8 replaced by ones (its output row 8 is
        // the column sum of dq), dB_v ranks 8 - 15 with rank 0 replaced by ones.
        const uint2 tr_ = *reinterpret_cast<const uint2*>(&timg[wave][0][fr][kg * 4]), dr_ = *reinterpret_cast<const uint2*>(&timg[wave][1][fr][kg * 4]);
        const uint2 ones = make_uint2(0x3f803f80u, 0x3f803f80u), zero = make_uint2(0u, 0u);
        const uint2 ta = fr < 8 ? tr_ : (fr == 8 ? ones : zero), tb = fr >= 8 ? tr_ : (fr == 0 ? ones : zero);
        const v4s_t opA = __builtin_bit_cast(v4s_t, dr_), opBa = __builtin_bit_cast(v4s_t, ta), opBb = __builtin_bit_cast(v4s_t, tb);
        // ---- dA += dt^T x, dB_q^T += t_q^T dq, dB_v^T += t_v^T dv over the wave's columns: contraction over the tile's 16 rows
        const int q = (lane >> 2) & 3, pc = lane & 3;
        const int toff = (kg * 4 + q) * SLAB_LD + pc * 8;      // + 32 ct: the 4-row x 16-column block of column tile ct, this lane's 8 bytes
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const v4s_t fx = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(sx + toff + ct * 32));
            const v4s_t fa = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(sa_ + toff + ct * 32));
            const v4s_t fb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(sb_ + toff + ct * 32));
            accA[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(opA, fx, accA[ct], 0, 0, 0);
            accBa[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(opBa, fa, accBa[ct], 0, 0, 0);
            accBb[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(opBb, fb, accBb[ct], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();                       // the slab and timg images are rewritten by the next tile
#pragma unroll
        for (int s = 0; s < KS; ++s) { xc[s] = xn[s]; ac[s] = an[s]; bc[s] = bn[s]; }
    }
    // ---- the workgroup's column sums go to its slice of the workspace with plain stores (round 4: flushed with atomics, 256 workgroups x 26 112 sums
    // onto the same addresses cost 110 us at the end of an 80 us launch); lane (col fr, kg) holds ranks 4 kg .. + 3 of column c0 + 16 ct + fr
    float* wsb = p.ws + (size_t)blockIdx.x * WS_ROWS * (CW * NW);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int col = c0 + ct * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rank = kg * 4 + r;
            wsb[(size_t)rank * (CW * NW) + col] = accA[ct][r];
            if (rank <= 8) wsb[(size_t)(16 + rank) * (CW * NW) + col] = accBa[ct][r];           // rank 8: ones^T dq
            if (rank >= 8 || rank == 0) wsb[(size_t)(rank == 0 ? 33 : 17 + rank) * (CW * NW) + col] = accBb[ct][r];      // rank 0: ones^T dv
        }
    }
}

// The same pass for ranks 9 - 15 (the image tower's default r = 12, CV/run_adapter.py:386-387): each LoRA gets a rank tile of its own (16 rows of its
// weight operands, rows past r zero; rank 15 of the dB products' operand is the row of ones), four gradient products instead of three (96 accumulator
// registers), 12 + 24 MFMAs per tile, the partial sums single-buffered (two barriers per tile: 133 KB of LDS).
constexpr int WS_ROWS2 = 64;                     // 0 - 15: dA_q; 16 - 31: dA_v; 32 - 46: dB_q ranks, 47: db_q; 48 - 62: dB_v ranks, 63: db_v
template <int CW, int NW>
__global__ void __launch_bounds__(NW * 64) lora_bwd2_kernel(const LoraBwdArgs p) {
    constexpr int KS = CW / 32, NCT = CW / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef float Part[4][16][16];                                                     // [t_q | t_v | dt_q | dt_v][row][rank]
    typedef char Slab[3][16 * SLAB_LD];
    typedef unsigned short Timg[4][16][16 + 4];
    Part* part = reinterpret_cast<Part*>(smem);                                        // [wave]
    Slab* slab = reinterpret_cast<Slab*>(smem + NW * sizeof(Part));
    Timg* timg = reinterpret_cast<Timg*>(smem + NW * sizeof(Part) + NW * sizeof(Slab));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, kg = lane >> 4;
    const int c0 = wave * CW, cl = c0 + kg * 8;
    uint4 wAa[KS], wAb[KS], wBa[KS], wBb[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const size_t off = (size_t)fr * p.ldw + cl + s * 32;
        wAa[s] = *reinterpret_cast<const uint4*>(p.Aa + off);
        wAb[s] = *reinterpret_cast<const uint4*>(p.Ab + off);
        wBa[s] = *reinterpret_cast<const uint4*>(p.BTa + off);
        wBb[s] = *reinterpret_cast<const uint4*>(p.BTb + off);
    }
    f32x4_t accAa[NCT], accAb[NCT], accBa[NCT], accBb[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) { accAa[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accAb[ct] = accAa[ct]; accBa[ct] = accAa[ct]; accBb[ct] = accAa[ct]; }
    const int ntiles = p.M / 16;
    uint4 xc[KS], ac[KS], bc[KS], xn[KS], an[KS], bn[KS];
    auto request = [&](int tile, uint4 (&xd)[KS], uint4 (&ad)[KS], uint4 (&bd)[KS]) {
        const size_t row = (size_t)tile * 16 + fr;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            xd[s] = *reinterpret_cast<const uint4*>(p.x + row * p.ldx + cl + s * 32);
            ad[s] = *reinterpret_cast<const uint4*>(p.dqa + row * p.lddq + cl + s * 32);
            bd[s] = *reinterpret_cast<const uint4*>(p.dqb + row * p.lddq + cl + s * 32);
        }
    };
    if ((int)blockIdx.x < ntiles) request(blockIdx.x, xc, ac, bc);
    char* sx = slab[wave][0];
    char* sa_ = slab[wave][1];
    char* sb_ = slab[wave][2];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        {
            const int tn = tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : tile;
            request(tn, xn, an, bn);
        }
        f32x4_t v4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v4[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            Mma<bf16_t>::mma(wAa[s], xc[s], v4[0]);
            Mma<bf16_t>::mma(wAb[s], xc[s], v4[1]);
            Mma<bf16_t>::mma(wBa[s], ac[s], v4[2]);
            Mma<bf16_t>::mma(wBb[s], bc[s], v4[3]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4_t*>(&part[wave][i][fr][kg * 4]) = v4[i];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            *reinterpret_cast<uint4*>(sx + fr * SLAB_LD + (s * 32 + kg * 8) * 2) = xc[s];
            *reinterpret_cast<uint4*>(sa_ + fr * SLAB_LD + (s * 32 + kg * 8) * 2) = ac[s];
            *reinterpret_cast<uint4*>(sb_ + fr * SLAB_LD + (s * 32 + kg * 8) * 2) = bc[s];
        }
        A4R_LORA_BARRIER();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4_t a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; ++w) a += *reinterpret_cast<const f32x4_t*>(&part[w][i][fr][kg * 4]);
            if (i == 2) a *= p.sa;
            if (i == 3) a *= p.sb;
#pragma unroll
            for (int r = 0; r < 4; ++r) timg[wave][i][kg * 4 + r][fr] = (unsigned short)f32_to_bf16_bits(a[r]);
        }
        A4R_LORA_BARRIER();                                    // everyone has read the partial sums (single-buffered); own timg writes landed
        const uint2 ones = make_uint2(0x3f803f80u, 0x3f803f80u);
        uint2 o4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) o4[i] = *reinterpret_cast<const uint2*>(&timg[wave][i][fr][kg * 4]);
        const v4s_t opBa = __builtin_bit_cast(v4s_t, fr == 15 ? ones : o4[0]), opBb = __builtin_bit_cast(v4s_t, fr == 15 ? ones : o4[1]);
        const v4s_t opAa = __builtin_bit_cast(v4s_t, o4[2]), opAb = __builtin_bit_cast(v4s_t, o4[3]);
        const int q = (lane >> 2) & 3, pc = lane & 3;
        const int toff = (kg * 4 + q) * SLAB_LD + pc * 8;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const v4s_t fx = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(sx + toff + ct * 32));
            const v4s_t fa = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(sa_ + toff + ct * 32));
            const v4s_t fb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(sb_ + toff + ct * 32));
            accAa[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(opAa, fx, accAa[ct], 0, 0, 0);
            accAb[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(opAb, fx, accAb[ct], 0, 0, 0);
            accBa[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(opBa, fa, accBa[ct], 0, 0, 0);
            accBb[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(opBb, fb, accBb[ct], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < KS; ++s) { xc[s] = xn[s]; ac[s] = an[s]; bc[s] = bn[s]; }
    }
    float* wsb = p.ws + (size_t)blockIdx.x * WS_ROWS2 * (CW * NW);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int col = c0 + ct * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rank = kg * 4 + r;
            wsb[(size_t)rank * (CW * NW) + col] = accAa[ct][r];
            wsb[(size_t)(16 + rank) * (CW * NW) + col] = accAb[ct][r];
            wsb[(size_t)(32 + rank) * (CW * NW) + col] = accBa[ct][r];
            wsb[(size_t)(48 + rank) * (CW * NW) + col] = accBb[ct][r];
        }
    }
}

// sums over the workgroups' slices, added into the destinations in their own layouts; grid (ceil(WS_ROWS H / 256), NCH): chunk y of the slices
__global__ void __launch_bounds__(256) lora_reduce_kernel(const LoraBwdArgs p, int nblk, int H, int wide) {
    const int rows = wide ? WS_ROWS2 : WS_ROWS;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * H) return;
    const int per = (nblk + (int)gridDim.y - 1) / (int)gridDim.y, b0 = blockIdx.y * per, b1 = b0 + per < nblk ? b0 + per : nblk;
    const size_t stride = (size_t)rows * H;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = b0;
    for (; b + 4 <= b1; b += 4) {
        s0 += p.ws[(size_t)b * stride + idx];
        s1 += p.ws[(size_t)(b + 1) * stride + idx];
        s2 += p.ws[(size_t)(b + 2) * stride + idx];
        s3 += p.ws[(size_t)(b + 3) * stride + idx];
    }
    for (; b < b1; ++b) s0 += p.ws[(size_t)b * stride + idx];
    const float v = (s0 + s1) + (s2 + s3);
    const int row = idx / H, col = idx % H;
    if (wide) {
        if (row < 16) atomicAdd(p.dAa + (size_t)row * p.lda + col, v);
        else if (row < 32) atomicAdd(p.dAb + (size_t)(row - 16) * p.lda + col, v);
        else if (row < 47) atomicAdd(p.dBa + (size_t)col * p.ldb + (row - 32), v);
        else if (row == 47) { if (p.dba) atomicAdd(p.dba + (size_t)col * p.ldbias, v); }
        else if (row < 63) atomicAdd(p.dBb + (size_t)col * p.ldb + (row - 48), v);
        else if (p.dbb) atomicAdd(p.dbb + (size_t)col * p.ldbias, v);
        return;
    }
    if (row < 8) atomicAdd(p.dAa + (size_t)row * p.lda + col, v);
    else if (row < 16) atomicAdd(p.dAb + (size_t)(row - 8) * p.lda + col, v);
    else if (row < 24) atomicAdd(p.dBa + (size_t)col * p.ldb + (row - 16), v);
    else if (row == 24) { if (p.dba) atomicAdd(p.dba + (size_t)col * p.ldbias, v); }
    else if (row < 33) atomicAdd(p.dBb + (size_t)col * p.ldb + (row - 25), v);
    else if (p.dbb) atomicAdd(p.dbb + (size_t)col * p.ldbias, v);
}

}  // namespace

int a4r_cu_count();       // a4r_gemm256.hip

extern "C" int a4r_lora_bwd_fused_ws_floats(int H) { return a4r_cu_count() * WS_ROWS2 * H; }

extern "C" int a4r_lora_bwd_fused(void* stream, const void* x, int ldx, const void* dqa, const void* dqb, int lddq,
                                  const void* Aa, const void* Ab, const void* BTa, const void* BTb, int ldw, float scale_a, float scale_b,
                                  float* dAa, float* dAb, int lda, float* dBa, float* dBb, int ldb, float* dbias_a, float* dbias_b, int ldbias,
                                  int M, int H, int dtype, int rank_rows, float* ws, int64_t ws_floats) {
    if (!x || !dqa || !dqb || !Aa || !Ab || !BTa || !BTb || !dAa || !dAb || !dBa || !dBb || !ws || M <= 0 || M % 16) return A4R_EINVAL;
    if (dtype != A4R_BF16 || H != 768 || (rank_rows != 8 && rank_rows != 16)) return A4R_EINVAL;     // (the one geometry LoRA runs on: BERT-base / ViT-B/16 / ViT-MAE-base)
    if (ldx < H || lddq < H || ldw < H || lda < H || ldb < rank_rows - (rank_rows == 16) || (ldx * 2) % 16 || (lddq * 2) % 16 || (ldw * 2) % 16) return A4R_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dqa) | reinterpret_cast<uintptr_t>(dqb) | reinterpret_cast<uintptr_t>(Aa) |
         reinterpret_cast<uintptr_t>(Ab) | reinterpret_cast<uintptr_t>(BTa) | reinterpret_cast<uintptr_t>(BTb)) & 15u) return A4R_EINVAL;
    if ((dbias_a || dbias_b) && ldbias <= 0) return A4R_EINVAL;
    const bool wide = rank_rows == 16;
    const int rows = wide ? WS_ROWS2 : WS_ROWS;
    int grid = a4r_cu_count();
    if (grid > M / 16) grid = M / 16;
    if (ws_floats < (int64_t)grid * rows * H) return A4R_EINVAL;
    const LoraBwdArgs p{(const bf16_t*)x, (const bf16_t*)dqa, (const bf16_t*)dqb, ldx, lddq, (const bf16_t*)Aa, (const bf16_t*)Ab, (const bf16_t*)BTa,
                        (const bf16_t*)BTb, ldw, scale_a, scale_b, dAa, dAb, lda, dBa, dBb, ldb, dbias_a, dbias_b, ldbias, M, ws};
    constexpr size_t lds1 = 2 * sizeof(float) * 8 * 2 * 16 * 16 + 8 * 3 * 16 * SLAB_LD + 8 * 2 * 16 * 20 * sizeof(unsigned short);
    constexpr size_t lds2 = sizeof(float) * 8 * 4 * 16 * 16 + 8 * 3 * 16 * SLAB_LD + 8 * 4 * 16 * 20 * sizeof(unsigned short);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lora_bwd_kernel<96, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess) return A4R_ELAUNCH;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lora_bwd2_kernel<96, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) return A4R_ELAUNCH;
        attr_set = true;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (wide) hipLaunchKernelGGL((lora_bwd2_kernel<96, 8>), dim3(grid), dim3(512), lds2, s, p);
    else hipLaunchKernelGGL((lora_bwd_kernel<96, 8>), dim3(grid), dim3(512), lds1, s, p);
    hipLaunchKernelGGL(lora_reduce_kernel, dim3((rows * H + 255) / 256, 4), dim3(256), 0, s, p, grid, H, wide ? 1 : 0);
    return a4r_launch_status();
}
