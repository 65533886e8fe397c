#!/usr/bin/env python
"""Generator of adapter4rec_amd/csrc/a4r_gemm256w4_loop.inc: the hand-scheduled K loop of the four-wave 256 x 256 NT GEMM
(gemm_nt_256w4_kernel, a4r_gemm256w4.hip) as ONE inline-asm text.

    python tools/gen_gemm_w4_loop.py [--out PATH] [--variant NAME]

Why text and not HIP C++: one wave per SIMD has nobody to cover an idle matrix pipe, so every LDS read, LDS-DMA and wait has to sit in
a chosen gap between two MFMAs; hipcc clumps the reads behind one s_waitcnt (profiles/LOG.md, "Four waves, one per SIMD, on the
weight-gradient kernel").  The text below fixes the order; the HIP side (a4r_gemm256w4.hip) hands over operands in PHYSICAL registers.

Register map of a wave (wave = (wm, wn), 128 x 128 of the 256 x 256 tile; MFMA 16x16x32 bf16, operands swapped as in the eight-wave kernel:
D = Bfrag x Afrag, so a lane's four accumulator registers are four consecutive columns of one output row):
    a[0:255]        accumulators, tile (mi, ni) -> a[(8 mi + ni) * 4 .. + 3]                                   (outputs, 8 x 32 registers)
    v[0:31]         F0.a[mi]   fragments of the K-tile's first 32-deep K step (ks = 0): A rows mi * 16 + (lane & 15), 16 bytes per lane
    v[32:63]        F0.b[ni]
    v[64:95]        F1.a[mi]   second K step (ks = 1)
    v[96:127]       F1.b[ni]
    v[128:131]      LDS read bases, ring buffer 0: A ks0, A ks1, B ks0, B ks1          (inputs)
    v[132:135]      the same for ring buffer 1
    v[136:143]      LDS-DMA source offsets of this wave's 4 + 4 pieces of the A0 / A1 units (bytes from the tile's first row at K-tile 0)
    v[144:151]      the same for B0 / B1
    s[64:65]        DMA stream pointer A: &A[tile row 0][K-tile being fetched]           (in / out)
    s[66:67]        DMA stream pointer B
    s[68:69], s[70:71]   the NEXT output tile's A / B tile bases (the stream continues there after this tile's last K-tile)
    s72             K-tiles of THIS tile still to be fetched (nk - 2 on entry)           (in, clobbered)
    s73             K-tile pairs after the first (nk / 2 - 1)                            (in, clobbered)
    s74             scratch (m0 saved)
    s75             LDS byte address of the ring

LDS ring: 2 buffers x [A0 | A1 | B0 | B1] x 16 KiB; a unit is 128 rows x 128 B, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7)
(swizzle on the DMA source address and on the fragment read; the LDS image itself is lane-linear).  Wave w = 2 wm + wn fetches pieces
4 w .. 4 w + 3 (rows 32 w .. 32 w + 31) of every unit and reads fragments from units A_wm and B_wn.

Schedule of K-tile u in buffer b (64 MFMAs per K step; "gap j" = behind MFMA j of the step):
    step 0 (F0):  gaps 0, 2, .., 30   ds_read F1(u) <- buffer b                     [16 reads]
                  gap BAR             s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier      [every wave's reads of buffer b are done (WAR) and every
                                                                                      wave's DMA of K-tile u + 1 has landed (RAW)]
                  gaps BAR+1 ..       LDS-DMA of K-tile u + 2 -> buffer b           [4 units x (m0, 4 pieces)] + stream bookkeeping
    step 1 (F1):  .. DMA continues; then ds_read F0(u + 1) <- buffer b ^ 1          [16 reads], s_waitcnt lgkmcnt(0) at the end
The first K-tile of an output tile multiplies into C = 0 (no accumulator zeroing) and skips the vmcnt wait (the HIP side guarantees
K-tiles 0 and 1 have landed).
"""
import argparse
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

UNIT = 16384
BUF = 4 * UNIT


def acc(mi, ni):
    b = (8 * mi + ni) * 4
    return f'a[{b}:{b + 3}]'


def frag(setno, which, idx):
    """setno 0/1 (K step), which 'a' / 'b', idx 0..7 -> v[lo:hi]"""
    lo = setno * 64 + (0 if which == 'a' else 32) + idx * 4
    return f'v[{lo}:{lo + 3}]'


def rd(setno, buf, k):
    """k-th read (0..15) of fragment set `setno` from ring buffer `buf`: a0 b0 a1 b1 ..."""
    which = 'a' if k % 2 == 0 else 'b'
    idx = k // 2
    base = 128 + 4 * buf + (0 if which == 'a' else 2) + setno        # v128..v135
    return f'ds_read_b128 {frag(setno, which, idx)}, v{base} offset:{idx * 2048}'


def mfma(setno, j, zero, order):
    mi, ni = order[j]
    c = '0' if zero else acc(mi, ni)
    return f'v_mfma_f32_16x16x32_bf16 {acc(mi, ni)}, {frag(setno, "b", ni)}, {frag(setno, "a", mi)}, {c}'


def dma_group(buf, per_piece_m0):
    """instructions that fetch one K-tile into ring buffer `buf` (this wave's 16 pieces) and advance the stream; one list entry = one gap"""
    out = []
    for unit in range(4):            # A0 A1 B0 B1
        sp = 's[64:65]' if unit < 2 else 's[66:67]'
        for i in range(4):
            vo = 136 + unit * 4 + i
            dst = buf * BUF + unit * UNIT + i * 1024      # + wave * 4096 folded into s75' (see HIP side: s75 = lds0 + wave * 4096)
            if per_piece_m0:
                out.append(f's_add_u32 m0, s75, {dst}')
                out.append(f'global_load_lds_dwordx4 v{vo}, {sp}')
            else:
                if i == 0:
                    out.append(f's_add_u32 m0, s75, {buf * BUF + unit * UNIT}')
                out.append(f'global_load_lds_dwordx4 v{vo}, {sp}' + (f' offset:{i * 1024}' if i else ''))
    out += ['s_add_u32 s64, s64, 128', 's_addc_u32 s65, s65, 0', 's_add_u32 s66, s66, 128', 's_addc_u32 s67, s67, 0',
            's_sub_u32 s72, s72, 1', 's_cmp_eq_u32 s72, 0', 's_cselect_b64 s[64:65], s[68:69], s[64:65]', 's_cselect_b64 s[66:67], s[70:71], s[66:67]']
    return out


def default_order():
    return [(mi, ni) for mi in range(8) for ni in range(8)]


def ktile(buf, first, cfg):
    """one K-tile from ring buffer buf; first: first K-tile of the output tile"""
    order = cfg['order']
    lines = []
    g0 = {}                                                   # gap -> list of instructions, K step 0
    g1 = {}
    for k in range(16):
        g0.setdefault(cfg['rd1_start'] + k * cfg['rd1_stride'], []).append(rd(1, buf, k))
    bar = cfg['bar']
    g0.setdefault(bar, []).append('s_waitcnt lgkmcnt(0)' if first else 's_waitcnt vmcnt(0) lgkmcnt(0)')
    g0[bar].append('s_barrier')
    dma = dma_group(buf, cfg['per_piece_m0'])
    pos = bar + 1
    step = 0
    for ins in dma:                                          # one instruction per gap, running over into K step 1
        if step == 0 and pos >= 64:
            step, pos = 1, 0
        (g0 if step == 0 else g1).setdefault(pos, []).append(ins)
        pos += cfg['dma_stride']
    dma_end = pos if step == 1 else 0
    r0 = max(dma_end, cfg['rd0_start']) if cfg['rd0_after_dma'] else cfg['rd0_start']
    for k in range(16):
        g1.setdefault(min(63, r0 + k * cfg['rd0_stride']), []).append(rd(0, buf ^ 1, k))
    for j in range(64):
        lines.append(mfma(0, j, first, order))
        lines += g0.get(j, [])
    for j in range(64):
        lines.append(mfma(1, j, False, order))
        lines += g1.get(j, [])
    lines.append('s_waitcnt lgkmcnt(0)')
    return lines


VARIANTS = {
    # rd1_*: reads of F1(u) in K step 0; bar: gap of the wait + barrier; dma_stride: gaps between two DMA-group instructions;
    # rd0_*: reads of F0(u + 1) in K step 1 (after the DMA group's last instruction when rd0_after_dma)
    'v1': dict(order=default_order(), rd1_start=0, rd1_stride=2, bar=40, per_piece_m0=True, dma_stride=1, rd0_start=0, rd0_stride=2, rd0_after_dma=True),
}


def build(cfg):
    L = []
    L.append('s_mov_b32 s74, m0')
    for k in range(16):                                      # pipeline fill: F0 of K-tile 0
        L.append(rd(0, 0, k))
    L.append('s_waitcnt lgkmcnt(0)')
    L += ktile(0, True, cfg)
    L += ktile(1, False, cfg)
    L.append('s_cmp_eq_u32 s73, 0')
    L.append('s_cbranch_scc1 L_a4r_w4_done_%=')
    L.append('L_a4r_w4_loop_%=:')
    L += ktile(0, False, cfg)
    L += ktile(1, False, cfg)
    L.append('s_sub_u32 s73, s73, 1')
    L.append('s_cmp_lg_u32 s73, 0')
    L.append('s_cbranch_scc1 L_a4r_w4_loop_%=')
    L.append('L_a4r_w4_done_%=:')
    L.append('s_nop 15')                                     # the last MFMAs' results must be architecturally visible to the v_accvgpr_read the compiler emits next
    L.append('s_nop 15')
    L.append('s_mov_b32 m0, s74')
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'adapter4rec_amd', 'csrc', 'a4r_gemm256w4_loop.inc'))
    ap.add_argument('--variant', default='v1')
    a = ap.parse_args()
    cfg = VARIANTS[a.variant]
    L = build(cfg)
    n_mfma = sum(1 for x in L if x.startswith('v_mfma'))
    with open(a.out, 'w') as f:
        f.write('// GENERATED by tools/gen_gemm_w4_loop.py --variant %s -- do not edit; the schedule and the register map are documented there.\n' % a.variant)
        f.write('// %d instructions, %d MFMAs.\n' % (sum(1 for x in L if not x.endswith(':')), n_mfma))
        f.write('#define A4R_W4_LOOP_ASM \\\n')
        for x in L:
            f.write('    "%s\\n" \\\n' % x)
        f.write('    ""\n')
    print('wrote', a.out, len(L), 'lines,', n_mfma, 'MFMAs')


if __name__ == '__main__':
    main()
