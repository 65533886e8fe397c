#!/usr/bin/env python
"""Generator of tools/w4/a4r_gemm256w4_loop.inc: the hand-scheduled K loop of the four-wave 256 x 256 NT GEMM
(gemm_nt_256w4_kernel, a4r_gemm256w4.hip) as ONE inline-asm text.

    python tools/w4/gen_gemm_w4_loop.py [--out PATH] [--variant NAME]

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
    s75             LDS byte address of the ring + wave * 4096 (this wave's 4 KiB of every unit)
    s76             wave id 0..3 (selects the wave's copy of the loop where the schedule staggers the waves)

LDS ring: 2 buffers x [A0 | A1 | B0 | B1] x 16 KiB; a unit is 128 rows x 128 B, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7)
(A units; B units: c ^ (2 ((r >> 3) & 3) + ((r >> 1) & 1)), see rd(); swizzle on the DMA source address and on the fragment read; the
LDS image itself is lane-linear).  Wave w = 2 wm + wn fetches pieces
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

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

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
    # A tile mi: rows 16 mi + (lane & 15).  B tile ni: MFMA row 4 a + r is unit row 32 (ni >> 1) + 8 a + r + 4 (ni & 1), so that a lane's
    # registers of tiles 2 p and 2 p + 1 are 8 consecutive output columns (a4r_gemm256_epi.inc, A4R_EPI_NATIVE8)
    off = idx * 2048 if which == 'a' else (idx >> 1) * 4096 + (idx & 1) * 512
    return f'ds_read_b128 {frag(setno, which, idx)}, v{base} offset:{off}'


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


def ktile(buf, first, cfg, wave=0):
    """one K-tile from ring buffer buf; first: first K-tile of the output tile.  Gaps are numbered 0..127 over both K steps."""
    order = cfg['order']
    g = {}                                                    # gap -> list of instructions

    def put(pos, ins):
        g.setdefault(max(0, min(127, pos)), []).append(ins)

    for k in range(16):
        put(cfg['rd1_start'] + k * cfg['rd1_stride'], rd(1, buf, k))
    bar = cfg['bar']
    put(bar, 's_waitcnt lgkmcnt(0)' if first else 's_waitcnt vmcnt(0) lgkmcnt(0)')
    put(bar, 's_barrier')
    dma = dma_group(buf, cfg['per_piece_m0'])
    if cfg.get('stagger', 0):
        # STAGGERED issue: the four waves leave the barrier together; with every wave's piece n in the same gap their requests meet at the CU's
        # one address unit (16 cycles per 1-KiB piece) and each wave's issue -- the in-order stream, MFMAs included -- waits for the other
        # three (measured: ~60 cycles per piece and wave, tools/w4/w4_sweep.sh).  Wave w owns the gaps = w (mod 4) behind the barrier.
        st = cfg['stagger']
        pos = bar + 1 + wave * (st // 4)
        n_glds = 0
        pend = []                                            # m0 writes / bookkeeping ride in the gap BEFORE the next piece's
        for ins in dma:
            if ins.startswith('global_load_lds'):
                for x in pend:
                    put(pos - 1, x)
                pend = []
                put(pos, ins)
                pos += st
                n_glds += 1
            else:
                pend.append(ins)
        tail = pos - st + 1
        for i, x in enumerate(pend):                        # stream bookkeeping behind the last piece, one per gap
            put(tail + i, x)
        dma_end = tail + len(pend)
    else:
        pos = bar + 1
        for ins in dma:                                      # one instruction per dma_stride gaps
            put(pos, ins)
            pos += cfg['dma_stride']
        dma_end = pos
    if cfg['rd0_after_dma']:
        r0 = max(dma_end, 64 + cfg['rd0_start'])
        for k in range(16):
            put(r0 + k * cfg['rd0_stride'], rd(0, buf ^ 1, k))
    else:                                                    # reads of F0(u + 1) in the gaps the DMA pieces leave (offset rd0_phase inside a stagger period)
        r0 = 64 + cfg['rd0_start']
        for k in range(16):
            put(r0 + k * cfg['rd0_stride'], rd(0, buf ^ 1, k))
    lines = []
    for j in range(128):
        lines.append(mfma(j // 64, j % 64, first and j < 64, order))
        lines += g.get(j, [])
    lines.append('s_waitcnt lgkmcnt(0)')
    return lines


def shell_order():
    """K step 0's MFMA order for a K-tile whose F0 reads (a0 b0 a1 b1 ..) may still be landing: shell k = the tiles that need a_k or b_k and
    nothing later, so MFMA k * k needs reads 0 .. 2 k + 1 only"""
    o = []
    for k in range(8):
        o += [(i, k) for i in range(k)] + [(k, j) for j in range(k + 1)]
    return o


def ktile4(buf, first, cfg, wave):
    """the two-barrier K-tile (variant family v4): B1 = WAR only (buffer b free for K-tile u + 2), B2 = RAW with a COUNTED vmcnt (K-tile u + 1
    has landed; this K-tile's 16 pieces stay in flight), so pieces have ~1.5 K-tiles to land and the DMA queue never drains.
    No lgkmcnt waits here except in front of the barriers: insert_lgkm_waits() adds the counted ones the MFMAs need."""
    g = {}

    def put(pos, ins):
        assert 0 <= pos < 128, pos
        g.setdefault(pos, []).append(ins)

    for k in range(16):
        put(cfg['rd1_start'] + k * cfg['rd1_stride'], rd(1, buf, k))
    b1, b2, st = cfg['b1'], cfg['b2'], cfg['stagger']
    assert b1 > cfg['rd1_start'] + 15 * cfg['rd1_stride']
    put(b1, 's_waitcnt lgkmcnt(0)')
    put(b1, 's_barrier')
    pos = b1 + 1 + wave * cfg.get('wshift', st // 4)
    pend = []
    for ins in dma_group(buf, False):
        if ins.startswith('global_load_lds'):
            for x in pend:
                put(pos - 1, x)
            pend = []
            put(pos, ins)
            pos += st
        else:
            pend.append(ins)
    last_piece = pos - st
    avail = 127 - last_piece                                 # stream bookkeeping behind the last piece (scalar instructions: several may share a gap)
    assert avail >= 1, last_piece
    per = -(-len(pend) // avail)
    for i, x in enumerate(pend):
        put(last_piece + 1 + i // per, x)
    # the counted wait: K-tile u + 1's pieces (all issued during K-tile u - 1) are older than the n_before pieces of K-tile u + 2 issued so far
    n_before = sum(1 for j in range(b2 + 1) for x in g.get(j, []) if x.startswith('global_load_lds'))
    if not first:
        g.setdefault(b2, []).insert(0, f's_waitcnt vmcnt({n_before})')
        g[b2].insert(1, 's_barrier')
    else:
        g.setdefault(b2, []).insert(0, 's_barrier')
    for k in range(16):
        put(b2 + 1 + k * cfg['rd0_stride'], rd(0, buf ^ 1, k))
    lines = []
    o0, o1 = shell_order(), default_order()
    for j in range(128):
        lines.append(mfma(j // 64, j % 64, first and j < 64, o0 if j < 64 else o1))
        lines += g.get(j, [])
    return lines


def insert_lgkm_waits(L):
    """Counted s_waitcnt lgkmcnt in front of every MFMA whose fragments may still be in flight.  LDS reads return in order: with n reads issued
    so far and read r the youngest one an MFMA needs, at most n - 1 - r may be outstanding.  `done` = reads known complete.  The state at the
    top of the loop body equals the state behind the peeled pair (same K-tile text), so one linear pass over the text is exact."""
    import re
    out = []
    issued = 0
    done = 0
    last_write = {}                                          # first register of a fragment -> index of the read that writes it
    for x in L:
        if x.startswith('ds_read_b128'):
            lo = int(re.match(r'ds_read_b128 v\[(\d+):', x).group(1))
            last_write[lo] = issued
            issued += 1
        elif x.startswith('v_mfma'):
            regs = [int(m) for m in re.findall(r'v\[(\d+):\d+\]', x)]
            need = max([last_write.get(r, -1) for r in regs] + [-1])
            if need >= done:
                allow = issued - 1 - need
                assert allow >= 0
                allow = min(allow, 15)
                out.append(f's_waitcnt lgkmcnt({allow})')
                done = max(done, issued - allow)
        elif x.startswith('s_waitcnt') and 'lgkmcnt(0)' in x:
            done = issued
        elif x.endswith(':') and 'loop' in x:               # loop head: reached from the peeled pair and from the back edge, same text behind both
            pass
        out.append(x)
    return out


VARIANTS = {
    # rd1_*: reads of F1(u) in K step 0; bar: gap of the wait + barrier; dma_stride: gaps between two DMA-group instructions;
    # rd0_*: reads of F0(u + 1) in K step 1 (after the DMA group's last instruction when rd0_after_dma); stagger: gaps between two pieces of
    # one wave, wave w shifted by w * stagger / 4 (0 = all waves issue in the same gaps; needs per-wave loop bodies)
    'v1': dict(order=default_order(), rd1_start=0, rd1_stride=2, bar=40, per_piece_m0=True, dma_stride=1, rd0_start=0, rd0_stride=2, rd0_after_dma=True),
    # one m0 per unit: the instruction offset moves BOTH the global and the LDS address (tools/_probe/lds_dma_offset_probe.hip), so piece i
    # carries offset:1024 i and its source offset is given 1024 i less by the HIP side (A4R_W4_PIECE_OFFSETS)
    'v2': dict(order=default_order(), rd1_start=0, rd1_stride=2, bar=40, per_piece_m0=False, dma_stride=1, rd0_start=0, rd0_stride=2, rd0_after_dma=True),
    'v2b32': dict(order=default_order(), rd1_start=0, rd1_stride=1, bar=32, per_piece_m0=False, dma_stride=1, rd0_start=0, rd0_stride=2, rd0_after_dma=True),
    'v2b24s2': dict(order=default_order(), rd1_start=0, rd1_stride=1, bar=24, per_piece_m0=False, dma_stride=2, rd0_start=0, rd0_stride=2, rd0_after_dma=True),
    # staggered pieces: wave w in gaps bar + 1 + w + 4 n (16 pieces over 64 gaps); F0 reads in K step 1 at even offsets between them
    'v3': dict(order=default_order(), rd1_start=0, rd1_stride=1, bar=24, per_piece_m0=False, stagger=4, rd0_start=26, rd0_stride=2, rd0_after_dma=False),
    'v3b': dict(order=default_order(), rd1_start=0, rd1_stride=1, bar=24, per_piece_m0=False, stagger=4, rd0_start=0, rd0_stride=2, rd0_after_dma=True),
    'v3s8': dict(order=default_order(), rd1_start=0, rd1_stride=1, bar=20, per_piece_m0=False, stagger=6, rd0_start=30, rd0_stride=2, rd0_after_dma=False),
    # same gaps for every wave (one loop body), pieces 4 gaps apart: what the per-wave shift alone is worth
    # two barriers per K-tile, counted vmcnt, shell-ordered first K step with counted lgkmcnt (ktile4)
    'v4': dict(family=4, rd1_start=8, rd1_stride=1, b1=28, stagger=4, b2=96, rd0_stride=1),
    'v4s': dict(family=4, rd1_start=8, rd1_stride=1, b1=28, stagger=4, b2=96, rd0_stride=1, nowave=1),
    'v4b': dict(family=4, rd1_start=4, rd1_stride=1, b1=24, stagger=4, b2=92, rd0_stride=2),
    'v4c': dict(family=4, rd1_start=8, rd1_stride=1, b1=28, stagger=4, b2=110, rd0_stride=1),
    # pieces spread over the whole rest of the K-tile (the burst of 16 pieces in 64 gaps asks the memory path for twice what it sustains;
    # the requests back up into the wave's issue, MFMAs included: SQ_WAIT_INST_ANY, not SQ_WAIT_ANY, grows)
    'v5a': dict(family=4, rd1_start=8, rd1_stride=1, b1=28, stagger=6, wshift=1, b2=96, rd0_stride=1),
    'v5b': dict(family=4, rd1_start=0, rd1_stride=1, b1=20, stagger=6, wshift=1, b2=96, rd0_stride=1),
    'v5c': dict(family=4, rd1_start=0, rd1_stride=1, b1=20, stagger=7, wshift=1, b2=100, rd0_stride=1),
    'v5d': dict(family=4, rd1_start=0, rd1_stride=1, b1=20, stagger=5, wshift=1, b2=96, rd0_stride=1),
    'v3x': dict(order=default_order(), rd1_start=0, rd1_stride=1, bar=24, per_piece_m0=False, stagger=4, nowave=1, rd0_start=26, rd0_stride=2, rd0_after_dma=False),
}


def ablate(L, abl):
    """timing-only builds (results wrong): 1 no LDS-DMA, 2 no vmcnt wait, 4 no fragment reads, 8 no barrier, 16 no MFMA"""
    out = []
    for x in L:
        if abl & 1 and x.startswith('global_load_lds'):
            continue
        if abl & 2 and x.startswith('s_waitcnt vmcnt(0) lgkmcnt(0)'):
            x = 's_waitcnt lgkmcnt(0)'
        if abl & 4 and x.startswith('ds_read'):
            continue
        if abl & 8 and x == 's_barrier':
            continue
        if abl & 16 and x.startswith('v_mfma'):
            continue
        out.append(x)
    return out


def body(cfg, wave, tag):
    L = []
    fam4 = cfg.get('family', 0) == 4
    kt = (lambda b, f: ktile4(b, f, cfg, wave)) if fam4 else (lambda b, f: ktile(b, f, cfg, wave))
    for k in range(16):                                      # pipeline fill: F0 of K-tile 0
        L.append(rd(0, 0, k))
    if not fam4:
        L.append('s_waitcnt lgkmcnt(0)')
    L += kt(0, True)
    L += kt(1, False)
    L.append('s_cmp_eq_u32 s73, 0')
    L.append('s_cbranch_scc1 L_a4r_w4_done_%=')
    L.append(f'L_a4r_w4_loop{tag}_%=:')
    L += kt(0, False)
    L += kt(1, False)
    L.append('s_sub_u32 s73, s73, 1')
    L.append('s_cmp_lg_u32 s73, 0')
    L.append(f's_cbranch_scc1 L_a4r_w4_loop{tag}_%=')
    if fam4:
        L = insert_lgkm_waits(L)
    return L


def build(cfg):
    L = ['s_mov_b32 s74, m0']
    if cfg.get('stagger', 0) and not cfg.get('nowave', 0):   # one copy of the loop per wave (s76 = wave id 0..3): its pieces sit in its own gaps
        for w in (1, 2, 3):
            L.append(f's_cmp_eq_u32 s76, {w}')
            L.append(f's_cbranch_scc1 L_a4r_w4_wave{w}_%=')
        for w in range(4):
            if w:
                L.append(f'L_a4r_w4_wave{w}_%=:')
            L += body(cfg, w, f'_w{w}')
            if w < 3:
                L.append('s_branch L_a4r_w4_done_%=')
    else:
        L += body(cfg, 0, '')
    L.append('L_a4r_w4_done_%=:')
    L.append('s_waitcnt lgkmcnt(0)')                         # the (unused) fragment reads of the tile's last K step must not land in registers the compiler has reused
    L.append('s_nop 15')                                     # the last MFMAs' results must be architecturally visible to the v_accvgpr_read that follow
    L.append('s_nop 15')
    L.append('s_mov_b32 m0, s74')
    return L


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), 'a4r_gemm256w4_loop.inc'))
    ap.add_argument('--variant', default='v1')
    ap.add_argument('--abl', type=int, default=0, help='timing-only ablation bits (see ablate()); never for the shipped library')
    a = ap.parse_args()
    cfg = VARIANTS[a.variant]
    L = ablate(build(cfg), a.abl)
    n_mfma = sum(1 for x in L if x.startswith('v_mfma'))
    with open(a.out, 'w') as f:
        f.write('// GENERATED by tools/w4/gen_gemm_w4_loop.py --variant %s -- do not edit; the schedule and the register map are documented there.\n' % a.variant)
        f.write('// %d instructions, %d MFMAs.\n' % (sum(1 for x in L if not x.endswith(':')), n_mfma))
        f.write('#define A4R_W4_PIECE_OFFSETS %d\n' % (0 if cfg.get('per_piece_m0', False) else 1))
        f.write('#define A4R_W4_LOOP_ASM \\\n')
        for x in L:
            f.write('    "%s\\n" \\\n' % x)
        f.write('    ""\n')
    print('wrote', a.out, len(L), 'lines,', n_mfma, 'MFMAs')


if __name__ == '__main__':
    main()
