"""The four-wave GEMM experiment (tools/w4/a4r_gemm256w4.hip; not part of the shipped library, so not part of tests/ either: python -m pytest tools/w4/check_isa.py) keeps its accumulators in AGPRs the compiler is told nothing about between the
hand-scheduled K loop and the epilogue's reads.  That is only sound while hipcc itself never touches an AGPR in those kernels and nothing spills:
checked here in the ISA hipcc emits for EVERY instantiation (device-only -S, ~25 s, no GPU).  Also: the committed loop text is what the generator
writes for the variant its header names."""
import os
import re
import shutil
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'adapter4rec_amd', 'csrc')
HIPCC = '/opt/rocm/bin/hipcc'


def test_loop_text_is_generator_output(tmp_path):
    inc = os.path.join(HERE, 'a4r_gemm256w4_loop.inc')
    head = open(inc).readline()
    m = re.search(r'--variant (\w+)', head)
    assert m, head
    out = tmp_path / 'loop.inc'
    subprocess.check_call([sys.executable, os.path.join(HERE, 'gen_gemm_w4_loop.py'), '--variant', m.group(1), '--out', str(out)], stdout=subprocess.DEVNULL)
    assert open(inc).read() == open(out).read(), 'a4r_gemm256w4_loop.inc is stale: run tools/w4/gen_gemm_w4_loop.py --variant ' + m.group(1)


def test_scoreboard_waits_cover_every_fragment():
    """independent re-check of insert_lgkm_waits: replay the text with in-order LDS returns and make sure no MFMA can read a fragment whose read
    is not provably complete (reads complete oldest first; a counted wait lgkmcnt(n) leaves at most the n youngest outstanding)"""
    sys.path.insert(0, HERE)
    import gen_gemm_w4_loop as G
    head = open(os.path.join(HERE, 'a4r_gemm256w4_loop.inc')).readline()
    cfg = G.VARIANTS[re.search(r'--variant (\w+)', head).group(1)]
    for wave in range(4):
        L = G.body(cfg, wave, '_t')
        for rounds in (1, 3):                       # the loop body once and three times behind the peeled pair
            i0 = next(i for i, x in enumerate(L) if x.endswith(':'))
            i1 = len(L) - 3                         # s_sub / s_cmp / s_cbranch
            text = L[:i0] + (L[i0 + 1:i1]) * rounds
            issued, complete, writer = 0, 0, {}
            for x in text:
                if x.startswith('ds_read_b128'):
                    writer[int(re.match(r'ds_read_b128 v\[(\d+):', x).group(1))] = issued
                    issued += 1
                elif x.startswith('s_waitcnt') and 'lgkmcnt' in x:
                    n = int(re.search(r'lgkmcnt\((\d+)\)', x).group(1))
                    complete = max(complete, issued - n)
                elif x.startswith('v_mfma'):
                    for r in re.findall(r'v\[(\d+):\d+\]', x):
                        assert writer.get(int(r), -1) < complete, (wave, rounds, x)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')
def test_no_agpr_outside_the_asm_blocks_and_no_scratch(tmp_path):
    s_path = tmp_path / 'w4.s'
    subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-function', '-Wno-unused-command-line-argument',
                           '--cuda-device-only', '-I' + CSRC, '-S', os.path.join(HERE, 'a4r_gemm256w4.hip'), '-o', str(s_path)], cwd=CSRC)
    kernels, cur, in_asm = {}, None, False
    areg = re.compile(r'(?<![\w.$])a\[?\d')
    meta_scratch = {}
    for line in open(s_path):
        m = re.match(r'^(_ZN\S*gemm_nt_256w4_kernel\S*):', line)
        if m:
            cur = m.group(1)
            kernels[cur] = dict(bad=[], asm=0)
            continue
        if line.startswith('.Lfunc_end'):
            cur = None
        if cur is None:
            m = re.match(r'\s+\.private_segment_fixed_size:\s+(\d+)', line)
            if m:
                meta_scratch.setdefault('sizes', []).append(int(m.group(1)))
            continue
        if '#ASMSTART' in line:
            in_asm = True
            continue
        if '#ASMEND' in line:
            in_asm = False
            continue
        code = line.split(';')[0]
        if in_asm:
            kernels[cur]['asm'] += 1
        elif code.strip() and not code.strip().startswith('.') and areg.search(code):
            kernels[cur]['bad'].append(line.strip())
    assert len(kernels) >= 17, list(kernels)
    for k, v in kernels.items():
        assert v['asm'] > 2000, (k, v['asm'])            # the K loop is there
        assert not v['bad'], (k, v['bad'][:5])
    assert meta_scratch.get('sizes') and all(x == 0 for x in meta_scratch['sizes']), meta_scratch
