"""CPU: liba4r_hip.so loads and exports every symbol include/a4r.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    txt = open(os.path.join(ROOT, 'include', 'a4r.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\bint\s+(a4r_\w+)\s*\(', txt)))


def test_header_declares_entry_points():
    names = declared()
    assert 'a4r_gemm_nt' in names and 'a4r_attn_bwd' in names and len(names) >= 19


def test_library_exports_every_declared_symbol():
    from adapter4rec_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f'{_lib.LIB_PATH} missing: run __graft_entry__.build()')
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared():
        assert hasattr(lib, name), name
    assert sorted(_lib.EXPORTS) == declared()
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'a4r.h')).read()
    import re as _re
    want = int(_re.search(r'#define A4R_ABI_VERSION (\d+)', hdr).group(1))
    assert lib.a4r_version() == want == _lib.ABI_VERSION


def test_binding_struct_sizes_match_header():
    """ctypes mirrors of the ABI structs must have the C layout (checked against a gcc-compiled probe)."""
    import subprocess
    import tempfile
    from adapter4rec_amd import _lib
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "a4r.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(a4r_gemm_t), sizeof(a4r_attn_t), '
           'sizeof(a4r_pack_desc_t), sizeof(a4r_sasrec_block_t), offsetof(a4r_sasrec_block_t, drop_seed), offsetof(a4r_gemm_t, c_scale_out), sizeof(a4r_layer_adapter_t), '
           'sizeof(a4r_encoder_layer_t), offsetof(a4r_encoder_layer_t, drop_seed), offsetof(a4r_encoder_layer_t, ad), offsetof(a4r_encoder_layer_t, dqkv));return 0;}\n')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 'p.c'), 'w').write(src)
        subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), os.path.join(d, 'p.c'), '-o', os.path.join(d, 'p')])
        out = subprocess.check_output([os.path.join(d, 'p')]).decode().split()
    assert [int(x) for x in out] == [ctypes.sizeof(_lib.GemmArgs), ctypes.sizeof(_lib.AttnArgs), ctypes.sizeof(_lib.PackDesc),
                                     ctypes.sizeof(_lib.SasrecBlock), _lib.SasrecBlock.drop_seed.offset, _lib.GemmArgs.c_scale_out.offset,
                                     ctypes.sizeof(_lib.LayerAdapter), ctypes.sizeof(_lib.EncoderLayer), _lib.EncoderLayer.drop_seed.offset, _lib.EncoderLayer.ad.offset,
                                     _lib.EncoderLayer.dqkv.offset]


def test_every_entry_point_refuses_null_pointers_before_launching():
    """Error behaviour at the boundary: called with NULL for every pointer and 0 for every scalar (the two argument structs zero-filled), each
    entry point that takes a pointer returns A4R_EINVAL (include/a4r.h) -- the checks sit in front of the first HIP call, so this runs without a GPU
    and nothing is enqueued.  (a4r_gemm_tail_plan is a host-side query whose outputs are optional: it returns its flag.)"""
    import torch
    from adapter4rec_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('argument-check probe is a CPU test')
    hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'a4r.h')).read(), flags=re.S)
    codes = dict(re.findall(r'#define (A4R_OK|A4R_EINVAL|A4R_ELAUNCH) \(?(-?\d+)\)?', hdr))
    assert {k: int(v) for k, v in codes.items()} == dict(A4R_OK=0, A4R_EINVAL=-1, A4R_ELAUNCH=-2)
    protos = re.findall(r'\n\s*int\s+(a4r_\w+)\s*\(([^;{]*?)\)\s*;', hdr)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    structs = dict(a4r_gemm_t=ctypes.sizeof(_lib.GemmArgs), a4r_attn_t=ctypes.sizeof(_lib.AttnArgs), a4r_encoder_layer_t=ctypes.sizeof(_lib.EncoderLayer))
    probed, keep = 0, []
    for name, args in protos:
        parts = [a.strip() for a in args.split(',')] if args.strip() not in ('', 'void') else []
        if not any('*' in a for a in parts) or name == 'a4r_gemm_tail_plan':
            continue
        vals, types = [], []
        for a in parts:
            m = re.match(r'const\s+(a4r_\w+_t)\s*\*', a)
            if m and m.group(1) in structs:
                keep.append(ctypes.create_string_buffer(structs[m.group(1)]))
                vals.append(ctypes.cast(keep[-1], ctypes.c_void_p)), types.append(ctypes.c_void_p)
            elif '*' in a:
                vals.append(None), types.append(ctypes.c_void_p)
            elif re.match(r'(const\s+)?float\b', a):
                vals.append(0.0), types.append(ctypes.c_float)
            elif re.match(r'(const\s+)?(int64_t|uint64_t|size_t)\b', a):
                vals.append(0), types.append(ctypes.c_int64)
            else:
                vals.append(0), types.append(ctypes.c_int32)
        f = getattr(lib, name)
        f.argtypes, f.restype = types, ctypes.c_int
        assert f(*vals) == -1, name
        probed += 1
    assert probed >= 44, probed


def test_no_cpu_fallback():
    import torch
    from adapter4rec_amd import _lib
    a = torch.zeros(128, 64)
    with pytest.raises(RuntimeError):
        _lib.gemm_nt(a, a, a)


def test_pack_fragment_layouts_are_permutations_of_the_kernels_read_order():
    """a4r_pack_desc_t layouts 1 / 2 (ABI 408), restated in tests/sim_lib.py: frag_index must be a permutation of the destination, and element
    (wave w, step s, [row tile nt | half h, step ks], lane, j) must be the matrix element the one-launch adapter kernels load for that fragment
    from a ROW-major copy (csrc/a4r_adapter_fused.hip: the two address forms of wd / wu)."""
    import torch
    import sim_lib
    for H in (128, 256, 512, 768, 1024):
        NW = 4 if H == 128 else 8
        CW, KS = H // NW, H // NW // 32
        i1, i2 = sim_lib.frag_index(1, 64, H), sim_lib.frag_index(2, H, 64)
        assert sorted(i1.reshape(-1).tolist()) == list(range(64 * H)) and sorted(i2.reshape(-1).tolist()) == list(range(64 * H))
        for w in (0, NW - 1):
            for lane in (0, 17, 63):
                fr, kg = lane & 15, lane >> 4
                for s in range(KS):
                    for nt in range(4):        # forward wd[s][nt] / backward wu[s][nt]: row nt*16 + fr, columns w*CW + s*32 + kg*8 + j
                        for j in (0, 7):
                            assert int(i1[nt * 16 + fr, w * CW + s * 32 + kg * 8 + j]) == ((((w * KS + s) * 4 + nt) * 64 + lane) * 8 + j)
                    for h in range(2):
                        for ks in range(2):    # forward wu[2s+h][ks] / backward image: row w*CW + s*32 + (fr>>2)*8 + h*4 + (fr&3), columns ks*32 + kg*8 + j
                            for j in (0, 7):
                                r = w * CW + s * 32 + (fr >> 2) * 8 + h * 4 + (fr & 3)
                                assert int(i2[r, ks * 32 + kg * 8 + j]) == (((((w * KS + s) * 2 + h) * 2 + ks) * 64 + lane) * 8 + j)

