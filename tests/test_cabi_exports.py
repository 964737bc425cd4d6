"""CPU checks of the drop-in boundary: libnnr_hip.so loads and exports every symbol include/nnr_hip.h declares, and the
ctypes mirrors of the argument structs have the C layout.  No compute calls (there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'nnr_hip.h')


def _declared():
    src = open(HEADER).read()
    return sorted(set(re.findall(r'^\s*(?:int|size_t)\s+(nnr_\w+)\s*\(', src, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from nnr_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = _declared()
    assert len(decl) >= 30
    for name in decl:
        assert hasattr(lib, name), 'symbol %s declared in include/nnr_hip.h but not exported' % name
    assert sorted(_lib.SYMBOLS) == decl, 'nnr_amd/_lib.py:SYMBOLS must list exactly the header\'s entry points'
    assert lib.nnr_version() >= 1


def test_ctypes_structs_match_c_layout(tmp_path):
    from nnr_amd import _lib
    src = tmp_path / 'sz.cpp'
    src.write_text('#include "%s"\n#include <stdio.h>\n#include <stddef.h>\nint main(){printf("%%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu %%zu\\n",'
                   'sizeof(nnr_gemm_args),sizeof(nnr_lstm_problem),sizeof(nnr_pool_args),offsetof(nnr_gemm_args,tile),'
                   'offsetof(nnr_pool_args,lddv),sizeof(nnr_corpus_tables),sizeof(nnr_batch_out),offsetof(nnr_corpus_tables,K1),'
                   'offsetof(nnr_lstm_problem,sync));}\n' % HEADER)
    exe = tmp_path / 'sz'
    subprocess.check_call(['hipcc', '-o', str(exe), str(src)], stderr=subprocess.DEVNULL)
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [ctypes.sizeof(_lib.GemmArgs), ctypes.sizeof(_lib.LstmProblem), ctypes.sizeof(_lib.PoolArgs), _lib.GemmArgs.tile.offset,
            _lib.PoolArgs.lddv.offset, ctypes.sizeof(_lib.CorpusTables), ctypes.sizeof(_lib.BatchOut), _lib.CorpusTables.K1.offset,
            _lib.LstmProblem.sync.offset]
    assert got == want


def _params(name):
    """Parameter list of an entry point as declared in the header (comments stripped)."""
    src = re.sub(r'/\*.*?\*/', ' ', open(HEADER).read(), flags=re.S)
    m = re.search(r'\b(?:int|size_t)\s+%s\s*\(([^;{]*?)\)\s*;' % re.escape(name), src, flags=re.S)
    assert m, name
    body = ' '.join(m.group(1).split())
    return [] if body in ('', 'void') else [q.strip() for q in body.split(',')]


def test_tape_registry_matches_the_header():
    """csrc/tape.hip records and replays entry points through one generated thunk each (8-byte argument slots -> typed call):
    every entry point that takes a trailing hipStream_t must be recordable, with exactly the header's argument count; host-only
    queries must not be.  The pool struct's round-3 fields sit behind the round-2 ones (the Python mirror appends them)."""
    from nnr_amd import _lib
    lib = _lib.lib()
    n_rec = 0
    for name in _declared():
        ps = _params(name)
        fid = lib.nnr_tape_fn_id(name.encode())
        if ps and ps[-1].startswith('hipStream_t') and not name.startswith('nnr_tape_'):
            assert fid >= 0, '%s takes a stream but has no thunk in csrc/tape.hip REGISTRY' % name
            assert lib.nnr_tape_fn_nargs(fid) == len(ps) - 1, (name, lib.nnr_tape_fn_nargs(fid), ps)
            n_rec += 1
        else:
            assert fid < 0, '%s has no stream argument and must not be recordable' % name
    assert n_rec >= 50
    assert _lib.PoolArgs.th.offset > _lib.PoolArgs.lddv.offset and _lib.PoolArgs.w2.offset > _lib.PoolArgs.th.offset


def test_product_path_refuses_cpu_tensors():
    import torch
    from nnr_amd import ops, _lib
    with pytest.raises(_lib.NnrHipError):
        ops.add_(torch.zeros(8), torch.zeros(8))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'nnr_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in txt.replace('oracle/nnr_oracle.py:length_order', ''), f + ' must not reference the oracle'
