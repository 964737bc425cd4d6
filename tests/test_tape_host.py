"""Host logic of the launch-sequence tape WITHOUT a GPU (csrc/tape.hip, nnr_amd/tape.py; include/nnr_hip.h `nnr_tape_*`).

Recording and replaying only moves argument bytes around on the host; what a replay launches is decided by the entry points it
calls.  Two of them return before their first HIP call for an empty problem -- `nnr_fill_zero(p, 0 bytes)` and `nnr_gemm_f32` with
M = 0 -- and refuse NULL operands for a non-empty one (NNR_ERR_ARG = -1).  That is enough to drive the whole mechanism on the
CPU: a replay of such calls succeeds or fails depending on the bytes the PATCHES wrote into the recorded arguments, and the failing
call is named by `nnr_tape_last_error`.  (The GPU twin -- same tape, real launches -- is tests/test_hip_tape_gpu.py.)"""
import ctypes as C

import pytest
import torch

from nnr_amd import _lib as L

ERR_ARG = -1


def _lib():
    lib = L.lib()
    lib.nnr_tape_patch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int64]
    return lib


def _call(lib, h, name, slots, stream=0, blobs=(), tag=-1):
    """nnr_tape_call with plain Python values; blobs: [(slot, ctypes object)].  -> (call index, slot byte offset, [blob byte offsets])"""
    fid = lib.nnr_tape_fn_id(name.encode())
    arr = (C.c_uint64 * max(1, len(slots)))(*slots)
    nb = len(blobs)
    bslot = (C.c_int * max(1, nb))(*[b[0] for b in blobs])
    bptr = (C.c_void_p * max(1, nb))(*[C.addressof(b[1]) for b in blobs])
    bbytes = (C.c_size_t * max(1, nb))(*[C.sizeof(b[1]) for b in blobs])
    soff, boff = C.c_size_t(), (C.c_size_t * max(1, nb))()
    rc = lib.nnr_tape_call(h, fid, C.c_void_p(stream), arr, len(slots), bslot, bptr, bbytes, nb, tag, C.byref(soff), boff)
    return rc, soff.value, [boff[i] for i in range(nb)]


def _replay(lib, h, seg, values=(), inputs=()):
    v = (C.c_uint64 * max(1, len(values)))(*values)
    p = (C.c_uint64 * max(1, len(inputs)))(*inputs)
    return lib.nnr_tape_replay(h, seg, v, len(values), p, len(inputs), -1)


def _last_error(lib, h):
    rc, call, name = C.c_int(), C.c_int(), C.create_string_buffer(64)
    assert lib.nnr_tape_last_error(h, C.byref(rc), C.byref(call), name, 64) == 0
    return rc.value, call.value, name.value.decode()


def _info(lib, h):
    calls, ops, segs, streams, nbytes = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
    assert lib.nnr_tape_info(h, C.byref(calls), C.byref(ops), C.byref(segs), C.byref(streams), C.byref(nbytes)) == 0
    return calls.value, ops.value, segs.value, streams.value, nbytes.value


def test_value_and_input_patches_segments_and_error_reporting():
    lib = _lib()
    h = C.c_void_p()
    assert lib.nnr_tape_create(C.byref(h)) == 0
    # argument validation of the recorder
    assert lib.nnr_tape_fn_id(b'nnr_version') < 0 and lib.nnr_tape_fn_id(b'no_such_entry_point') < 0
    assert _call(lib, h, 'nnr_fill_zero', [0])[0] == ERR_ARG                       # wrong argument count (takes p, bytes)
    assert _replay(lib, h, 0) == ERR_ARG                                           # not finalized yet
    # call 0 (segment 0): nnr_fill_zero(p = input[0] + 16, bytes = value[1] + 0): a value patch of width 8 and an input patch
    rc, s0, _ = _call(lib, h, 'nnr_fill_zero', [0xdead0, 0])
    assert rc == 0
    assert lib.nnr_tape_patch(h, s0, 1000, 8, 16) == 0
    assert lib.nnr_tape_patch(h, s0 + 8, 1, 8, 0) == 0
    assert lib.nnr_tape_patch(h, s0 + 8, 1, 3, 0) == ERR_ARG                       # width must be 4 or 8
    assert lib.nnr_tape_patch(h, 1 << 20, 1, 8, 0) == ERR_ARG                      # outside the recorded arguments
    assert lib.nnr_tape_segment(h) == 1                                            # the host does something of its own here
    # call 1 (segment 1): nnr_fill_zero(p = value[0] - 5 as a 4-byte patch over a zero slot, bytes = 1): fails unless p != NULL
    rc, s1, _ = _call(lib, h, 'nnr_fill_zero', [0, 0])
    assert rc == 1
    assert lib.nnr_tape_patch(h, s1, 0, 4, -5) == 0
    assert lib.nnr_tape_patch(h, s1 + 8, 2, 4, 0) == 0                             # bytes = value[2]
    assert _info(lib, h) == (2, 2, 2, 1, 32)
    assert lib.nnr_tape_finalize(h) == 0 and lib.nnr_tape_finalize(h) == ERR_ARG
    assert _call(lib, h, 'nnr_fill_zero', [0, 0])[0] == ERR_ARG                    # frozen
    assert _info(lib, h)[2] == 2
    # segment 0 with bytes = 0: returns before touching the (bogus) pointer; patches need all their values / inputs
    assert _replay(lib, h, 0, values=[5, 0, 0], inputs=[0x1000]) == 0
    assert _replay(lib, h, 0, values=[5], inputs=[0x1000]) == ERR_ARG              # value kind 1 / 2 missing
    assert _replay(lib, h, 0, values=[5, 0, 0], inputs=[]) == ERR_ARG              # input 0 missing
    assert _replay(lib, h, 2, values=[5, 0, 0], inputs=[0x1000]) == ERR_ARG        # no such segment
    # segment 1 sees the values patched in by segment 0 of the same step: p = 5 - 5 = NULL with bytes = 0 is fine ...
    assert _replay(lib, h, 1) == 0
    # ... and NULL with bytes = 1 is refused by the entry point: the replay reports which call
    assert _replay(lib, h, 0, values=[5, 0, 1], inputs=[0x1000]) == 0
    assert _replay(lib, h, 1) == ERR_ARG
    assert _last_error(lib, h) == (ERR_ARG, 1, 'nnr_fill_zero')
    assert lib.nnr_tape_destroy(h) == 0


def test_blob_arguments_are_copied_relocated_and_patchable():
    """Host structs (nnr_gemm_args) are copied into the tape; the argument slot is pointed at the copy on finalize (the arena may
    have moved while recording), and fields INSIDE the copy take patches."""
    lib = _lib()
    h = C.c_void_p()
    assert lib.nnr_tape_create(C.byref(h)) == 0
    g = L.GemmArgs()
    g.A, g.B, g.C, g.M, g.N, g.K = 0x1000, 0x2000, 0x3000, 0, 8, 8                 # empty problem: NNR_OK before any launch
    rc, s0, b = _call(lib, h, 'nnr_gemm_f32', [0], blobs=[(0, g)])
    assert rc == 0 and b[0] % 8 == 0
    assert lib.nnr_tape_patch(h, b[0] + L.GemmArgs.A.offset, 1000, 8, 0) == 0      # A = input[0]
    g.A = 0                                                                        # the caller's struct is free to change / die
    for _ in range(300):                                                           # grow the arena: earlier copies move with it
        assert _call(lib, h, 'nnr_fill_zero', [0, 0])[0] > 0
    assert lib.nnr_tape_finalize(h) == 0
    assert _replay(lib, h, 0, inputs=[0x5000]) == 0
    assert _replay(lib, h, 0, inputs=[0]) == ERR_ARG                               # A = NULL now: refused by nnr_gemm_f32 itself
    assert _last_error(lib, h) == (ERR_ARG, 0, 'nnr_gemm_f32')
    assert lib.nnr_tape_destroy(h) == 0


def test_tape_class_classifies_seeds_and_input_pointers():
    """nnr_amd.tape.Tape on the same two entry points: pointers into the batch tensors become input patches (with their offset),
    uint32 seeds within 64 of the step's dropout seeds become value patches, anything else is refused while recording."""
    from nnr_amd import tape as T
    buf = torch.zeros(64)
    t = T.Tape([buf], {'news_seed': 1000, 'user_seed': 5000})

    def step():
        lib = L.lib()                                                              # the recording proxy
        L.check(lib.nnr_fill_zero(C.c_void_p(buf.data_ptr() + 16), C.c_size_t(0), C.c_void_p(0)), 'nnr_fill_zero')
        g = L.GemmArgs()
        g.A, g.B, g.C, g.M, g.N, g.K = buf.data_ptr(), 0x2000, 0x3000, 0, 8, 8
        g.drop_target, g.drop_p, g.drop_seed = 3, 0.2, 1003
        L.check(lib.nnr_gemm_f32(C.byref(g), C.c_void_p(0)), 'nnr_gemm_f32')
        assert lib.nnr_version() >= 1                                              # host-only query: passes through, not recorded
        return 'done'

    assert t.record(step) == 'done' and not T.recording()
    info = t.info()
    assert (info['calls'], info['ops'], info['segments'], info['streams']) == (2, 2, 1, 1)
    assert L.lib() is not None and type(L.lib()).__name__ == 'CDLL'               # the proxy is gone

    class Batch:                                                                   # replay only asks a batch tensor for its address
        def __init__(self, p):
            self.p = p

        def data_ptr(self):
            return self.p

    vals = {'news_seed': 7, 'user_seed': 9, 'adam_step': 1}
    t.replay(vals, [buf])
    t.replay(vals, [Batch(0x7000)])
    with pytest.raises(L.NnrHipError, match=r'call 1 \(nnr_gemm_f32\)'):           # A = input[0] + 0 = NULL
        t.replay(vals, [Batch(0)])
    t.close()

    t2 = T.Tape([buf], {'news_seed': 1000, 'user_seed': 5000})

    def bad_seed():
        g = L.GemmArgs()
        g.A, g.B, g.C, g.M = 0x1000, 0x2000, 0x3000, 0
        g.drop_target, g.drop_seed = 3, 99999                                      # not derived from this step's seeds
        L.check(L.lib().nnr_gemm_f32(C.byref(g), C.c_void_p(0)), 'nnr_gemm_f32')

    with pytest.raises(T.TapeError, match='dropout seeds'):
        t2.record(bad_seed)
    assert not T.recording() and type(L.lib()).__name__ == 'CDLL'                  # a failed recording leaves nothing patched
    t2.close()


def test_pointer_provenance_guard_and_seed_separation():
    """Round-3 advisor: (1) a device pointer that is neither inside a batch tensor, nor inside a buffer the tape keeps alive, nor inside
    a `known` long-lived buffer (the trainer's flat parameter / gradient / moment buffers) must not be baked into a recording silently:
    it is listed in `violations` (the trainer then discards the tape and stays call by call); (2) a step whose two dropout seeds lie
    within 128 of each other cannot be recorded (derived seeds of one encoder would be attributed to the other)."""
    from nnr_amd import tape as T
    batch, flat, cache = torch.zeros(64), torch.zeros(256), torch.zeros(32)
    t = T.Tape([batch], {'news_seed': 1000, 'user_seed': 5000}, known=[flat])

    def step():
        lib = L.lib()
        t.keep.append(cache)                                                       # what ops.tape_keep / the torch.empty wrapper do
        g = L.GemmArgs()
        g.A, g.B, g.C, g.M, g.N, g.K = batch.data_ptr() + 8, flat.data_ptr() + 512, cache.data_ptr(), 0, 8, 8
        L.check(lib.nnr_gemm_f32(C.byref(g), C.c_void_p(0)), 'nnr_gemm_f32')      # input + known + kept: all vouched for
        g2 = L.GemmArgs()
        g2.A, g2.B, g2.C, g2.M, g2.N, g2.K = batch.data_ptr(), flat.data_ptr(), 0x7f0000001000, 0, 8, 8
        g2.bias = flat.data_ptr() + 4 * flat.numel()                               # one past the end of the known buffer
        L.check(lib.nnr_gemm_f32(C.byref(g2), C.c_void_p(0)), 'nnr_gemm_f32')
        L.check(lib.nnr_fill_zero(C.c_void_p(0x7f0000002000), C.c_size_t(0), C.c_void_p(0)), 'nnr_fill_zero')
        L.check(lib.nnr_fill_zero(C.c_void_p(cache.data_ptr() + 64), C.c_size_t(0), C.c_void_p(0)), 'nnr_fill_zero')

    t.record(step)
    assert sorted((n, str(w)) for n, w, _ in t.violations) == [('nnr_fill_zero', '0'), ('nnr_gemm_f32', 'C'), ('nnr_gemm_f32', 'bias')]
    assert {p for _, _, p in t.violations} == {0x7f0000001000, 0x7f0000002000, flat.data_ptr() + 4 * flat.numel()}
    t.close()
    for a, b in ((1000, 1100), (1100, 1000), (5, 0xFFFFFFF0), (7, 7)):
        with pytest.raises(T.TapeError, match='within 128'):
            T.Tape([batch], {'news_seed': a, 'user_seed': b})
    T.Tape([batch], {'news_seed': 1000, 'user_seed': 1128}).close()


def test_rejected_call_leaves_the_tape_untouched():
    """Round-3 advisor: nnr_tape_call validates every blob BEFORE it appends anything -- a refused call must not leave orphaned slots
    or blob references that finalize would later patch."""
    lib = _lib()
    h = C.c_void_p()
    assert lib.nnr_tape_create(C.byref(h)) == 0
    g = L.GemmArgs()
    g.A, g.B, g.C, g.M = 0x1000, 0x2000, 0x3000, 0
    before = _info(lib, h)
    fid = lib.nnr_tape_fn_id(b'nnr_gemm_f32')
    arr = (C.c_uint64 * 1)(0)
    bslot = (C.c_int * 2)(0, 5)                                                    # second blob names a slot the call does not have
    bptr = (C.c_void_p * 2)(C.addressof(g), C.addressof(g))
    bbytes = (C.c_size_t * 2)(C.sizeof(g), C.sizeof(g))
    assert lib.nnr_tape_call(h, fid, C.c_void_p(0), arr, 1, bslot, bptr, bbytes, 2, -1, None, None) == ERR_ARG
    bptr2 = (C.c_void_p * 2)(C.addressof(g), None)
    bslot2 = (C.c_int * 2)(0, 0)
    assert lib.nnr_tape_call(h, fid, C.c_void_p(0), arr, 1, bslot2, bptr2, bbytes, 2, -1, None, None) == ERR_ARG
    assert _info(lib, h) == before                                                 # no call, no op, no argument bytes were added
    assert _call(lib, h, 'nnr_gemm_f32', [0], blobs=[(0, g)])[0] == 0
    assert lib.nnr_tape_finalize(h) == 0 and _replay(lib, h, 0) == 0
    assert lib.nnr_tape_destroy(h) == 0
