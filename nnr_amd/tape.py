"""Host side of the launch-sequence tape (csrc/tape.hip, include/nnr_hip.h `nnr_tape_*`).

`Tape.record(fn)` runs `fn()` -- one training step written as a sequence of C-ABI calls (nnr_amd.step) -- EAGERLY, and while it
runs captures
  * every call into libnnr_hip.so (function, arguments, HIP stream): `_lib.lib()` hands out a recording proxy;
  * every cross-stream dependency: all of torch's stream joins end in `Event.record(stream)` / `Event.wait(stream)`, which are
    wrapped for the duration of the recording;
  * every device buffer the step allocates (`torch.empty / zeros / empty_like / zeros_like` results are kept alive by the tape, so
    the recorded addresses stay valid and are never handed to anybody else);
  * what changes from step to step: dropout seeds and Adam's step number (value patches, recognised by argument type and
    value), pointers into the batch tensors (input patches, recognised by address range);
  * host callbacks (`host_call`): points where the host has work of its own (torch.distributed's all-reduce); they split the
    tape into segments and are re-run between the segments of a replay, under the HIP stream that was current when recorded.
`Tape.replay(values, inputs)` then costs one C-ABI call per segment.  No CPU fallback: without the library nothing records."""
import ctypes as C
import struct

import torch

from . import _lib as L
from . import profile as _prof

MASK64 = (1 << 64) - 1
# entry points that are host-side queries / set-up (no stream argument): passed through, never recorded
_PASS = {'nnr_version', 'nnr_lstm_dims', 'nnr_lstm_sync_bytes', 'nnr_lstm_sync_diag_offset', 'nnr_lstm_set_timeout_counter',
         'nnr_slot_workspace_floats', 'nnr_dp_unique_id', 'nnr_dp_init', 'nnr_dp_destroy', 'nnr_dp_emulate_ranks', 'nnr_adam_skipped_steps', 'nnr_adam_skipped_peek',
         'nnr_token_sort_workspace_bytes', 'nnr_embed_scatter_sorted_workspace_floats'}
_INT_TYPES = (C.c_int, C.c_long, C.c_size_t, C.c_uint32, C.c_ulong, C.c_int64, C.c_uint64)
ACTIVE = [None]            # the tape that is recording right now
VALUE_KINDS = {'news_seed': 0, 'user_seed': 1, 'adam_step': 2}
_VALUE_ARGS = {('nnr_clip_adam', 13): 'adam_step'}            # (entry point, argument index) -> value kind, for plain integers
_HANDLE_ARGS = {'nnr_dp_allreduce', 'nnr_dp_broadcast'}       # their FIRST pointer is an opaque host handle (the communicator), not device memory


NEWS_SEED_STRIDE = 104729      # NewsEncoder._next_seed: the per-call seed advances by this much (news_encoders.py)
TAG_ALL = [False]


def _no_flops(vals=None):
    return 0.0


class TapeError(L.NnrHipError):
    pass


def recording():
    return ACTIVE[0] is not None


def host_call(fn):
    """Run `fn()` now; inside a recording, also make it a host callback of the tape (re-run at this point of every replay, under
    the HIP stream that is current now).  Stream / event operations `fn` performs are NOT recorded: they happen live again."""
    t = ACTIVE[0]
    if t is None or t._in_host_call:
        return fn()
    return t._host_call(fn)


class _Proxy:
    """Stands in for the ctypes library while a tape records: every entry point still runs, and is appended to the tape."""

    def __init__(self, real, tape):
        self._real, self._tape, self._cache = real, tape, {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            real = getattr(self._real, name)
            if name in _PASS or name.startswith('nnr_tape_'):
                fn = real
            else:
                tape = self._tape

                def fn(*args, _real=real, _name=name):
                    rc = _real(*args)
                    if rc == 0 and not tape._in_host_call:
                        tape._add_call(_name, args)
                    return rc
            self._cache[name] = fn
        return fn


class Tape:
    def __init__(self, inputs, seeds, known=()):
        """inputs: the batch tensors of the recorded step (pointers into them become input patches); seeds: {'news_seed': int,
        'user_seed': int} of the recorded step (uint32 arguments within 64 of one of them become value patches); known: tensors that
        outlive the tape by construction (the trainer's flat parameter / gradient / moment buffers, of which every parameter is a
        view).  Every device pointer that reaches a recorded call must resolve to an input, a `known` buffer or a buffer the tape
        itself keeps alive (torch.empty* results of the step, ops.tape_keep'd caches): anything else -- a temporary made by a torch
        op outside the library, e.g. an int64 -> int32 id conversion or a .contiguous() copy -- is listed in `violations` and the
        trainer discards the recording (round-3 advisor: such a pointer would be baked in as a constant and read stale memory)."""
        d = (int(seeds['news_seed']) - int(seeds['user_seed'])) & 0xFFFFFFFF
        if min(d, (1 << 32) - d) < 128:
            # the two per-step seeds advance by different strides; this close together a derived seed (base + small offset) of one
            # encoder would be attributed to the other's base and replays would draw wrong masks: record on another step
            raise TapeError('news / user dropout seeds of this step are within 128 of each other: not recordable (record the next step)')
        self.lib = L.lib()
        self.known = [t for t in known if torch.is_tensor(t)]
        self.violations = []                # (entry point, argument / field, pointer) that resolved to nothing the tape can vouch for
        self._ranges, self._ranged = [], 0
        self.h = C.c_void_p()
        L.check(self.lib.nnr_tape_create(C.byref(self.h)), 'nnr_tape_create')
        self.inputs = [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size(), tuple(t.shape), t.dtype) for t in inputs]
        self.seeds = dict(seeds)
        self.keep = []                      # tensors / events that must outlive the tape's recorded addresses
        self.host_calls = []                # (fn, raw stream handle) per segment boundary
        self.tags = []                      # (family, flops_fn) per tagged call
        self._pending_tag = None
        self._in_host_call = False
        self._fn_ids = {}
        self.skipped_event_waits = 0
        self.final = False
        self.calls = 0
        self._nsets = 0

    # ------------------------------------------------------------------------------------------------ recording
    def record(self, fn):
        assert ACTIVE[0] is None and not self.final
        real = L._lib
        ev_record, ev_wait = torch.cuda.Event.record, torch.cuda.Event.wait
        allocs = {k: getattr(torch, k) for k in ('empty', 'zeros', 'empty_like', 'zeros_like')}
        tape = self

        def record(ev, stream=None):
            if stream is None:
                stream = torch.cuda.current_stream()
            ev_record(ev, stream)
            if not tape._in_host_call:
                tape.keep.append(ev)
                L.check(tape.lib.nnr_tape_event_record(tape.h, C.c_uint64(id(ev)), C.c_void_p(stream.cuda_stream)), 'nnr_tape_event_record')

        def wait(ev, stream=None):
            if stream is None:
                stream = torch.cuda.current_stream()
            ev_wait(ev, stream)
            if not tape._in_host_call:
                rc = tape.lib.nnr_tape_event_wait(tape.h, C.c_void_p(stream.cuda_stream), C.c_uint64(id(ev)))
                if rc != 0:
                    # an event recorded before this step began (e.g. a cached W^T copy): complete by the time any replay starts, because
                    # every step ends with all of its streams joined into the stream the next one starts on
                    tape.skipped_event_waits += 1

        def keeping(f, fills):
            def g(*a, **k):
                t = f(*a, **k)
                if t.is_cuda and not tape._in_host_call:
                    if fills:
                        raise TapeError('torch.%s of a device tensor inside a recorded step: its fill kernel would not be part of the tape '
                                        '(first-use allocations belong to the warm-up steps)' % f.__name__)
                    tape.keep.append(t)
                return t
            return g

        ACTIVE[0] = self
        L._lib = _Proxy(real, self)
        torch.cuda.Event.record, torch.cuda.Event.wait = record, wait
        for k, f in allocs.items():
            setattr(torch, k, keeping(f, k.startswith('zeros')))
        _prof.TAPE_HOOK[0] = self._tag
        try:
            out = fn()
        finally:
            _prof.TAPE_HOOK[0] = None
            for k, f in allocs.items():
                setattr(torch, k, f)
            torch.cuda.Event.record, torch.cuda.Event.wait = ev_record, ev_wait
            L._lib = real
            ACTIVE[0] = None
        L.check(self.lib.nnr_tape_finalize(self.h), 'nnr_tape_finalize')
        self.final = True
        return out

    def _tag(self, family, flops_fn):
        self._pending_tag = (family, flops_fn)

    def _fn(self, name):
        i = self._fn_ids.get(name)
        if i is None:
            i = self.lib.nnr_tape_fn_id(name.encode())
            if i < 0:
                raise TapeError('%s was called while a tape was recording but is not a recordable entry point (csrc/tape.hip REGISTRY)' % name)
            self._fn_ids[name] = (i, self.lib.nnr_tape_fn_nargs(i))
            i = self._fn_ids[name]
        return i

    def _input_of(self, ptr):
        for k, (lo, hi, _, _) in enumerate(self.inputs):
            if lo <= ptr < hi:
                return k, ptr - lo
        return None

    def _vouched(self, ptr):
        """Is `ptr` inside a buffer that stays valid for the tape's lifetime (kept by the tape, or `known`)?"""
        if self._ranged == 0:
            for t in self.known:
                st = t.untyped_storage()
                self._ranges.append((st.data_ptr(), st.data_ptr() + st.nbytes()))
        keep = self.keep
        while self._ranged < len(keep):
            t = keep[self._ranged]
            self._ranged += 1
            if torch.is_tensor(t):
                st = t.untyped_storage()
                self._ranges.append((st.data_ptr(), st.data_ptr() + st.nbytes()))
        self._ranged = max(self._ranged, 1)
        for lo, hi in reversed(self._ranges):          # the most recent allocations are the likeliest hits
            if lo <= ptr < hi:
                return True
        return False

    def _check_ptr(self, name, where, ptr):
        if ptr and self._input_of(ptr) is None and not self._vouched(ptr):
            self.violations.append((name, where, ptr))

    def _seed_kind(self, v):
        for name in ('news_seed', 'user_seed'):
            d = (v - self.seeds[name]) & 0xFFFFFFFF
            if d < 64:
                return VALUE_KINDS[name], d
            if name == 'news_seed' and 0 <= d - NEWS_SEED_STRIDE < 64:
                return VALUE_KINDS[name], d      # the SECOND news-encoder call of the step (MHSA step: candidates, then history)
        raise TapeError('a uint32 argument (%d) that is not derived from this step\'s dropout seeds reached a recorded call' % v)

    def _add_call(self, name, args):
        fid, nargs = self._fn(name)
        if len(args) != nargs + 1:
            raise TapeError('%s: %d arguments recorded, the entry point takes %d + stream' % (name, len(args), nargs))
        stream = args[-1]
        slots = (C.c_uint64 * max(1, nargs))()
        blobs = []                           # (slot, ctypes object, nbytes)
        patches = []                         # ('slot' | blob index, byte offset inside, kind, width, addend)
        for i, a in enumerate(args[:-1]):
            if a is None:
                v = 0
            elif isinstance(a, int):
                v = a & MASK64
                kind = _VALUE_ARGS.get((name, i))
                if kind is not None:
                    patches.append(('slot', 8 * i, VALUE_KINDS[kind], 4, 0))
            elif isinstance(a, C.c_void_p):
                v = a.value or 0
                hit = self._input_of(v) if v else None
                if hit is not None:
                    patches.append(('slot', 8 * i, 1000 + hit[0], 8, hit[1]))
                elif v and not (i == 0 and name in _HANDLE_ARGS):
                    self._check_ptr(name, i, v)
            elif isinstance(a, C.c_float):
                v = struct.unpack('<I', struct.pack('<f', a.value))[0]
            elif isinstance(a, C.c_uint32):
                v = a.value
                if v:
                    k, d = self._seed_kind(v)
                    patches.append(('slot', 8 * i, k, 4, d))
            elif isinstance(a, _INT_TYPES):
                v = a.value & MASK64
            elif isinstance(a, (C.Structure, C.Array)) or hasattr(a, '_obj'):
                obj = a._obj if hasattr(a, '_obj') else a
                v = 0
                bi = len(blobs)
                blobs.append((i, obj, C.sizeof(obj)))
                self._blob_patches(obj, 0, bi, patches, name)
            else:
                raise TapeError('%s: argument %d of type %s cannot be recorded' % (name, i, type(a).__name__))
            slots[i] = v
        tag = -1
        if self._pending_tag is None and TAG_ALL[0]:
            self._pending_tag = (name[4:], _no_flops)          # diagnostics (tools/tape_timeline.py): every call carries timing events
        if self._pending_tag is not None:
            tag = len(self.tags)
            self.tags.append(self._pending_tag)
            self._pending_tag = None
        nb = len(blobs)
        bslot = (C.c_int * max(1, nb))(*[b[0] for b in blobs])
        bptr = (C.c_void_p * max(1, nb))(*[C.addressof(b[1]) for b in blobs])
        bbytes = (C.c_size_t * max(1, nb))(*[b[2] for b in blobs])
        slot_off = C.c_size_t()
        blob_off = (C.c_size_t * max(1, nb))()
        rc = self.lib.nnr_tape_call(self.h, fid, stream, slots, nargs, bslot, bptr, bbytes, nb, tag, C.byref(slot_off), blob_off)
        if rc < 0:
            raise TapeError('nnr_tape_call(%s) failed with %d' % (name, rc))
        for where, off, kind, width, addend in patches:
            base = slot_off.value if where == 'slot' else blob_off[where]
            L.check(self.lib.nnr_tape_patch(self.h, C.c_size_t(base + off), kind, width, C.c_int64(addend)), 'nnr_tape_patch')
        self.calls += 1

    def _blob_patches(self, obj, base, bi, patches, name=''):
        if isinstance(obj, C.Array):
            step = C.sizeof(obj._type_)
            if issubclass(obj._type_, C.Structure):
                for j in range(len(obj)):
                    self._blob_patches(obj[j], base + j * step, bi, patches, name)
            return
        for fname, ftype in obj._fields_:
            off = base + getattr(type(obj), fname).offset
            if ftype is C.c_void_p:
                v = getattr(obj, fname) or 0
                hit = self._input_of(v) if v else None
                if hit is not None:
                    patches.append((bi, off, 1000 + hit[0], 8, hit[1]))
                elif v:
                    self._check_ptr(name, fname, v)
            elif ftype is C.c_uint32 and fname.endswith('seed'):
                v = getattr(obj, fname)
                if v and (not hasattr(obj, 'drop_target') or obj.drop_target):
                    k, d = self._seed_kind(v)
                    patches.append((bi, off, k, 4, d))

    def _host_call(self, fn):
        seg = self.lib.nnr_tape_segment(self.h)
        if seg < 0:
            raise TapeError('nnr_tape_segment failed')
        self.host_calls.append((fn, torch.cuda.current_stream(), torch.cuda.current_device()))       # (the Stream OBJECT: see replay)
        self._in_host_call = True
        try:
            return fn()
        finally:
            self._in_host_call = False

    # ------------------------------------------------------------------------------------------------ replay
    def matches(self, inputs):
        return len(inputs) == len(self.inputs) and all(tuple(t.shape) == s and t.dtype == d and t.is_contiguous() for t, (_, _, s, d) in zip(inputs, self.inputs))

    def replay(self, values, inputs, timing=False):
        """values: {'news_seed', 'user_seed', 'adam_step'} of THIS step; inputs: the batch tensors.  timing: HIP events around every
        tagged (GEMM / recurrence) call of this replay (see timings())."""
        vals = (C.c_uint64 * 3)(*[int(values[k]) & MASK64 for k in ('news_seed', 'user_seed', 'adam_step')])
        ptrs = (C.c_uint64 * len(inputs))(*[t.data_ptr() for t in inputs])
        tset = -1
        if timing:
            assert self._nsets < 64, 'at most 64 timing replays per tape'
            tset = self._nsets
            self._nsets += 1
        for seg in range(len(self.host_calls) + 1):
            rc = self.lib.nnr_tape_replay(self.h, seg, vals, 3, ptrs, len(inputs), tset)
            L.CALLS[0] += 1
            if rc != 0:
                name = C.create_string_buffer(64)
                call = C.c_int()
                self.lib.nnr_tape_last_error(self.h, None, C.byref(call), name, 64)
                raise L.NnrHipError('tape replay: call %d (%s) failed with code %d' % (call.value, name.value.decode(), rc))
            if seg < len(self.host_calls):
                fn, stream, dev = self.host_calls[seg]
                with torch.cuda.stream(stream):          # the very stream object that was current when the callback was recorded
                    fn()

    def prepare_timing(self, nsets):
        """Create the events of the next `nsets` timing replays now (bench.py: before the timed window starts)."""
        L.check(self.lib.nnr_tape_prepare_timing(self.h, min(64, self._nsets + int(nsets))), 'nnr_tape_prepare_timing')

    def timings(self):
        """Per timing replay so far: [(family, flops_fn, ms)] of its tagged calls (synchronises)."""
        out = []
        n = len(self.tags)
        buf = (C.c_float * max(1, n))()
        for s in range(min(self._nsets, 64)):
            m = self.lib.nnr_tape_timings(self.h, s, buf, n)
            out.append([(self.tags[i][0], self.tags[i][1], float(buf[i])) for i in range(max(0, m)) if buf[i] >= 0])
        return out

    def timeline(self, set_index=0):
        """[(start_ms, dur_ms, stream index, family, shape tag)] of one timing replay, starts relative to the first tagged call."""
        n = len(self.tags)
        a, d, st = (C.c_float * max(1, n))(), (C.c_float * max(1, n))(), (C.c_int * max(1, n))()
        m = self.lib.nnr_tape_timeline(self.h, set_index, a, d, st, n)
        return [(float(a[i]), float(d[i]), int(st[i]), self.tags[i][0], getattr(self.tags[i][1], 'tag', '')) for i in range(max(0, m))]

    def info(self):
        calls, ops_, segs, streams = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        nbytes = C.c_size_t()
        self.lib.nnr_tape_info(self.h, C.byref(calls), C.byref(ops_), C.byref(segs), C.byref(streams), C.byref(nbytes))
        held = sum(t.numel() * t.element_size() for t in self.keep if torch.is_tensor(t))
        return {'calls': calls.value, 'ops': ops_.value, 'segments': segs.value, 'streams': streams.value, 'argument_bytes': nbytes.value,
                'buffers_held_gb': round(held / 2 ** 30, 3), 'event_waits_on_earlier_steps': self.skipped_event_waits, 'timed_calls': len(self.tags)}

    def close(self):
        if self.h:
            self.lib.nnr_tape_destroy(self.h)
            self.h = C.c_void_p()
        self.keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
