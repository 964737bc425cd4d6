"""Data parallelism of the hot path (reference: trainer.py:209-300, DistributedDataParallel over NCCL).

One process per GPU, `torch.distributed` with backend "nccl" (= RCCL over xGMI on ROCm) or "gloo" (CPU tests).
Impressions are independent, so the only exchange is ONE all-reduce(sum) of the flat fp32 gradient buffer per step
(4*P bytes); the 1/world_size average is folded into the fused clip+Adam kernel.  Parameters are broadcast from
rank 0 once (DDP's constructor does the same).  Every rank keeps a full optimizer replica, as the reference does."""
import ctypes as C
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun contract)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29512')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method='env://', world_size=world, rank=rank)
    return rank, local, world


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


class NativeExchange:
    """The same exchange through the C-ABI (`nnr_dp_*`, csrc/dp.hip: RCCL called directly on the launch stream, no
    ProcessGroup in between).  The 128-byte communicator id travels over the already-initialised torch.distributed group
    (any backend) when world > 1.  Default when the job's backend is "nccl" and world > 1 (see _native_wanted); NNR_DP_NATIVE=0 keeps
    `torch.distributed`'s all_reduce ("nccl" = the same RCCL)."""

    def __init__(self, rank, world, uid=None):
        """uid: the 128-byte communicator id every rank already holds (see exchange_unique_id); None = make / fetch it here."""
        from . import _lib as L
        self.L, self.rank, self.world = L, rank, world
        if uid is None:
            uid = exchange_unique_id(rank, world)
            if uid is None:
                raise RuntimeError('nnr_dp_unique_id failed on rank 0')
        self.ctx = C.c_void_p()
        L.check(L.lib().nnr_dp_init(uid, rank, world, C.byref(self.ctx)), 'nnr_dp_init')

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def allreduce(self, flat):
        self.L.check(self.L.lib().nnr_dp_allreduce(self.ctx, C.c_void_p(flat.data_ptr()), C.c_size_t(flat.numel()), self._stream()), 'nnr_dp_allreduce')

    def broadcast(self, flat, root=0):
        self.L.check(self.L.lib().nnr_dp_broadcast(self.ctx, C.c_void_p(flat.data_ptr()), C.c_size_t(flat.numel()), root, self._stream()), 'nnr_dp_broadcast')

    def close(self):
        if self.ctx:
            self.L.check(self.L.lib().nnr_dp_destroy(self.ctx), 'nnr_dp_destroy')
            self.ctx = C.c_void_p()


def exchange_unique_id(rank, world):
    """Rank 0 makes the RCCL unique id, every rank receives it over the torch.distributed group.  Rank 0 ALWAYS reaches the
    broadcast (an empty id on failure), so a rank-0 failure is seen by every rank as `None` instead of leaving the others blocked
    in the broadcast (round-4 advisor)."""
    from . import _lib as L
    raw = b''
    if rank == 0:
        uid = (C.c_ubyte * 128)()
        try:
            if L.lib().nnr_dp_unique_id(uid) == 0:
                raw = bytes(uid)
        except Exception:                 # noqa: BLE001
            raw = b''
    if world > 1:
        box = [raw]
        dist.broadcast_object_list(box, src=0)
        raw = box[0]
    if len(raw) != 128:
        return None
    return (C.c_ubyte * 128).from_buffer_copy(raw)


_native = None
_native_state = {'decided': False, 'why': ''}
_PROBE_TIMEOUT_S = float(os.environ.get('NNR_DP_PROBE_TIMEOUT', '90'))


def _agree(ok):
    """MIN of `ok` over the ranks of the torch.distributed group."""
    if world_size() > 1:
        flag = torch.tensor([int(ok)], device='cuda' if dist.get_backend() == 'nccl' else 'cpu', dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = int(flag.item())
    return int(ok)


def _probe_native(rank, world, uid, timeout_s):
    """ncclCommInitRank + an 8-float all-reduce on a WATCHDOG thread: both block without a time-out of their own, and a rank that
    never arrives (it failed elsewhere, or RCCL stalls beside torch's communicator) would otherwise hang every other rank before
    they reach the agreement.  Returns (NativeExchange | None, ok, why, timed_out).  A thread that is still alive after `timeout_s` is
    parked inside RCCL and cannot be cancelled: the CALLER decides what that means (_probe_timed_out: exit non-zero by default)."""
    import threading
    res = {'nx': None, 'ok': 0, 'why': 'timed out after %.0f s inside ncclCommInitRank / the probe all-reduce' % timeout_s, 'cancelled': False}
    dev_index = torch.cuda.current_device()

    def run():
        try:
            torch.cuda.set_device(dev_index)             # (the current device is per thread)
            nx = NativeExchange(rank, world, uid)
            if res['cancelled']:                         # the caller gave up while we were inside ncclCommInitRank: publish nothing, use nothing
                return
            res['nx'] = nx
            ok, why = 1, ''
            if world > 1:
                st = torch.cuda.Stream(device=dev_index)
                with torch.cuda.stream(st):
                    probe_t = torch.ones(8, device='cuda', dtype=torch.float32)
                    nx.allreduce(probe_t)
                st.synchronize()
                ok = 1 if float(probe_t[0]) == float(world) and float(probe_t[7]) == float(world) else 0
                why = '' if ok else 'probe all-reduce returned %r on rank %d' % (float(probe_t[0]), rank)
            if not res['cancelled']:
                res['ok'], res['why'] = ok, why
        except Exception as e:            # noqa: BLE001  (any failure of the optional binding means: use the fallback)
            res['ok'], res['why'] = 0, '%s: %s' % (type(e).__name__, e)

    th = threading.Thread(target=run, name='nnr-dp-probe', daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        res['cancelled'] = True
        return None, 0, res['why'], True
    return res['nx'], res['ok'], res['why'], False


def _probe_timed_out(why):
    """The watchdog gave up on a thread that is still blocked inside ncclCommInitRank / the probe all-reduce.  That thread cannot be
    cancelled, holds half-initialised RCCL state and may issue stream work whenever it wakes up; carrying on with torch.distributed's own
    RCCL communicator in the same process can deadlock against it (round-5 verdict / advisor).  Default: EXIT NON-ZERO -- every rank that
    timed out leaves with code 75 (EX_TEMPFAIL) and the launcher (torchrun) takes the job down; nothing is re-exec'ed.
    NNR_DP_PROBE_TIMEOUT_ACTION=fallback keeps round 5's behaviour (abandon the thread, use torch.distributed's all_reduce)."""
    action = os.environ.get('NNR_DP_PROBE_TIMEOUT_ACTION', 'exit')
    if action == 'fallback':
        return
    import sys
    sys.stderr.write('nnr_amd.dp: the C-ABI RCCL binding %s; a thread is parked inside RCCL and cannot be cancelled -- exiting with code 75 '
                     '(NNR_DP_PROBE_TIMEOUT_ACTION=fallback continues on torch.distributed instead; NNR_DP_NATIVE=0 skips the binding)\n' % why)
    sys.stderr.flush()
    os._exit(75)               # (not sys.exit: the parked daemon thread and RCCL's own threads must not get a chance to run atexit handlers)


def _native_wanted():
    """The C-ABI binding (RCCL called directly, `nnr_dp_allreduce` part of the recorded launch sequence: ONE tape segment, no host
    callback in the step) is the DEFAULT whenever the job exchanges over RCCL (backend "nccl", world > 1); `torch.distributed`'s
    all_reduce is the fallback (NNR_DP_NATIVE=0, the gloo test backend, or a box whose RCCL cannot be resolved).  NNR_DP_NATIVE=1
    forces it (one-rank ordering tests)."""
    e = os.environ.get('NNR_DP_NATIVE')
    if e is not None:
        return e == '1'
    return dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() > 1


def _native_exchange():
    """The process's NativeExchange, or None (torch.distributed binding).  Decided ONCE, identically on every rank: ncclCommInitRank
    is a blocking collective, so the ranks first agree (over the torch.distributed group) that each of them can resolve RCCL, and
    fall back together otherwise."""
    global _native
    if _native_state['decided']:
        return _native
    if not (_native_wanted() and torch.cuda.is_available()):
        return None                                   # (not latched: the process group may not be up yet)
    _native_state['decided'] = True
    from . import _lib as L
    rank = dist.get_rank() if dist.is_initialized() else 0
    try:
        scratch = (C.c_ubyte * 128)()
        ok = 1 if L.lib().nnr_dp_unique_id(scratch) == 0 else 0      # resolves RCCL (dlsym); the id itself is discarded
    except Exception:                     # noqa: BLE001
        ok = 0
    ok = _agree(ok)
    if not ok:
        _native_state['why'] = 'RCCL could not be resolved by libnnr_hip.so on every rank: torch.distributed binding'
        import warnings
        warnings.warn('nnr_amd.dp: ' + _native_state['why'])
        return None
    # step 1: the communicator id (rank 0 always reaches the broadcast); agreed before anyone enters the blocking init
    uid = exchange_unique_id(rank, world_size())
    if not _agree(uid is not None):
        _native_state['why'] = 'nnr_dp_unique_id failed: torch.distributed binding'
        import warnings
        warnings.warn('nnr_amd.dp: ' + _native_state['why'])
        return None
    # step 2: the communicator is exercised once before anything depends on it -- ncclCommInitRank + an 8-float all-reduce whose result
    # every rank can check, on a watchdog thread with a time-out -- and the ranks agree on the outcome over the torch group: a binding that
    # came up on some ranks only, stalls, or sums wrongly is dropped by ALL of them in favour of torch.distributed's all_reduce
    # (>= 2 RCCL ranks have never run where this build ran)
    nx, ok, why, timed_out = _probe_native(rank, world_size(), uid, _PROBE_TIMEOUT_S)
    if timed_out:
        _probe_timed_out(why)             # (default: does not return)
    ok = _agree(ok)
    if not ok:
        # a communicator that came up HERE while a peer failed or is stuck is leaked, not destroyed: ncclCommDestroy can block on a peer
        # that is still inside the collective (round-5 advisor); the process is about to use torch.distributed's communicator instead
        nx = None
        _native_state['why'] = 'the C-ABI RCCL binding did not come up on every rank (%s): torch.distributed binding' % (why or 'another rank failed')
        import warnings
        warnings.warn('nnr_amd.dp: ' + _native_state['why'])
        return None
    _native = nx
    return _native


def native_active(is_cuda=True):
    return bool(is_cuda) and _native_exchange() is not None


def broadcast_parameters(flat_params):
    if world_size() > 1:
        nx = _native_exchange() if flat_params.is_cuda else None
        if nx is not None:
            nx.broadcast(flat_params, 0)
        else:
            dist.broadcast(flat_params, src=0)


def allreduce_gradients(flat_grads):
    """Sum the flat gradient over ranks; returns the scale (1/world) the optimizer must apply."""
    w = world_size()
    if w > 1:
        nx = _native_exchange() if flat_grads.is_cuda else None
        if nx is not None:
            nx.allreduce(flat_grads)
        else:
            dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
    return 1.0 / w


TOUCHED_MAX_BATCH = int(os.environ.get('NNR_DP_TOUCHED_MAX_BATCH', '16'))


class GradientExchange:
    """The per-step gradient exchange of DistributedDataParallel (trainer.py:212-219,297) on the flat gradient buffer, in
    two buckets so that the first overlaps the rest of the backward pass the way DDP's bucketed all-reduce does:

      * EARLY bucket = the gradients of `early_modules` (the user encoder's own parameters: GCN, cluster attention, ...).
        The user encoder is the FIRST thing the backward pass finishes (loss -> user encoder -> news encoder), so its
        segment of the flat buffer is final while the whole news-encoder backward (several ms: recurrence + token GEMMs) is
        still ahead.  `early_ready()` -- called by the user encoder's backward function once its weight-gradient launches
        are ordered on the current stream -- starts an asynchronous all-reduce of that segment (torch.distributed: RCCL's
        own stream, ordered behind the current stream; C-ABI binding: a dedicated comm stream).
      * LATE bucket = everything else (word-embedding table, Bi-LSTM, attention / gate layers), all-reduced by `finish()`
        after the backward pass has joined its streams; `finish()` also makes the optimizer's stream wait for the early one.

    Both buckets are slices of ONE buffer: no packing copies.  The order of collectives (early, then late) is the same on
    every rank by construction.  With world_size 1 nothing is exchanged unless NNR_DP_FORCE=1 (single-GPU ordering tests on a
    one-rank communicator).  Returns the 1/world scale the fused clip+Adam kernel applies."""

    def __init__(self, flat, early_modules=(), table_param=None):
        self.flat = flat
        self.grad = flat.grad
        self.force = os.environ.get('NNR_DP_FORCE') == '1'
        total = flat.grad.numel()
        lo = hi = None
        ids = {id(p) for m in early_modules for p in m.parameters()}
        spans = sorted((o, o + (p.numel() + 3) // 4 * 4, id(p) in ids) for p, o in zip(flat.params, flat.offsets))
        early = [(a, b) for a, b, e in spans if e]
        if early:
            lo, hi = early[0][0], early[-1][1]
            if any(not e for a, b, e in spans if a >= lo and b <= hi):      # not contiguous in this layout: one bucket
                lo = hi = None
        self.early_span = (lo, hi) if lo is not None else None
        # TABLE bucket = the word-embedding table's gradient (70 % of the floats of CNE+SUE): final when the last embedding-row
        # scatter GEMM of the backward pass is done, while the LSTM weight-gradient GEMMs of the step's tail are still running
        self.table_span = None
        if table_param is not None and os.environ.get('NNR_DP_TABLE_BUCKET', '1') != '0':
            for p, o in zip(flat.params, flat.offsets):
                if p is table_param:
                    self.table_span = (o, o + (p.numel() + 3) // 4 * 4)
        cuts = sorted(x for x in (self.early_span, self.table_span) if x is not None)
        self.late_spans, pos = [], 0
        for a, b in cuts:
            if a > pos:
                self.late_spans.append((pos, a))
            pos = max(pos, b)
        if pos < total:
            self.late_spans.append((pos, total))
        # touched-row exchange of the table bucket (table_rows_exchange): decided BY RULE per step (begin_step) -- on when the per-GPU
        # batch is <= TOUCHED_MAX_BATCH (16) and world > 1: there a step touches a few thousand of the V table rows (per-GPU batch 8 x 8
        # ranks: ~18 of 72 MB) and is short enough (3 ms) for the dense 72 MB all-reduce to show; at larger per-GPU batches the dense
        # bucket hides behind the backward tail and the union grows.  NNR_DP_TOUCHED_ROWS=1 / 0 forces it on / off.
        e = os.environ.get('NNR_DP_TOUCHED_ROWS')
        self.touched_capable = self.table_span is not None and table_param is not None and table_param.dim() == 2
        self.touched_mode = 'auto' if e is None else ('on' if e == '1' else 'off')
        self.touched = self.touched_capable and self.touched_mode == 'on'
        self.table_shape = tuple(table_param.shape) if table_param is not None else None
        self._flags = self._pos = self._count = self._packed = None
        self._noted = []              # events behind the nnr_rows_touch launches of this step (one per token stream)
        self.last_touched = None      # (rows exchanged, rows of the table) of the last step that used the touched-row exchange
        self._pending = None
        self._pending_table = None
        self._table_events = []
        self._comm = None
        self._helper = None
        self.table_final_cb = None    # trainer hook (world 1): called on the helper stream once the table bucket's gradient is final (split norm)
        self._finish_events = []      # (start, end) HIP events around finish() on the caller's stream (bench.py: exposed exchange time)
        self.events = None            # tests: {'early_issued': Event, ...} recorded on the issuing stream when set to a dict

    def active(self):
        return world_size() > 1 or (self.force and dist.is_initialized())

    def describe(self):
        es = self.early_span
        ts = self.table_span
        extra = {'table_bucket_rule': 'touched rows when per-GPU batch <= %d and world > 1 (NNR_DP_TOUCHED_ROWS: %s)' % (TOUCHED_MAX_BATCH, self.touched_mode)}
        if self.touched:
            extra['table_bucket'] = 'touched rows only: flags [V] + packed [U, E] instead of [V, E]'
            if self.last_touched is not None:
                U, V = self.last_touched
                E = self.table_shape[1]
                extra['touched_rows_last_step'] = {'rows': U, 'of': V, 'bytes': 4 * (V + U * E), 'dense_bytes': 4 * V * E}
        elif ts is not None:
            extra['table_bucket'] = 'dense [V, E] all-reduce'
        return {**extra, 'buckets': ([{'name': 'early (user encoder)', 'floats': es[1] - es[0]}] if es else []) +
                           ([{'name': 'table (word embedding)', 'floats': ts[1] - ts[0]}] if ts else []) +
                           [{'name': 'late', 'floats': sum(b - a for a, b in self.late_spans)}],
                'binding': 'C-ABI nnr_dp_allreduce (RCCL, recorded in the launch tape)' if native_active(self.grad.is_cuda) else 'torch.distributed all_reduce',
                'overlap': ('early bucket is reduced while the news-encoder backward runs' if es else 'none (single bucket)') +
                           ('; table bucket while the LSTM weight-gradient GEMMs of the tail run' if ts else '')}

    def _reduce(self, view, async_op):
        nx = _native_exchange() if view.is_cuda else None
        if nx is None:
            return dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=async_op)
        if not async_op:
            nx.allreduce(view)
            return None
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=view.device)
        cur = torch.cuda.current_stream(view.device)
        self._comm.wait_stream(cur)
        with torch.cuda.stream(self._comm):
            nx.allreduce(view)
        return self._comm

    def _native(self):
        return native_active(self.grad.is_cuda)

    def _recordable(self):
        """May this step's exchange be RECORDED into the launch tape (its Python never runs again on a replay)?  Only the dense form over
        the C-ABI binding: nnr_dp_allreduce calls + stream joins are all it does, and the bookkeeping below (`_pending`, `_pending_table`)
        opens and closes inside the one recorded step.  The touched-row form needs the host every step (the union's row count sizes the
        collective), and then ALL THREE hooks -- early_ready, table_scatter_done, finish -- run as host callbacks of the tape, so that the
        bookkeeping they share is re-run together on every replay.  (Round-5 advisor, high: with finish() recorded and
        table_scatter_done a host callback, `_pending_table` was never cleared after the first replay and the table bucket was silently
        left un-reduced from the second replay on.)"""
        return self._native() and not self.touched

    def early_ready(self):
        """The early bucket's gradients are final on the CURRENT stream: start reducing them."""
        if self.early_span is None or not self.active():
            return
        if self._recordable():
            return self._early_ready()           # RCCL through the C-ABI: the call itself is part of a recorded launch sequence
        from .tape import host_call              # torch.distributed: the host's own work, re-run between the segments of a replay
        host_call(self._early_ready)

    def _early_ready(self):
        if self._pending is not None:
            return
        a, b = self.early_span
        if self.events is not None and self.grad.is_cuda:
            self.events['early_issued'] = torch.cuda.Event(enable_timing=True)
            self.events['early_issued'].record()
        self._pending = self._reduce(self.grad[a:b], True)

    # ------------------------------------------------------------------------------------------------ touched rows of the table
    def _touch_buffers(self):
        if self._flags is None:
            V, E = self.table_shape
            dev = self.grad.device
            self._flags = torch.zeros(V, device=dev, dtype=torch.float32)
            self._pos = torch.empty(V, device=dev, dtype=torch.int32)
            self._count = torch.zeros(1, device=dev, dtype=torch.int32)
            self._packed = torch.empty(V * E, device=dev, dtype=torch.float32)
        if self.grad.is_cuda:
            from . import ops
            ops.tape_keep(self._flags, self._pos, self._count, self._packed)      # long-lived buffers a recording tape may point at
        return self._flags

    def begin_step(self, per_gpu_batch=None):
        """Start of an optimizer step (before the forward pass): decide the form of the table bucket for THIS step (rule above; a
        launch tape is per batch shape, so the decision is part of what it records) and clear the touched-row flags."""
        self._table_events = []
        if self.touched_mode == 'auto' and per_gpu_batch is not None:
            self.touched = bool(self.touched_capable and self._rule(int(per_gpu_batch)))
        if not (self.touched and self.active()):
            return
        flags = self._touch_buffers()
        self._noted = []
        if flags.is_cuda:
            from . import ops
            ops.fill_zero(flags)
        else:
            flags.zero_()

    def _rule(self, per_gpu_batch):
        """The table bucket goes out as touched rows when ... (tests replace this to flip the form between batch shapes on one rank)."""
        return per_gpu_batch <= TOUCHED_MAX_BATCH and world_size() > 1

    def note_tokens(self, tok, total=None):
        """The word ids of one token stream of this step (device int32 `tok`, live count `total` on the device or None): mark their
        table rows as touched.  Called by the news encoder right after it planned the stream (forward pass)."""
        if not (self.touched and self.active()):
            return
        flags = self._touch_buffers()
        V = self.table_shape[0]
        if flags.is_cuda:
            from . import _lib as L, ops
            L.check(L.lib().nnr_rows_touch(ops._p(tok), C.c_long(tok.numel()), ops._p(total), V, ops._p(flags), ops._s()), 'nnr_rows_touch')
            ev = torch.cuda.Event()
            ev.record()
            self._noted.append(ev)
        else:
            ids = tok.reshape(-1)[:int(total.reshape(-1)[0])] if total is not None else tok.reshape(-1)
            ids = ids[(ids >= 0) & (ids < V)].long()
            flags[ids] = 1.0
            self._noted.append(None)

    def table_rows_exchange(self):
        """The table bucket as a TOUCHED-ROW exchange (trainer.py:297 all-reduces all V rows): the ranks sum
        their flag vectors (V floats), every rank derives the same packed order of the union, packs those rows of its table gradient,
        all-reduces [U, E] instead of [V, E] and writes the sums back.  Rows outside the union are zero on every rank, so the dense
        gradient -- and the dense clip + Adam after it -- is what the full all-reduce gives.  Needs the host once (U sizes the
        collective): runs on the helper stream behind the last embedding-row scatter, as a host callback of a recorded step."""
        V, E = self.table_shape
        a, _ = self.table_span
        flags, pos, count, packed = self._flags, self._pos, self._count, self._packed
        dense = self.grad[a:a + V * E]
        if flags.is_cuda:
            from . import _lib as L, ops
            cur = torch.cuda.current_stream(flags.device)
            for ev in self._noted:
                cur.wait_event(ev)
            self._reduce(flags, False)
            L.check(L.lib().nnr_rows_compact(ops._p(flags), V, ops._p(pos), ops._p(count), ops._s()), 'nnr_rows_compact')
            U = int(count.item())                                    # (synchronises THIS stream: the helper stream)
            if U > 0:
                L.check(L.lib().nnr_rows_pack(ops._p(dense), ops._p(pos), V, E, ops._p(packed), ops._s()), 'nnr_rows_pack')
                self._reduce(packed[:U * E], False)
                L.check(L.lib().nnr_rows_unpack(ops._p(packed), ops._p(pos), V, E, ops._p(dense), ops._s()), 'nnr_rows_unpack')
        else:
            dist.all_reduce(flags, op=dist.ReduceOp.SUM)
            rows = torch.nonzero(flags > 0).reshape(-1)
            U = int(rows.numel())
            if U > 0:
                d2 = dense.view(V, E)
                buf = d2.index_select(0, rows).contiguous()
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                d2.index_copy_(0, rows, buf)
        self._noted = []
        self.last_touched = (U, V)
        return U

    def table_scatter_done(self, expected):
        """One of the `expected` embedding-row scatter GEMMs of this backward pass is ordered on the CURRENT stream (they run on
        different HIP streams).  When the last one has reported, the table bucket is handed to the exchange on a helper stream that
        waits for all of them -- no stream of the backward pass waits for another one here."""
        if self.table_span is None:
            return
        if not self.active():
            # one GPU: nothing to exchange, but the table's gradient is FINAL here -- the trainer takes its share of the gradient norm on the
            # helper stream, beside the weight-gradient GEMMs of the step's tail (nnr_sumsq_part), instead of on the optimizer's serial tail
            if self.table_final_cb is not None and self.grad.is_cuda:
                ev = torch.cuda.Event()
                ev.record()
                self._table_events.append(ev)
                if len(self._table_events) >= expected:
                    if self._helper is None:
                        self._helper = torch.cuda.Stream(device=self.grad.device)
                    with torch.cuda.stream(self._helper):
                        for e in self._table_events:
                            self._helper.wait_event(e)
                        self.table_final_cb()
                    self._table_events = []
            return
        if not self.grad.is_cuda:
            # host tensors (gloo tests of the exchange logic): no streams to overlap with; only the touched-row form does anything here
            if self.touched and self._noted and self._pending_table is None:
                self._table_events.append(None)
                if len(self._table_events) >= expected:
                    self.table_rows_exchange()
                    self._pending_table = (None, None)
                    self._table_events = []
            return
        if self._recordable():
            return self._table_scatter_done(expected, False)
        from .tape import host_call              # (the touched-row exchange needs the host: the union's row count sizes the collective)
        # the FORM is captured here, when the step is issued / recorded: begin_step's Python does not run on a replay, and an eager step of
        # another batch shape in between (the epoch's last partial batch) may leave self.touched different from what this tape's launches
        # (flag fill, nnr_rows_touch) were recorded for (round-5 advisor, medium)
        touched = bool(self.touched)
        host_call(lambda: self._table_scatter_done(expected, touched))

    def _table_scatter_done(self, expected, touched=None):
        if touched is None:
            touched = self.touched
        if self._pending_table is not None:
            return
        ev = torch.cuda.Event()
        ev.record()
        self._table_events.append(ev)
        if len(self._table_events) < expected:
            return
        if self._helper is None:
            self._helper = torch.cuda.Stream(device=self.grad.device)
        a, b = self.table_span
        with torch.cuda.stream(self._helper):
            for e in self._table_events:
                self._helper.wait_event(e)
            if self.events is not None:
                self.events['table_issued'] = torch.cuda.Event(enable_timing=True)
                self.events['table_issued'].record()
            if touched:
                # (NOT `and self._noted`: on a REPLAYED step note_tokens' Python never runs -- nnr_rows_touch and the flag fill are part
                # of the tape --, and the exchange used to degrade silently to the dense all-reduce there; round-4 advisor.  Only CNE
                # calls this hook, and it notes every token stream it scatters.  The events in _noted are an extra ordering edge of
                # the eager step; the scatter events above already order this stream behind the forward pass that marked the rows.)
                self.table_rows_exchange()
                self._pending_table = (None, self._helper)
            else:
                self._pending_table = (self._reduce(self.grad[a:b], True), self._helper)
        self._table_events = []

    def finish(self):
        """All gradients are final on the current stream: reduce what is left, order the early bucket before the caller's next
        launch, return 1/world."""
        w = world_size()
        if not self.active():
            return 1.0 / w
        if self._recordable():
            return self._finish()
        from .tape import host_call
        return host_call(self._finish)

    def exposed_ms(self):
        """Mean time the caller's stream spent inside finish() over the steps since the last call (HIP events around it on that
        stream: waiting for the overlapped buckets + reducing the late one = the exchange time the backward pass did not hide)."""
        pairs, self._finish_events = self._finish_events, []
        if not pairs:
            return None
        torch.cuda.synchronize()
        return round(sum(a.elapsed_time(b) for a, b in pairs) / len(pairs), 4)

    def _finish(self):
        w = world_size()
        t0 = None
        if self.grad.is_cuda and len(self._finish_events) < 256:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record()
        pend, self._pending = self._pending, None
        pend_t, self._pending_table = self._pending_table, None
        self._table_events = []
        spans = list(self.late_spans)
        if pend is None and self.early_span is not None:       # early_ready was never called this step
            spans.append(self.early_span)
        if pend_t is None and self.table_span is not None:     # no scatter GEMM reported (another encoder, CPU tensors)
            spans.append(self.table_span)
        spans.sort()
        merged = []
        for a, b in spans:                                     # adjacent slices as one collective
            if merged and merged[-1][1] == a:
                merged[-1] = (merged[-1][0], b)
            else:
                merged.append((a, b))
        for a, b in merged:
            self._reduce(self.grad[a:b], False)
        cur = torch.cuda.current_stream(self.grad.device) if self.grad.is_cuda else None
        for p in (pend, pend_t[0] if pend_t is not None else None):
            if p is None:
                continue
            if isinstance(p, torch.cuda.Stream):
                cur.wait_stream(p)
            else:
                p.wait()                                       # NCCL: the current stream waits; gloo: the host waits
        if pend_t is not None and pend_t[1] is not None:
            cur.wait_stream(pend_t[1])
        if self.events is not None and self.grad.is_cuda:
            self.events['finished'] = torch.cuda.Event(enable_timing=True)
            self.events['finished'].record()
        if t0 is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            self._finish_events.append((t0, t1))
        return 1.0 / w


def sampler_indices(n, rank, world, epoch, seed=0):
    """The behaviour indices rank `rank` visits in epoch `epoch`, exactly as torch's DistributedSampler(shuffle=True, seed)
    after set_epoch(epoch) -- the sampler of trainer.py:256-257,264: randperm(n) from Generator(seed + epoch), padded by
    wrapping around to a multiple of `world`, then every world-th index from `rank`.  Feed slices of it to
    DeviceCorpus.train_batch."""
    g = torch.Generator()
    g.manual_seed(seed + epoch)
    idx = torch.randperm(n, generator=g).tolist()
    total = -(-n // world) * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * (-(-pad // len(idx))))[:pad]
    return torch.tensor(idx[rank:total:world], dtype=torch.int32)


def shard_batch(batch, rank, world):
    """Per-rank slice of a global batch: rank r takes samples r::world (DistributedSampler order, trainer.py:256-258)."""
    if world == 1:
        return batch
    return [t[rank::world].contiguous() for t in batch]


def barrier():
    if world_size() > 1:
        dist.barrier()
