"""Thin Python wrappers over the C-ABI (one function per entry point of include/nnr_hip.h).
Tensors are only used as (device pointer, size) carriers; views are fine as long as the leading dimension is passed."""
import ctypes as C
import math
import os
import weakref

import torch

from . import _lib as L
from . import profile as _prof

ACT_NONE, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
K_CHUNK = 1024      # reduction rows per slice of the token-reduction (weight-gradient) GEMMs, see nnr_gemm_args.k_chunk


class _NoSpan:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NOSPAN = _NoSpan()


def _hbm_span(family, per_row, rows, dyn=None, fixed=0.0, tag=''):
    """Live-profile span of an HBM-bound launch (bench.py `roofline.hbm`): ALGORITHMIC bytes = fixed + per_row x live rows, every
    operand read once and every result written once; `dyn` = device int32 holding the live row count (read after the timed region)."""
    if not _prof.active():
        return _NOSPAN

    def flops(vals=None):
        return 0.0

    def nbytes(vals=None, rows=rows, dyn=dyn):
        r = rows
        if dyn is not None:
            r = min(rows, int(vals[dyn.data_ptr()]) if vals is not None else int(dyn.reshape(-1)[0].item()))
        return float(fixed) + float(per_row) * r
    flops.dyn = [dyn] if dyn is not None else []
    flops.bytes_fn = nbytes
    flops.hbm = True
    flops.tag = tag or 'rows%d%s' % (rows, ' dyn' if dyn is not None else '')
    return _prof.span(family, flops)


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.NnrHipError('nnr_amd ops need device tensors (no CPU fallback on the product path)')
    return C.c_void_p(t.data_ptr())


_DEV_INDEX = []


def _s():
    """Raw handle of torch's current HIP stream on this process's device.  (`torch.cuda.current_stream().cuda_stream` costs
    ~9 us of host time per call -- device-index and availability checks --, 170 calls per step; the raw getter ~0.3 us.)"""
    if not _DEV_INDEX:
        _DEV_INDEX.append(torch.cuda.current_device())      # one process per GPU: fixed after set_device
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(_DEV_INDEX[0]))


# ---------------------------------------------------------------------------------------------- leaf work on its own stream
# Weight-gradient GEMMs are LEAVES of the backward pass: nothing downstream reads them before the optimizer.  Issued inline
# they sit on the critical chain of the (latency-bound) data-gradient kernels; here they go to a separate HIP stream that
# waits for the producer of their inputs, and the caller joins it before its backward returns (every accumulation into a
# parameter gradient is atomic, so concurrent leaves are safe).  The scope holds a reference to every input until the join,
# so a buffer the caller drops early cannot be handed out again under a pending read.  (`Tensor.record_stream` would do the
# same through the caching allocator, but a recorded multi-GB buffer that is freed while the leaf stream is still busy is
# not reusable until its event completes: the allocator then grows with hipMalloc and the step time turned bimodal,
# 14 ms or 50-67 ms per step -- measured when the 2.7 GB gate buffer was recorded.)
_LEAF = {}
_NO_DEFER = os.environ.get('NNR_LEAF_DEFER', '1') == '0'
SIDE_CALL = os.environ.get('NNR_SIDE_CALL', '1') != '0'      # model.Model.forward: candidate encoder call on a side stream
EXTRA_STREAMS = []          # every HIP stream this package created (side, title, leaf): see join_extra_streams()


ONE_STREAM = [os.environ.get('NNR_ONE_STREAM') == '1']      # diagnostic: every launch on the caller's stream (solo kernel durations)
STREAM_CACHES = []                                           # dicts of streams handed out by new_stream (dropped when the mode flips)


def set_one_stream(flag):
    """Serialise (True) / restore (False) the package's HIP streams: in serialised mode every launch goes to the caller's stream, so
    the HIP-event spans of nnr_amd.profile are SOLO kernel durations (bench.py's `roofline.isolated`)."""
    torch.cuda.synchronize()
    ONE_STREAM[0] = bool(flag)
    for c in STREAM_CACHES:
        c.clear()
    del EXTRA_STREAMS[:]


def new_stream(dev, critical=False):
    """critical: a stream that carries a piece of the dependent chain (candidate call, title chain) rather than leaf work.
    (Measured and rejected: giving the critical streams a high HIP stream priority -- 13.43 vs 13.05 ms/step.)"""
    if ONE_STREAM[0]:
        return torch.cuda.current_stream(dev)
    # NNR_PRIO=1 (A/B, round 3 -- the whole step is now enqueued at once by the native replay, so the hardware queues arbitrate): the
    # chain-carrying side streams get HIP's high priority.  Measured: no difference (11.20 / 10.80 vs 11.33 / 10.75 ms, 20 steps /
    # sustained).  Running the MAIN chain on a high-priority stream as well did not finish (the pair recurrence's partner workgroups
    # of lower-priority launches wait behind it): not offered.
    # HIP binds a stream to one of its 4 hardware queues (GPU_MAX_HW_QUEUES; 5 and up fall off a cliff: 13 ms per batch-64 step, 6.5 per batch-8 step) and
    # the step runs on 5-6 streams, so some share a queue; WHICH ones do depends on the order the process created and first used its streams, and it
    # decides up to 7 % of the latency-bound batch-8 step: 3.10-3.20 ms in the natural order of a fresh process, 3.3-4.1 ms with 1-4 idle streams in front
    # of the set or of one of its streams; at batch 64 every pattern tried is within 0.08 ms of the natural one (profiles/r06_ab.txt calls 42-46).
    # NNR_STREAM_BURN=a,b,...: the A/B knob -- a, b, ... idle streams (used once) in front of the 1st, 2nd, ... stream of the set.
    k = len(EXTRA_STREAMS)
    if _STREAM_BURN:
        from . import tape as _tape
        assert _tape.ACTIVE[0] is None, 'NNR_STREAM_BURN: a stream was created while a launch tape records'
        for _ in range(_STREAM_BURN[k] if k < len(_STREAM_BURN) else 0):
            _bind(torch.cuda.Stream(device=dev), dev)
    st = torch.cuda.Stream(device=dev, priority=-1) if (critical and STREAM_PRIO >= 1) else torch.cuda.Stream(device=dev)
    if _STREAM_BURN:
        _bind(st, dev)                                # first use = creation
    EXTRA_STREAMS.append(st)
    return st


_BURNT = []


def _bind(st, dev):
    with torch.cuda.stream(st):
        torch.empty(64, device=dev).zero_()           # (a torch kernel, not a C-ABI call)
    _BURNT.append(st)


STREAM_PRIO = int(os.environ.get('NNR_PRIO', '0'))
_STREAM_BURN = [int(x) for x in os.environ.get('NNR_STREAM_BURN', '').split(',') if x.strip()]


STREAM_CACHES.append(_LEAF)
_LEAF_ALT = {}
STREAM_CACHES.append(_LEAF_ALT)
LEAF2 = os.environ.get('NNR_LEAF2', '1') != '0'      # a second leaf stream for the title token stream's weight-gradient GEMMs (A/B)


def join_extra_streams(dev=None):
    """Make the current stream wait for everything enqueued on the package's own streams.  Every backward function joins
    the streams it used before it returns; the trainer calls this once more before the gradient exchange / optimizer
    (parameter gradients are written out of autograd's sight, so autograd's own stream bookkeeping does not cover them)."""
    cur = torch.cuda.current_stream(dev)
    for st in EXTRA_STREAMS:
        cur.wait_stream(st)
    _DEFER['keep'].clear()               # leaf_deferred: everything it issued is ordered before the caller's next launch now
    _DEFER['queued'] = False             # (also the recovery path if a backward pass died before its end-of-pass callback ran)


def join_leaf_streams(dev=None):
    """Make the current stream wait for the leaf streams only (weight-gradient launches issued so far), leaving the side / title
    streams alone.  The tensors those launches read stay held until the step's final join_extra_streams()."""
    if dev is None:
        dev = torch.device('cuda', torch.cuda.current_device())
    cur = torch.cuda.current_stream(dev)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    for table in (_LEAF, _LEAF_ALT):
        st = table.get(key)
        if st is not None and st is not cur:
            cur.wait_stream(st)


class leaf_scope:
    """with leaf_scope(device) as leaf:  leaf(fn, *input_tensors)  ...   -- joined on exit.
    defer_join: do NOT make the caller's stream wait for the leaf stream on exit; the leaf work (weight gradients nothing downstream
    reads) is joined by the step's final join_extra_streams(), and the input tensors are held until then.  (Round 4: the user
    encoder's backward ended with the main stream waiting ~150 us for its last weight-gradient GEMMs before the news encoder's
    backward could start -- a dependency only the data-parallel early bucket needs.)"""

    def __init__(self, dev, enable=True, defer_join=False):
        self.dev, self.enable, self.defer_join = dev, enable, defer_join

    def __enter__(self):
        self.keep = []
        if self.enable:
            key = (self.dev.type, self.dev.index)
            if key not in _LEAF:
                _LEAF[key] = new_stream(self.dev)
            self.leaf = _LEAF[key]
            self.leaf2 = None
            self.main = torch.cuda.current_stream(self.dev)
        return self

    def __call__(self, fn, *tensors, alt=False):
        """alt: the SECOND leaf stream (LEAF2, round 4): leaf work of the title token stream, so that it does not queue behind the
        content stream's on small, latency-bound steps."""
        if not self.enable:
            fn()
            return
        st = self.leaf
        if alt and LEAF2 and not ONE_STREAM[0]:
            if self.leaf2 is None:
                key = (self.dev.type, self.dev.index)
                if key not in _LEAF_ALT:
                    _LEAF_ALT[key] = new_stream(self.dev)
                self.leaf2 = _LEAF_ALT[key]
            st = self.leaf2
        st.wait_stream(torch.cuda.current_stream(self.dev))     # the producer of the inputs (may be a side stream)
        self.keep.extend(tensors)
        with torch.cuda.stream(st):
            fn()

    def sync(self):
        """Make the CURRENT stream wait for the leaf work issued so far (for a consumer in the middle of the scope)."""
        if self.enable:
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_stream(self.leaf)
            if self.leaf2 is not None:
                cur.wait_stream(self.leaf2)

    def __exit__(self, *a):
        if self.enable and self.defer_join and a[0] is None:
            _DEFER['keep'].extend(self.keep)      # held until join_extra_streams(): the leaf stream may still be reading them
        elif self.enable:
            self.main.wait_stream(self.leaf)
            if self.leaf2 is not None:
                self.main.wait_stream(self.leaf2)
        self.keep = []                   # (dropped on the host after the join was ENQUEUED: later main-stream work is ordered behind it)


_DEFER = {'keep': [], 'queued': False, 'calls': 0}     # calls: launches that went to the leaf stream (tests)


LEAF_MIN_ROWS = 49152       # defer only when the step is big enough to be GPU-bound (Model.forward posts the history call's token
STEP_ROWS = [0]             # rows here) or the reduction itself is this long: on small, launch-latency-bound steps the cross-stream
                            # dependencies cost more than the overlap gains (CNN+ATT at batch 16: 6 607 impressions/s inline vs
                            # 5 156 deferred; MHSA+MHSA at batch 64: 10 596 inline vs 11 120 with every weight gradient deferred)


def leaf_deferred(dev, rows, fn, *tensors):
    """Inside an autograd backward function: run `fn` (a weight-gradient launch, atomic accumulation) on the leaf stream
    behind the current stream's work, and join it when THIS backward pass ends (autograd's end-of-pass callback), so the data
    gradient chain on the main stream does not wait for it.  `tensors` (the inputs fn reads) are held until that join.
    NNR_LEAF_DEFER=0 runs fn inline."""
    if _NO_DEFER or _DEFER.get('off') or max(rows, STEP_ROWS[0]) < LEAF_MIN_ROWS:
        fn()
        return
    key = (dev.type, dev.index)
    if key not in _LEAF:
        _LEAF[key] = new_stream(dev)
    leaf, main = _LEAF[key], torch.cuda.current_stream(dev)
    leaf.wait_stream(main)
    _DEFER['keep'].extend(tensors)
    _DEFER['calls'] += 1
    with torch.cuda.stream(leaf):
        fn()
    if _DEFER.get('manual'):
        return                           # the native step (nnr_amd.step) joins the leaf stream itself when its backward sequence ends
    if not _DEFER['queued']:
        _DEFER['queued'] = True
        try:                             # end-of-pass callbacks run on the stream that surrounded the caller's backward()
            torch.autograd.Variable._execution_engine.queue_callback(lambda: join_extra_streams(dev))
        except RuntimeError:             # not inside a backward pass (a backward function called by hand): join right away
            join_extra_streams(dev)


def tape_keep(*tensors):
    """A launch tape that is recording right now (nnr_amd.tape) takes a reference to `tensors`: cached device buffers that live in
    module-level tables (W^T copies and their descriptor table, slot / exchange workspaces, packed-gradient accumulators) are replaced
    when a table is rebuilt -- e.g. after other models of the process were garbage-collected -- and a tape must not be left pointing
    at freed memory.  No-op when nothing records."""
    from . import tape as _tape
    t = _tape.ACTIVE[0]
    if t is not None:
        t.keep.extend(x for x in tensors if x is not None)


_WT = {}
USE_WT = os.environ.get('NNR_WT', '1') != '0'      # A/B switch: data-gradient GEMMs as NT products on cached W^T


class _WtEntry:
    __slots__ = ('ref', 'ptr', 'epoch', 'version', 't', 'event', 'stream')


_WT_TABLE = {'ids': None, 'dev': None, 'count': 0}      # device-resident descriptor table of the registered transposes


def wt(w):
    """W^T (contiguous [cols, rows]) of a 2-D contiguous PARAMETER, cached until the parameter changes (optimizer step:
    layers.PARAM_EPOCH; in-place edits: the tensor's version counter).  With it every data-gradient GEMM dX = dY . W becomes
    an NT product (both operands K-contiguous) and runs on the LDS-DMA staged kernels; the transposes are a few hundred KB per
    step against GBs of activations.  A stale copy is refreshed on the stream of the first user; a user on another HIP stream
    waits for the producer's event.  In training, Model.forward refreshes ALL registered copies in one launch on the leaf
    stream (wt_prefetch), so the backward pass only ever finds fresh ones."""
    from .layers import PARAM_EPOCH
    # identity of the cached object: the tensor OBJECT (weak reference) + its pointer.  A pointer alone is not an identity --
    # the caching allocator hands a freed parameter's address to the next model's parameter of the same shape.
    e = _WT.get(id(w))
    cur = torch._C._cuda_getCurrentRawStream(_DEV_INDEX[0] if _DEV_INDEX else torch.cuda.current_device())
    if e is not None and e.epoch == PARAM_EPOCH[0] and e.version == w._version and e.ptr == w.data_ptr() and e.ref() is w:
        if e.stream != cur:
            torch.cuda.current_stream(w.device).wait_event(e.event)
        tape_keep(e.t)
        return e.t
    rows, cols = w.shape
    if e is None or e.ref() is not w or e.ptr != w.data_ptr() or e.t.shape != (cols, rows):
        if len(_WT) > 256:                                   # entries of dead tensors (temporary views, discarded models)
            for k in [k for k, v in _WT.items() if v.ref() is None]:
                del _WT[k]
        e = _WtEntry()
        e.ref, e.ptr = weakref.ref(w), w.data_ptr()
        e.t = torch.empty((cols, rows), device=w.device, dtype=torch.float32)      # kept across refreshes: stable address
        _WT[id(w)] = e
        _WT_TABLE['ids'] = None
        mark_weight(e.t, prefetchable=True)
    tape_keep(e.t)
    transpose2d(w, e.t, rows, cols)
    e.event = torch.cuda.Event()
    e.event.record()
    e.epoch, e.version, e.stream = PARAM_EPOCH[0], w._version, cur
    return e.t


def wt_prefetch(dev):
    """Refresh, on the leaf stream and in ONE launch (nnr_transpose_batch over a device-resident descriptor table), the
    transposes of every parameter wt() has served before and that changed since (i.e. after an optimizer step).  Called at the
    start of a training forward pass: the copies are needed by the BACKWARD pass only, so they leave the critical chain --
    made lazily, the first user's stream does the copy and users on the other streams wait for it (measured: a 383 us stall
    of the history call's backward behind the candidate call's)."""
    if not USE_WT or not _WT or torch.cuda.is_current_stream_capturing():      # (under hipGraph capture the copies are refreshed lazily by wt():
        return                                                               # ending a capture that holds this fork segfaults in the HIP runtime)
    from .layers import PARAM_EPOCH
    live = [(k, e, e.ref()) for k, e in _WT.items()]
    live = [(k, e, w) for k, e, w in live if w is not None and isinstance(w, torch.nn.Parameter) and e.ptr == w.data_ptr()]
    if not live or all(e.epoch == PARAM_EPOCH[0] and e.version == w._version for _, e, w in live):
        return
    ids = tuple(k for k, _, _ in live)
    if _WT_TABLE['ids'] != ids:
        arr = (L.TransposeDesc * len(live))()
        for d, (_, e, w) in zip(arr, live):
            d.inp, d.out, d.rows, d.cols = w.data_ptr(), e.t.data_ptr(), w.shape[0], w.shape[1]
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        _WT_TABLE.update(ids=ids, dev=host.to(dev), count=len(live))
    key = (dev.type, dev.index)
    if key not in _LEAF:
        _LEAF[key] = new_stream(dev)
    leaf = _LEAF[key]
    leaf.wait_stream(torch.cuda.current_stream(dev))       # behind the optimizer step that changed the parameters
    tape_keep(_WT_TABLE['dev'], *[e.t for _, e, _ in live])
    with torch.cuda.stream(leaf):
        L.check(L.lib().nnr_transpose_batch(_p(_WT_TABLE['dev']), _WT_TABLE['count'], _s()), 'nnr_transpose_batch')
        ev = torch.cuda.Event()
        ev.record()
        cur = torch._C._cuda_getCurrentRawStream(_DEV_INDEX[0] if _DEV_INDEX else torch.cuda.current_device())
    for _, e, w in live:
        e.epoch, e.version, e.event, e.stream = PARAM_EPOCH[0], w._version, ev, cur


# LDS-DMA staged weight-gradient tiles (csrc/gemm.hip: gemm_tn_pipe_kernel): (tile id, rows, cols, workgroups to aim for)
TN_PIPE = os.environ.get('NNR_TN_PIPE', '1') != '0'
_TN_SQUARE = os.environ.get('NNR_TN_SQUARE', '1') == '1'      # LDS-DMA tile for the 900 x 900 weight gradients of the user encoder (same step time, 8 instead of 12 atomic slices)
_TN_WIDE = os.environ.get('NNR_TN_WIDE', '1') == '1'      # 128 x 160 tile for the 1664 x 300 weight gradient: counter traffic 2.0x -> 1.5x of its operands, step +0.05 ms
_TN_SMALL_LDS = os.environ.get('NNR_TN_SMALL_LDS', '0') == '1'      # 1: 53 KB 128 x 80 tiles everywhere.  Mid-round they won (12.35 vs 12.52 ms:
                                                                    # they fitted beside a 98 KB recurrence workgroup); since the recurrence holds
                                                                    # 120-130 KB and the split-K slices come in whole waves, the 78 KB 128 x 208
                                                                    # tile for N = 200 / 400 is ahead again (11.38 vs 11.43 ms, 4 rounds each)


_TN_T64 = int(os.environ.get('NNR_TN_T64', '3'))      # A/B (round 4): bit 0 = 64 x 208 tile for the M = 200 / 400 gate / attention weight gradients, bit 1 = for the gathered dW_hh (M = 832)


def tn_tile(M, N, K, gather=False):
    """Tile of a token-reduction (weight-gradient) GEMM C[M,N] += A[K,M]^T B[K,N] and the tile dims its split-K factor is sized
    for.  Measured on the step's shapes (tools/gemm_pipe_bench.py tn, TFLOP/s old -> new): 1664x300 82 -> 96 (128x80),
    832x200 with gathered rows 68 -> 86, 400x400 75 -> 82, 200x400 62 -> 73 (128x208); short reductions and the 900x900 SUE
    layers stay on the register-staged 64x80 tile."""
    if _TN_SQUARE and TN_PIPE and K >= 2048 and M >= 512 and N >= 512 and not gather and not ((M & 3) or (N & 3)):
        return 26, 128, 80, 2048          # SUE's 900 x 900 x 4 352 weight gradients: 8 slices of 544 rows instead of 12 of 363 on the 64 x 80 tile
    if not TN_PIPE or K < 8192 or (M & 3) or (N & 3) or (M >= 512 and N >= 512):
        return 0, 64, 80, 2048
    if _TN_WIDE and not gather and M >= 1024 and 160 < N <= 320:
        return 30, 128, 160, 2048         # dW_ih (1664 x 300): two 160-column blocks instead of four 80-column ones: A (d gates) is fetched twice, not 4x
    if _TN_SMALL_LDS:
        return (20 if gather else 26), 128, 80, 2048
    if N <= 208 or (N > 320 and N <= 416):
        if (_TN_T64 & 2) if gather else ((_TN_T64 & 1) and M % 128 != 0):
            return 32, 64, 208, 640       # gen-2 loop, 64 x 208: M = 200 / 400 / 832 in 4 / 7 / 13 row tiles (256 / 448 / 832 rows of MFMA work, not 256 / 512 / 896)
        return 27, 128, 208, 640          # gen-2 loop, 128 x 208 (one token row per DMA instruction: takes gathered rows)
    if gather:
        return 20, 128, 80, 2048          # gathered rows wider than 208 columns (dW_hh at --hidden_dim 212..256): tile 26 has no gather path
    return 26, 128, 80, 2048             # gen-2 loop, 128 x 80


def split_for(m, n, k, tile_m=64, tile_n=80, target_blocks=2048, kmin=256):
    """split-K factor for the token-reduction (weight-gradient) GEMMs: enough blocks to fill 256 CUs x ~3.  (Cutting short
    reductions finer, kmin 64, measured no better.)"""
    tiles = max(1, ((m + tile_m - 1) // tile_m) * ((n + tile_n - 1) // tile_n))
    if k <= 640:
        return 1      # a few hundred reduction rows (the per-candidate projections: 320 rows at batch 64): one slice -- no slab, no second launch
    return int(max(1, min((k + kmin - 1) // kmin, (target_blocks + tiles - 1) // tiles)))


# ---------------------------------------------------------------------------------------------- bf16x3 NT GEMMs (the default matrix path since round 6)
# The GPU-filling NT launches whose B operand is a weight (a parameter, a cached transpose, the packed LSTM input weights) run on
# csrc/gemm.hip:gemm_nt_bx3_kernel: fp32 arithmetic as six exact bf16 x bf16 products with fp32 accumulation: the same
# fp32 inputs, fp32 outputs, and a third of the fp32-MFMA kernel's error against fp64.  The weight's three bf16 images are cached per (storage,
# shape) and re-split when the parameters changed (layers.PARAM_EPOCH) -- one small launch per weight and step, recorded in the launch tape
# like any other call.  Round 5 measured it (off); round 6 made it the default after five interleaved same-box pairs + a per-shape-class A/B
# (profiles/r06_bx3_phases.md: -0.33 ms of the 10.27 ms headline step, every class positive).  NNR_BX3=0 = the pure fp32-MFMA path
# (bench.py keeps it as a `secondary` leg of the same workload so that both numbers are driver-timed).
BX3 = [os.environ.get('NNR_BX3', '1') == '1']
_BX3_MIN_ROWS = int(os.environ.get('NNR_BX3_MIN_ROWS', '2048'))
_BX3_TILE = int(os.environ.get('NNR_BX3_TILE', '50'))          # A/B: 50 = 128 x 80 (2 workgroups / CU), 51 = 64 x 80 (3), 52 = 128 x 64, 53 = 256 x 80 (1)
# shape classes the bf16x3 kernel takes (round 6: decided per class by same-box in-step A/Bs, profiles/r06_bx3_phases.md):
#   dx   = long reductions (K >= 1024: the embedding-row gradient dX = dGates . W_ih, K = 2 NP = 1664)
#   sue  = K >= 800 (the user encoder's 900 x 900 layers)
#   proj = N >= 1024 (the LSTM input projection x . W_ih^T, N = 1664)
#   gate = everything else (gate / attention projections, K, N = 200 .. 400)
_BX3_CLASSES = set(c for c in os.environ.get('NNR_BX3_CLASSES', 'dx,sue,proj,gate').split(',') if c)


def bx3_class(N, K):
    if K >= 1024:
        return 'dx'
    if K >= 800:
        return 'sue'
    if N >= 1024:
        return 'proj'
    return 'gate'


_B3 = {}
BX3_SEEN = {}                                                  # diagnostics: (M, N, K, 'weight' | 'other') -> launches that met every other condition


def mark_weight(*tensors, prefetchable=False):
    """`tensors` are derived weights (a cached transpose, a packed layout): rewritten only when the parameters change (layers.PARAM_EPOCH).
    prefetchable: the tensor is refreshed on the leaf stream at the START of a step (wt_prefetch's transposes), so bx3_prefetch may re-split
    it there; the packed LSTM weights are re-packed inside the forward pass and are split by their first user instead."""
    for t in tensors:
        t._nnr_weight = True
        if prefetchable:
            t._nnr_prefetch = True


BX3_PREFETCH = os.environ.get('NNR_BX3_PREFETCH', '1') != '0'


def bx3_prefetch(dev):
    """Re-split, on the leaf stream at the start of a training step (right behind wt_prefetch's transposes, same stream), the bf16 images of
    every weight bx3_images() has served before and that changed since (i.e. after an optimizer step): parameters and their cached transposes.
    Made lazily, each split is a 3-30 us launch on the stream of its first user -- five of them sat on the step's dependent chain in front of
    their GEMMs (profiles/r06_ab.txt: split_bf16x3_kernel 21 / 11 / 34 / 28 / 14 us on the main streams)."""
    if not (BX3[0] and BX3_PREFETCH) or not _B3 or torch.cuda.is_current_stream_capturing():
        return
    from .layers import PARAM_EPOCH
    stale = []
    for e in _B3.values():
        B = e[3]()
        if B is None or e[4][0] != B.data_ptr() or not (isinstance(B, torch.nn.Parameter) or getattr(B, '_nnr_prefetch', False)):
            continue
        if not e[7] or (e[1] == PARAM_EPOCH[0] and e[2] == B._version):
            continue                                  # (not served since the last prefetch -- another model's weight -- or still fresh)
        e[7] = False
        stale.append((e, B))
    if not stale:
        return
    key = (dev.type, dev.index)
    if key not in _LEAF:
        _LEAF[key] = new_stream(dev)
    leaf = _LEAF[key]
    leaf.wait_stream(torch.cuda.current_stream(dev))       # behind the optimizer step that changed the parameters
    with torch.cuda.stream(leaf):
        for e, B in stale:
            _, N, K, ldb = e[4]
            ldo = e[0].shape[2]
            tape_keep(e[0], B)                    # (B: a parameter -- inside the trainer's flat buffer -- or a long-lived cached transpose)
            L.check(L.lib().nnr_split_bf16x3(_p(B), N, K, ldb, ldo, _p(e[0]), C.c_long(N * ldo), _s()), 'nnr_split_bf16x3')
        ev = torch.cuda.Event()
        ev.record()
        cur = torch._C._cuda_getCurrentRawStream(_DEV_INDEX[0] if _DEV_INDEX else torch.cuda.current_device())
    for e, B in stale:
        e[1], e[2], e[5], e[6] = PARAM_EPOCH[0], B._version, ev, cur


def bx3_images(B, N, K, ldb):
    """(images [3, N, ldo] bf16-as-int16, image stride in elements, ldo) of the [N, K] weight `B`, re-split when the parameters changed.
    Identity of a cached entry = the tensor OBJECT (weak reference) + its pointer + the parameter epoch + the tensor's version counter: a
    pointer alone is not an identity (the caching allocator hands a freed model's addresses to the next model -- found by running the GPU
    suite with NNR_BX3=1: 10 tests multiplied by the previous test's weights).  The split runs on the stream of the first user; a user
    on another HIP stream waits for the producer's event, as wt() does (the candidate and the history calls of one step share every
    weight and run on two streams: without the wait the second one can read images that are still being written -- found by the
    two-ranks-on-one-GPU test under NNR_BX3=1, 7.7e-2 gradient error)."""
    from .layers import PARAM_EPOCH
    e = _B3.get(id(B))
    ldo = (K + 7) // 8 * 8
    cur = torch._C._cuda_getCurrentRawStream(_DEV_INDEX[0] if _DEV_INDEX else torch.cuda.current_device())
    if e is not None and (e[3]() is not B or e[4] != (B.data_ptr(), N, K, ldb)):
        e = None
    if e is None:
        if len(_B3) > 512:
            for k in [k for k, v in _B3.items() if v[3]() is None]:
                del _B3[k]
        e = _B3[id(B)] = [torch.empty((3, N, ldo), device=B.device, dtype=torch.int16), -1, None, weakref.ref(B), (B.data_ptr(), N, K, ldb), None, None, True]
    tape_keep(e[0])
    e[7] = True                                   # served since the last bx3_prefetch: worth re-splitting ahead of the next step's first user
    if e[1] != PARAM_EPOCH[0] or e[2] != B._version:
        # (write-after-read: last step's readers on every stream joined the main stream before the optimizer step that changed the epoch)
        L.check(L.lib().nnr_split_bf16x3(_p(B), N, K, ldb, ldo, _p(e[0]), C.c_long(N * ldo), _s()), 'nnr_split_bf16x3')
        e[5] = torch.cuda.Event()
        e[5].record()
        e[1], e[2], e[6] = PARAM_EPOCH[0], B._version, cur
    elif e[6] != cur:
        torch.cuda.current_stream(B.device).wait_event(e[5])
    return e[0], N * ldo, ldo


def _bx3_wanted(A, B, M, N, K, lda, ldb, trans_a, trans_b, a_idx, b_idx, c_idx, split_k, k_chunk, rowdot_w, colsum_out, atomic, batch, dyn_dim, drop):
    return (not trans_a and not trans_b and a_idx is None and b_idx is None and split_k <= 1 and k_chunk <= 0 and rowdot_w is None
            and colsum_out is None and batch <= 1 and dyn_dim in (0, 1) and M >= _BX3_MIN_ROWS and K >= 64 and N * K <= (1 << 22)
            and (K | lda | ldb) & 3 == 0 and (A.data_ptr() | B.data_ptr()) & 15 == 0 and (drop is None or drop[0] in (3, 4) or drop[1] <= 0.0))


def gemm(A, B, C_=None, *, M, N, K, lda, ldb, ldc=0, trans_a=False, trans_b=False, dyn=None, dyn_dim=0, a_idx=None, b_idx=None,
         drop=None, alpha=1.0, bias=None, rowvec=None, ldrv=0, rowvec_map=None, act=0, aux_out=None, ldaux=0, mul=None, ldmul=0,
         resid=None, ldres=0, accumulate=False, atomic=False, c_idx=None, split_k=1, rowdot_w=None, rowdot_out=None, batch=1,
         strideA=0, strideB=0, strideC=0, stride_aux=0, stride_res=0, tile=0, colsum_out=None, k_chunk=0, flop_scale=1.0, slab=None,
         pre_add=None, ldpre=0, gate_bwd=False, b3=None):
    # flop_scale: algorithmic / padded work of this launch (the LSTM gate columns are padded 800 -> 832 per direction; the live
    # profile counts the true 8H columns, not the padded 2*NP)
    # ctypes zero-initialises the struct: only the fields a call actually uses are written (a field store costs ~0.2 us of host
    # time and a step issues ~130 GEMMs; writing all ~50 fields was the largest single item of the host-side enqueue time)
    g = L.GemmArgs()
    g.A, g.B = A.data_ptr(), B.data_ptr()
    if C_ is not None:
        g.C = C_.data_ptr()
    g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.alpha = M, N, K, lda, ldb, ldc, alpha
    if trans_a:
        g.trans_a = 1
    if trans_b:
        g.trans_b = 1
    if dyn is not None:
        g.dyn_dev, g.dyn_dim = dyn.data_ptr(), dyn_dim
    if a_idx is not None:
        g.a_idx = a_idx.data_ptr()
    if b_idx is not None:
        g.b_idx = b_idx.data_ptr()
    if drop is not None and drop[1] > 0.0:
        g.drop_target, g.drop_p, g.drop_seed, g.drop_cols = drop[0], float(drop[1]), int(drop[2]) & 0xFFFFFFFF, int(drop[3])
    if bias is not None:
        g.bias = bias.data_ptr()
    if rowvec is not None:
        g.rowvec, g.ldrv, g.rowvec_map = rowvec.data_ptr(), ldrv, _p(rowvec_map)
    if act:
        g.act = act
    if aux_out is not None:
        g.aux_out, g.ldaux = aux_out.data_ptr(), ldaux
    if mul is not None:
        g.mul, g.ldmul = mul.data_ptr(), ldmul
    if resid is not None:
        g.resid, g.ldres = resid.data_ptr(), ldres
    if accumulate:
        g.accumulate = int(accumulate)
    if atomic:
        g.atomic = 1
    if c_idx is not None:
        g.c_idx = c_idx.data_ptr()
    g.split_k, g.batch = int(split_k), batch
    if rowdot_w is not None:
        g.rowdot_w, g.rowdot_out = rowdot_w.data_ptr(), rowdot_out.data_ptr()
    if batch > 1:
        g.strideA, g.strideB, g.strideC, g.stride_aux, g.stride_res = strideA, strideB, strideC, stride_aux, stride_res
    if tile:
        g.tile = tile
    if colsum_out is not None:
        g.colsum_out = colsum_out.data_ptr()
    if k_chunk:
        g.k_chunk = int(k_chunk)
    if pre_add is not None:
        g.pre_add, g.ldpre = pre_add.data_ptr(), ldpre
    if gate_bwd:
        g.gate_bwd = 1
    if b3 is None and BX3[0] and tile in (0, 9, 15, 16) and _bx3_wanted(A, B, M, N, K, lda, ldb, trans_a, trans_b, a_idx, b_idx, c_idx, split_k, k_chunk, rowdot_w,
                                                                    colsum_out, atomic, batch, dyn_dim, drop):
        # (NNR_BX3, default on): this NT launch on the BF16 matrix pipe, weights pre-split.  Only when B IS a weight: a parameter or a
        # marked derived weight -- an activation buffer is rewritten through the C-ABI without any version bump, so its cached images would go stale
        is_w = isinstance(B, torch.nn.Parameter) or getattr(B, '_nnr_weight', False)
        key = (M, N, K, 'weight' if is_w else 'other')
        BX3_SEEN[key] = BX3_SEEN.get(key, 0) + 1
        if is_w and bx3_class(N, K) in _BX3_CLASSES:
            b3 = bx3_images(B, N, K, ldb)
            tile = _BX3_TILE
            g.tile = _BX3_TILE
    if b3 is not None:
        g.B3, g.b3_stride, g.ldb3 = b3[0].data_ptr(), b3[1], b3[2]
    if slab is None and TN_SLAB and trans_a and trans_b and split_k > 1 and not k_chunk and c_idx is None and (N & 3) == 0 and C_ is not None:
        slab = _slab_ws(A.device, int(split_k) * (M * N + M))      # reproducible split-K: partial results to a slab + fixed-order reduction
    if slab is not None:
        g.slab, g.slab_floats = slab.data_ptr(), slab.numel()
    if not (A.is_cuda and B.is_cuda):
        raise L.NnrHipError('nnr_amd ops need device tensors (no CPU fallback on the product path)')
    if not _prof.active():
        L.check(L.lib().nnr_gemm_f32(C.byref(g), _s()), 'nnr_gemm_f32')
        return
    pipe_nt = (not trans_a and not trans_b and b_idx is None and split_k <= 1 and k_chunk <= 0 and rowdot_w is None and colsum_out is None
               and (drop is None or drop[0] in (3, 4) or drop[1] <= 0.0) and (K | lda | ldb) & 3 == 0 and (A.data_ptr() | B.data_ptr()) & 15 == 0)
    wg128 = ((M + 127) // 128) * ((N + 79) // 80) * (split_k if split_k > 1 else max(1, batch)) * (1 if k_chunk <= 0 else max(1, K // k_chunk))
    # mirror of the library's tile choice (csrc/gemm.hip:nnr_gemm_f32), only to NAME the kernel family in the live profile
    wg64 = ((M + 63) // 64) * ((N + 79) // 80) * (split_k if split_k > 1 else max(1, batch))
    if tile:
        t = tile
    elif rowdot_w is not None:
        t = 3
    elif (not trans_a and a_idx is None and b_idx is None and c_idx is None and dyn is None and split_k <= 1 and k_chunk <= 0
          and colsum_out is None and not atomic and wg64 <= 512 and K >= 64 and drop is None):
        t = 7
    elif pipe_nt and a_idx is None and K >= 800 and wg64 > 512:
        t = 9
        if dyn is None and batch <= 1 and os.environ.get('NNR_NT64', '0') == '1':
            nbm = (M + 127) // 128
            t80, t64 = nbm * ((N + 79) // 80), nbm * ((N + 63) // 64)
            if t80 < 1024 and ((t64 + 255) // 256) * 64 * 100 < ((t80 + 255) // 256) * 80 * 92:
                t = 31
    elif pipe_nt and (dyn is not None or wg128 >= 640):
        t = 15
    elif pipe_nt and wg64 > 512 and K >= 128:
        t = 16
    elif wg64 <= 512 and dyn is None and k_chunk <= 0 and K >= 128:
        t = 6
    elif M <= 512 or (wg128 < 640 and dyn is None) or trans_a:
        t = 2
    else:
        t = 5 if not trans_b else 4
    fam = 'gemm_%s_%s' % ('tn' if trans_a else ('nn' if trans_b else 'nt'),
                          {2: '64x80', 3: '128x208', 4: '128x80', 5: '128x80k32', 6: '64x80k64', 7: '16x80skinny', 9: 'pipe2_128x80', 15: 'pipe128x80', 16: 'pipe128x80s2',
                           20: 'pipe128x80', 26: 'pipe2_128x80', 27: 'pipe2_128x208', 30: 'pipe2_128x160', 32: 'pipe2_64x208', 50: 'bx3_128x80', 51: 'bx3_64x80'}.get(t, 'tile%d' % t))

    def flops(vals=None, M=M, N=N, K=K, dyn=dyn, dyn_dim=dyn_dim, batch=batch):
        # vals: {data_ptr of a device-side size: its value at the time of the launch} (replayed launches: the size buffers are
        # overwritten by the next step, so the trainer snapshots them per timed replay); None: read the buffer now
        m, k = M, K
        if dyn is not None:
            d = int(vals[dyn.data_ptr()]) if vals is not None else int(dyn.item())
            m, k = (min(M, d), K) if dyn_dim == 1 else (M, min(K, d))
        return 2.0 * m * N * k * max(1, batch) * flop_scale
    flops.dyn = [dyn] if dyn is not None else []
    flops.scale = flop_scale                         # algorithmic / executed (padded gate columns): bench.py reports both sums

    def op_bytes(vals=None, M=M, N=N, K=K, dyn=dyn, dyn_dim=dyn_dim, batch=batch):
        # algorithmic HBM bytes of the launch: A and B read once, C written once (+ read when accumulated into), every extra
        # epilogue stream (aux_out, mul, resid) once; true extents
        m, k = M, K
        if dyn is not None:
            d = int(vals[dyn.data_ptr()]) if vals is not None else int(dyn.item())
            m, k = (min(M, d), K) if dyn_dim == 1 else (M, min(K, d))
        outs = (1 if C_ is not None else 0) + (1 if accumulate else 0) + (1 if aux_out is not None else 0) + (1 if mul is not None else 0) + (1 if resid is not None else 0) + (1 if pre_add is not None else 0)
        return 4.0 * max(1, batch) * (m * k + N * k + m * N * outs)
    flops.bytes_fn = op_bytes
    if trans_a and trans_b and split_k > 1:
        flops.tn_dims = (M, N, flop_scale)           # token-reduction GEMM: operand bytes of the launch = live reduction rows x (M + N) x 4
    flops.tag = 'M%d N%d K%d%s%s%s' % (M, N, K, ' b%d' % batch if batch > 1 else '', (' sk%d' % split_k if split_k > 1 else '') + (' kc%d' % k_chunk if k_chunk > 0 else ''), ' dyn' if dyn is not None else '')
    with _prof.span(fam, flops):
        L.check(L.lib().nnr_gemm_f32(C.byref(g), _s()), 'nnr_gemm_f32')


def linear_fwd(x, w, bias=None, out=None, act=0, **kw):
    """out[M,N] = act(x[M,K] . w[N,K]^T + bias) for contiguous 2-D x."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=x.device, dtype=torch.float32)
    gemm(x, w, out, M=M, N=N, K=K, lda=x.stride(0), ldb=w.stride(0), ldc=out.stride(0), bias=bias, act=act, **kw)
    return out


def linear_bwd_data(dy, w, out=None, accumulate=False, **kw):
    """dx[M,K] (+)= dy[M,N] . w[N,K]"""
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty((M, K), device=dy.device, dtype=torch.float32)
    if USE_WT and M >= 1024 and w.is_contiguous() and (N & 3) == 0:
        gemm(dy, wt(w), out, M=M, N=K, K=N, lda=dy.stride(0), ldb=N, ldc=out.stride(0), accumulate=accumulate, **kw)      # NT on W^T
    else:
        gemm(dy, w, out, M=M, N=K, K=N, lda=dy.stride(0), ldb=w.stride(0), ldc=out.stride(0), trans_b=True, accumulate=accumulate, **kw)
    return out


def linear_bwd_weight(dy, x, dw, dyn=None, rows=None, db=None, small_lds=False, **kw):
    """dw[N,K] += dy[rows,N]^T . x[rows,K]   (split-K atomics; dw must already hold the running gradient);
    db[N] += column sums of dy, fused into the same launch.  small_lds: the 19 KB register-staged tile (for launches meant to run
    beside the recurrence, whose workgroups hold 120-130 KB of LDS per CU)."""
    R = dy.shape[0] if rows is None else rows
    N, K = dw.shape
    t, bm, bn, target = tn_tile(N, K, R)
    if small_lds:
        t, bm, bn, target = 2, 64, 80, 2048
    if t and ((dy.stride(0) | x.stride(0)) & 3 or (dy.data_ptr() | x.data_ptr()) & 15):
        t, bm, bn, target = 0, 64, 80, 2048
    gemm(dy, x, dw, M=N, N=K, K=R, lda=dy.stride(0), ldb=x.stride(0), ldc=dw.stride(0), trans_a=True, trans_b=True,
         atomic=True, dyn=dyn, dyn_dim=2, colsum_out=db, split_k=split_for(N, K, R, bm, bn, target), tile=t, **kw)


def rowdot(x, w, out, dyn=None, rows=None):
    """out[row] = <x[row], w> for the live rows of a 2-D x (the w2 . tanh(.) attention score)."""
    R = x.shape[0] if rows is None else rows
    L.check(L.lib().nnr_rowdot(_p(x), x.stride(0), _p(w), _p(dyn), R, x.shape[1], _p(out), _s()), 'nnr_rowdot')


_SLOT_WS = {}
_SLAB_WS = {}
TN_SLAB = os.environ.get('NNR_TN_SLAB', '1') != '0'      # split-K weight gradients through slabs + a fixed-order reduction (0: f32 atomics)
# fixed-order stream-K for the one-to-two-wave NT launches of the user encoder (csrc/gemm.hip: gemm_nt_sk_kernel; round 5, verdict item 1b)
def _slab_ws(dev, floats):
    """Split-K slab workspace of the CURRENT stream (nnr_gemm_args.slab): a launch's slices store their partial results there and the
    reduction that follows it on the same stream consumes them, so launches of one stream share one buffer (grown when a bigger
    launch comes along; a recording tape keeps every buffer it has seen alive)."""
    key = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
    ws = _SLAB_WS.get(key)
    if ws is None or ws.numel() < floats:
        ws = torch.empty(max(int(floats), 1 << 22), device=dev, dtype=torch.float32)
        _SLAB_WS[key] = ws
    tape_keep(ws)
    return ws



def _slot_ws(dev, n):
    """Slot workspace of the CURRENT stream (see nnr_slot_workspace_floats): kernels of one stream run in order and each call
    overwrites the slot rows it reads, so one buffer per stream serves every call; grown when a wider vector comes along."""
    key = torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())
    need = L.lib().nnr_slot_workspace_floats(int(n))
    ws = _SLOT_WS.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.zeros(max(need, 32 * 1024), device=dev, dtype=torch.float32)
        _SLOT_WS[key] = ws
    tape_keep(ws)
    return ws


def bias_grad(dy, db, dyn=None, rows=None):
    R = dy.shape[0] if rows is None else rows
    ws = _slot_ws(dy.device, db.numel())         # per-workgroup slot rows + fixed-order reduction (reproducible; round 3: only for R >= 1024)
    L.check(L.lib().nnr_colsum(_p(dy), dy.stride(0), _p(dyn), R, db.numel(), _p(db), _p(ws), _s()), 'nnr_colsum')


# ---------------------------------------------------------------------------------------------- planner / LSTM
class SeqPlan:
    """Device-resident plan of one token stream (see csrc/seq_plan.hip).  With mask1 / ids1 given the stream is the UNION of two
    encoder calls: sequences [0, n0) from (mask, ids), [n0, n0 + n1) from (mask1, ids1)."""

    def __init__(self, mask, ids, perm=None, mask1=None, ids1=None):
        n0, Lx = mask.shape
        n = n0 + (mask1.shape[0] if mask1 is not None else 0)
        dev = mask.device
        self.n, self.L, self.n0 = n, Lx, n0
        i32 = dict(device=dev, dtype=torch.int32)
        self.len = torch.empty(n, **i32)
        self.order = torch.empty(n, **i32)
        self.rank = torch.empty(n, **i32)
        self.slen = torch.empty(n, **i32)
        self.bs = torch.empty(Lx, **i32)
        self.off = torch.empty(Lx + 1, **i32)
        self.row_seq = torch.empty(n * Lx, **i32)
        self.tok = torch.empty(n * Lx, **i32) if ids is not None else None
        self.prev_f = torch.empty(n * Lx, **i32)
        self.prev_r = torch.empty(n * Lx, **i32)
        self.total = self.off[Lx:]                 # device int32 view: number of valid tokens
        self.cap = n * Lx
        m8 = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
        assert m8.is_contiguous() and (ids is None or (ids.is_contiguous() and ids.dtype == torch.int32))
        outs = (_p(self.len), _p(self.order), _p(self.rank), _p(self.slen), _p(self.bs), _p(self.off), _p(self.row_seq), _p(self.tok),
                _p(self.prev_f), _p(self.prev_r), _s())
        if mask1 is None:
            L.check(L.lib().nnr_seq_plan(_p(m8), _p(ids), n, Lx, _p(perm), *outs), 'nnr_seq_plan')
        else:
            m81 = mask1.view(torch.uint8) if mask1.dtype == torch.bool else mask1
            assert m81.is_contiguous() and m81.shape[1] == Lx and (ids is None) == (ids1 is None)
            assert ids1 is None or (ids1.is_contiguous() and ids1.dtype == torch.int32)
            L.check(L.lib().nnr_seq_plan_pair(_p(m8), _p(ids), n0, _p(m81), _p(ids1), n - n0, Lx, _p(perm), *outs), 'nnr_seq_plan_pair')


def cne_pair_map(plan_t, plan_c):
    """(pm_t, pm_c): per-call rank pairing of two union plans (csrc/seq_plan.hip, section D)."""
    n, n0 = plan_t.n, plan_t.n0
    assert plan_c.n == n and plan_c.n0 == n0
    buf = torch.empty(6 * n, device=plan_t.order.device, dtype=torch.int32)
    pm_t, pm_c = buf[:n], buf[n:2 * n]
    L.check(L.lib().nnr_cne_pair_map(_p(plan_t.order), _p(plan_c.order), n0, n, _p(pm_t), _p(pm_c), _p(buf[2 * n:]), _s()), 'nnr_cne_pair_map')
    return pm_t, pm_c


def lstm_dims(H):
    ub, hp, np_ = C.c_int(), C.c_int(), C.c_int()
    L.check(L.lib().nnr_lstm_dims(H, C.byref(ub), C.byref(hp), C.byref(np_)), 'nnr_lstm_dims(H=%d)' % H)
    return ub.value, hp.value, np_.value


class LstmPacked:
    """nn.LSTM parameters re-laid out for the recurrent kernels (csrc/lstm.hip)."""

    def __init__(self, p, H, E):
        ub, hp, np_ = lstm_dims(H)
        dev = p[0].device
        f = dict(device=dev, dtype=torch.float32)
        self.UB, self.HP, self.NP, self.H, self.E = ub, hp, np_, H, E
        self.w_ihp = torch.empty((2 * np_, E), **f)
        self.b_p = torch.empty(2 * np_, **f)
        self.wf = torch.empty(2 * ub * 4 * ub * 256, **f)
        self.wb = torch.empty(2 * ub * (np_ // 16) * 256, **f)
        self.w_ihp_t = torch.empty((E, 2 * np_), **f)             # [E, 2*NP]: K-contiguous B operand of the dX GEMM (NT form)
        mark_weight(self.w_ihp, self.w_ihp_t)
        L.check(L.lib().nnr_lstm_pack_weights(*[_p(t) for t in p], H, E, _p(self.w_ihp), _p(self.b_p), _p(self.wf), _p(self.wb),
                                              _p(self.w_ihp_t), _s()), 'nnr_lstm_pack_weights')


def lstm_unpack_grads(dw_ihp, db_p, dw_hhp, H, E, grads, zero_src=False):
    """grads: 8 tensors in nn.LSTM order (w_ih, w_hh, b_ih, b_hh, then *_reverse); accumulated into with f32 atomics
    (parameter gradients may be accumulated from several HIP streams at once).  zero_src: leave the packed buffers zeroed."""
    L.check(L.lib().nnr_lstm_unpack_grads(_p(dw_ihp), _p(db_p), _p(dw_hhp), H, E, *[_p(t) for t in grads], 1, int(zero_src), _s()),
            'nnr_lstm_unpack_grads')


LSTM_PAIR = os.environ.get('NNR_LSTM_PAIR', '1') != '0'      # 2-CU weights-stationary recurrence (lstm.hip) when H = 200
LAST_LSTM_SYNC = []                                           # exchange workspaces of the last launch (diagnostics)
_TMO = {}                                                     # per device: persistent exchange time-out counter (uint32 as int32)


def _timeout_counter(dev):
    """The device counter every pair-kernel launch of this process adds its exchange time-outs to (registered once with the
    library: nnr_lstm_set_timeout_counter).  One process drives one GPU."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _TMO.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int32, device=dev)
        _TMO[key] = t
        L.check(L.lib().nnr_lstm_set_timeout_counter(C.c_void_p(t.data_ptr())), 'nnr_lstm_set_timeout_counter')
    return t


def dp_busy(buf, workgroups, iters):
    """Diagnostics: `workgroups` resident 512-thread workgroups sweeping `buf` on the CURRENT stream (a stand-in for RCCL's ring kernels)."""
    L.check(L.lib().nnr_dp_busy(_p(buf), C.c_long(buf.numel()), int(workgroups), int(iters), _s()), 'nnr_dp_busy')


def lstm_sync_timeouts(reset=False):
    """Exchange time-outs of the CU-pair recurrence accumulated over EVERY launch since the process started (or since the
    last reset); must be 0.  A time-out poisons the step with NaN (the optimizer then skips it).  Synchronises."""
    n = sum(int(t.item()) for t in _TMO.values())
    if reset:
        for t in _TMO.values():
            t.zero_()
    return n


def lstm_last_launch_timeouts():
    """Time-outs recorded by the LAST pair-kernel launch only, read from the diagnostics block of its workspaces at the offset
    the library reports (nnr_lstm_sync_diag_offset).  Synchronises."""
    return sum(int(t[off // 4].item()) for t, off in LAST_LSTM_SYNC)


_SYNC_WS = {}


def _sync_workspace(dev, n, slot):
    """Exchange workspace of the CU-pair recurrence: zero-filled once, then reused launch after launch (the words carry a launch
    epoch, csrc/lstm.hip).  One per (stream kind, direction of the pass, HIP stream): launches that can be in flight together never share one."""
    key = (dev.index, n, slot, torch._C._cuda_getCurrentRawStream(_DEV_INDEX[0] if _DEV_INDEX else torch.cuda.current_device()))      # (per launch stream)
    ws = _SYNC_WS.get(key)
    if ws is None:
        ws = _SYNC_WS[key] = torch.zeros(L.lib().nnr_lstm_sync_bytes(n) // 4, dtype=torch.int32, device=dev)
    tape_keep(ws)
    return ws


def _lstm_probs(items, H=0, backward=False):
    arr = (L.LstmProblem * len(items))()
    del LAST_LSTM_SYNC[:]
    for a, it in zip(arr, items):
        pl = it['plan']
        if LSTM_PAIR and H == 200:
            _timeout_counter(it['gates'].device)
            it['sync'] = _sync_workspace(it['gates'].device, pl.n, (it.get('name'), len(LAST_LSTM_SYNC), backward))
            LAST_LSTM_SYNC.append((it['sync'], L.lib().nnr_lstm_sync_diag_offset(pl.n)))
        a.sync = _p(it.get('sync'))
        a.bs, a.off, a.slen, a.prev_f, a.prev_r = _p(pl.bs), _p(pl.off), _p(pl.slen), _p(pl.prev_f), _p(pl.prev_r)
        a.n, a.L = pl.n, pl.L
        a.gates, a.cell, a.hout, a.cn = _p(it['gates']), _p(it['cell']), _p(it.get('hout')), _p(it.get('cn'))
        a.wf, a.wb = _p(it['w'].wf), _p(it['w'].wb)
        a.dh, a.dcn = _p(it.get('dh')), _p(it.get('dcn'))
    return arr


def _lstm_flops(items, H):
    totals = [it['plan'].total for it in items]

    def flops(vals=None):
        tok = sum(float(vals[t.data_ptr()]) if vals is not None else float(t.item()) for t in totals)
        return tok * 2 * (2.0 * H * 4 * H)                                          # tokens x 2 directions x [1,H]x[H,4H]
    flops.dyn = totals
    flops.scale = 4.0 * H / max(4 * H, -(-H // 16) * 64)      # the kernels multiply NP = ceil(H / 16) x 64 gate columns per direction (H = 200: 832 for 800)
    return flops


def lstm_fwd(items, H):
    arr = _lstm_probs(items, H)
    with _prof.span('lstm_fwd', _lstm_flops(items, H)):
        L.check(L.lib().nnr_lstm_fwd(arr, len(items), H, _s()), 'nnr_lstm_fwd')


def lstm_bwd(items, H):
    arr = _lstm_probs(items, H, backward=True)
    with _prof.span('lstm_bwd', _lstm_flops(items, H)):
        L.check(L.lib().nnr_lstm_bwd(arr, len(items), H, _s()), 'nnr_lstm_bwd')


# ---------------------------------------------------------------------------------------------- pooling
def _pool_args(x, ldx, D, n, Lx, plan=None, mask=None, mask_div=1, score=None, v=None, ldv=0, scale=1.0, alpha=None, out=None,
               ldo=0, add_in=None, ldadd=0, dout=None, lddo=0, dout2=None, lddo2=0, dx=None, lddx=0, dx_accumulate=False,
               dscore=None, dv=None, lddv=0, th=None, w2=None, alpha_b=None, dout_b=None, lddo_b=0, dscore_b=None, v_b=None, ldv_b=0, scale_b=1.0):
    a = L.PoolArgs()
    a.x, a.ldx, a.D, a.n, a.L = _p(x), ldx, D, n, Lx
    a.packed = int(plan is not None)
    if plan is not None:
        a.off, a.slen, a.order = _p(plan.off), _p(plan.slen), _p(plan.order)
    if mask is not None:
        a.mask = _p(mask.view(torch.uint8) if mask.dtype == torch.bool else mask)
    a.mask_div = mask_div
    a.score, a.v, a.ldv, a.scale, a.alpha = _p(score), _p(v), ldv, float(scale), _p(alpha)
    a.out, a.ldo, a.add_in, a.ldadd = _p(out), ldo, _p(add_in), ldadd
    a.dout, a.lddo, a.dout2, a.lddo2 = _p(dout), lddo, _p(dout2), lddo2
    a.dx, a.lddx, a.dx_accumulate, a.dscore, a.dv, a.lddv = _p(dx), lddx, int(dx_accumulate), _p(dscore), _p(dv), lddv
    if th is not None:            # forward: score = <th[row], w2> inside the pool's pass (instead of a separate nnr_rowdot launch)
        a.th, a.ldth, a.A, a.w2 = _p(th), th.stride(0), th.shape[1], _p(w2)
    if alpha_b is not None:       # backward: a second pool's token gradient folded into this call's one write of dx (csrc/pool.hip)
        a.alpha_b, a.dout_b, a.lddo_b, a.dscore_b, a.v_b, a.ldv_b, a.scale_b = _p(alpha_b), _p(dout_b), lddo_b, _p(dscore_b), _p(v_b), ldv_b, float(scale_b)
    return a


def _pool_span(family, kw, per_token_arrays):
    """x [tokens, D] is the operand that matters: forward reads it once (+ the tanh projection when the score is fused), backward
    reads it and writes dx (+ reads dx when it accumulates); per-sequence vectors are n x D."""
    if not _prof.active():
        return _NOSPAN
    plan, D, n, Lx = kw.get('plan'), kw['D'], kw['n'], kw['Lx']
    th = kw.get('th')
    per_row = 4.0 * (D * per_token_arrays + (th.shape[1] if th is not None else 0) + 2)
    return _hbm_span(family, per_row, n * Lx, dyn=plan.total if plan is not None else None, fixed=4.0 * n * D * 3)


def pool_fwd(**kw):
    with _pool_span('pool_fwd', kw, 1):
        L.check(L.lib().nnr_attn_pool_fwd(C.byref(_pool_args(**kw)), _s()), 'nnr_attn_pool_fwd')


def pool_bwd(**kw):
    with _pool_span('pool_bwd', kw, (3 if kw.get('dx_accumulate') else 2) if kw.get('dx') is not None else 1):
        L.check(L.lib().nnr_attn_pool_bwd(C.byref(_pool_args(**kw)), _s()), 'nnr_attn_pool_bwd')


# ---------------------------------------------------------------------------------------------- elementwise
def add_(y, x, alpha=1.0):
    assert y.is_contiguous() and x.is_contiguous() and y.numel() == x.numel()
    L.check(L.lib().nnr_add(_p(y), _p(x), C.c_long(y.numel()), C.c_float(alpha), _s()), 'nnr_add')
    return y


def add_atomic_(y, x, alpha=1.0):
    assert y.is_contiguous() and x.is_contiguous() and y.numel() == x.numel()
    L.check(L.lib().nnr_add_atomic(_p(y), _p(x), C.c_long(y.numel()), C.c_float(alpha), _s()), 'nnr_add_atomic')
    return y


def expand_rows(x, N):
    """[B, D] -> [B, N, D]: x repeated over N (userEncoders.py:172,190), one launch."""
    B, D = x.shape
    y = torch.empty((B, N, D), device=x.device, dtype=torch.float32)
    L.check(L.lib().nnr_expand_rows_fwd(_p(x), _p(y), B, N, D, _s()), 'nnr_expand_rows_fwd')
    return y


def expand_rows_bwd(dy):
    """[B, N, D] -> [B, D]: the sum over N in ascending order, one launch."""
    B, N, D = dy.shape
    dx = torch.empty((B, D), device=dy.device, dtype=torch.float32)
    L.check(L.lib().nnr_expand_rows_bwd(_p(dy), _p(dx), B, N, D, _s()), 'nnr_expand_rows_bwd')
    return dx


def add2d(y, ldy, x, ldx, rows, cols, alpha=1.0, accumulate=False):
    L.check(L.lib().nnr_add2d(_p(y), ldy, _p(x), ldx, rows, cols, C.c_float(alpha), int(accumulate), _s()), 'nnr_add2d')


def gate_bwd(dHt, H, G, dH, dpre, plan, cols):
    with _hbm_span('gate_bwd', 5 * 4.0 * cols, plan.cap, dyn=plan.total):      # reads dHt, H, G; writes dH, dpre
        return _gate_bwd(dHt, H, G, dH, dpre, plan, cols)


def _gate_bwd(dHt, H, G, dH, dpre, plan, cols):
    L.check(L.lib().nnr_gate_bwd(_p(dHt), _p(H), _p(G), _p(dH), _p(dpre), _p(plan.total), plan.cap, cols, _s()), 'nnr_gate_bwd')


def packed_seq_sum(x, D, plan, out):
    L.check(L.lib().nnr_packed_seq_sum(_p(x), D, _p(plan.off), _p(plan.slen), plan.n, _p(out), _s()), 'nnr_packed_seq_sum')


def tanh_score_bwd(th, ds, w2, dw2, plan, A):
    rows = plan.cap if plan is not None else th.shape[0]
    ws = _slot_ws(th.device, A)                  # (reproducible dw2: see csrc/misc.hip slot_reduce_kernel)
    L.check(L.lib().nnr_tanh_score_bwd(_p(th), _p(ds), _p(w2), _p(dw2), _p(plan.total) if plan is not None else None, rows, A, _p(ws),
                                       _s()), 'nnr_tanh_score_bwd')


def small_embed_fwd(table, idx, out_view, ldo, p, seed):
    n, dim = idx.numel(), table.shape[1]
    L.check(L.lib().nnr_small_embed_fwd(_p(table), _p(idx), n, dim, _p(out_view), ldo, C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()),
            'nnr_small_embed_fwd')


def small_embed_bwd(idx, dim, dout_view, lddo, dtable, p, seed):
    L.check(L.lib().nnr_small_embed_bwd(_p(idx), idx.numel(), dim, _p(dout_view), lddo, _p(dtable), C.c_float(p),
                                        C.c_uint32(seed & 0xFFFFFFFF), _s()), 'nnr_small_embed_bwd')


def dropout(x, p, seed, out=None):
    if out is None:
        out = torch.empty_like(x)
    L.check(L.lib().nnr_dropout(_p(x), _p(out), C.c_long(x.numel()), C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()), 'nnr_dropout')
    return out


def layernorm_fwd(u, gamma, beta, eps, xhat, rstd, r_out, resid, y, p, seed):
    """y = dropout(relu(LayerNorm(u) * gamma + beta) + resid) over the last dimension of a contiguous [rows, D] u."""
    rows, D = u.numel() // u.shape[-1], u.shape[-1]
    L.check(L.lib().nnr_layernorm_fwd(_p(u), _p(gamma), _p(beta), C.c_float(eps), C.c_long(rows), D, _p(xhat), _p(rstd), _p(r_out), _p(resid),
                                      _p(y), C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()), 'nnr_layernorm_fwd')


def layernorm_bwd(dv, xhat, rstd, gamma, du, dgamma, dbeta):
    rows, D = dv.numel() // dv.shape[-1], dv.shape[-1]
    L.check(L.lib().nnr_layernorm_bwd(_p(dv), _p(xhat), _p(rstd), _p(gamma), C.c_long(rows), D, _p(du), _p(dgamma), _p(dbeta), _s()),
            'nnr_layernorm_bwd')


def relu_bwd(dy, y, dx=None):
    if dx is None:
        dx = torch.empty_like(dy)
    L.check(L.lib().nnr_relu_bwd(_p(dy), _p(y), _p(dx), C.c_long(dy.numel()), _s()), 'nnr_relu_bwd')
    return dx


def relu_drop_bwd(dy, r, ds, dx, p, seed):
    L.check(L.lib().nnr_relu_drop_bwd(_p(dy), _p(r), _p(ds), _p(dx), C.c_long(dy.numel()), C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF),
                                      _s()), 'nnr_relu_drop_bwd')


# ---------------------------------------------------------------------------------------------- SUE / loss / optimiser
def gcn_aggregate_fwd(graph, z, bias, resid, r_out, y, B, G, D, relu, p, seed):
    arrays = 3 + (1 if resid is not None else 0)                                  # z, (resid) read; r, y written
    with _hbm_span('gcn_aggregate_fwd', 4.0 * (G * D * arrays + G * G), B):
        L.check(L.lib().nnr_gcn_aggregate_fwd(_p(graph), _p(z), _p(bias), _p(resid), _p(r_out), _p(y), B, G, D, int(relu), C.c_float(p),
                                              C.c_uint32(int(seed) & 0xFFFFFFFF), _s()), 'nnr_gcn_aggregate_fwd')


def gcn_aggregate_bwd(graph, dy, r, ds, dx0, dz, B, G, D, p, seed):
    arrays = 4 + (1 if dx0 is not None else 0)                                    # dy, r read; dS, dz, (dx0) written
    with _hbm_span('gcn_aggregate_bwd', 4.0 * (G * D * arrays + G * G), B):
        L.check(L.lib().nnr_gcn_aggregate_bwd(_p(graph), _p(dy), _p(r), _p(ds), _p(dx0), _p(dz), B, G, D, C.c_float(p),
                                              C.c_uint32(int(seed) & 0xFFFFFFFF), _s()), 'nnr_gcn_aggregate_bwd')


def sue_x0_fwd(hist, proxy, x0, B, Hn, Kc, D, p, seed, cmask_fix=None):
    """cmask_fix: the [B, Kc + 1] cluster mask; its last column is set in place by the same launch (userEncoders.py:73)."""
    if cmask_fix is not None:
        assert cmask_fix.is_contiguous() and cmask_fix.element_size() == 1 and tuple(cmask_fix.shape) == (B, Kc + 1)
    L.check(L.lib().nnr_sue_x0_fwd(_p(hist), _p(proxy), _p(x0), B, Hn, Kc, D, C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _p(cmask_fix), _s()),
            'nnr_sue_x0_fwd')


def sue_x0_bwd(dx0, dhist, dproxy, B, Hn, Kc, D, p, seed, dx0_add=None):
    L.check(L.lib().nnr_sue_x0_bwd(_p(dx0), _p(dx0_add), _p(dhist), _p(dproxy), B, Hn, Kc, D, C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()),
            'nnr_sue_x0_bwd')


def sue_slice_fwd(gcn, x0, gfeat, B, Hn, G, D):
    L.check(L.lib().nnr_sue_slice_fwd(_p(gcn), _p(x0), _p(gfeat), B, Hn, G, D, _s()), 'nnr_sue_slice_fwd')


def sue_slice_bwd(dgfeat, dpad, B, Hn, G, D):
    L.check(L.lib().nnr_sue_slice_bwd(_p(dgfeat), _p(dpad), B, Hn, G, D, _s()), 'nnr_sue_slice_bwd')


def sue_intra_fwd(kf, qc, g, cidx, B, N, Hn, Cn, A, D, alpha, feat):
    # per user: keys [Hn, A] + queries [N, A] + features [Hn, D] + indices read; weights [N, Hn] + cluster features [N, Cn, D] written
    with _hbm_span('sue_intra_fwd', 4.0 * (Hn * A + N * A + Hn * D + N * Hn + N * Cn * D) + 8.0 * Hn, B):
        return _sue_intra_fwd(kf, qc, g, cidx, B, N, Hn, Cn, A, D, alpha, feat)


def _sue_intra_fwd(kf, qc, g, cidx, B, N, Hn, Cn, A, D, alpha, feat):
    L.check(L.lib().nnr_sue_intra_fwd(_p(kf), _p(qc), _p(g), _p(cidx), B, N, Hn, Cn, A, D, _p(alpha), _p(feat), _s()), 'nnr_sue_intra_fwd')


def sue_intra_bwd(kf, qc, g, cidx, alpha, dfeat, B, N, Hn, Cn, A, D, dg, dkf, dqc):
    ws = torch.empty(B * N * Hn, device=kf.device, dtype=torch.float32)
    # per user: d cluster features [N, Cn, D], features [Hn, D], keys, queries, weights read; d features [Hn, D], d keys, d queries written
    with _hbm_span('sue_intra_bwd', 4.0 * (N * Cn * D + 2 * Hn * D + 2 * Hn * A + 2 * N * A + N * Hn) + 8.0 * Hn, B):
        return _sue_intra_bwd(kf, qc, g, cidx, alpha, dfeat, B, N, Hn, Cn, A, D, dg, dkf, dqc, ws)


def _sue_intra_bwd(kf, qc, g, cidx, alpha, dfeat, B, N, Hn, Cn, A, D, dg, dkf, dqc, ws):
    L.check(L.lib().nnr_sue_intra_bwd(_p(kf), _p(qc), _p(g), _p(cidx), _p(alpha), _p(dfeat), B, N, Hn, Cn, A, D, _p(dg), _p(dkf), _p(dqc),
                                      _p(ws), _s()), 'nnr_sue_intra_bwd')


def logits_loss_fwd(user, cand, B, N, D, logits, loss, dlogits):
    L.check(L.lib().nnr_logits_loss_fwd(_p(user), _p(cand), B, N, D, _p(logits), _p(loss), _p(dlogits), _s()), 'nnr_logits_loss_fwd')


def logits_fwd(user, cand, B, N, D, logits):
    L.check(L.lib().nnr_logits_fwd(_p(user), _p(cand), B, N, D, _p(logits), _s()), 'nnr_logits_fwd')


def nls_loss(logits, B, N, loss, dlogits):
    L.check(L.lib().nnr_nls_loss(_p(logits), B, N, _p(loss), _p(dlogits), _s()), 'nnr_nls_loss')


def logits_bwd(dlogits, user, cand, B, N, D, duser, dcand, accumulate=False):
    L.check(L.lib().nnr_logits_bwd(_p(dlogits), _p(user), _p(cand), B, N, D, _p(duser), _p(dcand), int(accumulate), _s()), 'nnr_logits_bwd')


def sumsq(g, out):
    """out[0] = sum g^2 (stored; fixed-order sum)."""
    with _hbm_span('sumsq', 4.0, g.numel()):
        return _sumsq(g, out)


def _sumsq(g, out):
    L.check(L.lib().nnr_sumsq(_p(g), C.c_long(g.numel()), _p(out), _s()), 'nnr_sumsq')


def sumsq_part(g, out, add_in=None, slot=0):
    """out[0] = sum g^2 (+ add_in[0]) over one span of the flat gradient (fixed-order sum; `slot`: scratch set, one per concurrent stream)."""
    with _hbm_span('sumsq', 4.0, g.numel()):
        L.check(L.lib().nnr_sumsq_part(_p(g), C.c_long(g.numel()), _p(out), _p(add_in), int(slot), _s()), 'nnr_sumsq_part')


def fusion_rows_fwd(cat_table, sub_table, cat0, sub0, cat1, sub1, out_view, ldo, p, seed_cat, seed_sub):
    """feature_fusion's category / subCategory rows of one encoder call (cat1 = sub1 = None) or of the union of two calls."""
    n0, n1 = cat0.numel(), (cat1.numel() if cat1 is not None else 0)
    L.check(L.lib().nnr_fusion_rows_fwd(_p(cat_table), _p(sub_table), _p(cat0), _p(sub0), n0, _p(cat1), _p(sub1), n1, cat_table.shape[1],
                                        sub_table.shape[1], _p(out_view), ldo, C.c_float(p), C.c_uint32(seed_cat & 0xFFFFFFFF),
                                        C.c_uint32(seed_sub & 0xFFFFFFFF), _s()), 'nnr_fusion_rows_fwd')


DETERMINISTIC = os.environ.get('NNR_DETERMINISTIC', '1') != '0'      # reproducible forms of the small reductions (category tables, proxy nodes)


def fusion_rows_bwd(cat0, sub0, cat1, sub1, cd, sd, dout_view, lddo, dcat_table, dsub_table, p, seed_cat, seed_sub):
    n0, n1 = cat0.numel(), (cat1.numel() if cat1 is not None else 0)
    if DETERMINISTIC and cd <= 128 and sd <= 128:
        L.check(L.lib().nnr_fusion_rows_bwd_det(_p(cat0), _p(sub0), n0, _p(cat1), _p(sub1), n1, cd, sd, dcat_table.shape[0], dsub_table.shape[0],
                                                _p(dout_view), lddo, _p(dcat_table), _p(dsub_table), C.c_float(p), C.c_uint32(seed_cat & 0xFFFFFFFF),
                                                C.c_uint32(seed_sub & 0xFFFFFFFF), _s()), 'nnr_fusion_rows_bwd_det')
        return
    L.check(L.lib().nnr_fusion_rows_bwd(_p(cat0), _p(sub0), n0, _p(cat1), _p(sub1), n1, cd, sd, _p(dout_view), lddo, _p(dcat_table), _p(dsub_table),
                                        C.c_float(p), C.c_uint32(seed_cat & 0xFFFFFFFF), C.c_uint32(seed_sub & 0xFFFFFFFF), _s()), 'nnr_fusion_rows_bwd')


def click_loss(user, cand, B, N, D, logits, loss, dlogits, duser, dcand, terms_ws):
    """logits, loss, d loss / d logits, d user, d cand in one launch (model.py:126-127, trainer.py:64-66)."""
    L.check(L.lib().nnr_click_loss(_p(user), _p(cand), B, N, D, _p(logits), _p(loss), _p(dlogits), _p(duser), _p(dcand), _p(terms_ws), _s()),
            'nnr_click_loss')


def clip_adam(p, g, m, v, sumsq_buf, grad_scale, clip, lr, beta1, beta2, eps, wd, step):
    with _hbm_span('clip_adam', 7 * 4.0, p.numel()):                              # p, g, m, v read; p, m, v written
        return _clip_adam(p, g, m, v, sumsq_buf, grad_scale, clip, lr, beta1, beta2, eps, wd, step)


def _clip_adam(p, g, m, v, sumsq_buf, grad_scale, clip, lr, beta1, beta2, eps, wd, step):
    L.check(L.lib().nnr_clip_adam(_p(p), _p(g), _p(m), _p(v), C.c_long(p.numel()), _p(sumsq_buf), C.c_float(grad_scale), C.c_float(clip),
                                  C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(wd), int(step), _s()),
            'nnr_clip_adam')


def mhsa_fwd(qkv, mask, n, Lq, heads, dh, out, prob, p=0.0, seed=0):
    """p > 0: the dropout that follows the attention is applied in the output stage (== ops.dropout(out, p, seed))."""
    if mask is not None and mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    with _mhsa_span('mhsa_fwd', n, Lq, heads, dh, 4.0, 4.0):          # S = Q K^T and O = P V: 4 L^2 d FLOP per head; Q, K, V read, O written
        return _mhsa_fwd(qkv, mask, n, Lq, heads, dh, out, prob, p, seed)


def _mhsa_span(family, n, Lq, heads, dh, flop_units, arrays):
    if not _prof.active():
        return _NOSPAN

    def flops(vals=None):
        return flop_units * Lq * Lq * dh * heads * n

    def nbytes(vals=None):
        return arrays * 4.0 * n * Lq * heads * dh
    flops.dyn = []
    flops.bytes_fn = nbytes
    flops.tag = 'n%d L%d h%d d%d' % (n, Lq, heads, dh)
    return _prof.span(family, flops)


def _mhsa_fwd(qkv, mask, n, Lq, heads, dh, out, prob, p, seed):
    L.check(L.lib().nnr_mhsa_fwd(_p(qkv), _p(mask), n, Lq, heads, dh, C.c_float(1.0 / math.sqrt(dh)), _p(out), _p(prob),
                                 C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()), 'nnr_mhsa_fwd')


def mhsa_bwd(qkv, mask, prob, dout, n, Lq, heads, dh, dqkv, p=0.0, seed=0):
    if mask is not None and mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    # P recomputed (2), dP = dO V^T (2), dV = P^T dO (2), dQ = dS K (2), dK = dS^T Q (2): 10 L^2 d FLOP per head; Q, K, V, dO read, dQ, dK, dV written
    with _mhsa_span('mhsa_bwd', n, Lq, heads, dh, 10.0, 7.0):
        return _mhsa_bwd(qkv, mask, prob, dout, n, Lq, heads, dh, dqkv, p, seed)


def _mhsa_bwd(qkv, mask, prob, dout, n, Lq, heads, dh, dqkv, p, seed):
    L.check(L.lib().nnr_mhsa_bwd(_p(qkv), _p(mask), _p(prob), _p(dout), n, Lq, heads, dh, C.c_float(1.0 / math.sqrt(dh)), _p(dqkv),
                                 C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()), 'nnr_mhsa_bwd')


def mask_cover(mask):
    """cover[i][t] = 1 for t <= the last valid position of row i, all ones for a row without a valid position (csrc/seq_plan.hip)."""
    m8 = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
    assert m8.is_contiguous() and m8.dim() == 2
    out = torch.empty_like(m8)
    L.check(L.lib().nnr_mask_cover(_p(m8), m8.shape[0], m8.shape[1], _p(out), _s()), 'nnr_mask_cover')
    return out


def seq_rowmap(plan):
    """[n * L] int32: packed row of position t of sequence i (plan.off[t] + plan.rank[i]) or -1 beyond the sequence's length."""
    out = torch.empty(plan.n * plan.L, device=plan.off.device, dtype=torch.int32)
    L.check(L.lib().nnr_seq_rowmap(_p(plan.off), _p(plan.rank), _p(plan.len), plan.n, plan.L, _p(out), _s()), 'nnr_seq_rowmap')
    return out


def _mhsa_span_packed(family, plan, heads, dh, flop_units, arrays):
    """MFMA work as the dense form (the kernel still multiplies 32 x 32 tiles); HBM bytes over the LIVE rows only."""
    if not _prof.active():
        return _NOSPAN
    n, Lq = plan.n, plan.L

    def flops(vals=None):
        return flop_units * Lq * Lq * dh * heads * n

    def nbytes(vals=None, total=plan.total):
        rows = min(n * Lq, int(vals[total.data_ptr()]) if vals is not None else int(total.reshape(-1)[0].item()))
        return arrays * 4.0 * rows * heads * dh
    flops.dyn = [plan.total]
    flops.bytes_fn = nbytes
    flops.tag = 'n%d L%d h%d d%d packed' % (n, Lq, heads, dh)
    return _prof.span(family, flops)


def mhsa_fwd_packed(qkv, mask, rowmap, plan, heads, dh, out, p=0.0, seed=0):
    if mask is not None and mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    with _mhsa_span_packed('mhsa_fwd', plan, heads, dh, 4.0, 4.0):
        L.check(L.lib().nnr_mhsa_fwd_packed(_p(qkv), _p(mask), _p(rowmap), plan.n, plan.L, heads, dh, C.c_float(1.0 / math.sqrt(dh)), _p(out),
                                            C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()), 'nnr_mhsa_fwd_packed')


def mhsa_bwd_packed(qkv, mask, rowmap, plan, dout, heads, dh, dqkv, p=0.0, seed=0):
    if mask is not None and mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    with _mhsa_span_packed('mhsa_bwd', plan, heads, dh, 10.0, 7.0):
        L.check(L.lib().nnr_mhsa_bwd_packed(_p(qkv), _p(mask), _p(rowmap), _p(dout), plan.n, plan.L, heads, dh, C.c_float(1.0 / math.sqrt(dh)),
                                            _p(dqkv), C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()), 'nnr_mhsa_bwd_packed')


MHSA_PAIR = os.environ.get('NNR_MHSA_PAIR', '1') != '0'      # round 6: two titles of <= 16 positions per 32 x 32 attention problem (csrc/mhsa.hip: mhsa_pairing)


def mhsa_pair_map(plan, mask):
    """(vrowmap [n, 32] int32, vmask [n, 32] uint8) of the paired attention core over `plan` (a packed call with L = 32) and the ORIGINAL key mask."""
    if mask.dtype == torch.bool:
        mask = mask.view(torch.uint8)
    vrow = torch.empty((plan.n, plan.L), device=plan.off.device, dtype=torch.int32)
    vmask = torch.empty((plan.n, plan.L), device=plan.off.device, dtype=torch.uint8)
    L.check(L.lib().nnr_mhsa_pair_map(_p(plan.off), _p(plan.slen), _p(plan.order), _p(mask.contiguous()), plan.n, plan.L, _p(vrow), _p(vmask), _s()),
            'nnr_mhsa_pair_map')
    return vrow, vmask


def mhsa_fwd_paired(qkv, pair, plan, heads, dh, out, p=0.0, seed=0):
    vrow, vmask = pair
    with _mhsa_span_packed('mhsa_fwd', plan, heads, dh, 4.0, 4.0):
        L.check(L.lib().nnr_mhsa_fwd_paired(_p(qkv), _p(vmask), _p(vrow), _p(plan.off), plan.n, heads, dh, C.c_float(1.0 / math.sqrt(dh)), _p(out),
                                            C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()), 'nnr_mhsa_fwd_paired')


def mhsa_bwd_paired(qkv, pair, plan, dout, heads, dh, dqkv, p=0.0, seed=0):
    vrow, vmask = pair
    with _mhsa_span_packed('mhsa_bwd', plan, heads, dh, 10.0, 7.0):
        L.check(L.lib().nnr_mhsa_bwd_paired(_p(qkv), _p(vmask), _p(vrow), _p(plan.off), _p(dout), plan.n, heads, dh, C.c_float(1.0 / math.sqrt(dh)),
                                            _p(dqkv), C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()), 'nnr_mhsa_bwd_paired')


def mhsa_prob_size(n, Lq, heads):
    nb = 2 if Lq > 32 else 1
    return n * heads * nb * nb * 1024


def embed_gather(table, idx, p, seed, out=None, dyn=None):
    """out[row] = dropout(table[idx[row]]); `dyn`: device int32 holding the live row count (rows beyond it are left untouched)."""
    n, dim = idx.numel(), table.shape[1]
    if out is None:
        out = torch.empty((n, dim), device=table.device, dtype=torch.float32)
    with _hbm_span('embed_gather', 2 * 4.0 * dim + 4.0, n, dyn=dyn):              # a table row read, a row written, an id read
        return _embed_gather(table, idx, p, seed, out, dyn, n, dim)


def _embed_gather(table, idx, p, seed, out, dyn, n, dim):
    L.check(L.lib().nnr_embed_gather(_p(table), _p(idx), C.c_long(n), _p(dyn), dim, _p(out), C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()),
            'nnr_embed_gather')
    return out


def embed_scatter(dout, idx, dtable, p, seed, dyn=None):
    n, dim = idx.numel(), dtable.shape[1]
    with _hbm_span('embed_scatter', 2 * 4.0 * dim + 4.0, n, dyn=dyn):             # a gradient row read, a table-gradient row added to, an id read
        return _embed_scatter(dout, idx, dtable, p, seed, dyn, n, dim)


def _embed_scatter(dout, idx, dtable, p, seed, dyn, n, dim):
    if dyn is not None:
        L.check(L.lib().nnr_embed_scatter_dyn(_p(dout), _p(idx), C.c_long(n), _p(dyn), dim, _p(dtable), C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF),
                                              _s()), 'nnr_embed_scatter_dyn')
        return
    L.check(L.lib().nnr_embed_scatter(_p(dout), _p(idx), C.c_long(n), dim, _p(dtable), C.c_float(p), C.c_uint32(seed & 0xFFFFFFFF), _s()),
            'nnr_embed_scatter')


SCATTER_SORTED = os.environ.get('NNR_SCATTER_SORTED', '1') != '0'      # embedding-row gradient as a sorted segmented reduction (0: f32-atomic scatter)


class TokenSort:
    """The live rows of a packed token stream sorted by word id (csrc/sort.hip), issued on the LEAF stream behind the planner: it
    needs nothing but the planned ids, so it runs under the forward pass; the backward's nnr_embed_scatter_sorted waits for `event`."""

    def __init__(self, tok, total, vocab):
        dev, cap = tok.device, tok.numel()
        self.cap, self.vocab, self.total = cap, int(vocab), total
        i32 = dict(device=dev, dtype=torch.int32)
        buf = torch.empty(4 * cap, **i32)                        # keys_tmp | rows_tmp | keys_sorted | rows_sorted
        self.keys, self.rows = buf[2 * cap:3 * cap], buf[3 * cap:]
        nb = int(L.lib().nnr_token_sort_workspace_bytes(C.c_long(cap), self.vocab))
        temp = torch.empty(max(nb, 256), device=dev, dtype=torch.uint8)
        self.partial = torch.empty(int(L.lib().nnr_embed_scatter_sorted_workspace_floats(C.c_long(cap))), device=dev, dtype=torch.float32)
        key = (dev.type, dev.index)
        if key not in _LEAF:
            _LEAF[key] = new_stream(dev)
        leaf = _LEAF[key]
        leaf.wait_stream(torch.cuda.current_stream(dev))          # behind the planner that wrote `tok` / `total`
        with torch.cuda.stream(leaf):
            L.check(L.lib().nnr_token_sort(_p(tok), C.c_long(cap), _p(total), self.vocab, _p(buf[:cap]), _p(buf[cap:2 * cap]), _p(self.keys), _p(self.rows),
                                           _p(temp), C.c_size_t(temp.numel()), _s()), 'nnr_token_sort')
            self.event = torch.cuda.Event()
            self.event.record()
        self._keep = (buf, temp)


def embed_scatter_sorted(dout, ts, dtable, p, seed):
    """dtable[w] += sum of mask * dout[row] over the rows of word w in list order (reproducible; see TokenSort)."""
    dim = dtable.shape[1]
    torch.cuda.current_stream(dout.device).wait_event(ts.event)
    with _hbm_span('embed_scatter', 2 * 4.0 * dim + 8.0, ts.cap, dyn=ts.total, tag='sorted cap%d' % ts.cap):
        L.check(L.lib().nnr_embed_scatter_sorted(_p(dout), _p(ts.keys), _p(ts.rows), C.c_long(ts.cap), ts.vocab, dim, _p(dtable), C.c_float(p),
                                                 C.c_uint32(seed & 0xFFFFFFFF), _p(ts.partial), _s()), 'nnr_embed_scatter_sorted')


def fill_zero(t):
    """t.zero_() as an entry point of the library (hipMemsetAsync on the current stream): part of the launch tape."""
    assert t.is_contiguous()
    L.check(L.lib().nnr_fill_zero(_p(t), C.c_size_t(t.numel() * t.element_size()), _s()), 'nnr_fill_zero')
    return t


def copy_bytes(dst, src):
    """dst <- src (device to device, same byte size, both contiguous)."""
    nb = src.numel() * src.element_size()
    assert dst.is_contiguous() and src.is_contiguous() and dst.numel() * dst.element_size() == nb
    L.check(L.lib().nnr_copy_bytes(_p(dst), _p(src), C.c_size_t(nb), _s()), 'nnr_copy_bytes')
    return dst


def fill_column_u8(mask, col, value):
    """mask[:, col] = value for a contiguous 2-D bool / uint8 tensor (userEncoders.py:73)."""
    assert mask.dim() == 2 and mask.is_contiguous() and mask.element_size() == 1
    rows, cols = mask.shape
    L.check(L.lib().nnr_fill_column_u8(_p(mask), rows, cols, col % cols, int(value), _s()), 'nnr_fill_column_u8')


def transpose2d(x, out, rows, cols, accumulate=False):
    L.check(L.lib().nnr_transpose2d(_p(x), _p(out), C.c_long(rows), cols, int(accumulate), _s()), 'nnr_transpose2d')
