// Launch-sequence tape: record the C-ABI calls of ONE training step (arguments are plain pointers / sizes / structs; every
// data-dependent size already lives in device memory) together with the cross-stream dependencies of the step, then replay the
// whole sequence natively -- a plain loop of entry-point calls on the recorded HIP streams with hipEventRecord / hipStreamWaitEvent
// in between -- with ONE C-ABI call per segment.  The reference drives this path from Python, one ATen / cuDNN / torch_scatter call
// at a time (trainer.py:105-120); here the ~140 calls of a step cost the host ~4.4 ms of interpreter time, which bounds small
// per-GPU batches (batch 8 = the 8-GPU shard of `--batch_size 64`, trainer.py:218, is 3.4 ms = the host's launch rate).
// hipGraph replay of the same multi-stream step measured SLOWER than eager on this ROCm, hence a tape of
// ordinary launches: each replayed call runs the real entry point, so host-side launch logic (tile choice, the recurrence's launch
// epoch, workspace handling) stays exactly the code the eager path runs.
//
// What changes from step to step is patched into the recorded arguments before a replay:
//   * VALUE patches  : dropout seeds (per-call counters of the news / user encoder) and Adam's step number -- value[kind] + addend;
//   * INPUT patches  : pointers into the 21 batch tensors -- input[slot] + byte offset (the batch of step i lives elsewhere).
// A tape is split into SEGMENTS at the points where the host has work of its own (torch.distributed's bucketed all-reduce).
#include <cstring>
#include <string>
#include <tuple>
#include <utility>
#include <vector>
#include "common.h"

namespace {

typedef int (*thunk_t)(const uint64_t* slots, hipStream_t stream);

template <typename T> inline T slot_as(const uint64_t* s) { T v; std::memcpy(&v, s, sizeof(T)); return v; }

template <auto Fn> struct Thunk;
template <typename... P, int (*Fn)(P...)> struct Thunk<Fn> {
  static constexpr int nargs = (int)sizeof...(P) - 1;        // the trailing hipStream_t is supplied by the tape
  template <size_t... I> static int apply(const uint64_t* s, hipStream_t st, std::index_sequence<I...>) {
    using Tup = std::tuple<P...>;
    static_assert(std::is_same<std::tuple_element_t<sizeof...(P) - 1, Tup>, hipStream_t>::value, "entry point must end with hipStream_t");
    return Fn(slot_as<std::tuple_element_t<I, Tup>>(s + I)..., st);
  }
  static int call(const uint64_t* s, hipStream_t st) { return apply(s, st, std::make_index_sequence<sizeof...(P) - 1>{}); }
};

struct Entry { const char* name; thunk_t thunk; int nargs; };
#define R(fn) { #fn, &Thunk<&fn>::call, Thunk<&fn>::nargs }
const Entry REGISTRY[] = {
  R(nnr_gemm_f32), R(nnr_seq_plan), R(nnr_seq_plan_pair), R(nnr_cne_pair_map), R(nnr_lstm_pack_weights), R(nnr_lstm_unpack_grads),
  R(nnr_lstm_fwd), R(nnr_lstm_bwd), R(nnr_attn_pool_fwd), R(nnr_attn_pool_bwd), R(nnr_gate_bwd), R(nnr_packed_seq_sum),
  R(nnr_tanh_score_bwd), R(nnr_colsum), R(nnr_rowdot), R(nnr_small_embed_fwd), R(nnr_small_embed_bwd), R(nnr_embed_gather),
  R(nnr_embed_scatter), R(nnr_embed_scatter_dyn), R(nnr_transpose2d), R(nnr_transpose_batch), R(nnr_add), R(nnr_add_atomic), R(nnr_add2d), R(nnr_expand_rows_fwd), R(nnr_expand_rows_bwd),
  R(nnr_dropout), R(nnr_relu_bwd), R(nnr_gcn_aggregate_fwd), R(nnr_gcn_aggregate_bwd), R(nnr_relu_drop_bwd), R(nnr_mhsa_fwd), R(nnr_mhsa_bwd),
  R(nnr_sue_x0_fwd), R(nnr_sue_x0_bwd), R(nnr_sue_slice_fwd), R(nnr_sue_slice_bwd), R(nnr_sue_intra_fwd), R(nnr_sue_intra_bwd),
  R(nnr_corpus_batch), R(nnr_history_graph), R(nnr_logits_loss_fwd), R(nnr_logits_fwd), R(nnr_nls_loss), R(nnr_logits_bwd),
  R(nnr_layernorm_fwd), R(nnr_layernorm_bwd), R(nnr_sumsq), R(nnr_sumsq_part), R(nnr_clip_adam), R(nnr_dp_allreduce), R(nnr_dp_broadcast), R(nnr_dp_busy), R(nnr_split_bf16x3), R(nnr_mhsa_fwd_packed), R(nnr_mhsa_bwd_packed), R(nnr_mhsa_pair_map), R(nnr_mhsa_fwd_paired), R(nnr_mhsa_bwd_paired), R(nnr_mask_cover), R(nnr_seq_rowmap),
  R(nnr_fill_zero), R(nnr_copy_bytes), R(nnr_fill_column_u8), R(nnr_fusion_rows_fwd), R(nnr_fusion_rows_bwd), R(nnr_click_loss), R(nnr_rank_metrics),
  R(nnr_token_sort), R(nnr_embed_scatter_sorted), R(nnr_fusion_rows_bwd_det),
  R(nnr_rows_touch), R(nnr_rows_compact), R(nnr_rows_pack), R(nnr_rows_unpack),
};
#undef R
constexpr int NREG = (int)(sizeof(REGISTRY) / sizeof(REGISTRY[0]));

enum { OP_CALL = 0, OP_RECORD = 1, OP_WAIT = 2 };
struct Op { int kind, a, b; };                       // CALL: a = call index | RECORD: a = event, b = stream | WAIT: a = stream, b = event
struct Call { int fn, stream, nslots, tag; size_t slot_off; };          // slot_off: index into arena (uint64 units)
struct BlobRef { size_t slot; size_t blob_off; };   // arena[slot] = &arena[blob_off]   (fixed up by finalize)
struct Patch { size_t byte_off; int kind; int width; int64_t addend; };
constexpr int MAX_TIMING_SETS = 64;

}  // namespace

struct nnr_tape {
  std::vector<uint64_t> arena;                       // argument slots and deep copies of by-pointer structs, 8-byte units
  std::vector<Call> calls;
  std::vector<Op> ops;
  std::vector<size_t> seg_begin;                     // ops index where each segment starts (+ end sentinel after finalize)
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> events;
  std::vector<uint64_t> event_keys;                  // caller's key of each recorded event (0: internal)
  std::vector<BlobRef> blob_refs;
  std::vector<Patch> patches;
  std::vector<std::vector<hipEvent_t>> tset;         // timing event pairs, one set per timing replay (2 per tagged call)
  int ntagged = 0;
  bool final = false;
  int last_rc = 0, last_failed_call = -1;
};

namespace {
int stream_index(nnr_tape* t, hipStream_t s) {
  for (size_t i = 0; i < t->streams.size(); ++i) if (t->streams[i] == s) return (int)i;
  t->streams.push_back(s);
  return (int)t->streams.size() - 1;
}
int new_event(nnr_tape* t, uint64_t key) {
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return -1;
  t->events.push_back(e);
  t->event_keys.push_back(key);
  return (int)t->events.size() - 1;
}
}  // namespace

extern "C" int nnr_tape_create(nnr_tape** out) {
  if (!out) return NNR_ERR_ARG;
  *out = new nnr_tape();
  (*out)->seg_begin.push_back(0);
  return NNR_OK;
}

extern "C" int nnr_tape_destroy(nnr_tape* t) {
  if (!t) return NNR_OK;
  for (hipEvent_t e : t->events) (void)hipEventDestroy(e);
  for (auto& set : t->tset) for (hipEvent_t e : set) (void)hipEventDestroy(e);
  delete t;
  return NNR_OK;
}

// Index of a recordable entry point, or -1 (host-only queries and entry points without a stream are not recordable).
extern "C" int nnr_tape_fn_id(const char* name) {
  for (int i = 0; i < NREG; ++i) if (std::strcmp(REGISTRY[i].name, name) == 0) return i;
  return -1;
}
extern "C" int nnr_tape_fn_nargs(int fn) { return (fn >= 0 && fn < NREG) ? REGISTRY[fn].nargs : -1; }

// Append one call.  slots[nslots]: the arguments before the stream, one 8-byte slot each (integers sign-extended, floats in the
// low 4 bytes, pointers as is).  Arguments that point to HOST structs / arrays (nnr_gemm_args, nnr_lstm_problem[], ...) are
// passed as blobs: blob_slot[i] names the argument, blob_ptr[i] / blob_bytes[i] the bytes to copy into the tape.
// blob_off_out[i] receives the byte offset of copy i inside the arena (for nnr_tape_patch).  Returns the call index (>= 0).
extern "C" int nnr_tape_call(nnr_tape* t, int fn, hipStream_t stream, const uint64_t* slots, int nslots, const int* blob_slot,
                             const void* const* blob_ptr, const size_t* blob_bytes, int nblobs, int tag, size_t* slot_off_out,
                             size_t* blob_off_out) {
  if (!t || t->final || fn < 0 || fn >= NREG || nslots != REGISTRY[fn].nargs || (nslots > 0 && !slots)) return NNR_ERR_ARG;
  // every blob is validated BEFORE the arena is touched: a rejected call leaves no orphaned slots / blob references behind
  if (nblobs < 0 || (nblobs > 0 && (!blob_slot || !blob_ptr || !blob_bytes))) return NNR_ERR_ARG;
  for (int i = 0; i < nblobs; ++i)
    if (blob_slot[i] < 0 || blob_slot[i] >= nslots || !blob_ptr[i]) return NNR_ERR_ARG;
  Call c;
  c.fn = fn; c.stream = stream_index(t, stream); c.nslots = nslots; c.tag = tag; c.slot_off = t->arena.size();
  t->arena.insert(t->arena.end(), slots, slots + nslots);
  for (int i = 0; i < nblobs; ++i) {
    const size_t units = (blob_bytes[i] + 7) / 8, off = t->arena.size();
    t->arena.resize(off + units, 0);
    std::memcpy(&t->arena[off], blob_ptr[i], blob_bytes[i]);
    t->blob_refs.push_back({c.slot_off + (size_t)blob_slot[i], off});
    if (blob_off_out) blob_off_out[i] = off * 8;
  }
  if (slot_off_out) *slot_off_out = c.slot_off * 8;
  if (tag >= 0) t->ntagged = tag + 1 > t->ntagged ? tag + 1 : t->ntagged;
  t->calls.push_back(c);
  t->ops.push_back({OP_CALL, (int)t->calls.size() - 1, 0});
  return (int)t->calls.size() - 1;
}

// `waiter` waits for everything enqueued on `signaller` so far (torch's Stream.wait_stream).
extern "C" int nnr_tape_wait_stream(nnr_tape* t, hipStream_t waiter, hipStream_t signaller) {
  if (!t || t->final) return NNR_ERR_ARG;
  if (waiter == signaller) return NNR_OK;
  const int e = new_event(t, 0);
  if (e < 0) return NNR_ERR_LAUNCH;
  t->ops.push_back({OP_RECORD, e, stream_index(t, signaller)});
  t->ops.push_back({OP_WAIT, stream_index(t, waiter), e});
  return NNR_OK;
}
// Event.record(stream) / Stream.wait_event(event) with a caller-chosen non-zero key per event object; a key that is recorded again
// gets a fresh event (an event object re-recorded later in the step).
extern "C" int nnr_tape_event_record(nnr_tape* t, uint64_t key, hipStream_t s) {
  if (!t || t->final || !key) return NNR_ERR_ARG;
  const int e = new_event(t, key);
  if (e < 0) return NNR_ERR_LAUNCH;
  t->ops.push_back({OP_RECORD, e, stream_index(t, s)});
  return NNR_OK;
}
extern "C" int nnr_tape_event_wait(nnr_tape* t, hipStream_t s, uint64_t key) {
  if (!t || t->final || !key) return NNR_ERR_ARG;
  for (int i = (int)t->events.size() - 1; i >= 0; --i)
    if (t->event_keys[i] == key) { t->ops.push_back({OP_WAIT, stream_index(t, s), i}); return NNR_OK; }
  return NNR_ERR_ARG;                                // waiting for an event that was never recorded inside the tape
}
// End of a segment: the host runs something of its own here (returns the index of the segment that STARTS now).
extern "C" int nnr_tape_segment(nnr_tape* t) {
  if (!t || t->final) return NNR_ERR_ARG;
  t->seg_begin.push_back(t->ops.size());
  return (int)t->seg_begin.size() - 1;
}
// Before every replay, `width` bytes (4 or 8) at arena byte offset `byte_off` are set to  value[kind] + addend  (kind < 1000) or
// input[kind - 1000] + addend (kind >= 1000).
extern "C" int nnr_tape_patch(nnr_tape* t, size_t byte_off, int kind, int width, int64_t addend) {
  if (!t || t->final || kind < 0 || (width != 4 && width != 8) || byte_off + (size_t)width > t->arena.size() * 8) return NNR_ERR_ARG;
  t->patches.push_back({byte_off, kind, width, addend});
  return NNR_OK;
}
extern "C" int nnr_tape_finalize(nnr_tape* t) {
  if (!t || t->final) return NNR_ERR_ARG;
  for (const BlobRef& r : t->blob_refs) t->arena[r.slot] = (uint64_t)(uintptr_t)&t->arena[r.blob_off];
  t->seg_begin.push_back(t->ops.size());
  t->final = true;
  return NNR_OK;
}
extern "C" int nnr_tape_info(const nnr_tape* t, int* calls, int* ops, int* segments, int* streams, size_t* arena_bytes) {
  if (!t) return NNR_ERR_ARG;
  if (calls) *calls = (int)t->calls.size();
  if (ops) *ops = (int)t->ops.size();
  if (segments) *segments = (int)t->seg_begin.size() - (t->final ? 1 : 0);
  if (streams) *streams = (int)t->streams.size();
  if (arena_bytes) *arena_bytes = t->arena.size() * 8;
  return NNR_OK;
}

// Replay segment `segment`.  Patches are applied when segment == 0 (values / inputs of THIS step).  timing_set >= 0: HIP events
// around every tagged call on its own stream (set index < 64; read back with nnr_tape_timings).  Returns NNR_OK or the first
// failing call's code (nnr_tape_last_error names the call).
extern "C" int nnr_tape_replay(nnr_tape* t, int segment, const uint64_t* values, int nvalues, const uint64_t* inputs, int ninputs,
                               int timing_set) {
  if (!t || !t->final || segment < 0 || segment + 1 >= (int)t->seg_begin.size()) return NNR_ERR_ARG;
  if (segment == 0) {
    uint8_t* base = reinterpret_cast<uint8_t*>(t->arena.data());
    for (const Patch& p : t->patches) {
      uint64_t v;
      if (p.kind >= 1000) { if (p.kind - 1000 >= ninputs) return NNR_ERR_ARG; v = inputs[p.kind - 1000]; }
      else { if (p.kind >= nvalues) return NNR_ERR_ARG; v = values[p.kind]; }
      v += (uint64_t)p.addend;
      if (p.width == 4) { const uint32_t w = (uint32_t)v; std::memcpy(base + p.byte_off, &w, 4); }
      else std::memcpy(base + p.byte_off, &v, 8);
    }
  }
  std::vector<hipEvent_t>* ts = nullptr;
  if (timing_set >= 0 && t->ntagged > 0) {
    if (timing_set >= MAX_TIMING_SETS) timing_set = MAX_TIMING_SETS - 1;
    if ((int)t->tset.size() <= timing_set) t->tset.resize(timing_set + 1);
    ts = &t->tset[timing_set];
    if (ts->empty()) {
      // built aside and moved in only when EVERY create succeeded: a half-built set would be taken for a usable one by the next
      // timing replay and destroyed (uninitialised handles) by nnr_tape_destroy
      std::vector<hipEvent_t> fresh;
      fresh.reserve(2 * (size_t)t->ntagged);
      for (int i = 0; i < 2 * t->ntagged; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) {
          for (hipEvent_t d : fresh) (void)hipEventDestroy(d);
          return NNR_ERR_LAUNCH;
        }
        fresh.push_back(e);
      }
      *ts = std::move(fresh);
    }
  }
  for (size_t i = t->seg_begin[segment]; i < t->seg_begin[segment + 1]; ++i) {
    const Op& o = t->ops[i];
    if (o.kind == OP_CALL) {
      const Call& c = t->calls[o.a];
      hipStream_t s = t->streams[c.stream];
      const bool timed = ts && c.tag >= 0;
      if (timed) (void)hipEventRecord((*ts)[2 * c.tag], s);
      const int rc = REGISTRY[c.fn].thunk(&t->arena[c.slot_off], s);
      if (timed) (void)hipEventRecord((*ts)[2 * c.tag + 1], s);
      if (rc != NNR_OK) { t->last_rc = rc; t->last_failed_call = o.a; return rc; }
    } else if (o.kind == OP_RECORD) {
      if (hipEventRecord(t->events[o.a], t->streams[o.b]) != hipSuccess) return NNR_ERR_LAUNCH;
    } else {
      if (hipStreamWaitEvent(t->streams[o.a], t->events[o.b], 0) != hipSuccess) return NNR_ERR_LAUNCH;
    }
  }
  return NNR_OK;
}

// Create the HIP events of timing sets [0, nsets) now (host-side only): a caller that is about to time a window of replays does this
// BEFORE the window, so that the first instrumented replay does not spend a millisecond of host time in hipEventCreate while the GPU
// it has just synchronised with sits idle.
extern "C" int nnr_tape_prepare_timing(nnr_tape* t, int nsets) {
  if (!t || !t->final || nsets < 0) return NNR_ERR_ARG;
  if (nsets > MAX_TIMING_SETS) nsets = MAX_TIMING_SETS;
  if (t->ntagged <= 0) return NNR_OK;
  if ((int)t->tset.size() < nsets) t->tset.resize(nsets);
  for (int s = 0; s < nsets; ++s) {
    if (!t->tset[s].empty()) continue;
    std::vector<hipEvent_t> fresh;
    fresh.reserve(2 * (size_t)t->ntagged);
    for (int i = 0; i < 2 * t->ntagged; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) {
        for (hipEvent_t d : fresh) (void)hipEventDestroy(d);
        return NNR_ERR_LAUNCH;
      }
      fresh.push_back(e);
    }
    t->tset[s] = std::move(fresh);
  }
  return NNR_OK;
}

// ms[tag] = duration of the tagged call in timing set `set` (synchronises with those events).  Returns the number written.
extern "C" int nnr_tape_timings(nnr_tape* t, int set, float* ms, int n) {
  if (!t || set < 0 || set >= (int)t->tset.size() || t->tset[set].empty()) return NNR_ERR_ARG;
  const int m = n < t->ntagged ? n : t->ntagged;
  for (int i = 0; i < m; ++i) {
    float v = 0.f;
    if (hipEventSynchronize(t->tset[set][2 * i + 1]) != hipSuccess || hipEventElapsedTime(&v, t->tset[set][2 * i], t->tset[set][2 * i + 1]) != hipSuccess)
      v = -1.f;
    ms[i] = v;
  }
  return m;
}
// start_ms[tag] = start of the tagged call relative to the start of tagged call 0, stream_idx[tag] = index of its HIP stream in the
// tape (a per-call timeline of one replayed step, tools/tape_timeline.py).
extern "C" int nnr_tape_timeline(nnr_tape* t, int set, float* start_ms, float* dur_ms, int* stream_idx, int n) {
  if (!t || set < 0 || set >= (int)t->tset.size() || t->tset[set].empty()) return NNR_ERR_ARG;
  const int m = n < t->ntagged ? n : t->ntagged;
  std::vector<int> st(t->ntagged, -1);
  for (const Call& c : t->calls) if (c.tag >= 0) st[c.tag] = c.stream;
  for (int i = 0; i < m; ++i) {
    float a = -1.f, d = -1.f;
    if (hipEventSynchronize(t->tset[set][2 * i + 1]) == hipSuccess) {
      if (hipEventElapsedTime(&a, t->tset[set][0], t->tset[set][2 * i]) != hipSuccess) a = -1.f;
      if (hipEventElapsedTime(&d, t->tset[set][2 * i], t->tset[set][2 * i + 1]) != hipSuccess) d = -1.f;
    }
    start_ms[i] = a; dur_ms[i] = d; stream_idx[i] = st[i];
  }
  return m;
}
extern "C" int nnr_tape_last_error(const nnr_tape* t, int* rc, int* call, char* name, int name_cap) {
  if (!t) return NNR_ERR_ARG;
  if (rc) *rc = t->last_rc;
  if (call) *call = t->last_failed_call;
  if (name && name_cap > 0) {
    name[0] = 0;
    if (t->last_failed_call >= 0) std::strncpy(name, REGISTRY[t->calls[t->last_failed_call].fn].name, name_cap - 1), name[name_cap - 1] = 0;
  }
  return NNR_OK;
}

// ---- the small device ops a fully native step needs in place of the host framework's fill / copy / index-put kernels
namespace {
__global__ void fill_column_u8_kernel(uint8_t* m, int rows, int cols, int col, uint8_t v) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) m[(long)r * cols + col] = v;
}
}  // namespace
extern "C" int nnr_fill_zero(void* p, size_t bytes, hipStream_t stream) {
  if (!p && bytes) return NNR_ERR_ARG;
  if (bytes == 0) return NNR_OK;
  return hipMemsetAsync(p, 0, bytes, stream) == hipSuccess ? NNR_OK : NNR_ERR_LAUNCH;
}
extern "C" int nnr_copy_bytes(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if ((!dst || !src) && bytes) return NNR_ERR_ARG;
  if (bytes == 0) return NNR_OK;
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream) == hipSuccess ? NNR_OK : NNR_ERR_LAUNCH;
}
extern "C" int nnr_fill_column_u8(uint8_t* m, int rows, int cols, int col, int value, hipStream_t stream) {
  if (!m || rows < 0 || col < 0 || col >= cols) return NNR_ERR_ARG;
  if (rows == 0) return NNR_OK;
  hipLaunchKernelGGL(fill_column_u8_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, m, rows, cols, col, (uint8_t)value);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
