// Deterministic embedding-row gradient (nn.Embedding backward, newsEncoders.py:117-118 under autograd; config.py:125-130 asks for
// reproducible runs): dtable[w, :] += sum over the token rows t with id w of mask(t, :) * dout[t, :].
//
// The f32-atomic scatter (misc.hip: embed_scatter_kernel) is bound by the memory-side atomic units (~1.3 TB/s chip-wide, 14x slower on
// hot rows) and adds a word's rows in whatever order the waves arrive.  Here the token rows are SORTED BY WORD ID once per step --
// the ids are known as soon as the token stream is planned, so the sort runs on the leaf stream under the forward pass, off the
// critical chain -- and the backward pass is a segmented reduction over the sorted list:
//   * nnr_token_sort: keys = word id of each live packed row (rows beyond the live count / invalid ids -> the pad key V), values =
//     row index; stable LSD radix sort (rocPRIM's device radix sort, compiled into this library from its headers), only the
//     ceil(log2(V + 1)) significant bits.  Stable => inside a word's segment the rows stay in ascending row order.
//   * embed_scatter_sorted_kernel: one wave per chunk of 32 sorted rows; runs of equal ids are summed in registers in list order (4
//     rows in flight per wave).  A run that lies wholly inside its chunk is a complete segment and is added to the table row with ONE
//     atomic per element (the table gradient receives one such add per token stream and step -- two addends into a zeroed buffer
//     commute, so the result is bit-reproducible); a run that continues across a chunk boundary is stored as a partial row.
//   * embed_scatter_sorted_fix_kernel: the wave whose chunk holds the START of a multi-chunk segment adds the partial rows of the
//     following chunks in chunk order (8 rows in flight) and issues the one atomic add per element.
// Traffic: every gradient row is read once, every touched table row written once (+ 2 x 1.25 KB per chunk-crossing run).
#include <cstring>
#include <string.h>
#include "common.h"
#include <rocprim/device/device_radix_sort.hpp>

namespace {

constexpr int SS_CH = 32;        // sorted rows per wave
constexpr int SS_MAXJ = 5;       // columns per lane: dim <= 320
constexpr int SS_FL = 8;         // rows in flight per wave in the segmented reduction
constexpr int SS_PITCH = 64 * SS_MAXJ;

__global__ __launch_bounds__(256) void token_sort_keys_kernel(const int* __restrict__ tok, long cap, const int* __restrict__ n_dev, unsigned V,
                                                              unsigned* __restrict__ keys, int* __restrict__ rows) {
  const long n = n_dev ? min(cap, (long)*n_dev) : cap;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < cap; i += (long)gridDim.x * blockDim.x) {
    unsigned k = V;
    if (i < n) {
      const int t = tok[i];
      if (t >= 0 && (unsigned)t < V) k = (unsigned)t;
    }
    keys[i] = k;
    rows[i] = (int)i;
  }
}

__global__ __launch_bounds__(256) void embed_scatter_sorted_kernel(const float* __restrict__ dout, const unsigned* __restrict__ keys,
                                                                   const int* __restrict__ rows, long cap, unsigned V, int dim,
                                                                   float* __restrict__ dtable, uint32_t seed, uint32_t thr, float scale,
                                                                   float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long c = blockIdx.x * 4L + wv;
  const long p0 = c * SS_CH;
  if (p0 >= cap) return;
  const long p1 = min(cap, p0 + SS_CH);
  const int cnt = (int)(p1 - p0);
  unsigned myk = V;
  int myr = 0;
  if (lane < cnt) { myk = keys[p0 + lane]; myr = rows[p0 + lane]; }
  const unsigned k0 = __shfl(myk, 0, 64);
  if (k0 >= V) return;                                            // sorted: nothing but pad rows from here on
  const unsigned prevK = p0 > 0 ? keys[p0 - 1] : 0xFFFFFFFFu;     // (0xFFFFFFFF equals no valid key)
  const unsigned nextK = p1 < cap ? keys[p1] : 0xFFFFFFFFu;
  float acc[SS_MAXJ];
#pragma unroll
  for (int j = 0; j < SS_MAXJ; ++j) acc[j] = 0.f;
  unsigned run_key = k0;
  bool first = true;
  auto flush = [&](bool open_right) {
    const bool open_left = first && prevK == run_key;
    if (!open_left && !open_right) {                              // the whole segment of this word: one add per element
#pragma unroll
      for (int j = 0; j < SS_MAXJ; ++j) {
        const int col = lane + 64 * j;
        if (col < dim) atomicAdd(&dtable[(long)run_key * dim + col], acc[j]);
      }
    } else {
      float* dst = partial + (c * 2 + (first ? 0 : 1)) * (long)SS_PITCH;
#pragma unroll
      for (int j = 0; j < SS_MAXJ; ++j) dst[lane + 64 * j] = acc[j];
    }
  };
  bool done = false;
  const int dlast = dim - 1;
  for (int i0 = 0; i0 < cnt && !done; i0 += SS_FL) {
    // SS_FL (8; round 4 first had 4: half as many dependent round trips per chunk now) rows in flight: every load is issued UNCONDITIONALLY before any of them is used (a branch per row cuts the block, and the
    // wait for row u's loads then sits in front of row u + 1's: one row in flight -- 32 dependent round trips per wave, 85 us for the
    // 88 k-row content stream); pad entries read row 0 / a clamped column and are discarded below
    float v[SS_FL][SS_MAXJ];
    unsigned kk[SS_FL];
    int rr[SS_FL];
#pragma unroll
    for (int u = 0; u < SS_FL; ++u) {
      const int i = i0 + u;                                       // (< 64: lanes >= cnt hold the pad key and row 0)
      kk[u] = __shfl(myk, i, 64);
      rr[u] = __shfl(myr, i, 64);
    }
#pragma unroll
    for (int u = 0; u < SS_FL; ++u) {
      const float* src = dout + (long)rr[u] * dim;
#pragma unroll
      for (int j = 0; j < SS_MAXJ; ++j) v[u][j] = src[min(lane + 64 * j, dlast)];
    }
#pragma unroll
    for (int u = 0; u < SS_FL; ++u) {
      if (done) break;
      if (kk[u] >= V) { done = true; break; }
      if (kk[u] != run_key) {
        flush(false);
        run_key = kk[u];
        first = false;
#pragma unroll
        for (int j = 0; j < SS_MAXJ; ++j) acc[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < SS_MAXJ; ++j) {
        const int col = lane + 64 * j;
        float x = col < dim ? v[u][j] : 0.f;
        if (thr) x = nnr_keep(seed, (uint64_t)rr[u] * dim + col, thr) ? x * scale : 0.f;
        acc[j] += x;
      }
    }
  }
  flush(nextK == run_key);                                        // (a chunk that ran into pad rows has nextK = pad: closed)
}

__global__ __launch_bounds__(256) void embed_scatter_sorted_fix_kernel(const unsigned* __restrict__ keys, long cap, unsigned V, int dim,
                                                                       float* __restrict__ dtable, const float* __restrict__ partial) {
  // A workgroup looks after four consecutive chunks; for every multi-chunk segment that STARTS in one of them (at most two per chunk:
  // its first run and its last run) all four waves add the partial rows of the following chunks -- wave w takes chunks w, w + 4, ...
  // of the run, 8 rows in flight -- and the four sums are added in wave order: a frequent word's segment is hundreds of chunks long
  // (the top word of a Zipf vocabulary owns a seventh of the tokens), and one wave walking it alone was the kernel's long pole.
  __shared__ float red[4][SS_PITCH];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long nchunks = (cap + SS_CH - 1) / SS_CH;
  for (int ci = 0; ci < 4; ++ci) {
    const long c = blockIdx.x * 4L + ci;
    const long p0 = c * SS_CH;
    if (p0 >= cap) break;                                          // (uniform over the workgroup, like every branch below)
    const long p1 = min(cap, p0 + SS_CH);
    const unsigned k0 = keys[p0];
    if (k0 >= V) break;                                            // sorted: nothing but pad rows from here on
    const unsigned prevK = p0 > 0 ? keys[p0 - 1] : 0xFFFFFFFFu;
    const unsigned kl = keys[p1 - 1];
    const unsigned nextK = p1 < cap ? keys[p1] : 0xFFFFFFFFu;
    for (int cand = 0; cand < 2; ++cand) {
      // cand 0: the chunk's first run, if the segment STARTS here and runs on into the next chunk; cand 1: its last run, likewise
      const unsigned key = cand == 0 ? k0 : kl;
      const bool own = cand == 0 ? (prevK != k0 && kl == k0 && nextK == k0) : (kl != k0 && kl < V && nextK == kl);
      if (!own) continue;
      float acc[SS_MAXJ];
#pragma unroll
      for (int j = 0; j < SS_MAXJ; ++j) acc[j] = 0.f;
      long cc = c + 1;
      while (true) {
        const bool cont = (cc + lane < nchunks) && keys[(cc + lane) * SS_CH] == key;
        const unsigned long long m = __ballot(cont);
        const int nc = (m == ~0ull) ? 64 : __builtin_ctzll(~m);   // chunks cc .. cc + nc - 1 begin with this word: their slot 0 is its partial
        for (int q = wv; q < nc; q += 32) {                        // this wave: chunks q, q + 4, ..., q + 28 of the window, in order
          float v[8][SS_MAXJ];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float* sp = partial + ((cc + min(q + 4 * u, nc - 1)) * 2) * (long)SS_PITCH;
#pragma unroll
            for (int j = 0; j < SS_MAXJ; ++j) v[u][j] = sp[lane + 64 * j];
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (q + 4 * u < nc) {
#pragma unroll
              for (int j = 0; j < SS_MAXJ; ++j) acc[j] += v[u][j];
            }
          }
        }
        cc += nc;
        if (nc < 64) break;
      }
#pragma unroll
      for (int j = 0; j < SS_MAXJ; ++j) red[wv][lane + 64 * j] = acc[j];
      __syncthreads();
      if (wv == 0) {
        const float* src = partial + (c * 2 + cand) * (long)SS_PITCH;      // the run's own part in its first chunk
#pragma unroll
        for (int j = 0; j < SS_MAXJ; ++j) {
          const int col = lane + 64 * j;
          const float t = (((src[col] + red[0][col]) + red[1][col]) + red[2][col]) + red[3][col];
          if (col < dim) atomicAdd(&dtable[(long)key * dim + col], t);
        }
      }
      __syncthreads();
    }
  }
}

int key_bits(unsigned V) {
  int b = 1;
  while (b < 32 && (V >> b)) ++b;
  return b;
}

}  // namespace

extern "C" size_t nnr_token_sort_workspace_bytes(long cap, int vocab) {
  if (cap <= 0 || vocab <= 0) return 0;
  size_t bytes = 0;
  unsigned* k = nullptr;
  int* r = nullptr;
  if (rocprim::radix_sort_pairs(nullptr, bytes, k, k, r, r, (size_t)cap, 0u, (unsigned)key_bits((unsigned)vocab), (hipStream_t)0) != hipSuccess) return 0;
  return bytes + 256;
}

extern "C" int nnr_token_sort(const int* tok, long cap, const int* n_dev, int vocab, unsigned* keys_tmp, int* rows_tmp, unsigned* keys_sorted,
                              int* rows_sorted, void* temp, size_t temp_bytes, hipStream_t stream) {
  if (!tok || !keys_tmp || !rows_tmp || !keys_sorted || !rows_sorted || !temp || cap < 0 || vocab <= 0) return NNR_ERR_ARG;
  if (cap == 0) return NNR_OK;
  const int blocks = (int)((cap + 255) / 256 > 2048 ? 2048 : (cap + 255) / 256);
  hipLaunchKernelGGL(token_sort_keys_kernel, dim3(blocks), dim3(256), 0, stream, tok, cap, n_dev, (unsigned)vocab, keys_tmp, rows_tmp);
  NNR_CHECK_LAUNCH();
  size_t need = 0;
  if (rocprim::radix_sort_pairs(nullptr, need, keys_tmp, keys_sorted, rows_tmp, rows_sorted, (size_t)cap, 0u, (unsigned)key_bits((unsigned)vocab), stream) != hipSuccess)
    return NNR_ERR_LAUNCH;
  if (need > temp_bytes) return NNR_ERR_ARG;
  if (rocprim::radix_sort_pairs(temp, need, keys_tmp, keys_sorted, rows_tmp, rows_sorted, (size_t)cap, 0u, (unsigned)key_bits((unsigned)vocab), stream) != hipSuccess)
    return NNR_ERR_LAUNCH;
  return NNR_OK;
}

extern "C" size_t nnr_embed_scatter_sorted_workspace_floats(long cap) {
  return cap <= 0 ? 0 : (size_t)((cap + SS_CH - 1) / SS_CH) * 2 * SS_PITCH;
}

extern "C" int nnr_embed_scatter_sorted(const float* dout, const unsigned* keys_sorted, const int* rows_sorted, long cap, int vocab, int dim,
                                        float* dtable, float p, uint32_t seed, float* partial_ws, hipStream_t stream) {
  if (!dout || !keys_sorted || !rows_sorted || !dtable || !partial_ws || cap < 0 || vocab <= 0 || dim <= 0 || dim > SS_PITCH) return NNR_ERR_ARG;
  if (cap == 0) return NNR_OK;
  const uint32_t thr = nnr_drop_thresh(p);
  const float scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const long nchunks = (cap + SS_CH - 1) / SS_CH;
  const int blocks = (int)((nchunks + 3) / 4);
  hipLaunchKernelGGL(embed_scatter_sorted_kernel, dim3(blocks), dim3(256), 0, stream, dout, keys_sorted, rows_sorted, cap, (unsigned)vocab, dim, dtable,
                     seed, thr, scale, partial_ws);
  NNR_CHECK_LAUNCH();
  hipLaunchKernelGGL(embed_scatter_sorted_fix_kernel, dim3(blocks), dim3(256), 0, stream, keys_sorted, cap, (unsigned)vocab, dim, dtable, partial_ws);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
