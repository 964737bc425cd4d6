// Sequence planner: replaces newsEncoders.py:106-120 (mask fix, lengths, two torch.sort calls, index_select,
// pack_padded_sequence and its sorted_length.cpu() host sync) with three small kernels whose outputs stay in
// device memory.  Layout produced ("time-major packed", the PackedSequence order):
//   order[s]  : original row of the sequence at sorted position s (descending length, stable unless perm_in given)
//   rank[i]   : sorted position of original row i
//   slen[s]   : length of the sequence at sorted position s
//   bs[t]     : number of sequences with length > t;  off[t] = sum_{t'<t} bs[t'];  off[L] = total valid tokens
//   packed row of (s, t) = off[t] + s   (valid iff s < bs[t])
//   row_seq[row] = s, tok[row] = ids[order[s]][t], prev_f[row] / prev_r[row] = packed row of the previous step of
//   the forward / reverse recurrence (or -1 at the start of the sequence).
#include "common.h"

namespace {

constexpr int PLAN_THREADS = 1024;
constexpr int MAX_L = 512;
constexpr int MAX_SEG = 128;          // 64-item segments handled by the fast stable ranking (n <= 8192)

// ---- A. one wave per row: mask[:,0] = 1 in place (newsEncoders.py:108-109), length = popcount(mask row)
__global__ __launch_bounds__(256) void plan_len_kernel(uint8_t* __restrict__ mask, uint8_t* __restrict__ mask1, int n0, int n, int L,
                                                       int* __restrict__ len_out) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  uint8_t* row = (i < n0) ? mask + (long)i * L : mask1 + (long)(i - n0) * L;      // rows >= n0: the second encoder call's tensor
  int c = 0;
  for (int t = lane; t < L; t += 64) c += (t == 0 || row[t]) ? 1 : 0;
  c = (int)wave_sum((float)c);
  if (lane == 0) { row[0] = 1; len_out[i] = c; }
}

// ---- B. single workgroup: stable descending rank, batch sizes and row offsets
// rank(i) = #(len > len_i) + #(j < i : len_j == len_i).  The second term is a per-length prefix count: per 64-item
// segment histograms in LDS, a serial prefix over segments per length value, and a 64-step shuffle scan inside a segment.
__global__ __launch_bounds__(PLAN_THREADS) void plan_rank_kernel(const int* __restrict__ len_in, int n, int L,
                                                                 const int* __restrict__ perm_in, int* __restrict__ order,
                                                                 int* __restrict__ rank, int* __restrict__ slen,
                                                                 int* __restrict__ bs, int* __restrict__ off, int fast_ok) {
  extern __shared__ int sm[];          // [L+1] hist/gt, [L+1] offs, then fast path: [nseg][L+1] segment counts
  int* hist = sm;
  int* offs = sm + (L + 1);
  int* seg = offs + (L + 1);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nseg = (n + 63) >> 6;
  const bool fast = (perm_in == nullptr) && nseg <= MAX_SEG && fast_ok;
  for (int t = tid; t <= L; t += PLAN_THREADS) hist[t] = 0;
  if (fast) for (int t = tid; t < nseg * (L + 1); t += PLAN_THREADS) seg[t] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += PLAN_THREADS) {
    const int l = len_in[i];
    atomicAdd(&hist[l], 1);
    if (fast) atomicAdd(&seg[(i >> 6) * (L + 1) + l], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;                       // #(len > t), walking t downward
    for (int t = L; t >= 0; --t) { const int h = hist[t]; hist[t] = run; run += h; }
    int o = 0;
    for (int t = 0; t < L; ++t) { offs[t] = o; o += hist[t]; }
    offs[L] = o;
  }
  if (fast) {                          // exclusive prefix over segments, per length value
    for (int l = tid; l <= L; l += PLAN_THREADS) {
      int run = 0;
      for (int sgi = 0; sgi < nseg; ++sgi) { const int c = seg[sgi * (L + 1) + l]; seg[sgi * (L + 1) + l] = run; run += c; }
    }
  }
  __syncthreads();
  for (int t = tid; t < L; t += PLAN_THREADS) bs[t] = hist[t];
  for (int t = tid; t <= L; t += PLAN_THREADS) off[t] = offs[t];
  if (perm_in) {
    for (int s = tid; s < n; s += PLAN_THREADS) {
      const int i = perm_in[s];
      order[s] = i; rank[i] = s; slen[s] = len_in[i];
    }
  } else if (fast) {
    for (int sgi = wv; sgi < nseg; sgi += PLAN_THREADS / 64) {
      const int i = sgi * 64 + lane;
      const int li = (i < n) ? len_in[i] : -1;
      int before = 0;
      for (int k = 0; k < 64; ++k) {
        const int v = __shfl(li, k, 64);
        before += (k < lane && v == li) ? 1 : 0;
      }
      if (i < n) {
        const int pos = hist[li] + seg[sgi * (L + 1) + li] + before;
        order[pos] = i; rank[i] = pos; slen[pos] = li;
      }
    }
  } else {
    for (int i = tid; i < n; i += PLAN_THREADS) {
      const int li = len_in[i];
      int pos = 0;
      for (int j = 0; j < n; ++j) { const int lj = len_in[j]; pos += (lj > li) || (lj == li && j < i); }
      order[pos] = i; rank[i] = pos; slen[pos] = li;
    }
  }
}

// ---- C. one thread per (sorted position s, time t): per-row maps of the packed layout
__global__ void plan_rows_kernel(const int* __restrict__ ids, const int* __restrict__ ids1, int n0, int n, int L, const int* __restrict__ order,
                                 const int* __restrict__ slen, const int* __restrict__ off, int* __restrict__ row_seq,
                                 int* __restrict__ tok, int* __restrict__ prev_f, int* __restrict__ prev_r) {
  const long total = (long)n * L;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int t = idx / n, s = idx - (long)t * n;        // s fastest: consecutive threads -> consecutive packed rows
    const int l = slen[s];
    if (t >= l) continue;
    const int row = off[t] + s;
    row_seq[row] = s;
    if (tok) {
      const int i = order[s];
      tok[row] = !ids ? 0 : (i < n0 ? ids[(long)i * L + t] : ids1[(long)(i - n0) * L + t]);
    }
    prev_f[row] = (t > 0) ? off[t - 1] + s : -1;
    prev_r[row] = (t + 1 < l) ? off[t + 1] + s : -1;
  }
}

// ---- D. CNE rank pairing over a UNION of two encoder calls (newsEncoders.py:112-115,128-129 applied per call).  The reference
// pairs the title stream's sorted position r with the content stream's sorted position r INSIDE one call.  With the candidate
// call (rows [0, n0)) and the history call (rows [n0, n)) planned as one packed stream, "position inside the call" is the
// number of same-call sequences in front in the union order (the union order restricted to a call IS that call's order).
//   pm_t[s] = union position, in the content stream, of the partner of the title sequence at union position s; pm_c likewise.
// One workgroup; tmp = 4 n ints.
__global__ __launch_bounds__(PLAN_THREADS) void pair_map_kernel(const int* __restrict__ order_t, const int* __restrict__ order_c, int n0, int n,
                                                                int* __restrict__ pm_t, int* __restrict__ pm_c, int* __restrict__ tmp) {
  __shared__ int scan[2][PLAN_THREADS];
  const int tid = threadIdx.x;
  const int chunk = (n + PLAN_THREADS - 1) / PLAN_THREADS;
  const int lo = min(n, tid * chunk), hi = min(n, lo + chunk);
  int* key[2] = {tmp, tmp + n};                 // key_x[s] = slot of (call, position inside the call) of stream x's union position s
  int* inv[2] = {tmp + 2 * (long)n, tmp + 3 * (long)n};   // inv_x[slot] = s
  const int* order[2] = {order_t, order_c};
  int c0[2] = {0, 0};
  for (int x = 0; x < 2; ++x)
    for (int s = lo; s < hi; ++s) c0[x] += order[x][s] < n0 ? 1 : 0;
  scan[0][tid] = c0[0];
  scan[1][tid] = c0[1];
  __syncthreads();
  for (int d = 1; d < PLAN_THREADS; d <<= 1) {  // inclusive Hillis-Steele scan of the per-thread counts
    const int a = tid >= d ? scan[0][tid - d] : 0, b = tid >= d ? scan[1][tid - d] : 0;
    __syncthreads();
    scan[0][tid] += a;
    scan[1][tid] += b;
    __syncthreads();
  }
  for (int x = 0; x < 2; ++x) {
    int before0 = scan[x][tid] - c0[x];         // call-0 sequences in front of this thread's chunk
    for (int s = lo; s < hi; ++s) {
      const bool first = order[x][s] < n0;
      const int slot = first ? before0 : n0 + (s - before0);
      before0 += first ? 1 : 0;
      key[x][s] = slot;
      inv[x][slot] = s;
    }
  }
  __threadfence_block();
  __syncthreads();
  for (int s = tid; s < n; s += PLAN_THREADS) {
    pm_t[s] = inv[1][key[0][s]];
    pm_c[s] = inv[0][key[1][s]];
  }
}

}  // namespace

extern "C" int nnr_cne_pair_map(const int* order_t, const int* order_c, int n0, int n, int* pm_t, int* pm_c, int* tmp, hipStream_t stream) {
  if (!order_t || !order_c || !pm_t || !pm_c || !tmp || n <= 0 || n0 < 0 || n0 > n) return NNR_ERR_ARG;
  hipLaunchKernelGGL(pair_map_kernel, dim3(1), dim3(PLAN_THREADS), 0, stream, order_t, order_c, n0, n, pm_t, pm_c, tmp);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

static int seq_plan_impl(uint8_t* mask, const int* ids, int n0, uint8_t* mask1, const int* ids1, int n, int L, const int* perm_in, int* len_out,
                         int* order, int* rank, int* slen, int* bs, int* off, int* row_seq, int* tok, int* prev_f, int* prev_r,
                         hipStream_t stream) {
  if (!mask || n <= 0 || L <= 0 || L > MAX_L) return NNR_ERR_ARG;
  const int nseg = (n + 63) / 64;
  bool fast = !perm_in && nseg <= MAX_SEG;
  size_t shm = (size_t)(2 * (L + 1) + (fast ? nseg * (L + 1) : 0)) * sizeof(int);
  if (shm > 160 * 1024) {          // the per-segment histograms do not fit the LDS: the O(n^2) rank path needs only 2 * (L + 1) ints
    fast = false;
    shm = (size_t)(2 * (L + 1)) * sizeof(int);
  }
  hipLaunchKernelGGL(plan_len_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, mask, mask1, n0, n, L, len_out);
  NNR_CHECK_LAUNCH();
  if (shm > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(plan_rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
    return NNR_ERR_LAUNCH;
  hipLaunchKernelGGL(plan_rank_kernel, dim3(1), dim3(PLAN_THREADS), shm, stream, len_out, n, L, perm_in, order, rank, slen, bs, off, fast ? 1 : 0);
  NNR_CHECK_LAUNCH();
  const long total = (long)n * L;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(plan_rows_kernel, dim3(blocks), dim3(256), 0, stream, ids, ids1, n0, n, L, order, slen, off, row_seq, tok, prev_f, prev_r);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_seq_plan(uint8_t* mask, const int* ids, int n, int L, const int* perm_in, int* len_out, int* order,
                            int* rank, int* slen, int* bs, int* off, int* row_seq, int* tok, int* prev_f, int* prev_r,
                            hipStream_t stream) {
  return seq_plan_impl(mask, ids, n, nullptr, nullptr, n, L, perm_in, len_out, order, rank, slen, bs, off, row_seq, tok, prev_f, prev_r, stream);
}

// Two encoder calls (candidates: n0 sequences, history: n1) planned as ONE packed stream of n0 + n1 sequences: rows [0, n0) of
// every per-sequence array belong to the first call's tensors, rows [n0, n0 + n1) to the second's.
extern "C" int nnr_seq_plan_pair(uint8_t* mask0, const int* ids0, int n0, uint8_t* mask1, const int* ids1, int n1, int L, const int* perm_in,
                                 int* len_out, int* order, int* rank, int* slen, int* bs, int* off, int* row_seq, int* tok, int* prev_f,
                                 int* prev_r, hipStream_t stream) {
  if (!mask1 || n0 <= 0 || n1 <= 0 || (ids0 && !ids1)) return NNR_ERR_ARG;
  return seq_plan_impl(mask0, ids0, n0, mask1, ids1, n0 + n1, L, perm_in, len_out, order, rank, slen, bs, off, row_seq, tok, prev_f, prev_r, stream);
}

// ---- packed token rows for the MHSA news encoder (round 5; newsEncoders.py:187-200, layers.py:132-148,167-175).  The reference runs its
// Q/K/V and attention projections over all n * L padded positions; padded positions provably never reach the result (their keys are masked
// with -1e9, their pooled weight is exactly 0) unless a title is FULLY masked (softmax over 32 x -1e9 is uniform: all 32 positions count).
//   cover[i][t] = 1 for t <= (last valid position of title i), all ones for a title without any valid position
// is the set of rows that must exist; nnr_seq_plan on `cover` packs exactly those (for the prefix-shaped masks of the corpus,
// MIND_corpus.py:316,352: cover == mask), and the ORIGINAL mask still masks keys / pooled positions inside the kernels.
__global__ void mask_cover_kernel(const uint8_t* __restrict__ mask, int n, int L, uint8_t* __restrict__ cover) {
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  int last = -1;
  for (int t = lane; t < L; t += 64) if (mask[(long)i * L + t]) last = max(last, t);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
  if (last < 0) last = L - 1;
  for (int t = lane; t < L; t += 64) cover[(long)i * L + t] = t <= last ? 1 : 0;
}
// rowmap[i*L + t] = packed row of position t of sequence i (off[t] + rank[i]) for t < len[i], else -1
__global__ void seq_rowmap_kernel(const int* __restrict__ off, const int* __restrict__ rank, const int* __restrict__ len, int n, int L,
                                  int* __restrict__ rowmap) {
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)n * L) return;
  const int i = (int)(idx / L), t = (int)(idx - (long)i * L);
  rowmap[idx] = t < len[i] ? off[t] + rank[i] : -1;
}
// Virtual samples of the GROUPED attention core (csrc/mhsa.hip: mhsa_pairing): in the plan's sorted order, the n16 = off[17] - off[16] titles that
// cover more than 16 positions stay alone (virtual sample v = sorted position v); the n8 - n16 titles covering 9..16 positions go two to a sample
// (positions 0..15 / 16..31), the n - n8 covering <= 8 four to a sample (0..7 / 8..15 / 16..23 / 24..31).  vrowmap[v, q] = packed row of that
// position (off[t] + s) or -1, vmask[v, q] = the ORIGINAL key mask of that position; rows v >= n_virtual are filled with (-1, 0) and never read.
__global__ void mhsa_pair_map_kernel(const int* __restrict__ off, const int* __restrict__ slen, const int* __restrict__ order,
                                     const uint8_t* __restrict__ mask, int n, int* __restrict__ vrowmap, uint8_t* __restrict__ vmask) {
  constexpr int L = 32;
  const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (idx >= (long)n * L) return;
  const int v = (int)(idx / L), q = (int)(idx - (long)v * L);
  const int n16 = off[17] - off[16], n8 = off[9] - off[8];
  const int np = (n8 - n16 + 1) / 2, nv = n16 + np + (n - n8 + 3) / 4;
  int row = -1;
  uint8_t mk = 0;
  if (v < nv) {
    int s, t, end;
    if (v < n16) { s = v; t = q; end = n16; }
    else if (v < n16 + np) { s = n16 + 2 * (v - n16) + (q >> 4); t = q & 15; end = n8; }
    else { s = n8 + 4 * (v - n16 - np) + (q >> 3); t = q & 7; end = n; }
    if (s < end) {
      if (t < slen[s]) row = off[t] + s;
      mk = mask[(long)order[s] * L + t];
    }
  }
  vrowmap[idx] = row;
  vmask[idx] = mk;
}
extern "C" int nnr_mhsa_pair_map(const int* off, const int* slen, const int* order, const uint8_t* mask, int n, int L, int* vrowmap, uint8_t* vmask,
                                 hipStream_t stream) {
  if (!off || !slen || !order || !mask || !vrowmap || !vmask || n <= 0) return NNR_ERR_ARG;
  if (L != 32) return NNR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(mhsa_pair_map_kernel, dim3((unsigned)(((long)n * L + 255) / 256)), dim3(256), 0, stream, off, slen, order, mask, n, vrowmap, vmask);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
extern "C" int nnr_mask_cover(const uint8_t* mask, int n, int L, uint8_t* cover, hipStream_t stream) {
  if (!mask || !cover || n <= 0 || L <= 0) return NNR_ERR_ARG;
  hipLaunchKernelGGL(mask_cover_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, mask, n, L, cover);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
extern "C" int nnr_seq_rowmap(const int* off, const int* rank, const int* len, int n, int L, int* rowmap, hipStream_t stream) {
  if (!off || !rank || !len || !rowmap || n <= 0 || L <= 0) return NNR_ERR_ARG;
  hipLaunchKernelGGL(seq_rowmap_kernel, dim3((unsigned)(((long)n * L + 255) / 256)), dim3(256), 0, stream, off, rank, len, n, L, rowmap);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
