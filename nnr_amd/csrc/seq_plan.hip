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
__global__ __launch_bounds__(256) void plan_len_kernel(uint8_t* __restrict__ mask, int n, int L, int* __restrict__ len_out) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  uint8_t* row = mask + (long)i * L;
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
__global__ void plan_rows_kernel(const int* __restrict__ ids, int n, int L, const int* __restrict__ order,
                                 const int* __restrict__ slen, const int* __restrict__ off, int* __restrict__ row_seq,
                                 int* __restrict__ tok, int* __restrict__ prev_f, int* __restrict__ prev_r) {
  const long total = (long)n * L;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int t = idx / n, s = idx - (long)t * n;        // s fastest: consecutive threads -> consecutive packed rows
    const int l = slen[s];
    if (t >= l) continue;
    const int row = off[t] + s;
    row_seq[row] = s;
    if (tok) tok[row] = ids ? ids[(long)order[s] * L + t] : 0;
    prev_f[row] = (t > 0) ? off[t - 1] + s : -1;
    prev_r[row] = (t + 1 < l) ? off[t + 1] + s : -1;
  }
}

}  // namespace

extern "C" int nnr_seq_plan(uint8_t* mask, const int* ids, int n, int L, const int* perm_in, int* len_out, int* order,
                            int* rank, int* slen, int* bs, int* off, int* row_seq, int* tok, int* prev_f, int* prev_r,
                            hipStream_t stream) {
  if (!mask || n <= 0 || L <= 0 || L > MAX_L) return NNR_ERR_ARG;
  const int nseg = (n + 63) / 64;
  bool fast = !perm_in && nseg <= MAX_SEG;
  size_t shm = (size_t)(2 * (L + 1) + (fast ? nseg * (L + 1) : 0)) * sizeof(int);
  if (shm > 160 * 1024) {          // the per-segment histograms do not fit the LDS: the O(n^2) rank path needs only 2 * (L + 1) ints
    fast = false;
    shm = (size_t)(2 * (L + 1)) * sizeof(int);
  }
  hipLaunchKernelGGL(plan_len_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, mask, n, L, len_out);
  NNR_CHECK_LAUNCH();
  if (shm > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(plan_rank_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess)
    return NNR_ERR_LAUNCH;
  hipLaunchKernelGGL(plan_rank_kernel, dim3(1), dim3(PLAN_THREADS), shm, stream, len_out, n, L, perm_in, order, rank, slen, bs, off, fast ? 1 : 0);
  NNR_CHECK_LAUNCH();
  const long total = (long)n * L;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(plan_rows_kernel, dim3(blocks), dim3(256), 0, stream, ids, n, L, order, slen, off, row_seq, tok, prev_f, prev_r);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
