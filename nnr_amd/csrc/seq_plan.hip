// Sequence planner: replaces newsEncoders.py:106-120 (mask fix, lengths, two torch.sort calls, index_select,
// pack_padded_sequence and its sorted_length.cpu() host sync) with ONE single-workgroup kernel whose outputs stay
// in device memory.  Layout produced ("time-major packed", the PackedSequence order):
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

__global__ __launch_bounds__(PLAN_THREADS) void seq_plan_kernel(uint8_t* __restrict__ mask, const int* __restrict__ ids,
                                                                int n, int L, const int* __restrict__ perm_in,
                                                                int* __restrict__ len_out, int* __restrict__ order,
                                                                int* __restrict__ rank, int* __restrict__ slen,
                                                                int* __restrict__ bs, int* __restrict__ off,
                                                                int* __restrict__ row_seq, int* __restrict__ tok,
                                                                int* __restrict__ prev_f, int* __restrict__ prev_r) {
  extern __shared__ int sm[];          // [n] lengths, then [L+1] hist/bs, then [L+1] off
  int* lens = sm;
  int* hist = sm + n;
  int* offs = hist + (L + 1);
  const int tid = threadIdx.x;

  // 1. mask[:,0] = 1 (in place, newsEncoders.py:108-109) and lengths = sum(mask)
  for (int i = tid; i < n; i += PLAN_THREADS) {
    uint8_t* row = mask + (long)i * L;
    row[0] = 1;
    int c = 1;
    for (int t = 1; t < L; ++t) c += row[t] ? 1 : 0;
    lens[i] = c;
    len_out[i] = c;
  }
  for (int t = tid; t <= L; t += PLAN_THREADS) hist[t] = 0;
  __syncthreads();

  // 2. sorted position: descending length, ties by original index (stable), unless the caller supplies the order
  if (perm_in) {
    for (int s = tid; s < n; s += PLAN_THREADS) {
      const int i = perm_in[s];
      order[s] = i;
      rank[i] = s;
      slen[s] = lens[i];
    }
  } else {
    for (int i = tid; i < n; i += PLAN_THREADS) {
      const int li = lens[i];
      int pos = 0;
      for (int j = 0; j < n; ++j) {
        const int lj = lens[j];
        pos += (lj > li) || (lj == li && j < i);
      }
      order[pos] = i;
      rank[i] = pos;
      slen[pos] = li;
    }
  }
  // 3. bs[t] = #(len > t)
  for (int i = tid; i < n; i += PLAN_THREADS) atomicAdd(&hist[lens[i]], 1);   // hist[l] = #(len == l)
  __syncthreads();
  if (tid == 0) {
    int run = 0;                       // sequences with length > t, walking t downward
    for (int t = L; t >= 0; --t) {
      const int h = hist[t];           // #(len == t)
      hist[t] = run;                   // #(len > t)
      run += h;
    }
    int o = 0;
    for (int t = 0; t < L; ++t) { offs[t] = o; o += hist[t]; }
    offs[L] = o;
  }
  __syncthreads();
  for (int t = tid; t < L; t += PLAN_THREADS) bs[t] = hist[t];
  for (int t = tid; t <= L; t += PLAN_THREADS) off[t] = offs[t];
  __syncthreads();   // order[] / slen[] written above by other threads must be visible below
  __threadfence_block();

  // 4. per-row maps
  for (int s = tid; s < n; s += PLAN_THREADS) {
    const int i = order[s];
    const int l = lens[i];
    const int* idrow = ids ? ids + (long)i * L : nullptr;
    for (int t = 0; t < l; ++t) {
      const int row = offs[t] + s;
      row_seq[row] = s;
      if (tok) tok[row] = idrow ? idrow[t] : 0;
      prev_f[row] = (t > 0) ? offs[t - 1] + s : -1;
      prev_r[row] = (t + 1 < l) ? offs[t + 1] + s : -1;
    }
  }
}

}  // namespace

extern "C" int nnr_seq_plan(uint8_t* mask, const int* ids, int n, int L, const int* perm_in, int* len_out, int* order,
                            int* rank, int* slen, int* bs, int* off, int* row_seq, int* tok, int* prev_f, int* prev_r,
                            hipStream_t stream) {
  if (!mask || n <= 0 || L <= 0 || L > MAX_L) return NNR_ERR_ARG;
  const size_t shm = (size_t)(n + 2 * (L + 1)) * sizeof(int);
  if (shm > 64 * 1024) return NNR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(seq_plan_kernel, dim3(1), dim3(PLAN_THREADS), shm, stream, mask, ids, n, L, perm_in, len_out,
                     order, rank, slen, bs, off, row_seq, tok, prev_f, prev_r);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
