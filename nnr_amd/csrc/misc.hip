// HBM-bound elementwise / reduction kernels of the hot path (all float4-vectorised where the layout allows,
// grid capped at 2048 workgroups with grid-stride loops).
#include "common.h"

namespace {

constexpr int EW_BLOCKS = 2048;
__host__ __device__ inline int ew_grid(long n, int per_block) {
  long b = (n + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b > EW_BLOCKS ? EW_BLOCKS : b));
}
__device__ __forceinline__ int dyn_rows(const int* dev, int rows) { return dev ? min(rows, *dev) : rows; }

// ---- cross-selective gate backward (newsEncoders.py:128-131):  Ht = H * G, G = sigmoid(pre)
//      dH = dHt * G ;  dpre = dHt * H * G * (1 - G)
__global__ void gate_bwd_kernel(const float* __restrict__ dHt, const float* __restrict__ H, const float* __restrict__ G,
                                float* __restrict__ dH, float* __restrict__ dpre, const int* rows_dev, int rows, int cols) {
  const long n4 = (long)dyn_rows(rows_dev, rows) * cols / 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 d = reinterpret_cast<const f32x4*>(dHt)[i], h = reinterpret_cast<const f32x4*>(H)[i],
                g = reinterpret_cast<const f32x4*>(G)[i];
    reinterpret_cast<f32x4*>(dH)[i] = d * g;
    reinterpret_cast<f32x4*>(dpre)[i] = d * h * g * (1.f - g);
  }
}

// ---- out[s, :] = sum_{t < slen[s]} x[off[t] + s, :]
// One workgroup per sequence: thread (h, c) = (tid / 128, tid % 128) sums the float4 column c of the rows t = h, h + 2, ... with 8 rows in
// flight (the row address depends on off[t], so a plain loop pays one memory round trip per token), the two halves meet in LDS.  The launch
// is bound by its longest sequence: 128 rows are 8 round trips here (one wave per sequence and a second pass for columns 64..99: 32).
__global__ __launch_bounds__(256) void packed_seq_sum_kernel(const float* __restrict__ x, int D, const int* __restrict__ off,
                                                             const int* __restrict__ slen, int n, float* __restrict__ out) {
  __shared__ f32x4 part[128];
  const int s = blockIdx.x, h = threadIdx.x >> 7, c0 = threadIdx.x & 127;
  const int len = slen[s], nv = D >> 2;
  for (int cb = 0; cb < nv; cb += 128) {
    const int c = cb + c0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (c < nv) {
      int t = h;
      for (; t + 14 < len; t += 16) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(x + ((long)off[t + 2 * u] + s) * D + 4 * c);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; t < len; t += 2) acc += *reinterpret_cast<const f32x4*>(x + ((long)off[t] + s) * D + 4 * c);
    }
    if (cb) __syncthreads();
    if (h == 1) part[c0] = acc;
    __syncthreads();
    if (h == 0 && c < nv) *reinterpret_cast<f32x4*>(out + (long)s * D + 4 * c) = acc + part[c0];
  }
}

// ---- additive-attention score backward (layers.py:168-169): th = tanh(pre) saved; s = w2 . th
//      dpre = ds * w2 * (1 - th^2)  (in place over th) ;  dw2[a] += sum_rows ds * th
//
// Column sums that many workgroups add into ONE short vector: f32 atomics into the same 128-B line retire at ~2.2 ns each in
// L2 whatever the address within the line (measured: 1 600 workgroups x 200 columns = 93 us, 800 x 400 = 66 us, both far above
// their HBM time), so with a workspace the workgroups add into slot (blockIdx.x % NNR_SLOTS) of `ws` [NNR_SLOTS, N] -- 32x
// more lines -- and slot_reduce_kernel folds the slots into the destination and leaves the workspace zeroed for the next call.
// Round 4: the result must also be REPRODUCIBLE (config.py:125-130), and atomics into shared slots add in arrival order.  Now every
// workgroup b of the (at most NNR_SLOTS) workgroups of a launch owns a contiguous range of the LIVE rows and stores its column sums to
// ITS row ws[b][N]; slot_reduce_kernel adds the rows in a fixed tree order (thread (column, s) sums rows s, s + 32, ... in order, then
// the 32 partial sums are added in order) and issues ONE atomic add per column (the destination is a parameter gradient that a
// second launch -- the other encoder call of the plugin API -- may add into: two addends into a zeroed buffer commute).
constexpr int NNR_SLOTS = 1024;
__global__ __launch_bounds__(256) void slot_reduce_kernel(const float* __restrict__ ws, int nb, int N, float* __restrict__ out) {
  __shared__ float part[32][8];
  const int cl = threadIdx.x & 7, sg = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  float acc = 0.f;
  if (c < N)
    for (int b = sg; b < nb; b += 32) acc += ws[(long)b * N + c];
  part[sg][cl] = acc;
  __syncthreads();
  if (sg == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) t += part[q][cl];
    atomicAdd(&out[c], t);
  }
}

__global__ __launch_bounds__(256) void tanh_score_bwd_kernel(float* __restrict__ th, const float* __restrict__ ds,
                                                             const float* __restrict__ w2, float* __restrict__ dw2,
                                                             const int* rows_dev, int rows, int A, int rows_per_block, float* ws) {
  const int R = dyn_rows(rows_dev, rows);
  int r0, r1;
  if (ws) {                                   // reproducible form: gridDim.x workgroups share the LIVE rows, each stores to its own slot row
    const int chunk = (R + (int)gridDim.x - 1) / (int)gridDim.x;
    r0 = min(R, (int)blockIdx.x * chunk);
    r1 = min(R, r0 + chunk);
    dw2 = ws + (long)blockIdx.x * A;
  } else {
    r0 = blockIdx.x * rows_per_block;
    r1 = min(R, r0 + rows_per_block);
  }
  for (int a = threadIdx.x; a < A; a += blockDim.x) {
    const float w = w2[a];
    float acc = 0.f;
    int row = r0;
    for (; row + 8 <= r1; row += 8) {            // 8 rows in flight per thread
      float t[8], d[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { d[u] = ds[row + u]; t[u] = th[(long)(row + u) * A + a]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc += d[u] * t[u];
        th[(long)(row + u) * A + a] = d[u] * w * (1.f - t[u] * t[u]);
      }
    }
    for (; row < r1; ++row) {
      const float d = ds[row];
      const float t = th[(long)row * A + a];
      acc += d * t;
      th[(long)row * A + a] = d * w * (1.f - t * t);
    }
    if (ws) dw2[a] = acc;
    else if (r0 < r1) atomicAdd(&dw2[a], acc);
  }
}

// ---- out[row] = <x[row, :N], w>   (the w2 . tanh(.) score of the additive attention, layers.py:168; one wave per row, float4 lanes)
// The GEMM can fuse this into its epilogue only with a 208-wide tile (whole rows in one workgroup), which costs more than
// it saves: 486 us fused vs 300 us (64 x 80 tile) + 25 us for this HBM-bound pass on the history abstracts.
__global__ __launch_bounds__(256) void rowdot_kernel(const float* __restrict__ x, int ld, const float* __restrict__ w, const int* rows_dev,
                                                     int rows, int N, float* __restrict__ out) {
  const int R = dyn_rows(rows_dev, rows);
  const int lane = threadIdx.x & 63;
  const int nv = N >> 2;
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < R; row += gridDim.x * 4) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (long)row * ld);
    float p = 0.f;
    for (int c = lane; c < nv; c += 64) {
      const f32x4 a = xr[c], b = reinterpret_cast<const f32x4*>(w)[c];
      p += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    }
    p = wave_sum(p);
    if (lane == 0) out[row] = p;
  }
}

// ---- out[c] += sum_rows x[row, c]   (bias gradients)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ld, const int* rows_dev, int rows, int N,
                                                     float* __restrict__ out, int rows_per_block, float* ws) {
  const int R = dyn_rows(rows_dev, rows);
  int r0, r1;
  if (ws) {                                   // reproducible form (see slot_reduce_kernel): own slot row, written even when empty
    const int chunk = (R + (int)gridDim.x - 1) / (int)gridDim.x;
    r0 = min(R, (int)blockIdx.x * chunk);
    r1 = min(R, r0 + chunk);
    out = ws + (long)blockIdx.x * N;
  } else {
    r0 = blockIdx.x * rows_per_block;
    r1 = min(R, r0 + rows_per_block);
    if (r0 >= r1) return;
  }
  for (int c = blockIdx.y * blockDim.x + threadIdx.x; c < N; c += gridDim.y * blockDim.x) {
    float acc = 0.f;
    int row = r0;
    for (; row + 8 <= r1; row += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(long)(row + u) * ld + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; row < r1; ++row) acc += x[(long)row * ld + c];
    if (ws) out[c] = acc;
    else atomicAdd(&out[c], acc);
  }
}

// ---- small embedding tables (category / subCategory, newsEncoders.py:51-53): out[i, :dim] = drop(table[idx[i]])
__global__ void small_embed_kernel(const float* __restrict__ table, const int* __restrict__ idx, int n, int dim,
                                   float* __restrict__ out, int ldo, float* __restrict__ dtable,
                                   const float* __restrict__ dout, int lddo, uint32_t seed, uint32_t thr, float scale) {
  const long total = (long)n * dim;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = i / dim, c = i - (long)row * dim;
    const float m = nnr_keep(seed, (uint64_t)i, thr) ? scale : 0.f;
    if (out) out[(long)row * ldo + c] = table[(long)idx[row] * dim + c] * m;
    if (dtable) atomicAdd(&dtable[(long)idx[row] * dim + c], dout[(long)row * lddo + c] * m);
  }
}

// backward of the same: dtable[idx[row], c] += mask * dout[row, c].  The tables are tiny and the ids far from uniform (every padded
// history slot is news 0 -> category 0: half of the rows of a batch hit the same 50 addresses), and same-address f32 atomics
// serialise in L2 at ~60 ns each: one atomic per element took 97 us for 3 200 rows.  One wave walks `rpw` CONSECUTIVE rows
// (lane = column), merges runs of equal ids in a register -- padded slots are the contiguous tail of each history -- and emits
// one atomic per run.
__global__ __launch_bounds__(256) void small_embed_bwd_kernel(const int* __restrict__ idx, int n, int dim, const float* __restrict__ dout,
                                                              int lddo, float* __restrict__ dtable, uint32_t seed, uint32_t thr,
                                                              float scale, int rpw) {
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw, r1 = min(n, r0 + rpw);
  for (int c = lane; c < dim; c += 64) {
    int cur = -1;
    float acc = 0.f;
#pragma unroll 4
    for (int row = r0; row < r1; ++row) {
      const int id = idx[row];
      const float m = nnr_keep(seed, (uint64_t)((long)row * dim + c), thr) ? scale : 0.f;
      const float v = dout[(long)row * lddo + c] * m;
      if (id != cur) {
        if (cur >= 0) atomicAdd(&dtable[(long)cur * dim + c], acc);
        cur = id;
        acc = 0.f;
      }
      acc += v;
    }
    if (cur >= 0) atomicAdd(&dtable[(long)cur * dim + c], acc);
  }
}

// ---- embedding-row gather (nn.Embedding forward, newsEncoders.py:117-118,163,193) with fused dropout.
// One wave per row: the index is wave-uniform, the row is read as contiguous 16-byte lanes (a 300-float row = 75 float4 =
// two fully coalesced 1 KiB / 176 B wave accesses).  Pure HBM/L2 streaming: out bytes written once, table rows read once.
__global__ __launch_bounds__(256) void embed_gather_kernel(const float* __restrict__ table, const int* __restrict__ idx, long n,
                                                           const int* __restrict__ n_dev, int dim, float* __restrict__ out, uint32_t seed,
                                                           uint32_t thr, float scale) {
  // FOUR rows per wave and trip (round 3): a row costs two dependent memory round trips (its index, then the table row), and with one
  // row per trip a wave had a single one in flight -- 1.75 TB/s on the 84 k-row content stream of a batch-64 step, at the head of the
  // step's dependent chain.  The four indices, then the four rows' segments are loaded before any of them is used.
  constexpr int UR = 4;
  const int lane = threadIdx.x & 63;
  const int nv = dim >> 2;
  if (n_dev) n = min(n, (long)*n_dev);                 // live row count kept on the device (packed token streams)
  for (long row0 = (blockIdx.x * 4L + (threadIdx.x >> 6)) * UR; row0 < n; row0 += gridDim.x * 4L * UR) {
    int src[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) src[u] = (row0 + u < n) ? idx[row0 + u] : -1;
    for (int c = lane; c < nv; c += 64) {
      f32x4 v[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u)
        v[u] = (src[u] >= 0) ? reinterpret_cast<const f32x4*>(table + (long)src[u] * dim)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        if (row0 + u >= n) break;
        if (thr) {
          bool kp[4];
          nnr_keep4(seed, (uint64_t)(row0 + u) * dim + 4 * c, thr, kp);       // dim % 4 == 0 is checked by the launcher
#pragma unroll
          for (int e = 0; e < 4; ++e) v[u][e] = kp[e] ? v[u][e] * scale : 0.f;
        }
        __builtin_nontemporal_store(v[u], reinterpret_cast<f32x4*>(out + (row0 + u) * dim) + c);
      }
    }
  }
}
// backward: dtable[idx[row], :] += mask * dout[row, :]  -- f32 atomics, each wave-instruction = 256 contiguous bytes of one row.
// Global float atomics into ONE row run ~14x slower than spread ones (MI355X_MICROARCH.md), and word ids are far from uniform:
// dense id tensors are mostly <PAD> (id 0) past each title's length, and the vocabulary is frequency-ordered (MIND_corpus.py:
// rows >= 2 by descending frequency), so the hot destinations are the SMALL ids.  Rows whose id is < HOT are accumulated in an
// LDS table per workgroup (ds_add_f32) and leave the block as one atomic row per hot id; the workgroups are persistent
// (grid-stride) so there are few such flushes.
constexpr int SC_HOT = 32, SC_MAXD = 320;
__global__ __launch_bounds__(256) void embed_scatter_kernel(const float* __restrict__ dout, const int* __restrict__ idx, long n,
                                                            const int* __restrict__ n_dev, int dim,
                                                            float* __restrict__ dtable, uint32_t seed, uint32_t thr, float scale) {
  __shared__ float hot[SC_HOT * SC_MAXD];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool use_hot = dim <= SC_MAXD;
  if (n_dev) n = min(n, (long)*n_dev);                 // live row count kept on the device (packed token streams)
  if (use_hot) for (int i = threadIdx.x; i < SC_HOT * dim; i += 256) hot[i] = 0.f;
  __syncthreads();
  // (four rows in flight per wave, as in embed_gather_kernel, measured 2x SLOWER here -- 245 -> 494 us on the content stream: the
  // atomics of four rows issued back to back queue up behind each other; one row per trip it stays)
  for (long row = blockIdx.x * 4L + wv; row < n; row += gridDim.x * 4L) {
    const int dst = idx[row];
    if (dst < 0) continue;
    const bool h = use_hot && dst < SC_HOT;
    for (int c = lane; c < dim; c += 64) {
      float v = dout[row * dim + c];
      if (thr) v = nnr_keep(seed, (uint64_t)row * dim + c, thr) ? v * scale : 0.f;
      if (h) atomicAdd(&hot[dst * dim + c], v);
      else atomicAdd(&dtable[(long)dst * dim + c], v);
    }
  }
  __syncthreads();
  if (use_hot)
    for (int i = threadIdx.x; i < SC_HOT * dim; i += 256) {
      const float v = hot[i];
      if (v != 0.f) atomicAdd(&dtable[i], v);      // hot rows are rows 0..HOT-1 of the table: same flat offset
    }
}
// ---- out[c, r] = in[r, c]  (weight re-layouts, e.g. Conv1d [C_out*C_in, k] -> [k, C_out*C_in])
__global__ void transpose2d_kernel(const float* __restrict__ in, float* __restrict__ out, long rows, int cols, int accumulate) {
  const long total = rows * cols;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long c = i / rows, r = i - c * rows;             // consecutive threads -> consecutive r : coalesced writes
    const float v = in[r * cols + c];
    if (accumulate) atomicAdd(&out[i], v);                 // parameter gradients: two HIP streams may add concurrently
    else out[i] = v;
  }
}

// ---- several transposes in one launch (the W^T copies of every weight the backward pass multiplies by: ~20 small matrices
// per step; blockIdx.y selects the matrix, tiles of 32 x 32 go through LDS so that reads and writes are both coalesced)
__global__ __launch_bounds__(256) void transpose_batch_kernel(const nnr_transpose_desc* __restrict__ descs) {
  __shared__ float tile[32][33];
  const nnr_transpose_desc d = descs[blockIdx.y];
  const int tr = (d.rows + 31) / 32, tc = (d.cols + 31) / 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8 threads
  for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
    const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
    for (int j = ty; j < 32; j += 8) {
      const int r = r0 + j, c = c0 + tx;
      tile[j][tx] = (r < d.rows && c < d.cols) ? d.in[(long)r * d.cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
      const int c = c0 + j, r = r0 + tx;
      if (r < d.rows && c < d.cols) d.out[(long)c * d.rows + r] = tile[tx][j];
    }
    __syncthreads();
  }
}

// ---- generic y (op)= x
__global__ void add_kernel(float* __restrict__ y, const float* __restrict__ x, long n, float alpha) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += alpha * x[i];
}
// same with f32 atomics: for accumulators that two HIP streams add into concurrently (parameter gradients)
__global__ void add_atomic_kernel(float* __restrict__ y, const float* __restrict__ x, long n, float alpha) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) atomicAdd(&y[i], alpha * x[i]);
}
__global__ void add2d_kernel(float* __restrict__ y, int ldy, const float* __restrict__ x, int ldx, int rows, int cols,
                             float alpha, int accumulate) {
  const long total = (long)rows * cols;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = i / cols, c = i - (long)r * cols;
    const float v = alpha * x[(long)r * ldx + c];
    float* p = y + (long)r * ldy + c;
    *p = accumulate ? *p + v : v;
  }
}
// ---- [B, D] -> [B, N, D] (the user vector repeated over the N candidates, userEncoders.py:172,190) and its backward, the sum over the N copies in
//      ascending candidate order; one launch each (the add2d form took N launches per direction on the dependent chain of the MHSA step)
__global__ void expand_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int N, int D) {
  const long total = (long)B * N * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / D;
    const int c = (int)(i - row * D);
    y[i] = x[(row / N) * D + c];
  }
}
__global__ void expand_rows_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int N, int D) {
  const long total = (long)B * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / D;
    const int c = (int)(i - b * D);
    const float* p = dy + b * N * D + c;
    float acc = p[0];
    for (int j = 1; j < N; ++j) acc += p[(long)j * D];
    dx[i] = acc;
  }
}

// ---- SUE graph input (userEncoders.py:80): X0[b, :Hn] = hist[b] ; X0[b, Hn + k] = dropout_(proxy[k])  (mask per sample)
//      backward: dhist = dX0[:, :Hn] ; dproxy[k] += sum_b mask * dX0[b, Hn + k]
//      cmask_fix (forward, optional): the [B, Kc + 1] cluster mask whose last column the reference sets in place (userEncoders.py:73);
//      dx0_add (backward, optional): a second addend of the upstream gradient (the outer residual gcn(X0) + X0, :81)
__global__ void sue_x0_kernel(const float* __restrict__ hist, const float* __restrict__ proxy, float* __restrict__ x0, int B,
                              int Hn, int Kc, int D, const float* __restrict__ dx0, const float* __restrict__ dx0_add,
                              float* __restrict__ dhist, float* __restrict__ dproxy, uint8_t* __restrict__ cmask_fix, uint32_t seed,
                              uint32_t thr, float scale) {
  const int G = Hn + Kc;
  const long total = (long)B * G * D;
  if (cmask_fix)
    for (long b = blockIdx.x * (long)blockDim.x + threadIdx.x; b < B; b += (long)gridDim.x * blockDim.x) cmask_fix[b * (Kc + 1) + Kc] = 1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % D; const long q = i / D; const int j = q % G; const int b = q / G;
    const float up = dx0 ? (dx0_add ? dx0[i] + dx0_add[i] : dx0[i]) : 0.f;
    if (j < Hn) {
      if (x0) x0[i] = hist[((long)b * Hn + j) * D + c];
      if (dhist) dhist[((long)b * Hn + j) * D + c] = up;
    } else {
      const int k = j - Hn;
      if (x0) {
        const float m = nnr_keep(seed, (uint64_t)((long)b * Kc + k) * D + c, thr) ? scale : 0.f;
        x0[i] = proxy[(long)k * D + c] * m;
      }
    }
  }
  if (dproxy) {
    // dproxy[k, c] += sum_b mask * dX0[b, Hn + k, c]: one thread per (k, c) adds the B samples IN ORDER (reproducible; f32 atomics from
    // B x Kc x D threads added in arrival order) and issues one atomic add (the destination is a parameter gradient)
    for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < (long)Kc * D; t += (long)gridDim.x * blockDim.x) {
      const int c = t % D, k = t / D;
      float acc = 0.f;
      for (int b = 0; b < B; ++b) {
        const long i = ((long)b * G + Hn + k) * D + c;
        const float up = dx0_add ? dx0[i] + dx0_add[i] : dx0[i];
        const float m = nnr_keep(seed, (uint64_t)((long)b * Kc + k) * D + c, thr) ? scale : 0.f;
        acc += up * m;
      }
      atomicAdd(&dproxy[t], acc);
    }
  }
}

// ---- gfeat[b, j, :] = gcn[b, j, :] + x0[b, j, :], j < Hn (userEncoders.py:81-82); backward scatters dgfeat back
//      into the padded [B, G, D] gradient (proxy rows get zero) -- used for both addends.
__global__ void sue_slice_kernel(const float* __restrict__ gcn, const float* __restrict__ x0, float* __restrict__ gfeat,
                                 const float* __restrict__ dgfeat, float* __restrict__ dpad, int B, int Hn, int G, int D) {
  const long total = (long)B * G * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = i % D; const long q = i / D; const int j = q % G; const int b = q / G;
    if (gfeat) { if (j < Hn) gfeat[((long)b * Hn + j) * D + c] = gcn[i] + x0[i]; }
    if (dpad) dpad[i] = (j < Hn) ? dgfeat[((long)b * Hn + j) * D + c] : 0.f;
  }
}

// ---- backward of  y = dropout(relu(z) + x)  given r = relu(z):  dym = mask(dy) ; ds = dym * (r > 0) ; dx = dym
__global__ void relu_drop_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ r, float* __restrict__ ds,
                                     float* __restrict__ dx, long n, int cols, uint32_t seed, uint32_t thr, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float m = nnr_keep(seed, (uint64_t)i, thr) ? scale : 0.f;
    const float g = dy[i] * m;
    ds[i] = (r[i] > 0.f) ? g : 0.f;
    if (dx) dx[i] = g;
  }
}

// ---- standalone dropout  y = mask(x)   (same mask forward / backward)
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, uint32_t seed, uint32_t thr,
                               float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = nnr_keep(seed, (uint64_t)i, thr) ? x[i] * scale : 0.f;
}

// ---- relu backward in place-free form: dx = dy * (y > 0)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = (y[i] > 0.f) ? dy[i] : 0.f;
}

// ---- SUE intra-cluster attention (userEncoders.py:85-89; replaces torch_scatter.scatter_softmax / scatter_sum).
// One workgroup per (sample b, candidate n).  scores over the Hn history items, softmax WITHIN each cluster id
// (segments held in LDS), cluster-wise weighted sum of the [Hn, D] features -> [C, D]; empty clusters give 0.
constexpr int SUE_MAXH = 64, SUE_MAXC = 32;
// cluster member lists of one sample in LDS: order[cstart[c] .. cstart[c] + ccnt[c]) = the items of cluster c, ascending
__device__ __forceinline__ void sue_members(const int* cid, int Hn, int C, int* ccnt, int* cstart, int* order) {
  const int tid = threadIdx.x;
  if (tid < C) {
    int n = 0;
    for (int j = 0; j < Hn; ++j) n += (cid[j] == tid);
    ccnt[tid] = n;
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int c = 0; c < C; ++c) { cstart[c] = acc; acc += ccnt[c]; }
  }
  __syncthreads();
  if (tid < C) {
    int k = cstart[tid];
    for (int j = 0; j < Hn; ++j) if (cid[j] == tid) order[k++] = j;
  }
  __syncthreads();
}

// forward: one workgroup per (b, n, 256-column slice) -- every workgroup recomputes the (tiny) scores and segment softmax of its (b, n) and
// produces one slice of the C cluster features, walking each cluster's member list once.
// Round 6: the N candidates of a user all read the user's g [Hn, D] (180 KB; the launch's algorithmic bytes count it ONCE), and consecutive
// workgroup ids go to different XCDs, whose L2s do not share: as a (B N, slices) grid the five readers of a g slice sat in five L2s and the counter
// traffic was 2.6x the algorithmic bytes.  The grid is 1-D now and workgroup id -> (XCD x = id & 7, slot = id >> 3) -> user b = 8 (slot / (N S)) + x:
// every workgroup of a user runs on ONE XCD, the N readers of a slice back to back (n fastest).
__device__ __forceinline__ bool sue_xcd_map(int id, int B, int per_user, int* b, int* w) {
  const int x = id & 7, slot = id >> 3;
  *b = (slot / per_user) * 8 + x;
  *w = slot % per_user;
  return *b < B;
}
__global__ __launch_bounds__(256) void sue_intra_fwd_kernel(const float* __restrict__ kf, const float* __restrict__ qc,
                                                            const float* __restrict__ g, const long* __restrict__ cidx,
                                                            int B, int N, int Hn, int C, int A, int D, float inv_scale,
                                                            float* __restrict__ alpha, float* __restrict__ feat) {
  __shared__ float sc[SUE_MAXH], al[SUE_MAXH], cmax[SUE_MAXC], csum[SUE_MAXC];
  __shared__ int cid[SUE_MAXH], order[SUE_MAXH], ccnt[SUE_MAXC], cstart[SUE_MAXC];
  int b, wi;
  if (!sue_xcd_map(blockIdx.x, B, N * ((D + 255) / 256), &b, &wi)) return;
  const int slice = wi / N, bn = b * N + (wi - slice * N);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int j = tid; j < Hn; j += 256) cid[j] = (int)cidx[(long)b * Hn + j];
  const float* q = qc + (long)bn * A;
  for (int j0 = w; j0 < Hn; j0 += 16) {                     // 4 items per wave and trip: their loads are in flight together
    float p[4] = {0.f, 0.f, 0.f, 0.f};
    const float* k[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) k[u] = kf + ((long)b * Hn + min(j0 + 4 * u, Hn - 1)) * A;
    for (int x = lane; x < A; x += 64) {
      const float qv = q[x];
      float kv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) kv[u] = k[u][x];
#pragma unroll
      for (int u = 0; u < 4; ++u) p[u] += kv[u] * qv;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float t = wave_sum(p[u]);
      if (lane == 0 && j0 + 4 * u < Hn) sc[j0 + 4 * u] = t * inv_scale;
    }
  }
  __syncthreads();
  sue_members(cid, Hn, C, ccnt, cstart, order);
  if (tid < C) {
    float m = -INFINITY;
    for (int k = 0; k < ccnt[tid]; ++k) m = fmaxf(m, sc[order[cstart[tid] + k]]);
    float sm = 0.f;
    for (int k = 0; k < ccnt[tid]; ++k) sm += expf(sc[order[cstart[tid] + k]] - m);
    cmax[tid] = m; csum[tid] = sm;
  }
  __syncthreads();
  for (int j = tid; j < Hn; j += 256) {
    const float v = expf(sc[j] - cmax[cid[j]]) / csum[cid[j]];
    al[j] = v;
    if (slice == 0) alpha[(long)bn * Hn + j] = v;
  }
  __syncthreads();
  const int col = slice * 256 + tid;
  if (col >= D) return;
  const float* gb = g + (long)b * Hn * D + col;
  float* fo = feat + (long)bn * C * D + col;
  // One walk over the cluster-sorted member list, 8 loads in flight per trip (cluster by cluster the walk was up to Hn dependent
  // loads: clusters hold 2-3 members on average).  Every cluster still sums its members in ascending order.
  for (int c = 0; c < C; ++c)
    if (ccnt[c] == 0) fo[(long)c * D] = 0.f;            // empty cluster -> 0, as scatter_sum
  const int total = cstart[C - 1] + ccnt[C - 1];
  if (total == 0) return;
  int cur = cid[order[0]];
  float acc = 0.f;
  for (int p0 = 0; p0 < total; p0 += 8) {
    int jj[8];
    float gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      jj[u] = order[min(p0 + u, total - 1)];
      gv[u] = gb[(long)jj[u] * D];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (p0 + u < total) {
        const int cl = cid[jj[u]];
        if (cl != cur) {
          fo[(long)cur * D] = acc;
          acc = 0.f;
          cur = cl;
        }
        acc += al[jj[u]] * gv[u];
      }
    }
  }
  fo[(long)cur * D] = acc;
}

// backward part 1, grid B*N: d alpha, segment-softmax backward -> ds[b, n, :] (workspace), dqc[b, n, :].
// (B*N = 320 workgroups of 4 waves at batch 64.  16 waves per workgroup take the Hn items in one trip and are no faster alone
// (30 vs 27 us) but 126 vs 82 us inside the step: a 1024-thread workgroup needs a whole CU at once, and the CUs are shared with the
// weight-gradient GEMM of the other stream.  The kernel is written for any multiple of 64 threads.)
__global__ __launch_bounds__(256) void sue_intra_bwd_ds_kernel(const float* __restrict__ kf, const float* __restrict__ g,
                                                                const long* __restrict__ cidx, const float* __restrict__ alpha,
                                                                const float* __restrict__ dfeat, int B, int N, int Hn, int C, int A, int D,
                                                                float inv_scale, float* __restrict__ ds_ws, float* __restrict__ dqc) {
  __shared__ float al[SUE_MAXH], da[SUE_MAXH], ds[SUE_MAXH], csum[SUE_MAXC];
  __shared__ int cid[SUE_MAXH];
  int b, ni;
  if (!sue_xcd_map(blockIdx.x, B, N, &b, &ni)) return;             // (the N workgroups of a user read the same g[b]: one XCD, see the forward)
  const int bn = b * N + ni, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int nt = blockDim.x, nw = nt >> 6;
  for (int j = tid; j < Hn; j += nt) { cid[j] = (int)cidx[(long)b * Hn + j]; al[j] = alpha[(long)bn * Hn + j]; }
  __syncthreads();
  const float* gb = g + (long)b * Hn * D;
  const float* df = dfeat + (long)bn * C * D;
  // dalpha_j = <dfeat[cid_j], g_j>
  if (!(D & 3)) {
    // 4 items per wave at a time, float4 lanes, two column trips per pass: 16 independent 16-byte loads in flight, 2 passes at D = 900
    const int D4 = D >> 2;
    for (int j0 = w; j0 < Hn; j0 += 4 * nw) {
      float p[4] = {0.f, 0.f, 0.f, 0.f};
      const f32x4* dr[4];
      const f32x4* gr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(j0 + nw * u, Hn - 1);
        dr[u] = reinterpret_cast<const f32x4*>(df + (long)cid[j] * D);
        gr[u] = reinterpret_cast<const f32x4*>(gb + (long)j * D);
      }
      for (int x = lane; x < D4; x += 128) {
        const int x2 = x + 64;
        const bool two = x2 < D4;
        f32x4 a[4], c[4], a2[4], c2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = dr[u][x]; c[u] = gr[u][x]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a2[u] = f32x4{0.f, 0.f, 0.f, 0.f};
          c2[u] = a2[u];
          if (two) { a2[u] = dr[u][x2]; c2[u] = gr[u][x2]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] += a[u][0] * c[u][0] + a[u][1] * c[u][1] + a[u][2] * c[u][2] + a[u][3] * c[u][3];
        if (two) {
#pragma unroll
          for (int u = 0; u < 4; ++u) p[u] += a2[u][0] * c2[u][0] + a2[u][1] * c2[u][1] + a2[u][2] * c2[u][2] + a2[u][3] * c2[u][3];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float t = wave_sum(p[u]);
        if (lane == 0 && j0 + nw * u < Hn) da[j0 + nw * u] = t;
      }
    }
  } else {
    for (int j = w; j < Hn; j += nw) {
      const float* dr = df + (long)cid[j] * D;
      const float* gr = gb + (long)j * D;
      float p = 0.f;
      for (int x = lane; x < D; x += 64) p += dr[x] * gr[x];
      p = wave_sum(p);
      if (lane == 0) da[j] = p;
    }
  }
  __syncthreads();
  if (tid < C) {
    float sm = 0.f;
    for (int j = 0; j < Hn; ++j) if (cid[j] == tid) sm += al[j] * da[j];
    csum[tid] = sm;
  }
  __syncthreads();
  for (int j = tid; j < Hn; j += nt) {
    const float v = al[j] * (da[j] - csum[cid[j]]) * inv_scale;
    ds[j] = v;
    ds_ws[(long)bn * Hn + j] = v;
  }
  __syncthreads();
  // dqc[b, n, :] = sum_j ds[j] * kf[b, j, :]
  for (int x = tid; x < A; x += nt) {
    float acc = 0.f;
    int j = 0;
    for (; j + 8 <= Hn; j += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = kf[((long)b * Hn + j + u) * A + x];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += ds[j + u] * v[u];
    }
    for (; j < Hn; ++j) acc += ds[j] * kf[((long)b * Hn + j) * A + x];
    dqc[(long)bn * A + x] = acc;
  }
}

// backward part 2, grid (B, ceil(D/256) + 1): slices y < last -> dg[b, j, cols] = sum_n alpha[b,n,j] * dfeat[b,n,cid_j,cols]
// (each row written exactly once, no read-modify-write); the last slice -> dkf[b, j, :] = sum_n ds[b,n,j] * qc[b,n,:]
__global__ __launch_bounds__(256) void sue_intra_bwd_dg_kernel(const float* __restrict__ qc, const long* __restrict__ cidx,
                                                               const float* __restrict__ alpha, const float* __restrict__ dfeat,
                                                               const float* __restrict__ ds_ws, int N, int Hn, int C, int A, int D,
                                                               float* __restrict__ dg, float* __restrict__ dkf) {
  __shared__ float al[8][SUE_MAXH];
  __shared__ int cid[SUE_MAXH], order[SUE_MAXH], ccnt[SUE_MAXC], cstart[SUE_MAXC];
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool last = blockIdx.y == gridDim.y - 1;
  for (int i = tid; i < N * Hn; i += 256) {
    const int n = i / Hn, j = i - n * Hn;
    al[n][j] = last ? ds_ws[((long)b * N + n) * Hn + j] : alpha[((long)b * N + n) * Hn + j];
  }
  for (int j = tid; j < Hn; j += 256) cid[j] = (int)cidx[(long)b * Hn + j];
  __syncthreads();
  if (last) {
    if (blockIdx.z) return;
    for (int x = tid; x < A; x += 256) {
      float qv[8];
#pragma unroll
      for (int n = 0; n < 8; ++n) qv[n] = n < N ? qc[((long)b * N + n) * A + x] : 0.f;
      for (int j = 0; j < Hn; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < 8; ++n) if (n < N) acc += al[n][j] * qv[n];      // rows n >= N of `al` are never written (stale LDS may hold NaN)
        dkf[((long)b * Hn + j) * A + x] = acc;
      }
    }
    return;
  }
  sue_members(cid, Hn, C, ccnt, cstart, order);
  const int col = blockIdx.y * 256 + tid;
  if (col >= D) return;
  // blockIdx.z deals the clusters (c = z, z + Z, ...): B * slices workgroups alone are 1.25 waves per SIMD at batch 64, and their C
  // dependent load groups were pure latency.  The next cluster's d feat values are loaded before this cluster's rows are stored.
  const int Z = gridDim.z;
  int c = blockIdx.z;
  while (c < C && ccnt[c] == 0) c += Z;
  float nx[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) nx[n] = (n < N && c < C) ? dfeat[(((long)b * N + n) * C + c) * D + col] : 0.f;
  while (c < C) {
    float dv[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) dv[n] = nx[n];
    int cn = c + Z;
    while (cn < C && ccnt[cn] == 0) cn += Z;
#pragma unroll
    for (int n = 0; n < 8; ++n) nx[n] = (n < N && cn < C) ? dfeat[(((long)b * N + n) * C + cn) * D + col] : 0.f;
    for (int k = 0; k < ccnt[c]; ++k) {
      const int j = order[cstart[c] + k];
      float acc = 0.f;
#pragma unroll
      for (int n = 0; n < 8; ++n) if (n < N) acc += al[n][j] * dv[n];
      dg[((long)b * Hn + j) * D + col] = acc;
    }
    c = cn;
  }
}

// ---- click predictor + loss (model.py:126-127, trainer.py:64-66), one workgroup for the whole (tiny) batch tail.
// logits[b, n] = <user[b, n], cand[b, n]> ; loss = mean_b( -log_softmax(logits[b])[0] )
// dlogits[b, n] = (softmax(logits[b])[n] - [n == 0]) * loss_scale / B   (loss_scale = 1, or 1 for DP: grads are averaged later)
__global__ __launch_bounds__(256) void logits_kernel(const float* __restrict__ user, const float* __restrict__ cand, int BN, int D,
                                                     float* __restrict__ logits) {
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= BN) return;
  float p = 0.f;
  for (int x = lane; x < D; x += 64) p += user[(long)i * D + x] * cand[(long)i * D + x];
  p = wave_sum(p);
  if (lane == 0) logits[i] = p;
}
__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ logits, int B, int N, float* __restrict__ loss,
                                                   float* __restrict__ dlogits) {
  __shared__ float part[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* l = logits + (long)b * N;
    float m = -INFINITY;
    for (int n = 0; n < N; ++n) m = fmaxf(m, l[n]);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += expf(l[n] - m);
    const float lse = m + logf(s);
    acc += lse - l[0];
    if (dlogits)
      for (int n = 0; n < N; ++n) dlogits[(long)b * N + n] = (expf(l[n] - lse) - (n == 0 ? 1.f : 0.f)) / (float)B;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = part[0] / (float)B;
}
// duser = dlogit * cand ; dcand (+)= dlogit * user
__global__ void logits_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ user, const float* __restrict__ cand,
                                  long BN, int D, float* __restrict__ duser, float* __restrict__ dcand, int dcand_acc) {
  const long total = BN * D;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float d = dlogits[i / D];
    duser[i] = d * cand[i];
    const float v = d * user[i];
    dcand[i] = dcand_acc ? dcand[i] + v : v;
  }
}

// ---- optimiser (trainer.py:118-120): global L2 norm -> clip coefficient -> Adam (torch.optim.Adam defaults), one flat buffer
// The norm must be a DETERMINISTIC function of the gradient: under data parallelism every rank computes it from the same all-reduced buffer,
// and a clip coefficient that differs in the last bit (f32 atomics add the block sums in arrival order) lets the ranks' parameters drift apart
// (tools/dp_two_rank_check.py).  Block sums go to fixed slots; the last block to arrive adds the slots in a fixed order.
constexpr int SUMSQ_BLOCKS = 1024;
constexpr int SUMSQ_SLOTS = 4;               // independent scratch sets: launches on DIFFERENT streams use different slots (nnr_sumsq_part)
__device__ float g_sumsq_part[SUMSQ_SLOTS][SUMSQ_BLOCKS];
__device__ unsigned g_sumsq_arrived[SUMSQ_SLOTS];
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out, const float* __restrict__ add_in, int slot) {
  __shared__ float part[4];
  __shared__ bool last;
  float acc = 0.f;
  const long n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; acc += v * v; }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    g_sumsq_part[slot][blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
    __threadfence();
    last = atomicAdd(&g_sumsq_arrived[slot], 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float t = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += __builtin_nontemporal_load(&g_sumsq_part[slot][i]);
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float own = (part[0] + part[1]) + (part[2] + part[3]);
    *out = add_in ? own + *add_in : own;                    // (round 3: STORED, not accumulated; round 5: + the partial sum of another span, fixed order)
    g_sumsq_arrived[slot] = 0;                              // ready for the next launch on this slot (launches that share a slot are stream-ordered)
  }
}
__device__ unsigned g_adam_skipped;          // optimizer steps skipped because the gradient norm was not finite (nnr_adam_skipped_steps)
__device__ unsigned* g_adam_skipped_mirror;  // host-pinned, device-mapped copy of the count: read by nnr_adam_skipped_peek WITHOUT a synchronisation
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, const float* __restrict__ sumsq, float grad_scale, float clip, float lr, float b1, float b2,
                            float eps, float wd, float bc1, float bc2_sqrt) {
  // clip_grad_norm_: coef = clip / (norm + 1e-6), applied only when < 1 (torch clamps the coefficient to 1)
  // a gradient whose norm is not finite (overflow, or the NaN poison of a timed-out recurrence exchange, lstm.hip) must not
  // reach the parameters or the moments: the step is skipped as a whole (torch's clip_grad_norm_ + Adam would write NaN into
  // all three, permanently)
  if (!isfinite(*sumsq)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      const unsigned c = atomicAdd(&g_adam_skipped, 1u) + 1u;
      unsigned* mirror = g_adam_skipped_mirror;
      if (mirror) __hip_atomic_store(mirror, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  float coef = grad_scale;
  if (clip > 0.f) {
    const float norm = sqrtf(*sumsq) * grad_scale;
    const float c = clip / (norm + 1e-6f);
    if (c < 1.f) coef *= c;
  }
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * coef;
    if (wd != 0.f) gi += wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}


// ---- LayerNorm over the last dimension for --gcn_layer_norm (layers.py:273-274,287-288: nn.LayerNorm([out_dim]) between the
// graph convolution and the ReLU), fused with what follows it in GCNLayer.forward / GCN.forward:
//   y = dropout(relu(LN(u) * gamma + beta) + resid).   One wave per row (D <= 1280), the row lives in registers; biased variance,
//   1 / sqrt(var + eps) exactly as ATen.  Saved for backward: xhat = (u - mean) * rstd, rstd, and r = relu(.) (as the GEMM
//   epilogue of the LayerNorm-free path saves it).
constexpr int LN_MAXPL = 20;      // elements per lane (D <= 1280: --hidden_dim 256 gives D = 1124)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ u, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, long rows, int D, float* __restrict__ xhat, float* __restrict__ rstd_out,
                                                     float* __restrict__ r_out, const float* __restrict__ resid, float* __restrict__ y,
                                                     uint32_t seed, uint32_t thr, float scale) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = u + row * D;
  float v[LN_MAXPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < D ? x[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    const int c = lane + 64 * i;
    const float d = c < D ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) rstd_out[row] = rstd;
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    const int c = lane + 64 * i;
    if (c < D) {
      const float xh = (v[i] - mean) * rstd;
      const float r = fmaxf(xh * gamma[c] + beta[c], 0.f);
      xhat[row * D + c] = xh;
      r_out[row * D + c] = r;
      float o = r + (resid ? resid[row * D + c] : 0.f);
      if (thr) o = nnr_keep(seed, (uint64_t)(row * D + c), thr) ? o * scale : 0.f;
      y[row * D + c] = o;
    }
  }
}

// du = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dv * gamma;   dgamma += sum_rows dv * xhat;  dbeta += sum_rows dv
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dv, const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, long rows, int D, float* __restrict__ du,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[2][LN_MAXPL * 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * LN_MAXPL * 64; i += 256) (&red[0][0])[i] = 0.f;
  __syncthreads();
  float ag[LN_MAXPL], ab[LN_MAXPL];
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) ag[i] = ab[i] = 0.f;
  for (int k = 0; k < 4; ++k) {                        // 16 rows per workgroup: 4 per wave
    const long row = (long)blockIdx.x * 16 + w * 4 + k;
    if (row >= rows) break;
    float g[LN_MAXPL], xh[LN_MAXPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXPL; ++i) {
      const int c = lane + 64 * i;
      const float d = c < D ? dv[row * D + c] : 0.f;
      xh[i] = c < D ? xhat[row * D + c] : 0.f;
      g[i] = c < D ? d * gamma[c] : 0.f;
      s1 += g[i];
      s2 += g[i] * xh[i];
      ag[i] += d * xh[i];
      ab[i] += d;
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
    const float rs = rstd[row];
#pragma unroll
    for (int i = 0; i < LN_MAXPL; ++i) {
      const int c = lane + 64 * i;
      if (c < D) du[row * D + c] = rs * (g[i] - s1 - xh[i] * s2);
    }
  }
#pragma unroll
  for (int i = 0; i < LN_MAXPL; ++i) {
    atomicAdd(&red[0][lane + 64 * i], ag[i]);
    atomicAdd(&red[1][lane + 64 * i], ab[i]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    atomicAdd(&dgamma[c], red[0][c]);
    atomicAdd(&dbeta[c], red[1][c]);
  }
}

}  // namespace

#define EW_LAUNCH(kern, n, ...)                                                             \
  do {                                                                                      \
    hipLaunchKernelGGL(kern, dim3(ew_grid((n), 256)), dim3(256), 0, stream, __VA_ARGS__);   \
    NNR_CHECK_LAUNCH();                                                                     \
    return NNR_OK;                                                                          \
  } while (0)

extern "C" int nnr_version(void) { return 1; }

extern "C" int nnr_gate_bwd(const float* dHt, const float* H, const float* G, float* dH, float* dpre, const int* rows_dev,
                            int rows, int cols, hipStream_t stream) {
  if (cols & 3) return NNR_ERR_UNSUPPORTED;
  EW_LAUNCH(gate_bwd_kernel, (long)rows * cols / 4, dHt, H, G, dH, dpre, rows_dev, rows, cols);
}

extern "C" int nnr_packed_seq_sum(const float* x, int D, const int* off, const int* slen, int n, float* out, hipStream_t stream) {
  if (D & 3) return NNR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(packed_seq_sum_kernel, dim3(n), dim3(256), 0, stream, x, D, off, slen, n, out);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_slot_workspace_floats(int N) { return NNR_SLOTS * N; }

extern "C" int nnr_tanh_score_bwd(float* th, const float* ds, const float* w2, float* dw2, const int* rows_dev, int rows, int A,
                                  float* ws, hipStream_t stream) {
  const int rpb = 64;
  int blocks = (rows + rpb - 1) / rpb;
  if (ws) blocks = blocks > NNR_SLOTS ? NNR_SLOTS : (blocks < 1 ? 1 : blocks);      // (the live rows are shared out on the device)
  hipLaunchKernelGGL(tanh_score_bwd_kernel, dim3(blocks), dim3(256), 0, stream, th, ds, w2, dw2, rows_dev, rows, A, rpb, ws);
  NNR_CHECK_LAUNCH();
  if (ws) {
    hipLaunchKernelGGL(slot_reduce_kernel, dim3((A + 7) / 8), dim3(256), 0, stream, (const float*)ws, blocks, A, dw2);
    NNR_CHECK_LAUNCH();
  }
  return NNR_OK;
}

extern "C" int nnr_rowdot(const float* x, int ld, const float* w, const int* rows_dev, int rows, int N, float* out, hipStream_t stream) {
  if (!x || !w || !out || rows <= 0) return rows <= 0 ? NNR_OK : NNR_ERR_ARG;
  if ((N & 3) || (ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return NNR_ERR_UNSUPPORTED;
  const int blocks = (rows + 3) / 4;
  hipLaunchKernelGGL(rowdot_kernel, dim3(blocks > 8192 ? 8192 : blocks), dim3(256), 0, stream, x, ld, w, rows_dev, rows, N, out);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_colsum(const float* x, int ld, const int* rows_dev, int rows, int N, float* out, float* ws, hipStream_t stream) {
  if (rows <= 0 || N <= 0) return NNR_OK;
  // rows per workgroup: few enough that mid-size inputs (3 200 .. 6 080 rows here) still spread over the chip -- with 128 rows
  // per block a [3200, 200] bias gradient was 25 workgroups walking 128 rows each (43 us); many enough to bound the atomics
  int rpb = rows / 512;
  rpb = rpb < 8 ? 8 : (rpb > 128 ? 128 : rpb);
  int blocks = (rows + rpb - 1) / rpb;
  if (ws) blocks = blocks > NNR_SLOTS ? NNR_SLOTS : blocks;
  hipLaunchKernelGGL(colsum_kernel, dim3(blocks, (N + 255) / 256), dim3(256), 0, stream, x, ld, rows_dev, rows, N, out, rpb, ws);
  NNR_CHECK_LAUNCH();
  if (ws) {
    hipLaunchKernelGGL(slot_reduce_kernel, dim3((N + 7) / 8), dim3(256), 0, stream, (const float*)ws, blocks, N, out);
    NNR_CHECK_LAUNCH();
  }
  return NNR_OK;
}

extern "C" int nnr_small_embed_fwd(const float* table, const int* idx, int n, int dim, float* out, int ldo, float p, uint32_t seed,
                                   hipStream_t stream) {
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  EW_LAUNCH(small_embed_kernel, (long)n * dim, table, idx, n, dim, out, ldo, (float*)nullptr, (const float*)nullptr, 0, seed,
            nnr_drop_thresh(p), sc);
}
extern "C" int nnr_small_embed_bwd(const int* idx, int n, int dim, const float* dout, int lddo, float* dtable, float p, uint32_t seed,
                                   hipStream_t stream) {
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  if (n <= 0) return NNR_OK;
  const int rpw = n >= 16 * 1024 ? 16 : (n >= 2048 ? 8 : 4);          // >= ~256 waves on the chip, runs long enough to merge
  hipLaunchKernelGGL(small_embed_bwd_kernel, dim3((n + 4 * rpw - 1) / (4 * rpw)), dim3(256), 0, stream, idx, n, dim, dout, lddo, dtable, seed,
                     nnr_drop_thresh(p), sc, rpw);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}


extern "C" int nnr_embed_gather(const float* table, const int* idx, long n, const int* n_dev, int dim, float* out, float p, uint32_t seed,
                                hipStream_t stream) {
  if (dim & 3) return NNR_ERR_UNSUPPORTED;
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  hipLaunchKernelGGL(embed_gather_kernel, dim3(ew_grid((n + 3) / 4, 4)), dim3(256), 0, stream, table, idx, n, n_dev, dim, out, seed,
                     nnr_drop_thresh(p), sc);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
static int embed_scatter_launch(const float* dout, const int* idx, long n, const int* n_dev, int dim, float* dtable, float p, uint32_t seed,
                                hipStream_t stream) {
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const long want = (n + 3) / 4;
  hipLaunchKernelGGL(embed_scatter_kernel, dim3((int)(want < 1 ? 1 : (want > 1024 ? 1024 : want))), dim3(256), 0, stream, dout, idx, n, n_dev, dim, dtable,
                     seed, nnr_drop_thresh(p), sc);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
extern "C" int nnr_embed_scatter(const float* dout, const int* idx, long n, int dim, float* dtable, float p, uint32_t seed, hipStream_t stream) {
  return embed_scatter_launch(dout, idx, n, nullptr, dim, dtable, p, seed, stream);
}
extern "C" int nnr_embed_scatter_dyn(const float* dout, const int* idx, long n, const int* n_dev, int dim, float* dtable, float p, uint32_t seed,
                                     hipStream_t stream) {
  return embed_scatter_launch(dout, idx, n, n_dev, dim, dtable, p, seed, stream);
}
extern "C" int nnr_transpose2d(const float* in, float* out, long rows, int cols, int accumulate, hipStream_t stream) {
  EW_LAUNCH(transpose2d_kernel, rows * cols, in, out, rows, cols, accumulate);
}

extern "C" int nnr_transpose_batch(const nnr_transpose_desc* descs_dev, int count, hipStream_t stream) {
  if (count <= 0) return NNR_OK;
  if (!descs_dev) return NNR_ERR_ARG;
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(64, count), dim3(256), 0, stream, descs_dev);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_add(float* y, const float* x, long n, float alpha, hipStream_t stream) { EW_LAUNCH(add_kernel, n, y, x, n, alpha); }
extern "C" int nnr_add_atomic(float* y, const float* x, long n, float alpha, hipStream_t stream) {
  EW_LAUNCH(add_atomic_kernel, n, y, x, n, alpha);
}
extern "C" int nnr_add2d(float* y, int ldy, const float* x, int ldx, int rows, int cols, float alpha, int accumulate,
                         hipStream_t stream) {
  EW_LAUNCH(add2d_kernel, (long)rows * cols, y, ldy, x, ldx, rows, cols, alpha, accumulate);
}

extern "C" int nnr_expand_rows_fwd(const float* x, float* y, int B, int N, int D, hipStream_t stream) {
  if (!x || !y || B <= 0 || N <= 0 || D <= 0) return NNR_ERR_ARG;
  EW_LAUNCH(expand_rows_kernel, (long)B * N * D, x, y, B, N, D);
}
extern "C" int nnr_expand_rows_bwd(const float* dy, float* dx, int B, int N, int D, hipStream_t stream) {
  if (!dy || !dx || B <= 0 || N <= 0 || D <= 0) return NNR_ERR_ARG;
  EW_LAUNCH(expand_rows_bwd_kernel, (long)B * D, dy, dx, B, N, D);
}

extern "C" int nnr_sue_x0_fwd(const float* hist, const float* proxy, float* x0, int B, int Hn, int Kc, int D, float p, uint32_t seed,
                              uint8_t* cmask_fix, hipStream_t stream) {
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  EW_LAUNCH(sue_x0_kernel, (long)B * (Hn + Kc) * D, hist, proxy, x0, B, Hn, Kc, D, (const float*)nullptr, (const float*)nullptr, (float*)nullptr,
            (float*)nullptr, cmask_fix, seed, nnr_drop_thresh(p), sc);
}
extern "C" int nnr_sue_x0_bwd(const float* dx0, const float* dx0_add, float* dhist, float* dproxy, int B, int Hn, int Kc, int D, float p,
                              uint32_t seed, hipStream_t stream) {
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  EW_LAUNCH(sue_x0_kernel, (long)B * (Hn + Kc) * D, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, B, Hn, Kc, D, dx0, dx0_add,
            dhist, dproxy, (uint8_t*)nullptr, seed, nnr_drop_thresh(p), sc);
}
extern "C" int nnr_sue_slice_fwd(const float* gcn, const float* x0, float* gfeat, int B, int Hn, int G, int D, hipStream_t stream) {
  EW_LAUNCH(sue_slice_kernel, (long)B * G * D, gcn, x0, gfeat, (const float*)nullptr, (float*)nullptr, B, Hn, G, D);
}
extern "C" int nnr_sue_slice_bwd(const float* dgfeat, float* dpad, int B, int Hn, int G, int D, hipStream_t stream) {
  EW_LAUNCH(sue_slice_kernel, (long)B * G * D, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, dgfeat, dpad, B, Hn, G, D);
}

extern "C" int nnr_relu_drop_bwd(const float* dy, const float* r, float* ds, float* dx, long n, float p, uint32_t seed,
                                 hipStream_t stream) {
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  EW_LAUNCH(relu_drop_bwd_kernel, n, dy, r, ds, dx, n, 0, seed, nnr_drop_thresh(p), sc);
}
extern "C" int nnr_dropout(const float* x, float* y, long n, float p, uint32_t seed, hipStream_t stream) {
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  EW_LAUNCH(dropout_kernel, n, x, y, n, seed, nnr_drop_thresh(p), sc);
}

extern "C" int nnr_layernorm_fwd(const float* u, const float* gamma, const float* beta, float eps, long rows, int D, float* xhat, float* rstd,
                                 float* r_out, const float* resid, float* y, float p, uint32_t seed, hipStream_t stream) {
  if (D > LN_MAXPL * 64 || rows <= 0) return D > LN_MAXPL * 64 ? NNR_ERR_UNSUPPORTED : NNR_OK;
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, u, gamma, beta, eps, rows, D, xhat, rstd, r_out, resid, y,
                     seed, nnr_drop_thresh(p), sc);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
extern "C" int nnr_layernorm_bwd(const float* dv, const float* xhat, const float* rstd, const float* gamma, long rows, int D, float* du,
                                 float* dgamma, float* dbeta, hipStream_t stream) {
  if (D > LN_MAXPL * 64 || rows <= 0) return D > LN_MAXPL * 64 ? NNR_ERR_UNSUPPORTED : NNR_OK;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, dv, xhat, rstd, gamma, rows, D, du, dgamma, dbeta);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
extern "C" int nnr_relu_bwd(const float* dy, const float* y, float* dx, long n, hipStream_t stream) {
  EW_LAUNCH(relu_bwd_kernel, n, dy, y, dx, n);
}

extern "C" int nnr_sue_intra_fwd(const float* kf, const float* qc, const float* g, const long* cidx, int B, int N, int Hn, int C, int A,
                                 int D, float* alpha, float* feat, hipStream_t stream) {
  if (Hn > SUE_MAXH || C > SUE_MAXC) return NNR_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(sue_intra_fwd_kernel, dim3((B + 7) / 8 * 8 * N * ((D + 255) / 256)), dim3(256), 0, stream, kf, qc, g, cidx, B, N, Hn, C, A, D,
                     1.f / sqrtf((float)A), alpha, feat);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
extern "C" int nnr_sue_intra_bwd(const float* kf, const float* qc, const float* g, const long* cidx, const float* alpha,
                                 const float* dfeat, int B, int N, int Hn, int C, int A, int D, float* dg, float* dkf, float* dqc,
                                 float* ds_ws, hipStream_t stream) {
  if (Hn > SUE_MAXH || C > SUE_MAXC || N > 8) return NNR_ERR_UNSUPPORTED;
  if (!ds_ws) return NNR_ERR_ARG;
  hipLaunchKernelGGL(sue_intra_bwd_ds_kernel, dim3((B + 7) / 8 * 8 * N), dim3(256), 0, stream, kf, g, cidx, alpha, dfeat, B, N, Hn, C, A, D,
                     1.f / sqrtf((float)A), ds_ws, dqc);
  NNR_CHECK_LAUNCH();
  hipLaunchKernelGGL(sue_intra_bwd_dg_kernel, dim3(B, (D + 255) / 256 + 1, C >= 3 ? 3 : 1), dim3(256), 0, stream, qc, cidx, alpha, dfeat, ds_ws, N, Hn, C,
                     A, D, dg, dkf);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_logits_loss_fwd(const float* user, const float* cand, int B, int N, int D, float* logits, float* loss, float* dlogits,
                                   hipStream_t stream) {
  hipLaunchKernelGGL(logits_kernel, dim3((B * N + 3) / 4), dim3(256), 0, stream, user, cand, B * N, D, logits);
  NNR_CHECK_LAUNCH();
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, stream, logits, B, N, loss, dlogits);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
extern "C" int nnr_logits_fwd(const float* user, const float* cand, int B, int N, int D, float* logits, hipStream_t stream) {
  hipLaunchKernelGGL(logits_kernel, dim3((B * N + 3) / 4), dim3(256), 0, stream, user, cand, B * N, D, logits);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
extern "C" int nnr_nls_loss(const float* logits, int B, int N, float* loss, float* dlogits, hipStream_t stream) {
  hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(256), 0, stream, logits, B, N, loss, dlogits);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
extern "C" int nnr_logits_bwd(const float* dlogits, const float* user, const float* cand, int B, int N, int D, float* duser, float* dcand,
                              int dcand_accumulate, hipStream_t stream) {
  EW_LAUNCH(logits_bwd_kernel, (long)B * N * D, dlogits, user, cand, (long)B * N, D, duser, dcand, dcand_accumulate);
}

static int sumsq_launch(const float* g, long n, float* out, const float* add_in, int slot, hipStream_t stream) {
  if (!g || !out || n < 0 || slot < 0 || slot >= SUMSQ_SLOTS || (((uintptr_t)g) & 15)) return NNR_ERR_ARG;
  const long want = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)(want < 1 ? 1 : (want > SUMSQ_BLOCKS ? SUMSQ_BLOCKS : want))), dim3(256), 0, stream, g, n, out, add_in, slot);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
extern "C" int nnr_sumsq(const float* g, long n, float* out_zeroed, hipStream_t stream) { return sumsq_launch(g, n, out_zeroed, nullptr, 0, stream); }
extern "C" int nnr_sumsq_part(const float* g, long n, float* out, const float* add_in, int slot, hipStream_t stream) {
  return sumsq_launch(g, n, out, add_in, slot, stream);
}

// The mirror of g_adam_skipped in pinned host memory (one word per process, never freed): set up by the first nnr_clip_adam /
// nnr_adam_skipped_peek call of the process.
static unsigned* g_skip_mirror_host = nullptr;
static int skip_mirror_init() {
  static int state = 0;                      // 0 = not tried, 1 = ready, -1 = unavailable (peek then reports NNR_ERR_LAUNCH, the sync read still works)
  if (state != 0) return state;
  unsigned* host = nullptr;
  unsigned* dev = nullptr;
  if (hipHostMalloc(reinterpret_cast<void**>(&host), 64, hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); state = -1; return state; }
  *host = 0;
  if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(g_adam_skipped_mirror), &dev, sizeof(dev)) != hipSuccess) {
    (void)hipGetLastError(); (void)hipHostFree(host); state = -1; return state;
  }
  g_skip_mirror_host = host;
  state = 1;
  return state;
}
extern "C" int nnr_clip_adam(float* p, const float* g, float* m, float* v, long n, const float* sumsq, float grad_scale, float clip,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int step, hipStream_t stream) {
  skip_mirror_init();
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  EW_LAUNCH(adam_kernel, n, p, g, m, v, n, sumsq, grad_scale, clip, lr, beta1, beta2, eps, weight_decay, bc1, bc2s);
}
extern "C" int nnr_adam_skipped_steps(unsigned* host_out, int reset) {
  // synchronous read of the device counter (diagnostics: the training loop polls it every few hundred steps, never per step)
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_adam_skipped), sizeof(v)) != hipSuccess) return NNR_ERR_LAUNCH;
  if (host_out) *host_out = v;
  if (reset) {
    const unsigned z = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_adam_skipped), &z, sizeof(z)) != hipSuccess) return NNR_ERR_LAUNCH;
    if (g_skip_mirror_host) __atomic_store_n(g_skip_mirror_host, 0u, __ATOMIC_RELAXED);
  }
  return NNR_OK;
}
extern "C" int nnr_adam_skipped_peek(unsigned* host_out) {
  // NO synchronisation: the count as of the last nnr_clip_adam launch that has COMPLETED and skipped (the kernel stores it to pinned host memory)
  if (skip_mirror_init() != 1) return NNR_ERR_LAUNCH;
  if (host_out) *host_out = __atomic_load_n(g_skip_mirror_host, __ATOMIC_RELAXED);
  return NNR_OK;
}
