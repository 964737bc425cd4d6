// Fused forms of the small, latency-bound launches on the critical chain of the training step (round 3).  At per-GPU batch 8 the
// step is ~140 dependent launches of 8-40 us with ~7 us of dispatch latency between them; at batch 64 the same launches still sit
// between the big GEMM phases.  Each kernel here replaces 2-6 of them and computes bit-identical values (same dropout masks: the
// counter-based generator is keyed by the same (seed, flat element index) as the kernels it replaces).
#include "common.h"

namespace {

// ---- feature fusion (newsEncoders.py:50-54) for the UNION of the candidate call and the history call, both tables in one launch:
//   out[row, 0:cd]      = dropout(category_table[cat(row)])        mask index row * cd + c, seed_cat
//   out[row, cd:cd+sd]  = dropout(subCategory_table[sub(row)])     mask index row * sd + c, seed_sub
// with cat(row) = row < n0 ? cat0[row] : cat1[row - n0] (the two calls' id tensors are read where they lie: no concatenation copies).
// Replaces 4 x nnr_copy_bytes + 2 x nnr_small_embed_fwd.
__global__ void fusion_rows_kernel(const float* __restrict__ ctab, const float* __restrict__ stab, const int* __restrict__ cat0,
                                   const int* __restrict__ sub0, int n0, const int* __restrict__ cat1, const int* __restrict__ sub1, int n,
                                   int cd, int sd, float* __restrict__ out, int ldo, uint32_t seed_cat, uint32_t seed_sub, uint32_t thr,
                                   float scale) {
  const int w = cd + sd;
  const long total = (long)n * w;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int row = i / w, c = i - (long)row * w;
    const bool is_cat = c < cd;
    const int cl = is_cat ? c : c - cd, dim = is_cat ? cd : sd;
    const int* ip = row < n0 ? (is_cat ? cat0 : sub0) + row : (is_cat ? cat1 : sub1) + (row - n0);
    const float m = nnr_keep(is_cat ? seed_cat : seed_sub, (uint64_t)((long)row * dim + cl), thr) ? scale : 0.f;
    out[(long)row * ldo + c] = (is_cat ? ctab : stab)[(long)(*ip) * dim + cl] * m;
  }
}

// backward of the same (replaces 2 x nnr_small_embed_bwd): blockIdx.y = table; one wave walks `rpw` consecutive rows and merges runs
// of equal ids in a register (padded history slots are news 0 -> category 0: half of the rows hit the same 50 addresses, and
// same-address f32 atomics serialise in L2), one atomic per run.
__global__ __launch_bounds__(256) void fusion_rows_bwd_kernel(const int* __restrict__ cat0, const int* __restrict__ sub0, int n0,
                                                              const int* __restrict__ cat1, const int* __restrict__ sub1, int n, int cd,
                                                              int sd, const float* __restrict__ dout, int lddo, float* __restrict__ dctab,
                                                              float* __restrict__ dstab, uint32_t seed_cat, uint32_t seed_sub,
                                                              uint32_t thr, float scale, int rpw) {
  const bool is_cat = blockIdx.y == 0;
  const int dim = is_cat ? cd : sd, col0 = is_cat ? 0 : cd;
  const int* i0 = is_cat ? cat0 : sub0;
  const int* i1 = is_cat ? cat1 : sub1;
  float* dtab = is_cat ? dctab : dstab;
  const uint32_t seed = is_cat ? seed_cat : seed_sub;
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw, r1 = min(n, r0 + rpw);
  for (int c = lane; c < dim; c += 64) {
    int cur = -1;
    float acc = 0.f;
#pragma unroll 4
    for (int row = r0; row < r1; ++row) {
      const int id = row < n0 ? i0[row] : i1[row - n0];
      const float m = nnr_keep(seed, (uint64_t)((long)row * dim + c), thr) ? scale : 0.f;
      const float v = dout[(long)row * lddo + col0 + c] * m;
      if (id != cur) {
        if (cur >= 0) atomicAdd(&dtab[(long)cur * dim + c], acc);
        cur = id;
        acc = 0.f;
      }
      acc += v;
    }
    if (cur >= 0) atomicAdd(&dtab[(long)cur * dim + c], acc);
  }
}

// Reproducible form (round 4; config.py:125-130): one workgroup per TABLE ROW t.  Its sixteen waves scan the id list in 64-row chunks
// (wave w takes chunks w, w + 16, ...: one coalesced id load, a ballot, then the matching rows in ascending order, eight in flight), lane =
// column; the sixteen partial sums are added in wave order and the table row gets ONE f32 atomic per element (a second launch -- the other encoder call
// of the plugin API -- may add into the same gradient: two addends into a zeroed buffer commute).  No run merging, no arrival order.
constexpr int FRD_WAVES = 16;
__global__ __launch_bounds__(FRD_WAVES * 64) void fusion_rows_bwd_det_kernel(const int* __restrict__ cat0, const int* __restrict__ sub0, int n0,
                                                                             const int* __restrict__ cat1, const int* __restrict__ sub1, int n, int cd,
                                                                             int sd, int ncat, const float* __restrict__ dout, int lddo,
                                                                             float* __restrict__ dctab, float* __restrict__ dstab, uint32_t seed_cat,
                                                                             uint32_t seed_sub, uint32_t thr, float scale) {
  __shared__ float part[FRD_WAVES][128];
  const bool is_cat = (int)blockIdx.x < ncat;
  const int t = is_cat ? blockIdx.x : blockIdx.x - ncat;
  const int dim = is_cat ? cd : sd, col0 = is_cat ? 0 : cd;
  const int* i0 = is_cat ? cat0 : sub0;
  const int* i1 = is_cat ? cat1 : sub1;
  float* dtab = is_cat ? dctab : dstab;
  const uint32_t seed = is_cat ? seed_cat : seed_sub;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c0 = min(lane, dim - 1), c1 = min(lane + 64, dim - 1);      // (clamped: the loads below are unconditional)
  float acc0 = 0.f, acc1 = 0.f;                       // columns lane and lane + 64 (dim <= 128, checked by the launcher)
  for (int base = w * 64; base < n; base += FRD_WAVES * 64) {
    const int row = base + lane;
    int id = -1;
    if (row < n) id = row < n0 ? i0[row] : i1[row - n0];
    unsigned long long m = __ballot(id == t);
    while (m) {                                         // eight matching rows in flight, added in ascending row order
      long rr[8];
      float wgt[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {                     // (wave-uniform arithmetic: which rows, and whether the slot is used at all)
        wgt[q] = m ? 1.f : 0.f;
        rr[q] = m ? base + __builtin_ctzll(m) : base;
        m &= m - 1;
      }
      float v0[8], v1[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {                     // all sixteen loads are issued before the first one is used
        v0[q] = dout[rr[q] * lddo + col0 + c0];
        v1[q] = dout[rr[q] * lddo + col0 + c1];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float k0 = (lane < dim && nnr_keep(seed, (uint64_t)(rr[q] * dim + lane), thr)) ? scale * wgt[q] : 0.f;
        const float k1 = (lane + 64 < dim && nnr_keep(seed, (uint64_t)(rr[q] * dim + lane + 64), thr)) ? scale * wgt[q] : 0.f;
        acc0 += v0[q] * k0;
        acc1 += v1[q] * k1;
      }
    }
  }
  part[w][lane] = acc0;
  part[w][lane + 64] = acc1;
  __syncthreads();
  if (w == 0) {
    for (int c = lane; c < dim; c += 64) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < FRD_WAVES; ++q) v += part[q][c];
      if (v != 0.f) atomicAdd(&dtab[(long)t * dim + c], v);
    }
  }
}

// ---- click predictor + loss + their backward in ONE launch (model.py:126-127, trainer.py:64-66): one workgroup per sample.
//   logits[b, n] = <user[b, n], cand[b, n]>;  loss = mean_b(-log_softmax(logits[b])[0]);  dlogits = (softmax - onehot_0) / B
//   duser[b, n] = dlogits[b, n] * cand[b, n];  dcand[b, n] = dlogits[b, n] * user[b, n]
// The mean over the batch is a DETERMINISTIC fixed-order sum: every workgroup stores its sample's term, the last one to arrive adds
// them in index order.  Replaces logits_kernel + loss_kernel + logits_bwd_kernel (3 launches on the chain).
__device__ unsigned g_click_arrived;
constexpr int CLICK_MAXN = 64;
__global__ __launch_bounds__(256) void click_loss_kernel(const float* __restrict__ user, const float* __restrict__ cand, int B, int N, int D,
                                                         float* __restrict__ logits, float* __restrict__ loss, float* __restrict__ dlogits,
                                                         float* __restrict__ duser, float* __restrict__ dcand, float* __restrict__ terms) {
  __shared__ float l[CLICK_MAXN], d[CLICK_MAXN];
  __shared__ float part[256];
  __shared__ bool last;
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float* u = user + (long)b * N * D;
  const float* c = cand + (long)b * N * D;
  for (int n = w; n < N; n += 4) {
    float p = 0.f;
    for (int x = lane; x < D; x += 64) p += u[(long)n * D + x] * c[(long)n * D + x];
    p = wave_sum(p);
    if (lane == 0) l[n] = p;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = -INFINITY;
    for (int n = 0; n < N; ++n) m = fmaxf(m, l[n]);
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += expf(l[n] - m);
    const float lse = m + logf(s);
    for (int n = 0; n < N; ++n) {
      const float dv = (expf(l[n] - lse) - (n == 0 ? 1.f : 0.f)) / (float)B;
      d[n] = dv;
      logits[(long)b * N + n] = l[n];
      if (dlogits) dlogits[(long)b * N + n] = dv;
    }
    terms[b] = lse - l[0];
  }
  __syncthreads();
  if (duser) {
    const int total = N * D;
    for (int i = threadIdx.x; i < total; i += 256) {
      const float dv = d[i / D];
      duser[(long)b * total + i] = dv * c[i];
      dcand[(long)b * total + i] = dv * u[i];
    }
  }
  if (threadIdx.x == 0) {
    __threadfence();
    last = atomicAdd(&g_click_arrived, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  float t = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) t += __builtin_nontemporal_load(&terms[i]);
  part[threadIdx.x] = t;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *loss = part[0] / (float)B;
    g_click_arrived = 0;                                   // (launches of one process are stream-ordered)
  }
}

}  // namespace

extern "C" int nnr_fusion_rows_fwd(const float* cat_table, const float* sub_table, const int* cat0, const int* sub0, int n0, const int* cat1,
                                   const int* sub1, int n1, int cd, int sd, float* out, int ldo, float p, uint32_t seed_cat,
                                   uint32_t seed_sub, hipStream_t stream) {
  if (!cat_table || !sub_table || !cat0 || !sub0 || !out || n0 < 0 || n1 < 0 || (n1 > 0 && (!cat1 || !sub1))) return NNR_ERR_ARG;
  const int n = n0 + n1;
  if (n == 0) return NNR_OK;
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const long total = (long)n * (cd + sd);
  const int grid = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(fusion_rows_kernel, dim3(grid), dim3(256), 0, stream, cat_table, sub_table, cat0, sub0, n0, cat1, sub1, n, cd, sd, out, ldo,
                     seed_cat, seed_sub, nnr_drop_thresh(p), sc);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_fusion_rows_bwd(const int* cat0, const int* sub0, int n0, const int* cat1, const int* sub1, int n1, int cd, int sd,
                                   const float* dout, int lddo, float* dcat_table, float* dsub_table, float p, uint32_t seed_cat,
                                   uint32_t seed_sub, hipStream_t stream) {
  if (!cat0 || !sub0 || !dout || !dcat_table || !dsub_table || n0 < 0 || n1 < 0 || (n1 > 0 && (!cat1 || !sub1))) return NNR_ERR_ARG;
  const int n = n0 + n1;
  if (n == 0) return NNR_OK;
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const int rpw = n >= 16 * 1024 ? 16 : (n >= 2048 ? 8 : 4);
  hipLaunchKernelGGL(fusion_rows_bwd_kernel, dim3((n + 4 * rpw - 1) / (4 * rpw), 2), dim3(256), 0, stream, cat0, sub0, n0, cat1, sub1, n, cd, sd,
                     dout, lddo, dcat_table, dsub_table, seed_cat, seed_sub, nnr_drop_thresh(p), sc, rpw);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_fusion_rows_bwd_det(const int* cat0, const int* sub0, int n0, const int* cat1, const int* sub1, int n1, int cd, int sd,
                                       int ncat, int nsub, const float* dout, int lddo, float* dcat_table, float* dsub_table, float p,
                                       uint32_t seed_cat, uint32_t seed_sub, hipStream_t stream) {
  if (!cat0 || !sub0 || !dout || !dcat_table || !dsub_table || n0 < 0 || n1 < 0 || (n1 > 0 && (!cat1 || !sub1)) || ncat < 0 || nsub < 0 || cd > 128 || sd > 128)
    return NNR_ERR_ARG;
  const int n = n0 + n1;
  if (n == 0 || ncat + nsub == 0) return NNR_OK;
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  hipLaunchKernelGGL(fusion_rows_bwd_det_kernel, dim3(ncat + nsub), dim3(FRD_WAVES * 64), 0, stream, cat0, sub0, n0, cat1, sub1, n, cd, sd, ncat, dout, lddo,
                     dcat_table, dsub_table, seed_cat, seed_sub, nnr_drop_thresh(p), sc);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_click_loss(const float* user, const float* cand, int B, int N, int D, float* logits, float* loss, float* dlogits,
                              float* duser, float* dcand, float* terms_ws, hipStream_t stream) {
  if (!user || !cand || !logits || !loss || !terms_ws || B <= 0 || N <= 0 || N > CLICK_MAXN || D <= 0 || ((duser == nullptr) != (dcand == nullptr)))
    return NNR_ERR_ARG;
  hipLaunchKernelGGL(click_loss_kernel, dim3(B), dim3(256), 0, stream, user, cand, B, N, D, logits, loss, dlogits, duser, dcand, terms_ws);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
