// SUE's per-user graph aggregate (layers.py:285-292, GCNLayer.forward: graph @ feature): Y_b = A_b . Z_b for B users, A_b a dense
// G x G adjacency (G = max_history_num + category_num = 68 with the reference's defaults), Z_b [G, D].  As a batched 64 x 80
// tile GEMM this is 1 536 workgroups of which half own FOUR rows (68 = 64 + 4) and every one runs a 17-step reduction: 38-42 us
// for 31 MB of traffic.  Here one workgroup owns (user b, 64 columns): A_b sits in LDS (18 KB), every wave keeps the whole
// reduction of its 16 columns in registers (17 B fragments) and multiplies the five row tiles against it -- the launch is bound
// by its HBM traffic.  The forward epilogue is GCNLayer's (bias, ReLU -> r, residual, dropout), the backward variant applies the
// ReLU / dropout mask to dY while it loads it (and writes dS and the residual branch's gradient), then multiplies by A_b^T.
#include "common.h"

namespace {

constexpr int GCN_MAXG = 128;                   // 8 row tiles, 32 k-steps

// MODE 0: y = drop(relu?(A z + bias) [-> r_out] + resid)          MODE 1: ds = mask(dy) * (r > 0), dx0 = mask(dy), dz = A^T ds
// Global traffic is float4 per thread, 256 B per row of the 64-column tile, both ways: the input tile is staged in LDS (the
// backward applies its mask on the way in and writes dS / dx0 from the same registers), the product goes back through the same
// LDS tile and the epilogue runs on float4s.
template <int MODE, int MT>      // MT = ceil(G / 16) row tiles: every loop below is static for the graph size at hand
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8))) void gcn_aggregate_kernel(
    const float* __restrict__ graph, const float* __restrict__ in, const float* __restrict__ bias, const float* __restrict__ aux_in,
    float* __restrict__ out0, float* __restrict__ out1, float* __restrict__ out2, int G, int D, int relu, uint32_t seed, uint32_t thr, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // Only the GR = roundup4(G) rows that exist are kept in LDS (not the MT * 16 of the row tiles): G = 68 -> 36.4 KB instead of
  // 46.6 KB, FOUR workgroups per CU instead of three, so the 15 x B workgroups of a batch-64 launch (960) are resident at once
  // instead of 768 + a tail of 192.  Fragment rows / columns beyond GR are clamped onto GR - 1: they only feed accumulator rows
  // >= G, which are never written back.
  const int ks_n = (G + 3) >> 2, GR = ks_n * 4;
  const int A_LD = GR | 1;                      // odd row stride: the row-major fragment read (16 rows x 4 columns) spreads over the banks
  constexpr int Z_LD = 68;                      // 64 columns + 4: rows stay 16-byte aligned
  float* As = sm;                               // [GR][A_LD], zero beyond G
  float* Zs = sm + ((GR * A_LD + 3) & ~3);      // [GR][Z_LD]
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r16 = lane & 15, kk = lane >> 4;
  const float* A = graph + (long)b * G * G;
  if ((G & 3) == 0) {
    // float4 rows (G = 68: 17 per row); rows / columns beyond G are never read into a stored result (they only feed the
    // accumulator rows >= G, which the epilogue drops)
    const int g4 = G >> 2;
    for (int i = tid; i < G * g4; i += 256) {
      const int row = i / g4, c = (i - row * g4) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(A + row * G + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) As[row * A_LD + c + e] = v[e];
    }
  } else {
    for (int i = tid; i < GR * A_LD; i += 256) {
      const int row = i / A_LD, col = i - row * A_LD;
      As[i] = (row < G && col < G) ? A[row * G + col] : 0.f;
    }
  }
  const int c0 = blockIdx.x * 64;
  const int c4 = (tid & 15) * 4, rr = tid >> 4;            // this thread's float4 column inside the tile, first row
  const bool cok = c0 + c4 < D;                            // (D % 4 == 0)
  const long base = (long)b * G * D;
#pragma unroll
  for (int pass = 0; pass < MT; ++pass) {
    const int row = pass * 16 + rr;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < G && cok) {
      const long idx = base + (long)row * D + c0 + c4;
      v = *reinterpret_cast<const f32x4*>(in + idx);
      if (MODE == 1 && aux_in) {
        if (thr) {
          bool kp[4];
          nnr_keep4(seed, (uint64_t)idx, thr, kp);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = kp[e] ? v[e] * scale : 0.f;
        }
        if (out1) *reinterpret_cast<f32x4*>(out1 + idx) = v;           // gradient of the residual branch
        const f32x4 rv = *reinterpret_cast<const f32x4*>(aux_in + idx);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = rv[e] > 0.f ? v[e] : 0.f;
        *reinterpret_cast<f32x4*>(out0 + idx) = v;                     // dS
      }
    }
    if (row < GR) *reinterpret_cast<f32x4*>(&Zs[row * Z_LD + c4]) = v;
  }
  __syncthreads();
  float bf[MT * 4];                                        // lane (column r16, kk): in[4 ks + kk][column] for every k-step
#pragma unroll
  for (int ks = 0; ks < MT * 4; ++ks) bf[ks] = ks < ks_n ? Zs[(4 * ks + kk) * Z_LD + w * 16 + r16] : 0.f;
  f32x4 acc[MT];
  int arow[MT];                                            // this lane's row (forward) / column (backward) of A per row tile, clamped
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    arow[m] = min(m * 16 + r16, GR - 1);
  }
#pragma unroll
  for (int ks = 0; ks < MT * 4; ++ks) {
    if (ks < ks_n) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float a = MODE == 0 ? As[arow[m] * A_LD + 4 * ks + kk] : As[(4 * ks + kk) * A_LD + arow[m]];
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf[ks], acc[m], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                         // every wave has its B fragments: the tile is reused for the product
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (m * 16 + 4 * kk + e < GR) Zs[(m * 16 + 4 * kk + e) * Z_LD + w * 16 + r16] = acc[m][e];
  __syncthreads();
  if (!cok) return;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0 && bias) bv = *reinterpret_cast<const f32x4*>(bias + c0 + c4);
#pragma unroll
  for (int pass = 0; pass < MT; ++pass) {
    const int row = pass * 16 + rr;
    if (row < G) {
      const long idx = base + (long)row * D + c0 + c4;
      f32x4 x = *reinterpret_cast<const f32x4*>(&Zs[row * Z_LD + c4]);
      if (MODE == 0) {
        x += bv;
        if (relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
        }
        if (out1) *reinterpret_cast<f32x4*>(out1 + idx) = x;           // r: input of the backward mask
        if (aux_in) x += *reinterpret_cast<const f32x4*>(aux_in + idx);      // residual
        if (thr) {
          bool kp[4];
          nnr_keep4(seed, (uint64_t)idx, thr, kp);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = kp[e] ? x[e] * scale : 0.f;
        }
        *reinterpret_cast<f32x4*>(out0 + idx) = x;
      } else {
        *reinterpret_cast<f32x4*>(out2 + idx) = x;                     // dZ
      }
    }
  }
}


}  // namespace

extern "C" int nnr_gcn_aggregate_fwd(const float* graph, const float* z, const float* bias, const float* resid, float* r_out, float* y, int B,
                                     int G, int D, int relu, float p, uint32_t seed, hipStream_t stream) {
  if (!graph || !z || !y || B <= 0 || G <= 0 || D <= 0) return NNR_ERR_ARG;
  if (G > GCN_MAXG || (D & 3)) return NNR_ERR_UNSUPPORTED;          // float4 rows
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const int gp = (G + 15) / 16 * 16, gr = (G + 3) / 4 * 4;
#define GCN_CASE(MT_) case MT_: hipLaunchKernelGGL((gcn_aggregate_kernel<0, MT_>), dim3((D + 63) / 64, B), dim3(256), (size_t)(((gr * (gr | 1) + 3) & ~3) + gr * 68) * sizeof(float), \
                                                   stream, graph, z, bias, resid, y, r_out, nullptr, G, D, relu, seed, nnr_drop_thresh(p), sc); break;
  switch (gp / 16) { GCN_CASE(1) GCN_CASE(2) GCN_CASE(3) GCN_CASE(4) GCN_CASE(5) GCN_CASE(6) GCN_CASE(7) GCN_CASE(8) }
#undef GCN_CASE
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_gcn_aggregate_bwd(const float* graph, const float* dy, const float* r, float* ds, float* dx0, float* dz, int B, int G, int D,
                                     float p, uint32_t seed, hipStream_t stream) {
  if (!graph || !dy || !dz || B <= 0 || G <= 0 || D <= 0 || (r && !ds)) return NNR_ERR_ARG;
  if (G > GCN_MAXG || (D & 3)) return NNR_ERR_UNSUPPORTED;          // float4 rows
  const float sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const int gp = (G + 15) / 16 * 16, gr = (G + 3) / 4 * 4;
#define GCN_CASE(MT_) case MT_: hipLaunchKernelGGL((gcn_aggregate_kernel<1, MT_>), dim3((D + 63) / 64, B), dim3(256), (size_t)(((gr * (gr | 1) + 3) & ~3) + gr * 68) * sizeof(float), \
                                                   stream, graph, dy, nullptr, r, ds, dx0, dz, G, D, 0, seed, nnr_drop_thresh(p), sc); break;
  switch (gp / 16) { GCN_CASE(1) GCN_CASE(2) GCN_CASE(3) GCN_CASE(4) GCN_CASE(5) GCN_CASE(6) GCN_CASE(7) GCN_CASE(8) }
#undef GCN_CASE
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
