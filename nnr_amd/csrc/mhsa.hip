// Multi-head self-attention core on the matrix cores (replaces MultiHeadAttention.forward, layers.py:137-147, and its
// backward): per (sample, head)   S = Q K^T / sqrt(d_k) ; key mask (-1e9) ; softmax ; O = P V.
//
// One wave (64 lanes) per (sample, head); exact-fp32 v_mfma_f32_32x32x2_f32.  The products are computed TRANSPOSED,
//   S^T[key j][query i] = sum_k K[j][k] Q[i][k]        (A = K, B = Q^T)
// so a lane owns ONE query column: its 16 accumulator registers (x NB key blocks) are 16 keys of that query, the other
// 16 keys sit in lane^32.  The softmax over keys is therefore in-lane + one __shfl_xor(32) -- no LDS, no 32-lane scans.
// The normalised P^T accumulators are then used DIRECTLY as the B operand of
//   O^T[d][i] = sum_j V^T[d][j] P^T[j][i]
// with the k-order of each MFMA step permuted to the accumulator's row map (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)),
// so P never leaves registers.  Sequence lengths 32 (news titles, NB=1) and 50 (user history, NB=2, padded to 64).
// Backward recomputes nothing: it reloads P^T in the same register layout, forms dP^T = V dO^T and dS^T in registers,
// uses dS^T as a register operand for dQ^T = K^T dS^T, and goes through one LDS tile for the two products that reduce
// over the query (lane) index: dV = P^T dO and dK = dS^T Q.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

struct MhsaArgs {
  const float* qkv; const uint8_t* mask; int n, Lq, heads, dh; float scale;
  float* out; float* prob; const float* dout; float* dqkv;
};

// stage the [Lq, dh] slice (row stride ld) of one head into LDS as [LP][SD], zero rows >= Lq
__device__ __forceinline__ void stage(float* dst, const float* src, int ld, int Lq, int dh, int LP, int SD, int lane) {
  for (int idx = lane; idx < LP * dh; idx += 64) {
    const int q = idx / dh, d = idx - q * dh;
    dst[q * SD + d] = (q < Lq) ? src[(long)q * ld + d] : 0.f;
  }
}

// C^T[x][y] (NB x NB blocks of 32x32) = sum_k X[x][k] Y[y][k], X and Y staged [LP][SD], k < dh (dh even)
template <int NB>
__device__ __forceinline__ void rows_dot(const float* X, const float* Y, int dh, int SD, int lane, f32x16 (&c)[NB][NB]) {
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int xb = 0; xb < NB; ++xb)
#pragma unroll
    for (int yb = 0; yb < NB; ++yb) {
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int ks = 0; ks < dh; ks += 2) {
        const float a = X[(xb * 32 + l31) * SD + ks + half];
        const float b = Y[(yb * 32 + l31) * SD + ks + half];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      c[xb][yb] = acc;
    }
}

// R^T[d][i] (one 32-row block of d < dh, NB column blocks) = sum_j X[j][d] * P[jb][ib][reg](j = acc rows), X staged [LP][SD]
template <int NB>
__device__ __forceinline__ void acc_as_b(const float* X, const f32x16 (&P)[NB][NB], int dh, int SD, int lane, f32x16 (&o)[NB]) {
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float a = (l31 < dh) ? X[(jb * 32 + acc_row(s, half)) * SD + l31] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, P[jb][ib][s], acc, 0, 0, 0);
      }
    o[ib] = acc;
  }
}

// write R^T accumulators (rows = feature d, cols = sequence position i) to dst[i*ld + d], d < dh, i < Lq
template <int NB>
__device__ __forceinline__ void store_T(float* dst, int ld, const f32x16 (&o)[NB], int Lq, int dh, int lane) {
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
    const int i = ib * 32 + l31;
    if (i < Lq) {
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int d = acc_row(reg, half);
        if (d < dh) dst[(long)i * ld + d] = o[ib][reg];
      }
    }
  }
}

template <int NB>
__global__ __launch_bounds__(256) void mhsa_fwd_kernel(MhsaArgs a) {
  constexpr int LP = 32 * NB;
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int pair = blockIdx.x * (blockDim.x >> 6) + wv;
  if (pair >= a.n * a.heads) return;
  const int smp = pair / a.heads, head = pair - smp * a.heads;
  const int dh = a.dh, SD = dh | 1, HD = a.heads * dh, ld = 3 * HD;
  float* Qs = smem + wv * (3 * LP * SD);
  float* Ks = Qs + LP * SD;
  float* Vs = Ks + LP * SD;
  const float* base = a.qkv + (long)smp * a.Lq * ld + head * dh;
  stage(Qs, base, ld, a.Lq, dh, LP, SD, lane);
  stage(Ks, base + HD, ld, a.Lq, dh, LP, SD, lane);
  stage(Vs, base + 2 * HD, ld, a.Lq, dh, LP, SD, lane);
  __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's own LDS writes (no cross-wave sharing)
  f32x16 p[NB][NB];                        // p[jb][ib] = S^T block: rows keys, cols queries
  rows_dot<NB>(Ks, Qs, dh, SD, lane, p);
  // ---- key mask + softmax over keys (rows) for this lane's query column(s)
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
    float m = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int j = jb * 32 + acc_row(reg, half);
        float s = p[jb][ib][reg] * a.scale;
        if (j >= a.Lq) s = -INFINITY;
        else if (a.mask && !a.mask[(long)smp * a.Lq + j]) s = -1e9f;
        p[jb][ib][reg] = s;
        m = fmaxf(m, s);
      }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const float e = expf(p[jb][ib][reg] - m);
        p[jb][ib][reg] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) p[jb][ib][reg] *= inv;
  }
  if (a.prob) {
    float* pp = a.prob + (long)pair * (NB * NB * 1024);
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int ib = 0; ib < NB; ++ib)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) pp[((jb * NB + ib) * 16 + reg) * 64 + lane] = p[jb][ib][reg];
  }
  f32x16 o[NB];
  acc_as_b<NB>(Vs, p, dh, SD, lane, o);
  store_T<NB>(a.out + (long)smp * a.Lq * HD + head * dh, HD, o, a.Lq, dh, lane);
}

template <int NB>
__global__ __launch_bounds__(256) void mhsa_bwd_kernel(MhsaArgs a) {
  constexpr int LP = 32 * NB, ST = LP + 1;
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int pair = blockIdx.x * (blockDim.x >> 6) + wv;
  if (pair >= a.n * a.heads) return;
  const int smp = pair / a.heads, head = pair - smp * a.heads;
  const int dh = a.dh, SD = dh | 1, HD = a.heads * dh, ld = 3 * HD;
  float* Qs = smem + wv * (4 * LP * SD + LP * ST);
  float* Ks = Qs + LP * SD;
  float* Vs = Ks + LP * SD;
  float* Gs = Vs + LP * SD;                // dO
  float* T = Gs + LP * SD;                 // [LP][ST] transpose tile
  const float* base = a.qkv + (long)smp * a.Lq * ld + head * dh;
  stage(Qs, base, ld, a.Lq, dh, LP, SD, lane);
  stage(Ks, base + HD, ld, a.Lq, dh, LP, SD, lane);
  stage(Vs, base + 2 * HD, ld, a.Lq, dh, LP, SD, lane);
  stage(Gs, a.dout + (long)smp * a.Lq * HD + head * dh, HD, a.Lq, dh, LP, SD, lane);
  f32x16 p[NB][NB], dp[NB][NB];
  {
    const float* pp = a.prob + (long)pair * (NB * NB * 1024);
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int ib = 0; ib < NB; ++ib)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) p[jb][ib][reg] = pp[((jb * NB + ib) * 16 + reg) * 64 + lane];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  // P^T -> LDS tile (rows keys j, cols queries i) for dV = P^T dO
#pragma unroll
  for (int jb = 0; jb < NB; ++jb)
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) T[(jb * 32 + acc_row(reg, half)) * ST + ib * 32 + l31] = p[jb][ib][reg];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  float* dbase = a.dqkv + (long)smp * a.Lq * ld + head * dh;
  // generic: R[x][d] = sum_i T[x][i] * Y[i][d]   (rows x = keys, reduce over queries i)
  auto t_times = [&](const float* Y, float* dst) {
#pragma unroll
    for (int xb = 0; xb < NB; ++xb) {
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < LP; i += 2) {
        const float av = T[(xb * 32 + l31) * ST + i + half];
        const float bv = (l31 < dh) ? Y[(i + half) * SD + l31] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
      // acc: rows x (keys), cols d
      if (l31 < dh) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int j = xb * 32 + acc_row(reg, half);
          if (j < a.Lq) dst[(long)j * ld + l31] = acc[reg];
        }
      }
    }
  };
  t_times(Gs, dbase + 2 * HD);             // dV
  // dP^T = V dO^T ;  dS^T = P^T * (dP^T - delta_i) * scale
  rows_dot<NB>(Vs, Gs, dh, SD, lane, dp);
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
    float delta = 0.f;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) delta += p[jb][ib][reg] * dp[jb][ib][reg];
    delta += __shfl_xor(delta, 32, 64);
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        // masked_fill(mask == 0, -1e9) makes the score a constant: no gradient reaches Q/K through a masked key.  P is
        // exactly 0 there except when EVERY key of the sample is masked (uniform softmax) -- that case needs the explicit zero.
        const int j = jb * 32 + acc_row(reg, half);
        const bool dead = a.mask && j < a.Lq && !a.mask[(long)smp * a.Lq + j];
        dp[jb][ib][reg] = dead ? 0.f : p[jb][ib][reg] * (dp[jb][ib][reg] - delta) * a.scale;
      }
  }
  // dQ^T[k][i] = sum_j K[j][k] dS^T[j][i]  (register operand)
  f32x16 dq[NB];
  acc_as_b<NB>(Ks, dp, dh, SD, lane, dq);
  store_T<NB>(dbase, ld, dq, a.Lq, dh, lane);
  // dK = dS^T Q through the LDS tile
  __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
  for (int jb = 0; jb < NB; ++jb)
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) T[(jb * 32 + acc_row(reg, half)) * ST + ib * 32 + l31] = dp[jb][ib][reg];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  t_times(Qs, dbase + HD);
}

}  // namespace

extern "C" int nnr_mhsa_fwd(const float* qkv, const uint8_t* mask, int n, int Lq, int heads, int dh, float scale, float* out,
                            float* prob, hipStream_t stream) {
  if (!qkv || !out || n <= 0) return NNR_ERR_ARG;
  if (Lq > 64 || dh > 32 || (dh & 1)) return NNR_ERR_UNSUPPORTED;
  MhsaArgs a{qkv, mask, n, Lq, heads, dh, scale, out, prob, nullptr, nullptr};
  const int NB = Lq > 32 ? 2 : 1, LP = 32 * NB, SD = dh | 1;
  const int waves = 4;
  const size_t shm = (size_t)waves * 3 * LP * SD * sizeof(float);
  const int blocks = (n * heads + waves - 1) / waves;
  if (NB == 1) hipLaunchKernelGGL((mhsa_fwd_kernel<1>), dim3(blocks), dim3(64 * waves), shm, stream, a);
  else hipLaunchKernelGGL((mhsa_fwd_kernel<2>), dim3(blocks), dim3(64 * waves), shm, stream, a);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_mhsa_bwd(const float* qkv, const uint8_t* mask, const float* prob, const float* dout, int n, int Lq, int heads, int dh,
                            float scale, float* dqkv, hipStream_t stream) {
  if (!qkv || !prob || !dout || !dqkv || n <= 0) return NNR_ERR_ARG;
  if (Lq > 64 || dh > 32 || (dh & 1)) return NNR_ERR_UNSUPPORTED;
  MhsaArgs a{qkv, mask, n, Lq, heads, dh, scale, nullptr, const_cast<float*>(prob), dout, dqkv};
  const int NB = Lq > 32 ? 2 : 1, LP = 32 * NB, SD = dh | 1;
  const int waves = NB == 1 ? 4 : 1;
  const size_t shm = (size_t)waves * (4 * LP * SD + LP * (LP + 1)) * sizeof(float);
  const int blocks = (n * heads + waves - 1) / waves;
  if (NB == 1) hipLaunchKernelGGL((mhsa_bwd_kernel<1>), dim3(blocks), dim3(64 * waves), shm, stream, a);
  else hipLaunchKernelGGL((mhsa_bwd_kernel<2>), dim3(blocks), dim3(64 * waves), shm, stream, a);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
