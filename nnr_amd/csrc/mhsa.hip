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
// Backward saves nothing but Q, K, V: it recomputes P^T with the forward's code (prob == NULL; a caller may still pass the
// probabilities the forward stored), forms dP^T = V dO^T and dS^T in registers, uses dS^T as a register operand for
// dQ^T = K^T dS^T, and goes through one LDS tile for the two products that reduce over the query (lane) index:
// dV = P^T dO and dK = dS^T Q.
//
// Memory side (this is what bounds the kernel -- QK^T / PV are ~5 % of the encoder's FLOPs): a workgroup is 4 waves = 4
// ADJACENT heads of one sample, so it moves [Lq] row segments of 4*dh floats (320 B) as float4, every load of every operand
// issued before the first LDS write (one HBM latency per wave, not one per loop trip), and the results leave through the LDS
// tiles the same way.  The key mask is one ballot word per wave.  The dropout that follows the attention in the news encoder
// is applied in the output stage / while staging dO (same counter-based mask as nnr_dropout).  Shapes that do not fit the
// 4-head grouping (heads % 4, dh % 4) take the one-head-per-wave path with the same batched loads.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

struct MhsaArgs {
  const float* qkv; const uint8_t* mask; int n, Lq, heads, dh; float scale;
  float* out; float* prob; const float* dout; float* dqkv;
  uint32_t seed, thr; float dscale;        // dropout on the attention output (thr == 0: off): out = keep(idx) ? O * dscale : 0
                                           // with idx = the flat element index in out / dout [n*Lq, heads*dh]
  const int* rowmap;                       // round 5, PACKED token rows (NULL: dense): position q of sample s lives at row rowmap[s*Lq + q] of
                                           // qkv / out / dout / dqkv, -1 = the row does not exist (a padded position: reads as zero, is not stored)
  const int* pair_off;                     // round 6, PAIRED short titles (NULL: off; needs rowmap, Lq == 32): the plan's off[] -- see mhsa_pairing()
};

// Round 6: SHORT titles share one 32 x 32 attention problem -- two of <= 16 positions, four of <= 8.  The attention core multiplies 32 x 32 blocks
// whatever the title's length; 85 % of MIND-shaped titles cover <= 16 positions and a third <= 8: their problems were mostly padding.
// nnr_mhsa_pair_map lays the samples out as VIRTUAL samples in the plan's sorted order (descending cover length), with n16 = off[17] - off[16]
// titles covering more than 16 positions (fully masked titles cover all 32) and n8 = off[9] - off[8] covering more than 8:
//   v < n16                : the title at sorted position v alone;
//   n16 <= v < n16 + P     : P = ceil((n8 - n16) / 2) pairs -- sorted positions n16 + 2 k (+1) at positions 0..15 / 16..31;
//   n16 + P <= v < nv      : Q = ceil((n - n8) / 4) quads -- sorted positions n8 + 4 k (+1, +2, +3) at positions 0..7 / 8..15 / 16..23 / 24..31.
// A query of one title must not see the keys of another: their scores are -inf (weight exactly 0 -- unlike the -1e9 of a masked key, which matters
// only inside a fully masked title, and those are never grouped), so every product that follows (P V, dP, dS, dQ, dK, dV) sees exact zeros across
// the titles of a group.  `xmask` of a sample: query i and key j belong to different titles iff (i ^ j) & xmask  (0 / 16 / 24).
struct MhsaPairing { int n16, np, nv; };
__device__ __forceinline__ MhsaPairing mhsa_pairing(const MhsaArgs& a) {
  if (!a.pair_off) return MhsaPairing{a.n, 0, a.n};
  const int n16 = a.pair_off[17] - a.pair_off[16], n8 = a.pair_off[9] - a.pair_off[8];
  const int np = (n8 - n16 + 1) / 2;
  return MhsaPairing{n16, np, n16 + np + (a.n - n8 + 3) / 4};
}
__device__ __forceinline__ int mhsa_xmask(const MhsaArgs& a, const MhsaPairing& pg, int smp) {
  return !a.pair_off || smp < pg.n16 ? 0 : (smp < pg.n16 + pg.np ? 16 : 24);
}

// stage the [Lq, dh] slice (row stride ld) of one head into LDS as [LP][SD], zero rows >= Lq
__device__ __forceinline__ void stage(float* dst, const float* src, int ld, int Lq, int dh, int LP, int SD, int lane) {
  for (int idx = lane; idx < LP * dh; idx += 64) {
    const int q = idx / dh, d = idx - q * dh;
    dst[q * SD + d] = (q < Lq) ? src[(long)q * ld + d] : 0.f;
  }
}

// the same for the upstream gradient: dO = keep(idx) ? dout * dscale : 0   (e0 = flat index of src[0] in dout)
__device__ __forceinline__ void stage_drop(float* dst, const float* src, long e0, int ld, int Lq, int dh, int LP, int SD, int lane,
                                           const MhsaArgs& a) {
  for (int idx = lane; idx < LP * dh; idx += 64) {
    const int q = idx / dh, d = idx - q * dh;
    float v = 0.f;
    if (q < Lq) {
      v = src[(long)q * ld + d];
      if (a.thr) v = nnr_keep(a.seed, (uint64_t)(e0 + (long)q * ld + d), a.thr) ? v * a.dscale : 0.f;
    }
    dst[q * SD + d] = v;
  }
}

// Per-wave variant of the batched float4 staging below (one head per wave, dh % 4 == 0): NMAT tiles of [LP][SD] from row
// slices of dh floats; all loads of a batch are in flight before the first LDS write.  DM as in stage4v.
template <int NMAT, int LP, int DM = -1>
__device__ __forceinline__ void stage1v(float* tiles, const float* const (&src)[NMAT], const int (&ld)[NMAT], int Lq, int dh, int SD,
                                        int lane, const MhsaArgs* a = nullptr, long e0 = 0) {
  const int dh4 = dh >> 2, n4 = LP * dh4;
  constexpr int U = 5;
  for (int b0 = 0; b0 < n4; b0 += 64 * U) {
    float4 r[NMAT][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = b0 + u * 64 + lane, q = idx / dh4, c4 = idx - q * dh4;
      const bool live = idx < n4 && q < Lq;
#pragma unroll
      for (int m = 0; m < NMAT; ++m)
        r[m][u] = live ? *(const float4*)(src[m] + (long)q * ld[m] + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (DM >= 0 && a->thr && live) {
        bool k[4];
        nnr_keep4(a->seed, (uint64_t)(e0 + (long)q * ld[DM >= 0 ? DM : 0] + 4 * c4), a->thr, k);
        float4& v = r[DM >= 0 ? DM : 0][u];
        v.x = k[0] ? v.x * a->dscale : 0.f; v.y = k[1] ? v.y * a->dscale : 0.f;
        v.z = k[2] ? v.z * a->dscale : 0.f; v.w = k[3] ? v.w * a->dscale : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = b0 + u * 64 + lane, q = idx / dh4, d = 4 * (idx - q * dh4);
      if (idx < n4) {
#pragma unroll
        for (int m = 0; m < NMAT; ++m) {
          float* t = tiles + m * LP * SD + q * SD + d;
          t[0] = r[m][u].x; t[1] = r[m][u].y; t[2] = r[m][u].z; t[3] = r[m][u].w;
        }
      }
    }
  }
}

// Cooperative path (heads % 4 == 0, dh % 4 == 0): the 4 waves of a workgroup own 4 CONSECUTIVE heads of one sample, whose
// slices are adjacent in memory, so the workgroup moves [Lq] row segments of 4*dh contiguous floats (320 B at dh = 20) as
// float4 -- ALL loads of all NMAT matrices are issued before the first LDS write, so a wave pays the HBM latency once
// instead of once per loop trip -- and scatters them into the per-wave tiles (tile m of wave w at w*wstride + m*LP*SD).
// DM >= 0: matrix DM is the upstream gradient and gets the dropout mask (e0 = flat index of src[DM][0] in dout).
// rmap (packed rows): the sample's slice of MhsaArgs::rowmap -- position q is row rmap[q] (src / e0 then are relative to ROW 0 of the buffers)
template <int NMAT, int LP, int DM = -1>
__device__ __forceinline__ void stage4v(float* smem, int wstride, const float* const (&src)[NMAT], const int (&ld)[NMAT],
                                        int Lq, int dh, int SD, int tid, const MhsaArgs* a = nullptr, long e0 = 0, const int* rmap = nullptr) {
  const int n4 = LP * dh;                  // float4 per matrix: LP rows x (4 heads * dh / 4)
  constexpr int U = 3;
  for (int b0 = 0; b0 < n4; b0 += 256 * U) {
    float4 r[NMAT][U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = b0 + u * 256 + tid, q0 = idx / dh, c4 = idx - q0 * dh;
      bool live = idx < n4 && q0 < Lq;
      long q = q0;
      if (rmap) { q = live ? rmap[q0] : -1; live = q >= 0; }
#pragma unroll
      for (int m = 0; m < NMAT; ++m)
        r[m][u] = live ? *(const float4*)(src[m] + (long)q * ld[m] + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (DM >= 0 && a->thr && live) {
        bool k[4];
        nnr_keep4(a->seed, (uint64_t)(e0 + (long)q * ld[DM >= 0 ? DM : 0] + 4 * c4), a->thr, k);
        float4& v = r[DM >= 0 ? DM : 0][u];
        v.x = k[0] ? v.x * a->dscale : 0.f; v.y = k[1] ? v.y * a->dscale : 0.f;
        v.z = k[2] ? v.z * a->dscale : 0.f; v.w = k[3] ? v.w * a->dscale : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = b0 + u * 256 + tid, q = idx / dh, c = 4 * (idx - q * dh), w = c / dh, d = c - w * dh;
      if (idx < n4) {
#pragma unroll
        for (int m = 0; m < NMAT; ++m) {
          float* t = smem + w * wstride + m * LP * SD + q * SD + d;
          t[0] = r[m][u].x; t[1] = r[m][u].y; t[2] = r[m][u].z; t[3] = r[m][u].w;
        }
      }
    }
  }
}

// the reverse: per-wave result tiles [LP][SD] (rows = sequence position) -> [Lq] row segments of 4*dh floats, float4 stores
template <int LP, bool DROP = false>
__device__ __forceinline__ void unstage4v(const float* tile0, int wstride, float* dst, int ld, int Lq, int dh, int SD, int tid,
                                          const MhsaArgs* a = nullptr, long e0 = 0, const int* rmap = nullptr) {
  const int n4 = Lq * dh;
  for (int idx = tid; idx < n4; idx += 256) {
    const int q0 = idx / dh, c = 4 * (idx - q0 * dh), w = c / dh, d = c - w * dh;
    const float* t = tile0 + w * wstride + q0 * SD + d;
    long q = q0;
    if (rmap) { q = rmap[q0]; if (q < 0) continue; }         // a padded position: no row to store to
    float4 v = make_float4(t[0], t[1], t[2], t[3]);
    if (DROP && a->thr) {
      bool k[4];
      nnr_keep4(a->seed, (uint64_t)(e0 + (long)q * ld + c), a->thr, k);
      v.x = k[0] ? v.x * a->dscale : 0.f; v.y = k[1] ? v.y * a->dscale : 0.f;
      v.z = k[2] ? v.z * a->dscale : 0.f; v.w = k[3] ? v.w * a->dscale : 0.f;
    }
    *(float4*)(dst + (long)q * ld + c) = v;
  }
}

// R^T accumulators (rows = feature d, cols = sequence position i) -> LDS tile[i][d]
template <int NB>
__device__ __forceinline__ void tile_T(float* tile, int SD, const f32x16 (&o)[NB], int dh, int lane) {
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int ib = 0; ib < NB; ++ib)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int d = acc_row(reg, half);
      if (d < dh) tile[(ib * 32 + l31) * SD + d] = o[ib][reg];
    }
}

// the sample's key mask as one wave-uniform 64-bit word (bit j = key j is live): one byte load per lane + a ballot, instead
// of 16 byte loads per lane in every consumer
__device__ __forceinline__ unsigned long long key_bits(const MhsaArgs& a, int smp, int lane) {
  if (!a.mask) return ~0ull;
  const bool live = lane < a.Lq && a.mask[(long)smp * a.Lq + lane] != 0;
  return __ballot(live);
}

// key mask + softmax over keys (rows) of S^T for this lane's query column(s); p: raw scores in, probabilities out
template <int NB>
__device__ __forceinline__ void softmax_T(f32x16 (&p)[NB][NB], const MhsaArgs& a, unsigned long long live, int half, int xmask = 0, int qcol = 0) {
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
    float m = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int j = jb * 32 + acc_row(reg, half);
        float sc = p[jb][ib][reg] * a.scale;
        if (j >= a.Lq) sc = -INFINITY;
        else if ((qcol ^ j) & xmask) sc = -INFINITY;                    // (NB == 1) a key of ANOTHER title of the group: this lane's query is qcol
        else if (!((live >> j) & 1)) sc = -1e9f;
        p[jb][ib][reg] = sc;
        m = fmaxf(m, sc);
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
}

// C^T[x][y] (NB x NB blocks of 32x32) = sum_k X[x][k] Y[y][k], X and Y staged [LP][SD], k < dh (dh even)
// DH > 0: the head dimension is a compile-time constant (20 in every BASELINE config): the k loop unrolls, all of a product's LDS reads are
// issued ahead of its dependent MFMA chain instead of one (read, wait ~100 clk, MFMA 64 clk) round per step -- with two waves per SIMD
// that chain, not the matrix pipe (0.23 busy) or HBM, was what a head's ~14 us were made of (round 4)
template <int NB, int DH = 0>
__device__ __forceinline__ void rows_dot(const float* X, const float* Y, int dh, int SD, int lane, f32x16 (&c)[NB][NB]) {
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int xb = 0; xb < NB; ++xb)
#pragma unroll
    for (int yb = 0; yb < NB; ++yb) {
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (DH > 0) {
        float av[DH / 2 > 0 ? DH / 2 : 1], bv[DH / 2 > 0 ? DH / 2 : 1];
#pragma unroll
        for (int k2 = 0; k2 < DH / 2; ++k2) {
          av[k2] = X[(xb * 32 + l31) * SD + 2 * k2 + half];
          bv[k2] = Y[(yb * 32 + l31) * SD + 2 * k2 + half];
        }
#pragma unroll
        for (int k2 = 0; k2 < DH / 2; ++k2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k2], bv[k2], acc, 0, 0, 0);
      } else {
        for (int ks = 0; ks < dh; ks += 2) {
          const float a = X[(xb * 32 + l31) * SD + ks + half];
          const float b = Y[(yb * 32 + l31) * SD + ks + half];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
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
    for (int jb = 0; jb < NB; ++jb) {
      float av[16];
#pragma unroll
      for (int s = 0; s < 16; ++s) av[s] = X[(jb * 32 + acc_row(s, half)) * SD + l31];      // reads first, then the chain.  (Lanes l31 >= dh read up to 11
                                                                                          // floats past the row -- inside the workgroup's tiles -- and only feed output ROWS >= dh, which nobody stores)
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], P[jb][ib][s], acc, 0, 0, 0);
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

template <int NB, int DH, bool FULL = false>
__global__ __launch_bounds__(256) void mhsa_fwd_kernel(MhsaArgs a_in, int coop) {
  constexpr int LP = 32 * NB;
  MhsaArgs a = a_in;
  if (FULL) a.Lq = LP;
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5;
  const int nw = blockDim.x >> 6;
  const int pair = blockIdx.x * nw + wv;
  const int dh = DH > 0 ? DH : a.dh, SD = dh | 1, HD = a.heads * dh, ld = 3 * HD;      // (DH > 0: every / dh, % dh and * SD below folds)
  const int wstride = 3 * LP * SD;
  float* Qs = smem + wv * wstride;
  float* Ks = Qs + LP * SD;
  float* Vs = Ks + LP * SD;
  const int pair0 = blockIdx.x * 4, smp0 = pair0 / a.heads, head0 = pair0 - smp0 * a.heads;    // coop: the workgroup's 4 heads
  const int* rmap = a.rowmap ? a.rowmap + (long)smp0 * a.Lq : nullptr;        // (packed rows: coop path only, checked by the entry point)
  const MhsaPairing pg = mhsa_pairing(a);
  if (a.pair_off && smp0 >= pg.nv) return;                                     // (paired: coop path only; the grid is sized for n samples)
  if (coop) {
    const float* base0 = a.qkv + (rmap ? 0 : (long)smp0 * a.Lq * ld) + head0 * dh;
    const float* const src[3] = {base0, base0 + HD, base0 + 2 * HD};
    const int lds[3] = {ld, ld, ld};
    stage4v<3, LP>(smem, wstride, src, lds, a.Lq, dh, SD, threadIdx.x, nullptr, 0, rmap);
    __syncthreads();
  }
  if (pair >= a.n * a.heads) return;       // never taken on the coop path (n * heads is a multiple of 4 there)
  const int smp = pair / a.heads, head = pair - smp * a.heads;
  if (!coop) {
    const float* base = a.qkv + (long)smp * a.Lq * ld + head * dh;
    if (!(dh & 3) && !(HD & 3)) {
      const float* const src[3] = {base, base + HD, base + 2 * HD};
      const int lds[3] = {ld, ld, ld};
      stage1v<3, LP>(Qs, src, lds, a.Lq, dh, SD, lane);
    } else {
      stage(Qs, base, ld, a.Lq, dh, LP, SD, lane);
      stage(Ks, base + HD, ld, a.Lq, dh, LP, SD, lane);
      stage(Vs, base + 2 * HD, ld, a.Lq, dh, LP, SD, lane);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): this wave's own LDS writes (no cross-wave sharing)
  }
  f32x16 p[NB][NB];                        // p[jb][ib] = S^T block: rows keys, cols queries
  rows_dot<NB, DH>(Ks, Qs, dh, SD, lane, p);
  softmax_T<NB>(p, a, key_bits(a, smp, lane), half, mhsa_xmask(a, pg, smp), lane & 31);
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
  if (coop) {                              // O -> this wave's Q tile (free since the score product) -> 320-B row segments
    tile_T<NB>(Qs, SD, o, dh, lane);
    __syncthreads();
    const long e0 = (rmap ? 0 : (long)smp0 * a.Lq * HD) + head0 * dh;
    unstage4v<LP, true>(smem, wstride, a.out + e0, HD, a.Lq, dh, SD, threadIdx.x, &a, e0, rmap);
  } else {
    const long e0 = (long)smp * a.Lq * HD + head * dh;
    if (a.thr) {
#pragma unroll
      for (int ib = 0; ib < NB; ++ib)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const long e = e0 + (long)(ib * 32 + (lane & 31)) * HD + acc_row(reg, half);
          o[ib][reg] = nnr_keep(a.seed, (uint64_t)e, a.thr) ? o[ib][reg] * a.dscale : 0.f;
        }
    }
    store_T<NB>(a.out + e0, HD, o, a.Lq, dh, lane);
  }
}

template <int NB, int DH>
__global__ __launch_bounds__(256) void mhsa_bwd_kernel(MhsaArgs a, int coop) {
  constexpr int LP = 32 * NB, ST = LP + 1;
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int pair = blockIdx.x * (blockDim.x >> 6) + wv;
  const int dh = DH > 0 ? DH : a.dh, SD = dh | 1, HD = a.heads * dh, ld = 3 * HD;      // (DH > 0: every / dh, % dh and * SD below folds)
  const int wstride = 4 * LP * SD + LP * ST;
  float* Qs = smem + wv * wstride;
  float* Ks = Qs + LP * SD;
  float* Vs = Ks + LP * SD;
  float* Gs = Vs + LP * SD;                // dO
  float* T = Gs + LP * SD;                 // [LP][ST] transpose tile
  const int pair0 = blockIdx.x * 4, smp0 = pair0 / a.heads, head0 = pair0 - smp0 * a.heads;
  const int* rmap = a.rowmap ? a.rowmap + (long)smp0 * a.Lq : nullptr;
  if (coop) {
    const float* base0 = a.qkv + (rmap ? 0 : (long)smp0 * a.Lq * ld) + head0 * dh;
    const long e0c = (rmap ? 0 : (long)smp0 * a.Lq * HD) + head0 * dh;
    const float* const src[4] = {base0, base0 + HD, base0 + 2 * HD, a.dout + e0c};
    const int lds[4] = {ld, ld, ld, HD};
    stage4v<4, LP, 3>(smem, wstride, src, lds, a.Lq, dh, SD, threadIdx.x, &a, e0c, rmap);
    __syncthreads();
  }
  if (pair >= a.n * a.heads) return;
  const int smp = pair / a.heads, head = pair - smp * a.heads;
  if (!coop) {
    const float* base = a.qkv + (long)smp * a.Lq * ld + head * dh;
    const long e0 = (long)smp * a.Lq * HD + head * dh;
    if (!(dh & 3) && !(HD & 3)) {
      const float* const src[4] = {base, base + HD, base + 2 * HD, a.dout + e0};
      const int lds[4] = {ld, ld, ld, HD};
      stage1v<4, LP, 3>(Qs, src, lds, a.Lq, dh, SD, lane, &a, e0);
    } else {
      stage(Qs, base, ld, a.Lq, dh, LP, SD, lane);
      stage(Ks, base + HD, ld, a.Lq, dh, LP, SD, lane);
      stage(Vs, base + 2 * HD, ld, a.Lq, dh, LP, SD, lane);
      stage_drop(Gs, a.dout + e0, e0, HD, a.Lq, dh, LP, SD, lane, a);
    }
  }
  f32x16 p[NB][NB], dp[NB][NB];
  const unsigned long long live = key_bits(a, smp, lane);
  if (a.prob) {
    const float* pp = a.prob + (long)pair * (NB * NB * 1024);
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int ib = 0; ib < NB; ++ib)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) p[jb][ib][reg] = pp[((jb * NB + ib) * 16 + reg) * 64 + lane];
    __builtin_amdgcn_s_waitcnt(0xc07f);
  } else {
    // recompute P^T from Q, K: one more 32 x 32 x dh product + softmax instead of 4 KB of HBM traffic each way per head
    __builtin_amdgcn_s_waitcnt(0xc07f);
    rows_dot<NB, DH>(Ks, Qs, dh, SD, lane, p);
    softmax_T<NB>(p, a, live, half);
  }
  // P^T -> LDS tile (rows keys j, cols queries i) for dV = P^T dO
#pragma unroll
  for (int jb = 0; jb < NB; ++jb)
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) T[(jb * 32 + acc_row(reg, half)) * ST + ib * 32 + l31] = p[jb][ib][reg];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  float* dbase = a.dqkv + (long)smp * a.Lq * ld + head * dh;
  // generic: R[x][d] = sum_i T[x][i] * Y[i][d]   (rows x = keys, reduce over queries i); rows go to dst[x*dld + d]
  auto t_times = [&](const float* Y, float* dst, int dld) {
#pragma unroll
    for (int xb = 0; xb < NB; ++xb) {
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i0 = 0; i0 < LP; i0 += 32) {
        float av[16], bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          av[u] = T[(xb * 32 + l31) * ST + i0 + 2 * u + half];
          bv[u] = (l31 < dh) ? Y[(i0 + 2 * u + half) * SD + l31] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
      }
      // acc: rows x (keys), cols d
      if (l31 < dh) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int j = xb * 32 + acc_row(reg, half);
          if (j < a.Lq) dst[(long)j * dld + l31] = acc[reg];
        }
      }
    }
  };
  // dP^T = V dO^T first: afterwards the V tile is free and takes dV (coop path: results leave through LDS as row segments)
  rows_dot<NB, DH>(Vs, Gs, dh, SD, lane, dp);
  if (coop) t_times(Gs, Vs, SD);           // dV (LDS ops of one wave execute in order: the reads of V above are done)
  else t_times(Gs, dbase + 2 * HD, ld);
  // dS^T = P^T * (dP^T - delta_i) * scale
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
        const bool dead = j < a.Lq && !((live >> j) & 1);
        dp[jb][ib][reg] = dead ? 0.f : p[jb][ib][reg] * (dp[jb][ib][reg] - delta) * a.scale;
      }
  }
  // dQ^T[k][i] = sum_j K[j][k] dS^T[j][i]  (register operand)
  f32x16 dq[NB];
  acc_as_b<NB>(Ks, dp, dh, SD, lane, dq);
  if (coop) tile_T<NB>(Gs, SD, dq, dh, lane);            // the dO tile is free now
  else store_T<NB>(dbase, ld, dq, a.Lq, dh, lane);
  // dK = dS^T Q through the LDS tile
  __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
  for (int jb = 0; jb < NB; ++jb)
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) T[(jb * 32 + acc_row(reg, half)) * ST + ib * 32 + l31] = dp[jb][ib][reg];
  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (coop) {
    t_times(Qs, Ks, SD);                   // dK into the K tile (its last reader was the dQ product)
    __syncthreads();
    float* d0 = a.dqkv + (rmap ? 0 : (long)smp0 * a.Lq * ld) + head0 * dh;
    unstage4v<LP>(smem + 3 * LP * SD, wstride, d0, ld, a.Lq, dh, SD, threadIdx.x, nullptr, 0, rmap);            // dQ
    unstage4v<LP>(smem + 1 * LP * SD, wstride, d0 + HD, ld, a.Lq, dh, SD, threadIdx.x, nullptr, 0, rmap);       // dK
    unstage4v<LP>(smem + 2 * LP * SD, wstride, d0 + 2 * HD, ld, a.Lq, dh, SD, threadIdx.x, nullptr, 0, rmap);   // dV
  } else {
    t_times(Qs, dbase + HD, ld);
  }
}


// ---- persistent backward (round 4): NB = 1, 4-head groups, P recomputed.  A workgroup walks `gp` consecutive 4-head groups; the
// global loads of group g + 1 (Q, K, V, dO row segments + the key-mask byte: 13 registers of float4 per lane) are issued BEFORE the
// products of group g and land in LDS after them, and the results of group g leave LDS through registers so that their global stores
// are issued after the next group's tiles are in place -- per group a wave no longer sits through a load latency, a workgroup launch
// and a store drain with only one other wave on its SIMD to cover for it (the one-group kernel: 8 us per head for ~4 us of work).
template <int NMAT, int DM>
__device__ __forceinline__ void load4v(float4 (&r)[NMAT][3], const float* const (&src)[NMAT], const int (&ld)[NMAT], int Lq, int dh, int tid,
                                       const MhsaArgs& a, long e0, const int* rmap = nullptr) {
  const int n4 = 32 * dh;                  // <= 768: one batch of three float4 per thread and matrix
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int idx = u * 256 + tid, q0 = idx / dh, c4 = idx - q0 * dh;
    bool live = idx < n4 && q0 < Lq;
    long q = q0;
    if (rmap) { q = live ? rmap[q0] : -1; live = q >= 0; }
#pragma unroll
    for (int m = 0; m < NMAT; ++m)
      r[m][u] = live ? *(const float4*)(src[m] + (long)q * ld[m] + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (DM >= 0 && a.thr && live) {
      bool k[4];
      nnr_keep4(a.seed, (uint64_t)(e0 + (long)q * ld[DM >= 0 ? DM : 0] + 4 * c4), a.thr, k);
      float4& v = r[DM >= 0 ? DM : 0][u];
      v.x = k[0] ? v.x * a.dscale : 0.f; v.y = k[1] ? v.y * a.dscale : 0.f;
      v.z = k[2] ? v.z * a.dscale : 0.f; v.w = k[3] ? v.w * a.dscale : 0.f;
    }
  }
}
template <int NMAT>
__device__ __forceinline__ void put4v(float* smem, int wstride, const float4 (&r)[NMAT][3], int dh, int SD, int tid) {
  const int n4 = 32 * dh;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int idx = u * 256 + tid, q = idx / dh, c = 4 * (idx - q * dh), w = c / dh, d = c - w * dh;
    if (idx < n4) {
#pragma unroll
      for (int m = 0; m < NMAT; ++m) {
        float* t = smem + w * wstride + m * 32 * SD + q * SD + d;
        t[0] = r[m][u].x; t[1] = r[m][u].y; t[2] = r[m][u].z; t[3] = r[m][u].w;
      }
    }
  }
}

template <int DH, bool FULL>
__global__ __launch_bounds__(256, 2) void mhsa_bwd_persist_kernel(MhsaArgs a_in, int gp) {
  constexpr int NB = 1, LP = 32, ST = LP + 1;
  MhsaArgs a = a_in;
  if (FULL) a.Lq = LP;                       // every `< Lq` predicate of the tiles (~70 per group) folds: titles are padded to 32 tokens in every config
  extern __shared__ float smem[];
  const int tid0 = threadIdx.x, lane0 = tid0 & 63, wv = tid0 >> 6;
  const int dh = DH > 0 ? DH : a.dh, SD = dh | 1, HD = a.heads * dh, ld = 3 * HD;      // (DH > 0: every / dh, % dh and * SD below folds)
  const int wstride = 4 * LP * SD + LP * ST;
  float* Qs = smem + wv * wstride;
  float* Ks = Qs + LP * SD;
  float* Vs = Ks + LP * SD;
  float* Gs = Vs + LP * SD;                // dO
  float* T = Gs + LP * SD;                 // [LP][ST] transpose tile
  const MhsaPairing pg = mhsa_pairing(a);
  const int ngroups = pg.nv * a.heads / 4;                     // (paired short titles: nv virtual samples, the grid is sized for n)
  const int g_lo = blockIdx.x * gp, g_hi = min(ngroups, g_lo + gp);
  if (g_lo >= g_hi) return;
  float4 r[4][3];
  int mb = 0;                              // the key-mask byte of this lane's key position, loaded with the group's tiles
  auto issue = [&](int g) __attribute__((always_inline)) {
    const int pair0 = g * 4, smp0 = pair0 / a.heads, head0 = pair0 - smp0 * a.heads;
    const int* rmap = a.rowmap ? a.rowmap + (long)smp0 * a.Lq : nullptr;
    const float* base0 = a.qkv + (rmap ? 0 : (long)smp0 * a.Lq * ld) + head0 * dh;
    const long e0 = (rmap ? 0 : (long)smp0 * a.Lq * HD) + head0 * dh;
    const float* const src[4] = {base0, base0 + HD, base0 + 2 * HD, a.dout + e0};
    const int lds[4] = {ld, ld, ld, HD};
    mb = (a.mask && lane0 < a.Lq) ? (int)a.mask[(long)smp0 * a.Lq + lane0] : (a.mask ? 0 : 1);
    load4v<4, 3>(r, src, lds, a.Lq, dh, tid0, a, e0, rmap);
  };
  issue(g_lo);
  put4v<4>(smem, wstride, r, dh, SD, tid0);
#pragma nounroll
  for (int g = g_lo; g < g_hi; ++g) {
    // the lane id goes through an opaque move once per group: every LDS address below depends on it, so the compiler cannot hoist the
    // ~100 loop-invariant per-lane offsets of the transposes out of the group loop (it did, and spilled 26-59 of them)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int pair0 = g * 4, smp0 = pair0 / a.heads, head0 = pair0 - smp0 * a.heads;
    const unsigned long long live = a.mask ? __ballot(mb != 0) : ~0ull;
    __syncthreads();                                            // this group's tiles are in LDS
    f32x16 p[NB][NB], dp[NB][NB];
    rows_dot<NB, DH>(Ks, Qs, dh, SD, lane, p);
    softmax_T<NB>(p, a, live, half, mhsa_xmask(a, pg, smp0), l31);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) T[acc_row(reg, half) * ST + l31] = p[0][0][reg];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    auto t_times = [&](const float* Y, float* dst, int dld) __attribute__((always_inline)) {
      f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float av[16], bv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        av[u] = T[l31 * ST + 2 * u + half];
        bv[u] = Y[(2 * u + half) * SD + l31];                   // (lanes l31 >= dh: junk columns, not stored)
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
      if (l31 < dh) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int j = acc_row(reg, half);
          if (j < a.Lq) dst[(long)j * dld + l31] = acc[reg];
        }
      }
    };
    rows_dot<NB, DH>(Vs, Gs, dh, SD, lane, dp);
    t_times(Gs, Vs, SD);                                        // dV into the V tile
    {
      float delta = 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) delta += p[0][0][reg] * dp[0][0][reg];
      delta += __shfl_xor(delta, 32, 64);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int j = acc_row(reg, half);
        const bool dead = j < a.Lq && !((live >> j) & 1);
        dp[0][0][reg] = dead ? 0.f : p[0][0][reg] * (dp[0][0][reg] - delta) * a.scale;
      }
    }
    // the next group's loads go out HERE: P is dead, the remaining products (dQ, dK: half of the group's MFMA work) and the result
    // stores cover their latency, and the 52 registers they land in do not overlap the scores / dP phase (issued at the top of the
    // group they cost 26-59 spilled VGPRs at two waves per SIMD)
    if (g + 1 < g_hi) issue(g + 1);
    f32x16 dq[NB];
    acc_as_b<NB>(Ks, dp, dh, SD, lane, dq);
    tile_T<NB>(Gs, SD, dq, dh, lane);                           // dQ into the dO tile
    __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) T[acc_row(reg, half) * ST + l31] = dp[0][0][reg];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    t_times(Qs, Ks, SD);                                        // dK into the K tile
    __syncthreads();
    // results leave through LDS as row segments; then the NEXT group's tiles (already in registers) go in
    const int* rmap_g = a.rowmap ? a.rowmap + (long)smp0 * a.Lq : nullptr;
    float* d0 = a.dqkv + (rmap_g ? 0 : (long)smp0 * a.Lq * ld) + head0 * dh;
    unstage4v<LP>(smem + 3 * LP * SD, wstride, d0, ld, a.Lq, dh, SD, tid, nullptr, 0, rmap_g);            // dQ
    unstage4v<LP>(smem + 1 * LP * SD, wstride, d0 + HD, ld, a.Lq, dh, SD, tid, nullptr, 0, rmap_g);       // dK
    unstage4v<LP>(smem + 2 * LP * SD, wstride, d0 + 2 * HD, ld, a.Lq, dh, SD, tid, nullptr, 0, rmap_g);   // dV
    __syncthreads();                                            // every wave has read the result tiles
    if (g + 1 < g_hi) put4v<4>(smem, wstride, r, dh, SD, tid);
  }
}

}  // namespace

static int mhsa_fwd_launch(const float* qkv, const uint8_t* mask, const int* rowmap, int n, int Lq, int heads, int dh, float scale, float* out,
                           float* prob, float drop_p, uint32_t seed, hipStream_t stream, const int* pair_off = nullptr) {
  if (!qkv || !out || n <= 0) return NNR_ERR_ARG;
  if (Lq > 64 || dh > 32 || (dh & 1)) return NNR_ERR_UNSUPPORTED;
  if (pair_off && (!rowmap || Lq != 32)) return NNR_ERR_UNSUPPORTED;
  MhsaArgs a{qkv, mask, n, Lq, heads, dh, scale, out, prob, nullptr, nullptr, seed, nnr_drop_thresh(drop_p),
             drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, rowmap, pair_off};
  const int NB = Lq > 32 ? 2 : 1, LP = 32 * NB, SD = dh | 1;
  const int waves = 4;
  const int coop = (heads % 4 == 0 && dh % 4 == 0) ? 1 : 0;        // a workgroup's 4 waves then are 4 adjacent heads of one sample
  if (rowmap && (!coop || prob)) return NNR_ERR_UNSUPPORTED;       // packed rows (and paired titles): the 4-head cooperative staging only
  const size_t shm = ((size_t)waves * 3 * LP * SD + 16) * sizeof(float);      // + 16: operand reads of lanes >= dh run up to 11 floats past the last row
  const int blocks = (n * heads + waves - 1) / waves;
  if (NB == 1 && dh == 20 && Lq == 32) hipLaunchKernelGGL((mhsa_fwd_kernel<1, 20, true>), dim3(blocks), dim3(64 * waves), shm, stream, a, coop);
  else if (NB == 1 && dh == 20) hipLaunchKernelGGL((mhsa_fwd_kernel<1, 20>), dim3(blocks), dim3(64 * waves), shm, stream, a, coop);
  else if (NB == 1) hipLaunchKernelGGL((mhsa_fwd_kernel<1, 0>), dim3(blocks), dim3(64 * waves), shm, stream, a, coop);
  else if (dh == 20) hipLaunchKernelGGL((mhsa_fwd_kernel<2, 20>), dim3(blocks), dim3(64 * waves), shm, stream, a, coop);
  else hipLaunchKernelGGL((mhsa_fwd_kernel<2, 0>), dim3(blocks), dim3(64 * waves), shm, stream, a, coop);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_mhsa_fwd(const float* qkv, const uint8_t* mask, int n, int Lq, int heads, int dh, float scale, float* out,
                            float* prob, float drop_p, uint32_t seed, hipStream_t stream) {
  return mhsa_fwd_launch(qkv, mask, nullptr, n, Lq, heads, dh, scale, out, prob, drop_p, seed, stream);
}
extern "C" int nnr_mhsa_fwd_packed(const float* qkv, const uint8_t* mask, const int* rowmap, int n, int Lq, int heads, int dh, float scale,
                                   float* out, float drop_p, uint32_t seed, hipStream_t stream) {
  if (!rowmap) return NNR_ERR_ARG;
  return mhsa_fwd_launch(qkv, mask, rowmap, n, Lq, heads, dh, scale, out, nullptr, drop_p, seed, stream);
}

static int mhsa_bwd_launch(const float* qkv, const uint8_t* mask, const int* rowmap, const float* prob, const float* dout, int n, int Lq, int heads,
                           int dh, float scale, float* dqkv, float drop_p, uint32_t seed, hipStream_t stream, const int* pair_off = nullptr) {
  if (!qkv || !dout || !dqkv || n <= 0) return NNR_ERR_ARG;         // prob == NULL: P is recomputed from Q, K
  if (Lq > 64 || dh > 32 || (dh & 1)) return NNR_ERR_UNSUPPORTED;
  if (pair_off && (!rowmap || Lq != 32)) return NNR_ERR_UNSUPPORTED;
  MhsaArgs a{qkv, mask, n, Lq, heads, dh, scale, nullptr, const_cast<float*>(prob), dout, dqkv, seed, nnr_drop_thresh(drop_p),
             drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, rowmap, pair_off};
  const int NB = Lq > 32 ? 2 : 1, LP = 32 * NB, SD = dh | 1;
  const int waves = NB == 1 ? 4 : 1;
  const int coop = (waves == 4 && heads % 4 == 0 && dh % 4 == 0) ? 1 : 0;
  if (rowmap && (!coop || prob)) return NNR_ERR_UNSUPPORTED;
  const size_t shm = ((size_t)waves * (4 * LP * SD + LP * (LP + 1)) + 16) * sizeof(float);
  const int blocks = (n * heads + waves - 1) / waves;
  static const int persist = [] { const char* e = getenv("NNR_MHSA_PERSIST"); return e ? atoi(e) : 1; }();      // A/B: 0 = one 4-head group per workgroup
  if (pair_off && !(coop && !prob && 32 * dh <= 768)) return NNR_ERR_UNSUPPORTED;      // paired titles: the persistent 4-head kernel only
  if ((persist || pair_off) && coop && !prob && 32 * dh <= 768) {
    // one workgroup per sample's heads (heads / 4 groups), or fewer groups when that leaves the chip short of workgroups
    const int ngroups = n * heads / 4;
    int gp = heads / 4;
    while (gp > 1 && (ngroups + gp - 1) / gp < 1024) --gp;
    const dim3 grid((ngroups + gp - 1) / gp);
    if (dh == 20 && Lq == 32) hipLaunchKernelGGL((mhsa_bwd_persist_kernel<20, true>), grid, dim3(256), shm, stream, a, gp);
    else if (dh == 20) hipLaunchKernelGGL((mhsa_bwd_persist_kernel<20, false>), grid, dim3(256), shm, stream, a, gp);
    else hipLaunchKernelGGL((mhsa_bwd_persist_kernel<0, false>), grid, dim3(256), shm, stream, a, gp);
    NNR_CHECK_LAUNCH();
    return NNR_OK;
  }
  if (NB == 1 && dh == 20) hipLaunchKernelGGL((mhsa_bwd_kernel<1, 20>), dim3(blocks), dim3(64 * waves), shm, stream, a, coop);
  else if (NB == 1) hipLaunchKernelGGL((mhsa_bwd_kernel<1, 0>), dim3(blocks), dim3(64 * waves), shm, stream, a, coop);
  else if (dh == 20) hipLaunchKernelGGL((mhsa_bwd_kernel<2, 20>), dim3(blocks), dim3(64 * waves), shm, stream, a, coop);
  else hipLaunchKernelGGL((mhsa_bwd_kernel<2, 0>), dim3(blocks), dim3(64 * waves), shm, stream, a, coop);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_mhsa_bwd(const float* qkv, const uint8_t* mask, const float* prob, const float* dout, int n, int Lq, int heads, int dh,
                            float scale, float* dqkv, float drop_p, uint32_t seed, hipStream_t stream) {
  return mhsa_bwd_launch(qkv, mask, nullptr, prob, dout, n, Lq, heads, dh, scale, dqkv, drop_p, seed, stream);
}
extern "C" int nnr_mhsa_bwd_packed(const float* qkv, const uint8_t* mask, const int* rowmap, const float* dout, int n, int Lq, int heads, int dh,
                                   float scale, float* dqkv, float drop_p, uint32_t seed, hipStream_t stream) {
  if (!rowmap) return NNR_ERR_ARG;
  return mhsa_bwd_launch(qkv, mask, rowmap, nullptr, dout, n, Lq, heads, dh, scale, dqkv, drop_p, seed, stream);
}

// Paired short titles (mhsa_pairing above): vmask / vrowmap [n, 32] from nnr_mhsa_pair_map (csrc/seq_plan.hip), off = the plan's offsets.
extern "C" int nnr_mhsa_fwd_paired(const float* qkv, const uint8_t* vmask, const int* vrowmap, const int* off, int n, int heads, int dh, float scale,
                                   float* out, float drop_p, uint32_t seed, hipStream_t stream) {
  if (!vrowmap || !vmask || !off) return NNR_ERR_ARG;
  return mhsa_fwd_launch(qkv, vmask, vrowmap, n, 32, heads, dh, scale, out, nullptr, drop_p, seed, stream, off);
}
extern "C" int nnr_mhsa_bwd_paired(const float* qkv, const uint8_t* vmask, const int* vrowmap, const int* off, const float* dout, int n, int heads,
                                   int dh, float scale, float* dqkv, float drop_p, uint32_t seed, hipStream_t stream) {
  if (!vrowmap || !vmask || !off) return NNR_ERR_ARG;
  return mhsa_bwd_launch(qkv, vmask, vrowmap, nullptr, dout, n, 32, heads, dh, scale, dqkv, drop_p, seed, stream, off);
}
