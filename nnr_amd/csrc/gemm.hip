// fp32 GEMM for the NNR hot path on gfx950:  C[M,N] = epilogue( A_op[M,K] . B_op[K,N] )
//
// * v_mfma_f32_16x16x4_f32 (exact fp32, bitwise an fmaf chain) -- 157 TFLOP/s roofline.
// * Workgroup = 4 waves; wave w owns rows [w*16*TM, (w+1)*16*TM) x all 16*TN columns of the block tile.
// * BK = 16 per stage; global -> registers -> LDS double buffer, one barrier per stage.
// * K-contiguous operands ("weights" [N,K], activations [M,K]) live in LDS as [row][16] with an XOR swizzle
//   on the 16-byte chunk so a fragment is ONE conflict-free ds_read_b128 feeding 4 MFMAs;
//   K-major operands ([K,M] / [K,N], used by the backward GEMMs) live as [16][R+4] and are read with
//   conflict-free ds_read_b32.
// * Epilogue activations use the hardware transcendentals (v_exp_f32 / v_rcp_f32, common.h: abs error ~2e-7) -- the
//   tanh / sigmoid GEMMs of the attention and gate layers apply them to 30-60 M elements per launch.
// * blockIdx -> tile mapping is XCD-aware: the 8 XCDs each walk a contiguous range of tiles, column blocks
//   fastest, so the A row-panel of a tile row stays in ONE XCD's L2 while its column blocks run.
// * Row gather on A (embedding rows feeding the LSTM input projection) / on B's k-rows (backward weight
//   gradient), counter-based dropout recomputed in the loaders, row scatter with f32 atomics in the epilogue
//   (embedding gradient), split-K with atomics for the token-reduction GEMMs, dynamic token count read from
//   device memory (no host sync).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

template <int BK>
__device__ __forceinline__ int swz(int r16) {  // chunk XOR for row r (0..15) of a 16-row block
  // BK=16 (4 chunks / row): f(r>>2) = {0,3,2,1};  BK=32 (8 chunks / row): (r>>1)&7;  BK=64 (16 chunks = one 256-B bank row
  // per row): r itself.  Each makes the four ds_read_b128 lane groups hit 16 distinct 16-B slots (brute-force checked).
  return BK == 16 ? ((4 - (r16 >> 2)) & 3) : BK == 32 ? ((r16 >> 1) & 7) : (r16 & 15);
}

// ---------------------------------------------------------------- shared epilogue (all tiled kernels of this file)
// Wave w of the 4-wave workgroup holds rows [(w*TM + m)*16, +16) of the block tile in acc[m][*]; `stage` is LDS of at least
// 64 * (16*TN + 4) floats that no wave still reads.
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const nnr_gemm_args& g, f32x4 (&acc)[TM][TN], float* lds, float* __restrict__ C, int m0,
                                              int n0, int M, int N, int z) {
  constexpr int BN = 16 * TN, E_LD = BN + 4;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;
  const uint32_t dthr = g.drop_thresh;
  const float dscale = g.drop_scale;
  // ---------------------------------------------------------------- epilogue
  // Accumulators go through LDS in TM passes of 64 rows (tiny fully-unrolled store loop: the MFMA registers are only
  // ever indexed statically), then a rolled, runtime-flagged loop applies the epilogue with consecutive lanes on
  // consecutive columns: every global access (C, aux, mul, resid, atomics) is a contiguous 256-B wave access.
  const bool use_atomic = !g.slab_mode && (g.atomic || g.split_k > 1 || g.k_chunk > 0);      // (slab mode: a slice stores its tile plainly)
  float* aux = g.aux_out;
  const float* res = g.resid;
  const float* mulp = g.mul;
  if (g.batch > 1 && g.split_k <= 1 && g.k_chunk <= 0) {
    if (aux) aux += (long)z * g.stride_aux;
    if (res) res += (long)z * g.stride_res;
  }
  float* stage = lds;
#pragma unroll
  for (int m = 0; m < TM; ++m) {
    if (m > 0) __syncthreads();
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) stage[(w * 16 + kk * 4 + reg) * E_LD + n * 16 + r] = acc[m][n][reg];
    __syncthreads();
    if (g.rowdot_w) {
      // one wave per row: lanes stride the columns, wave-reduce the w2-weighted sum (needs the whole row in this tile)
      for (int lr = w; lr < 64; lr += 4) {
        const int row = m0 + (lr >> 4) * (TM * 16) + m * 16 + (lr & 15);
        float dot = 0.f;
        if (row < M) {
          for (int c = lane; c < BN; c += 64) {
            const int col = n0 + c;
            if (col < N) {
              float x = stage[lr * E_LD + c] * g.alpha;
              if (g.bias) x += g.bias[col];
              if (g.act == 1) x = fmaxf(x, 0.f);
              else if (g.act == 2) x = fast_tanh(x);
              else if (g.act == 3) x = fast_sigmoid(x);
              if (aux) aux[(long)row * g.ldaux + col] = x;
              dot += g.rowdot_w[col] * x;
              if (C) C[(long)row * g.ldc + col] = x;
            }
          }
        }
        dot = wave_sum(dot);
        if (row < M && lane == 0) g.rowdot_out[row] = dot;
      }
    } else if (g.vec_epi) {
      // float4 path (all operands 16-B aligned, no scatter / atomics): a wave writes 1 KiB contiguous pieces of C rows
      constexpr int NV = BN / 4;
      for (int idx = tid; idx < 64 * NV; idx += 256) {
        const int lr = idx / NV, c4 = idx - lr * NV;
        const int row = m0 + (lr >> 4) * (TM * 16) + m * 16 + (lr & 15);
        const int col = n0 + 4 * c4;
        if (row >= M || col >= N) continue;
        f32x4 x = *reinterpret_cast<const f32x4*>(&stage[lr * E_LD + 4 * c4]) * g.alpha;
        if (g.pre_add) x += *reinterpret_cast<const f32x4*>(g.pre_add + (long)row * g.ldpre + col);
        if (g.gate_bwd) {
          // d(Ht = H * G): this GEMM completes dHt = x; dH = x * G and d pre = x * H * G * (1 - G) leave from the same registers
          const f32x4 gv = *reinterpret_cast<const f32x4*>(mulp + (long)row * g.ldmul + col);
          const f32x4 hv = *reinterpret_cast<const f32x4*>(res + (long)row * g.ldres + col);
          *reinterpret_cast<f32x4*>(aux + (long)row * g.ldaux + col) = x * hv * gv * (1.f - gv);
          *reinterpret_cast<f32x4*>(C + (long)row * g.ldc + col) = x * gv;
          continue;
        }
        if (g.accumulate == 2) x += *reinterpret_cast<const f32x4*>(C + (long)row * g.ldc + col);
        if (g.bias) x += *reinterpret_cast<const f32x4*>(g.bias + col);
        if (g.rowvec) x += *reinterpret_cast<const f32x4*>(g.rowvec + (long)(g.rowvec_map ? g.rowvec_map[row] : row) * g.ldrv + col);
        if (g.act == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = fmaxf(x[e], 0.f);
        } else if (g.act == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = fast_tanh(x[e]);
        } else if (g.act == 3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = fast_sigmoid(x[e]);
        }
        if (aux) *reinterpret_cast<f32x4*>(aux + (long)row * g.ldaux + col) = x;
        if (mulp) x *= *reinterpret_cast<const f32x4*>(mulp + (long)row * g.ldmul + col);
        if (res) x += *reinterpret_cast<const f32x4*>(res + (long)row * g.ldres + col);
        if (g.drop_target == 3) {
          bool kp[4];
          nnr_keep4(g.drop_seed, (uint64_t)(row + (long)z * g.M) * g.drop_cols + col, dthr, kp);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[e] = kp[e] ? x[e] * dscale : 0.f;
        }
        if (C) {
          f32x4* cp = reinterpret_cast<f32x4*>(C + (long)row * g.ldc + col);
          if (g.accumulate == 1) x += *cp;
          *cp = x;
        }
      }
    } else {
      for (int idx = tid; idx < 64 * BN; idx += 256) {
        const int lr = idx / BN, c = idx - lr * BN;
        const int row = m0 + (lr >> 4) * (TM * 16) + m * 16 + (lr & 15);
        const int col = n0 + c;
        if (row >= M || col >= N) continue;
        int crow = row;
        if (g.c_idx) { crow = g.c_idx[row]; if (crow < 0) continue; }
        float x = stage[lr * E_LD + c] * g.alpha;
        if (g.accumulate == 2) x += C[(long)crow * g.ldc + col];      // running sum BEFORE bias / activation
        if (g.bias) x += g.bias[col];
        if (g.rowvec) x += g.rowvec[(long)(g.rowvec_map ? g.rowvec_map[row] : row) * g.ldrv + col];
        if (g.act == 1) x = fmaxf(x, 0.f);
        else if (g.act == 2) x = fast_tanh(x);
        else if (g.act == 3) x = fast_sigmoid(x);
        if (aux) aux[(long)row * g.ldaux + col] = x;
        if (mulp) x *= mulp[(long)row * g.ldmul + col];
        if (res) x += res[(long)row * g.ldres + col];
        if (g.drop_target == 3)
          x = nnr_keep(g.drop_seed, (uint64_t)(row + (long)z * g.M) * g.drop_cols + col, dthr) ? x * dscale : 0.f;
        if (C) {
          float* cp = C + (long)crow * g.ldc + col;
          if (use_atomic) {
            if (g.drop_target == 4)   // scatter of d(dropout(emb)) : mask keyed by the token row
              x = nnr_keep(g.drop_seed, (uint64_t)row * g.drop_cols + col, dthr) ? x * dscale : 0.f;
            atomicAdd(cp, x);
          } else {
            if (g.accumulate == 1) x += *cp;
            *cp = x;
          }
        }
      }
    }
  }
}

template <int TM, int TN, bool TA, bool TB, int BK>
__global__ __launch_bounds__(256) void gemm_kernel(nnr_gemm_args g) {
  constexpr int BM = 64 * TM, BN = 16 * TN;
  constexpr int A_LD = BM + 4, B_LD = BN + 4;
  constexpr int KQ = BK / 4, NKG = BK / 16;      // 16-byte chunks per K-contiguous row; 16-wide k-groups per stage
  constexpr int A_SZ = TA ? BK * A_LD : BM * BK;
  constexpr int B_SZ = TB ? BK * B_LD : BN * BK;
  constexpr int NAL = BM * KQ / 256;             // A float4 loads per thread
  constexpr int NBL = (BN * KQ + 255) / 256;     // B float4 loads per thread
  constexpr int E_LD = BN + 4;                // epilogue staging: 64 rows x BN (+pad) per pass
  constexpr int LDS_FLOATS = (2 * (A_SZ + B_SZ) > 64 * E_LD) ? 2 * (A_SZ + B_SZ) : 64 * E_LD;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 15, kk = lane >> 4;

  // ---- dynamic extents
  int M = g.M, K = g.K;
  if (g.dyn_dim == 1) M = min(M, *g.dyn_dev);
  if (g.dyn_dim == 2) K = min(K, *g.dyn_dev);
  const int N = g.N;

  // ---- XCD-aware tile id over the LIVE tiles only (with a device-side M most of the static grid is empty; the
  //      remap must spread the live row-panels over all 8 XCDs, not pack them onto the first ones)
  const int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN;
  const int nblk = nbm * nbn;
  if ((int)blockIdx.x >= nblk) return;
  int v;
  {
    const int b = blockIdx.x, q = nblk >> 3, rem = nblk & 7, x = b & 7, slot = b >> 3;
    v = x * q + min(x, rem) + slot;
  }
  const int bm = v / nbn, bn = v - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  // ---- batch / split-K
  const float* __restrict__ A = g.A;
  const float* __restrict__ B = g.B;
  float* __restrict__ C = g.C;
  int kbeg = 0, kend = K;
  const int z = blockIdx.z;
  if (g.k_chunk > 0) {
    // fixed-size reduction slices: the slice count follows the LIVE token count (device side), the grid covers capacity
    kbeg = z * g.k_chunk;
    kend = min(K, kbeg + g.k_chunk);
    if (kbeg >= kend) return;
  } else if (g.split_k > 1) {
    const int ktiles = (K + BK - 1) / BK;
    const int per = (ktiles + g.split_k - 1) / g.split_k;
    kbeg = z * per * BK;
    kend = min(K, kbeg + per * BK);
    if (kbeg >= kend) return;
  } else if (g.batch > 1) {
    A += (long)z * g.strideA;
    B += (long)z * g.strideB;
    C += (long)z * g.strideC;
  }
  const bool vecA = ((g.lda & 3) == 0) && ((((uintptr_t)A) & 15) == 0);
  const bool vecB = ((g.ldb & 3) == 0) && ((((uintptr_t)B) & 15) == 0);
  const uint32_t dthr = g.drop_thresh;
  const float dscale = g.drop_scale;

  // ---- per-thread load coordinates that do not change over k
  long a_src[NAL];   // TA=false: source row offset (elements) or -1
  if (!TA) {
#pragma unroll
    for (int j = 0; j < NAL; ++j) {
      const int f = tid + 256 * j, row = f / KQ, gm = m0 + row;
      long s = -1;
      if (gm < M) {
        int src = g.a_idx ? g.a_idx[gm] : gm;
        if (src >= 0) s = (long)src * g.lda;
      }
      a_src[j] = s;
    }
  }

  f32x4 ra[NAL];
  f32x4 rb[NBL];
  // optional fused bias gradient (trans_a only): colsum_out[m] += sum_k A[k][m], taken from the A^T tile registers of
  // the column-block-0 workgroups -- the weight-gradient GEMM streams exactly the tensor whose column sums are db
  const bool do_colsum = TA && g.colsum_out != nullptr && bn == 0;
  f32x4 csum[NAL];
#pragma unroll
  for (int j = 0; j < NAL; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto load_tiles = [&](int k0) {
    // ---------------- A
#pragma unroll
    for (int j = 0; j < NAL; ++j) {
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      const int f = tid + 256 * j;
      if (!TA) {
        const int kq = f % KQ, k = k0 + 4 * kq;
        if (a_src[j] >= 0 && k < kend) {
          const float* p = A + a_src[j] + k;
          if (vecA && k + 3 < kend) {
            val = *reinterpret_cast<const f32x4*>(p);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (k + e < kend) val[e] = p[e];
          }
          if (g.drop_target == 1) {
            const int gm = m0 + f / KQ;
            bool kp[4];
            if ((g.drop_cols & 3) == 0) nnr_keep4(g.drop_seed, (uint64_t)gm * g.drop_cols + k, dthr, kp);
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e) kp[e] = nnr_keep(g.drop_seed, (uint64_t)gm * g.drop_cols + (k + e), dthr);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) val[e] = kp[e] ? val[e] * dscale : 0.f;
          }
        }
      } else {
        const int kr = f / (BM / 4), mq = f - kr * (BM / 4);
        const int gk = k0 + kr, gm = m0 + 4 * mq;
        if (gk < kend && gm < M) {
          const float* p = A + (long)gk * g.lda + gm;
          if (vecA && gm + 3 < M) {
            val = *reinterpret_cast<const f32x4*>(p);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (gm + e < M) val[e] = p[e];
          }
        }
      }
      ra[j] = val;
    }
    // ---------------- B
#pragma unroll
    for (int j = 0; j < NBL; ++j) {
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      const int f = tid + 256 * j;
      if (f < BN * KQ || TB) {
        if (!TB) {
          const int row = f / KQ, kq = f % KQ, gn = n0 + row, k = k0 + 4 * kq;
          if (gn < N && k < kend) {
            const float* p = B + (long)gn * g.ldb + k;
            if (vecB && k + 3 < kend) {
              val = *reinterpret_cast<const f32x4*>(p);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) if (k + e < kend) val[e] = p[e];
            }
          }
        } else {
          const int kr = f / (BN / 4), nq = f - kr * (BN / 4);
          const int gk = k0 + kr, gn = n0 + 4 * nq;
          if (kr < BK && gk < kend && gn < N) {
            int src = g.b_idx ? g.b_idx[gk] : gk;
            if (src >= 0) {
              const float* p = B + (long)src * g.ldb + gn;
              if (vecB && gn + 3 < N) {
                val = *reinterpret_cast<const f32x4*>(p);
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (gn + e < N) val[e] = p[e];
              }
              if (g.drop_target == 2) {
                bool kp[4];
                if ((g.drop_cols & 3) == 0) nnr_keep4(g.drop_seed, (uint64_t)gk * g.drop_cols + gn, dthr, kp);
                else {
#pragma unroll
                  for (int e = 0; e < 4; ++e) kp[e] = nnr_keep(g.drop_seed, (uint64_t)gk * g.drop_cols + (gn + e), dthr);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) val[e] = kp[e] ? val[e] * dscale : 0.f;
              }
            }
          }
        }
      }
      rb[j] = val;
    }
  };

  auto store_tiles = [&](int buf) {
    float* As = lds + buf * (A_SZ + B_SZ);
    float* Bs = As + A_SZ;
#pragma unroll
    for (int j = 0; j < NAL; ++j) {
      const int f = tid + 256 * j;
      if (!TA) {
        const int row = f / KQ, kq = f % KQ;
        *reinterpret_cast<f32x4*>(&As[row * BK + 4 * (kq ^ swz<BK>(row & 15))]) = ra[j];
      } else {
        const int kr = f / (BM / 4), mq = f - kr * (BM / 4);
        *reinterpret_cast<f32x4*>(&As[kr * A_LD + 4 * mq]) = ra[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NBL; ++j) {
      const int f = tid + 256 * j;
      if (f < BN * KQ || TB) {
        if (!TB) {
          const int row = f / KQ, kq = f % KQ;
          *reinterpret_cast<f32x4*>(&Bs[row * BK + 4 * (kq ^ swz<BK>(row & 15))]) = rb[j];
        } else {
          const int kr = f / (BN / 4), nq = f - kr * (BN / 4);
          if (kr < BK) *reinterpret_cast<f32x4*>(&Bs[kr * B_LD + 4 * nq]) = rb[j];
        }
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_tiles(kbeg);
  if (do_colsum) {
#pragma unroll
    for (int j = 0; j < NAL; ++j) csum[j] += ra[j];
  }
  store_tiles(0);
  __syncthreads();

  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = (k0 + BK) < kend;
    if (more) {
      load_tiles(k0 + BK);
      if (do_colsum) {
#pragma unroll
        for (int j = 0; j < NAL; ++j) csum[j] += ra[j];
      }
    }

    const float* As = lds + buf * (A_SZ + B_SZ);
    const float* Bs = As + A_SZ;
#pragma unroll
    for (int kg2 = 0; kg2 < NKG; ++kg2) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        if (!TA) {
          af[m] = *reinterpret_cast<const f32x4*>(&As[((w * TM + m) * 16 + r) * BK + 4 * ((kg2 * 4 + kk) ^ swz<BK>(r))]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) af[m][i] = As[(kg2 * 16 + 4 * kk + i) * A_LD + (w * TM + m) * 16 + r];
        }
      }
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        if (!TB) {
          bf[n] = *reinterpret_cast<const f32x4*>(&Bs[(n * 16 + r) * BK + 4 * ((kg2 * 4 + kk) ^ swz<BK>(r))]);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) bf[n][i] = Bs[(kg2 * 16 + 4 * kk + i) * B_LD + n * 16 + r];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][i], bf[n][i], acc[m][n], 0, 0, 0);
    }

    if (more) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  if (do_colsum) {
    // thread (kr, mq) holds partial sums for columns 4*mq..4*mq+3; threads with equal mq differ in kr (0 .. BK-1): every partial goes
    // to its own LDS word [kr][BM] (the tile buffers are free now: BK x BM floats <= A_SZ) and is summed over kr in FIXED order -- LDS
    // float atomics here would add in arrival order, i.e. round differently from run to run
    float* red = lds;
#pragma unroll
    for (int j = 0; j < NAL; ++j) {
      const int f = tid + 256 * j, kr = f / (BM / 4), mq = f - kr * (BM / 4);
      *reinterpret_cast<f32x4*>(&red[kr * BM + 4 * mq]) = csum[j];
    }
    __syncthreads();
    for (int i = tid; i < BM; i += 256) {
      float t = 0.f;
      for (int kr = 0; kr < BK; ++kr) t += red[kr * BM + i];
      if (m0 + i < M) {
        if (g.slab_mode) g.colsum_out[(long)z * M + m0 + i] = t;      // slice z's own row of the column-sum slab
        else atomicAdd(&g.colsum_out[m0 + i], t);
      }
    }
    __syncthreads();
  }

  gemm_epilogue<TM, TN>(g, acc, lds, g.slab_mode ? C + (long)z * M * N : C, m0, n0, M, N, z);
}

template <int TM, int TN, int BK>
int launch_cfg(const nnr_gemm_args& g, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 16 * TN;
  const int nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
  if (g.rowdot_w && nbn != 1) return NNR_ERR_ARG;
  dim3 grid(nbm * nbn, 1, g.k_chunk > 0 ? (g.K + g.k_chunk - 1) / g.k_chunk : (g.split_k > 1 ? g.split_k : (g.batch > 1 ? g.batch : 1)));
  dim3 block(256);
  if (!g.trans_a && !g.trans_b) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, false, BK>), grid, block, 0, s, g);
  else if (!g.trans_a && g.trans_b) hipLaunchKernelGGL((gemm_kernel<TM, TN, false, true, BK>), grid, block, 0, s, g);
  else if (g.trans_a && g.trans_b) hipLaunchKernelGGL((gemm_kernel<TM, TN, true, true, BK>), grid, block, 0, s, g);
  else return NNR_ERR_ARG;   // (A^T, B[N,K]) never occurs on this path
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}


// ------------------------------------------------------------------------------------------------ pipelined NT GEMM (LDS-DMA)
// C[M,N] = epilogue(A[M,K] . B[N,K]^T), both operands K-contiguous -- the forward GEMMs, and every data-gradient GEMM once the
// (small) weight has been transposed.  Same tile / wave / fragment layout as gemm_kernel, different staging:
//  * global -> LDS by LDS-DMA (`global_load_lds_dwordx4`: 64 lanes x 16 B = 1 KiB straight into LDS, no staging registers,
//    no ds_write pass).  The LDS image of a stage is LINEAR in (row, 16-byte position); the XOR swizzle that makes the
//    fragment reads conflict-free is applied to the per-lane SOURCE address (position p of row r holds k-chunk p ^ swz(r))
//    and again on the ds_read_b128 side -- the same involution on both sides.
//  * NS stage buffers; the loads of stage s + NS - 1 are issued right after the barrier that opens stage s, so NS - 1
//    stages (2-3 us of MFMA work) cover the L2 / HBM round trip even when only one or two workgroups share the CU -- the
//    regime of the mid-size GEMMs (SUE: 4 352 x 900 x 900) and of the tail of every launch, where the two-buffer
//    register-staged kernel exposes one round trip per 16-deep stage.
//  * one `s_waitcnt vmcnt(n)` + one raw `s_barrier` per stage: the wait retires this wave's own DMAs of the stage about to
//    be read (the newer stages stay in flight ACROSS the barrier), the barrier makes every wave's DMAs of that stage visible
//    and doubles as the write-after-read fence for the buffer that is refilled next.
//  * rows beyond M / N and k-chunks beyond K read a zero page instead (the source address is per lane), so the tile needs
//    no bounds logic in the loop.  Requirements (checked by the dispatcher): K, lda, ldb multiples of 4, 16-byte aligned bases.
__device__ __attribute__((aligned(1024))) float nnr_zero_page[512] = {};      // 2 KiB: a whole (gathered) tile row of zeros

__device__ __forceinline__ void lds_dma16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;      // M0 carries the wave-uniform LDS base of the instruction; it is compiler-reserved: saved and restored here
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void set_prio_dyn(unsigned p) {      // s_setprio takes an immediate
  switch (p & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wait until at most AHEAD stages of CNT instructions each (this wave's own DMAs) are outstanding, AHEAD = min(MAXA, ahead)
template <int CNT, int MAXA>
__device__ __forceinline__ void wait_stages(int ahead) {
  if constexpr (MAXA == 0) wait_vmcnt<0>();
  else {
    if (ahead >= MAXA) wait_vmcnt<CNT * MAXA>();
    else wait_stages<CNT, MAXA - 1>(ahead);
  }
}

template <int TM, int TN, int BK, int NS, int OCC, int PRIO = 0>
__global__ __launch_bounds__(256, OCC) void gemm_nt_pipe_kernel(nnr_gemm_args g) {
  // PRIO 1: static, distinct wave priorities for the workgroups that (most likely) share a CU.  With equal priorities the SIMD
  // interleaves the MFMAs of its resident waves instruction by instruction, so they all reach their per-stage wait + barrier +
  // fragment-read phase together and the matrix pipe idles through it; with distinct priorities one wave runs its stage at full
  // rate while the others queue, and the phases stay staggered.
  if (PRIO == 1) set_prio_dyn(blockIdx.x >> 8);
  if (PRIO == 2) set_prio_dyn(blockIdx.x >> 5);
  if (PRIO == 3) set_prio_dyn(blockIdx.x);
  constexpr int BM = 64 * TM, BN = 16 * TN, ROWS = BM + BN;
  constexpr int KQ = BK / 4, NKG = BK / 16;             // 16-byte chunks per row; 16-deep k-groups per stage
  constexpr int RPI = 64 / KQ;                          // tile rows one DMA instruction covers (1 KiB)
  constexpr int NI = ROWS / RPI;                        // DMA instructions per stage
  constexpr int NPW = (NI + 3) / 4;                     // ... per wave (wave w issues instructions w, w + 4, ...)
  constexpr int STAGE = ROWS * BK;                      // floats
  constexpr int E_LD = BN + 4;
  constexpr int LDS_FLOATS = (NS * STAGE > 64 * E_LD) ? NS * STAGE : 64 * E_LD;
  static_assert(ROWS % RPI == 0, "tile rows must fill whole DMA instructions");
  __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kk = lane >> 4;

  int M = g.M;
  if (g.dyn_dim == 1) M = min(M, *g.dyn_dev);
  const int N = g.N, K = g.K;
  const int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN;
  const int nblk = nbm * nbn;
  if ((int)blockIdx.x >= nblk) return;
  int v;
  {
    const int b = blockIdx.x, q = nblk >> 3, rem = nblk & 7, x = b & 7, slot = b >> 3;
    v = x * q + min(x, rem) + slot;
  }
  const int bm = v / nbn, bn = v - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int z = blockIdx.z;
  const float* __restrict__ A = g.A;
  const float* __restrict__ B = g.B;
  float* __restrict__ C = g.C;
  if (g.batch > 1) {
    A += (long)z * g.strideA;
    B += (long)z * g.strideB;
    C += (long)z * g.strideC;
  }

  // ---- per-lane DMA sources: instruction q covers tile rows [q*RPI, +RPI) (A rows first, then B rows); this lane feeds
  //      position (lane % KQ) of row q*RPI + lane / KQ, i.e. k-chunk (lane % KQ) ^ swz(row & 15) of that row
  const float* rowp[NPW];
  int kch[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = w + 4 * i;
    const int tr = q * RPI + lane / KQ;
    const int c = (lane % KQ) ^ swz<BK>(tr & 15);
    kch[i] = 4 * c;
    const float* p = nullptr;
    if (q < NI) {
      if (tr < BM) {
        const int gm = m0 + tr;
        if (gm < M) {
          const int src = g.a_idx ? g.a_idx[gm] : gm;
          if (src >= 0) p = A + (long)src * g.lda + 4 * c;
        }
      } else {
        const int gn = n0 + (tr - BM);
        if (gn < N) p = B + (long)gn * g.ldb + 4 * c;
      }
    }
    rowp[i] = p;
  }
  const float* zero = nnr_zero_page;
  asm volatile("" : "+s"(zero));                       // pinned in SGPRs: the compiler would re-fetch the symbol's address per use
  const unsigned lds_base = (unsigned)(uintptr_t)lds;      // low 32 bits of a flat LDS address = the LDS byte offset
  const int S = (K + BK - 1) / BK;

  auto issue = [&](int s) {
    const int k0 = s * BK;
    const unsigned sb = lds_base + (unsigned)((s % NS) * STAGE * 4);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int q = w + 4 * i;
      if (q < NI) {
        const float* src = (rowp[i] != nullptr && k0 + kch[i] < K) ? rowp[i] + k0 : zero;
        lds_dma16(src, sb + (unsigned)(q * 1024));
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < S) issue(s);

  // Ragged last column block (round 5): N = 200 / 300 / 900 are 2.5 / 3.75 / 11.25 tiles of 80 columns; the 16-column MFMA tiles of the last block that lie
  // wholly beyond N (2 of 5 at N = 200: 13 % of the launch's matrix work; 1 of 5 at N = 300, 3 of 5 at N = 900: 5 %) only ever multiplied clamped rows into
  // columns that are never stored.  Those workgroups take a second instantiation of the stage loop that skips them (uniform branches per tile); every
  // accumulator that IS stored sees the same MFMAs in the same order: results are bit-identical.
  const int nv = (g.sched & 1) ? min(TN, (N - n0 + 15) >> 4) : TN;      // (g.sched bit 0: set by the launcher unless NNR_RAGGED=0 -- A/B)
  auto run = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    for (int s = 0; s < S; ++s) {
      // this wave's DMAs of stage s have landed once at most `ahead` newer stages of its own are still outstanding
      const int ahead = S - 1 - s;
      if (NI % 4 == 0 || w < NI % 4) wait_stages<NPW, NS - 2>(ahead);
      else wait_stages<NPW - 1, NS - 2>(ahead);
      __builtin_amdgcn_s_barrier();
      if (s + NS - 1 < S) issue(s + NS - 1);              // into the buffer every wave finished reading before that barrier

      const float* As = lds + (s % NS) * STAGE;
      const float* Bs = As + BM * BK;
#pragma unroll
      for (int kg = 0; kg < NKG; ++kg) {
        f32x4 af[TM], bf[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
          af[m] = *reinterpret_cast<const f32x4*>(&As[((w * TM + m) * 16 + r) * BK + 4 * ((kg * 4 + kk) ^ swz<BK>(r))]);
        if constexpr (FULL) {
#pragma unroll
          for (int n = 0; n < TN; ++n)
            bf[n] = *reinterpret_cast<const f32x4*>(&Bs[(n * 16 + r) * BK + 4 * ((kg * 4 + kk) ^ swz<BK>(r))]);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
              for (int n = 0; n < TN; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][i], bf[n][i], acc[m][n], 0, 0, 0);
        } else {
#pragma unroll
          for (int n = 0; n < TN; ++n)
            if (n < nv) bf[n] = *reinterpret_cast<const f32x4*>(&Bs[(n * 16 + r) * BK + 4 * ((kg * 4 + kk) ^ swz<BK>(r))]);
#pragma unroll
          for (int n = 0; n < TN; ++n)
            if (n < nv) {
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int m = 0; m < TM; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m][i], bf[n][i], acc[m][n], 0, 0, 0);
            }
        }
      }
    }
  };
  if (nv == TN) run(std::true_type{});
  else run(std::false_type{});
  __syncthreads();        // every wave is done with the stage buffers: the epilogue reuses them
  gemm_epilogue<TM, TN>(g, acc, lds, C, m0, n0, M, N, z);
}

// ---- lean LDS-DMA issue: N loads of one operand in ONE statement.  The global address is SGPR base (advanced by the caller per
// stage) + a per-lane 32-bit byte offset that never changes, so a stage costs no vector ALU work at all; M0 (the LDS base) is
// saved once, stepped by `step` bytes between the loads and restored.  (s_nop 0: the wait state between an M0 write and its use.)
template <int N>
__device__ __forceinline__ void lds_dma16_block(const float* sbase, unsigned lds_dst, const unsigned (&voff)[8]) {
  unsigned keep;
  static_assert(N >= 1 && N <= 8, "1..8 loads per block");
#define NNR_DMA_HEAD "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\t"
#define NNR_DMA_LD(I) "s_nop 0\n\tglobal_load_lds_dwordx4 %[v" #I "], %[sb]\n\ts_add_u32 m0, m0, 0x1000\n\t"
#define NNR_DMA_TAIL "s_mov_b32 m0, %[keep]"
#define NNR_DMA_OPS : [keep] "=&s"(keep) : [dst] "s"(lds_dst), [sb] "s"(sbase), [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]), [v3] "v"(voff[3]), \
                      [v4] "v"(voff[4]), [v5] "v"(voff[5]), [v6] "v"(voff[6]), [v7] "v"(voff[7]) : "memory", "scc"
  if constexpr (N == 1) asm volatile(NNR_DMA_HEAD NNR_DMA_LD(0) NNR_DMA_TAIL NNR_DMA_OPS);
  if constexpr (N == 2) asm volatile(NNR_DMA_HEAD NNR_DMA_LD(0) NNR_DMA_LD(1) NNR_DMA_TAIL NNR_DMA_OPS);
  if constexpr (N == 3) asm volatile(NNR_DMA_HEAD NNR_DMA_LD(0) NNR_DMA_LD(1) NNR_DMA_LD(2) NNR_DMA_TAIL NNR_DMA_OPS);
  if constexpr (N == 4) asm volatile(NNR_DMA_HEAD NNR_DMA_LD(0) NNR_DMA_LD(1) NNR_DMA_LD(2) NNR_DMA_LD(3) NNR_DMA_TAIL NNR_DMA_OPS);
  if constexpr (N == 5) asm volatile(NNR_DMA_HEAD NNR_DMA_LD(0) NNR_DMA_LD(1) NNR_DMA_LD(2) NNR_DMA_LD(3) NNR_DMA_LD(4) NNR_DMA_TAIL NNR_DMA_OPS);
  if constexpr (N == 6) asm volatile(NNR_DMA_HEAD NNR_DMA_LD(0) NNR_DMA_LD(1) NNR_DMA_LD(2) NNR_DMA_LD(3) NNR_DMA_LD(4) NNR_DMA_LD(5) NNR_DMA_TAIL NNR_DMA_OPS);
  if constexpr (N == 7) asm volatile(NNR_DMA_HEAD NNR_DMA_LD(0) NNR_DMA_LD(1) NNR_DMA_LD(2) NNR_DMA_LD(3) NNR_DMA_LD(4) NNR_DMA_LD(5) NNR_DMA_LD(6) NNR_DMA_TAIL NNR_DMA_OPS);
  if constexpr (N == 8) asm volatile(NNR_DMA_HEAD NNR_DMA_LD(0) NNR_DMA_LD(1) NNR_DMA_LD(2) NNR_DMA_LD(3) NNR_DMA_LD(4) NNR_DMA_LD(5) NNR_DMA_LD(6) NNR_DMA_LD(7) NNR_DMA_TAIL NNR_DMA_OPS);
#undef NNR_DMA_HEAD
#undef NNR_DMA_LD
#undef NNR_DMA_TAIL
#undef NNR_DMA_OPS
}

// ---- second-generation NT loop: the same LDS-DMA staging, BK = 32, with the fragment reads software-pipelined ACROSS the stage
// barrier.  A stage is two 16-deep k-groups; the fragments of group 1 are read while group 0's MFMAs run, and the wait + barrier
// + DMA issue + fragment reads of the NEXT stage's group 0 sit between the two MFMA blocks of the current stage -- so a single
// wave keeps the matrix pipe fed (no read latency in front of any MFMA block, the barrier costs only the skew between the
// four waves).  The first-generation loop above needs 3-4 co-resident workgroups per CU to hide those gaps; this one is
// meant for 1-2 (bigger tiles, deeper prefetch).
template <int TM, int TN, int NS, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_nt_pipe2_kernel(nnr_gemm_args g) {
  constexpr int BK = 32, BM = 64 * TM, BN = 16 * TN, ROWS = BM + BN;
  constexpr int KQ = BK / 4, RPI = 64 / KQ, NI = ROWS / RPI, NPW = (NI + 3) / 4, STAGE = ROWS * BK, E_LD = BN + 4;
  constexpr int LDS_FLOATS = (NS * STAGE > 64 * E_LD) ? NS * STAGE : 64 * E_LD;
  static_assert(ROWS % RPI == 0 && NS >= 3, "tile shape");
  __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kk = lane >> 4;
  int M = g.M;
  if (g.dyn_dim == 1) M = min(M, *g.dyn_dev);
  const int N = g.N, K = g.K;
  const int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN;
  const int nblk = nbm * nbn;
  if ((int)blockIdx.x >= nblk) return;
  int v;
  {
    const int b = blockIdx.x, q = nblk >> 3, rem = nblk & 7, x = b & 7, slot = b >> 3;
    v = x * q + min(x, rem) + slot;
  }
  const int bm = v / nbn, bn = v - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int z = blockIdx.z;
  const float* __restrict__ A = g.A;
  const float* __restrict__ B = g.B;
  float* __restrict__ C = g.C;
  if (g.batch > 1) {
    A += (long)z * g.strideA;
    B += (long)z * g.strideB;
    C += (long)z * g.strideC;
  }
  // ---- DMA geometry.  Instruction q covers tile rows [q*8, +8): A rows for q < NIA, then B rows; wave w issues q = w, w + 4, ...
  // i.e. its i-th A instruction is q = w + 4i (all waves have NIA / 4 of them) and its j-th B instruction q = NIA + w + 4j.
  // Per lane and instruction a constant byte offset from the operand's stage base: (row * ld + chunk) -- rows past the matrix
  // edge are CLAMPED to the last valid row: they only feed output rows / columns that are never stored.  Only the k-tail needs
  // real zeros; it goes through the per-lane-pointer path with the zero page (last stage, K % 32 != 0 only).
  constexpr int NIA = BM / RPI, NIB = BN / RPI, NA = NIA / 4, NBMAX = (NIB + 3) / 4;
  static_assert(NIA % 4 == 0 && NA <= 8 && NBMAX <= 8, "tile shape");
  unsigned voffA[8], voffB[8];
  int kchA[NA], kchB[NBMAX];
#pragma unroll
  for (int i = 0; i < 8; ++i) voffA[i] = voffB[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int tr = (w + 4 * i) * RPI + lane / KQ;
    const int c = (lane % KQ) ^ swz<BK>(tr & 15);
    kchA[i] = 4 * c;
    voffA[i] = (unsigned)(((long)min(tr, M - 1 - m0) * g.lda + 4 * c) * 4);
  }
#pragma unroll
  for (int j = 0; j < NBMAX; ++j) {
    const int tr = (w + 4 * j) * RPI + lane / KQ;           // row inside the B part of the tile
    const int c = (lane % KQ) ^ swz<BK>(tr & 15);
    kchB[j] = 4 * c;
    voffB[j] = (unsigned)(((long)min(tr, N - 1 - n0) * g.ldb + 4 * c) * 4);
  }
  const float* zero = nnr_zero_page;
  asm volatile("" : "+s"(zero));
  const unsigned lds_base = (unsigned)(uintptr_t)lds;
  const int S = (K + BK - 1) / BK;
  const float* Abase = A + (long)m0 * g.lda;               // wave-uniform (SGPR) stage bases, advanced by BK floats per stage
  const float* Bbase = B + (long)n0 * g.ldb;
  const bool ktail = (K % BK) != 0;
  const bool nb_hi = (NIB % 4 == 0) || (w < NIB % 4);      // this wave has NBMAX (else NBMAX - 1) B instructions
  auto issue_lean = [&](int s) {                             // any stage but a k-tail one: no vector ALU work
    const int k0 = s * BK;
    const unsigned sb = lds_base + (unsigned)((s % NS) * STAGE * 4) + (unsigned)(w * 1024);
    lds_dma16_block<NA>(Abase + k0, sb, voffA);
    if (nb_hi) lds_dma16_block<NBMAX>(Bbase + k0, sb + NIA * 1024, voffB);
    else if constexpr (NBMAX > 1) lds_dma16_block<NBMAX - 1>(Bbase + k0, sb + NIA * 1024, voffB);
  };
  auto issue = [&](int s) {
    if (!(ktail && s == S - 1)) { issue_lean(s); return; }
    const int k0 = s * BK;                                     // zero the k-chunks beyond K (per-lane source select)
    const unsigned sb = lds_base + (unsigned)((s % NS) * STAGE * 4) + (unsigned)(w * 1024);
#pragma unroll
    for (int i = 0; i < NA; ++i)
      lds_dma16((k0 + kchA[i] < K) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(Abase + k0) + voffA[i]) : zero, sb + i * 4096);
#pragma unroll
    for (int j = 0; j < NBMAX; ++j)
      if (j < NBMAX - 1 || nb_hi)
        lds_dma16((k0 + kchB[j] < K) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(Bbase + k0) + voffB[j]) : zero,
                  sb + NIA * 1024 + j * 4096);
  };
  auto wait_landed = [&](int ahead) {       // this wave's DMAs of the stage about to be read: at most `ahead` newer stages stay in flight
    if (NI % 4 == 0 || w < NI % 4) wait_stages<NPW, NS - 3>(ahead);
    else wait_stages<NPW - 1, NS - 3>(ahead);
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  // Ragged last column block / reduction tail (round 5; see gemm_nt_pipe_kernel): the workgroups of a last column block with nv < TN live 16-column tiles run a
  // second instantiation of the loop that skips the dead tiles; the last stage only multiplies the 16-deep k-groups / MFMA steps that hold a k < K (K = 900: 4 live k in the last stage, its second group is skipped).
  // Skipped MFMAs only ever added exact zeros / fed columns that are never stored: results are bit-identical.
  const int nv = (g.sched & 1) ? min(TN, (N - n0 + 15) >> 4) : TN;
  const int klast = (g.sched & 1) ? K - (S - 1) * BK : BK;         // valid reduction depth of the last stage (1 .. 32)
  auto run = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    auto rd = [&](int s, int kg, f32x4 (&a)[TM], f32x4 (&b)[TN]) {
      const float* As = lds + (s % NS) * STAGE;
      const float* Bs = As + BM * BK;
#pragma unroll
      for (int m = 0; m < TM; ++m)
        a[m] = *reinterpret_cast<const f32x4*>(&As[((w * TM + m) * 16 + r) * BK + 4 * ((kg * 4 + kk) ^ swz<BK>(r))]);
#pragma unroll
      for (int n = 0; n < TN; ++n)
        if (FULL || n < nv) b[n] = *reinterpret_cast<const f32x4*>(&Bs[(n * 16 + r) * BK + 4 * ((kg * 4 + kk) ^ swz<BK>(r))]);
    };
    auto mm = [&](const f32x4 (&a)[TM], const f32x4 (&b)[TN], int steps) {      // steps: 4-deep k-steps of this 16-deep group to multiply (4 everywhere but in the tail)
      if constexpr (FULL) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < steps) {
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
              for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][i], b[n][i], acc[m][n], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int n = 0; n < TN; ++n)
          if (n < nv) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (i < steps) {
#pragma unroll
                for (int m = 0; m < TM; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][i], b[n][i], acc[m][n], 0, 0, 0);
              }
          }
      }
    };
    auto mm4 = [&](const f32x4 (&a)[TM], const f32x4 (&b)[TN]) {               // a whole 16-deep group: no step test in the steady-state body
      if constexpr (FULL) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][i], b[n][i], acc[m][n], 0, 0, 0);
      } else {
        mm(a, b, 4);
      }
    };

#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < S) issue(s);
    // stage 0 must be complete: with min(NS - 1, S) stages issued, the newer ones (at most NS - 2) may stay in flight
    if (NI % 4 == 0 || w < NI % 4) wait_stages<NPW, NS - 2>(S - 1); else wait_stages<NPW - 1, NS - 2>(S - 1);
    __builtin_amdgcn_s_barrier();
    rd(0, 0, fa0, fb0);
    // Three loops, so that the steady-state body is branch-free between the fragment reads and the MFMA blocks (any branch that
    // merges there makes the compiler's lgkmcnt bookkeeping wait for the NEWEST reads): (1) stages whose refill is an ordinary
    // stage, (2) the one whose refill is the last stage (possibly a k-tail), (3) the drain, no refill; the last stage is peeled.
    // lgkmcnt(0), visible to the compiler (a builtin, not asm): scalar loads of kernel arguments still pending at the loop header
    // would otherwise force EVERY in-loop LDS wait to lgkmcnt(0) (scalar loads return out of order), i.e. to wait for the reads
    // just issued instead of the older ones the MFMAs need.
    __builtin_amdgcn_s_waitcnt(0xC07F);
    // After every MFMA block an lgkmcnt(0) the compiler can see: the reads it covers were issued a whole MFMA block earlier, so it
    // never stalls, and it empties the compiler's pending-read list before the barrier / DMA / branch section -- where that list
    // otherwise degrades to "wait for everything" at the next use.
#define NNR_LGKM0() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); } while (0)
    int s = 0;
    for (; s + NS < S; ++s) {                                   // refill = stage s + NS - 1 <= S - 2
      rd(s, 1, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);                        // keep the reads AHEAD of the MFMA block that hides them
      mm4(fa0, fb0);
      NNR_LGKM0();
      wait_landed(NS);                                          // stage s + 1: up to NS - 3 newer stages stay in flight
      __builtin_amdgcn_s_barrier();
      issue_lean(s + NS - 1);                                   // into the buffer of stage s - 1: free since the previous barrier
      rd(s + 1, 0, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      mm4(fa1, fb1);
      NNR_LGKM0();
    }
    for (; s + 1 < S; ++s) {                                    // at most NS - 1 iterations
      rd(s, 1, fa1, fb1);
      __builtin_amdgcn_sched_barrier(0);
      mm4(fa0, fb0);
      NNR_LGKM0();
      wait_landed(S - 1 - (s + 1));
      __builtin_amdgcn_s_barrier();
      if (s + NS - 1 < S) issue(s + NS - 1);
      rd(s + 1, 0, fa0, fb0);
      __builtin_amdgcn_sched_barrier(0);
      mm4(fa1, fb1);
      NNR_LGKM0();
    }
#undef NNR_LGKM0
    rd(S - 1, 1, fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    // (MFMA step i of a 16-deep group multiplies k = 16 kg + 4 kk + i over the four lane groups kk: it holds a live k iff 16 kg + i < klast)
    mm(fa0, fb0, min(4, klast));
    if (klast > 16) mm(fa1, fb1, min(4, klast - 16));
  };
  if (nv == TN) run(std::true_type{});
  else run(std::false_type{});
  __syncthreads();
  gemm_epilogue<TM, TN>(g, acc, lds, C, m0, n0, M, N, z);
}

static int nt_ragged_bit() {
  static const int on = [] { const char* e = getenv("NNR_RAGGED"); return (e && atoi(e) == 0) ? 0 : 1; }();      // A/B: 0 = every tile of a ragged last column block / reduction tail is multiplied
  return on;
}

template <int TM, int TN, int NS, int OCC>
int launch_pipe2(const nnr_gemm_args& g0, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 16 * TN;
  nnr_gemm_args g = g0;
  g.sched = nt_ragged_bit();
  const int nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
  dim3 grid(nbm * nbn, 1, g.batch > 1 ? g.batch : 1), block(256);
  hipLaunchKernelGGL((gemm_nt_pipe2_kernel<TM, TN, NS, OCC>), grid, block, 0, s, g);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

template <int TM, int TN, int BK, int NS, int OCC, int PRIO = 0>
int launch_pipe(const nnr_gemm_args& g0, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 16 * TN;
  nnr_gemm_args g = g0;
  g.sched = nt_ragged_bit();
  const int nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
  dim3 grid(nbm * nbn, 1, g.batch > 1 ? g.batch : 1), block(256);
  hipLaunchKernelGGL((gemm_nt_pipe_kernel<TM, TN, BK, NS, OCC, PRIO>), grid, block, 0, s, g);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

// what the pipelined NT kernel accepts (everything else stays on gemm_kernel)
static bool pipe_ok(const nnr_gemm_args& g) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  return !g.trans_a && !g.trans_b && !g.b_idx && g.split_k <= 1 && g.k_chunk <= 0 && !g.rowdot_w && !g.colsum_out &&
         (g.drop_target == 0 || g.drop_target == 3 || g.drop_target == 4) && (g.K & 3) == 0 && (g.lda & 3) == 0 && (g.ldb & 3) == 0 &&
         al(g.A) && al(g.B) && (g.batch <= 1 || (((g.strideA | g.strideB) & 3) == 0)) && (g.dyn_dim == 0 || g.dyn_dim == 1);
}

// ------------------------------------------------------------------------------------------------ NT GEMM on the BF16 matrix pipe (round 5)
// Tile 50, selected only when the caller passes pre-split weights (nnr_amd/ops.py: the default for weight-operand NT launches since round 6; NNR_BX3=0
// keeps the fp32-MFMA kernels).  fp32 arithmetic without narrowing:
// an fp32 value is EXACTLY x1 + x2 + x3 with three bf16 values (8 + 8 + 8 significant bits); the six products a_i b_j with i + j <= 4 carry everything
// above 2^-26 |a b| (below the rounding of an fp32 product), a bf16 x bf16 product is exact in fp32 and v_mfma_f32_16x16x32_bf16 accumulates in
// fp32 -- at 16x the issue rate of v_mfma_f32_16x16x4_f32.  Measured error vs fp64: a third of the fp32-MFMA kernel's (12 roundings of the hi
// accumulator per 400-long dot product instead of 400); tools/micro/bf16x3_gemm.hip, profiles/r05_bf16x3.txt.
//  * B (a weight matrix [N, K]) arrives PRE-SPLIT: three bf16 images [N, ldb3] (nnr_split_bf16x3, once per optimizer step);
//  * A (activations, fp32) is LDS-DMA'd as fp32 and split in registers by the wave that owns the rows, right in front of its MFMAs;
//  * two fp32 accumulators per output block: a1 b1 | the five small terms (added smallest first);
//  * staging = gemm_nt_pipe_kernel's (per-lane-source LDS-DMA, zero page for k-chunks past K, counted vmcnt + one barrier per stage), BK = 32,
//    2 stages x 31.7 KB, 2 workgroups per CU; epilogue = gemm_epilogue (every element-wise feature of the NT kernels).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// Edge behaviour (tests/test_hip_ops_gpu.py::test_gemm_bf16x3_edge_values):
//  * |x| >= 0x7F7F8000 (the top half-ulp below bf16's largest value and everything above it) would ROUND to +-Inf as a bf16: the first image is
//    TRUNCATED there instead, so a finite fp32 value -- FLT_MAX included -- stays three finite images that sum to it exactly;
//  * +-Inf / NaN: first image = the value itself (NaN canonical), the other two 0.  Every output a non-finite operand reaches is NON-FINITE (Inf x b1
//    is +-Inf, Inf x b2 has b2's sign: the partial sums meet as +-Inf or as Inf - Inf = NaN), as with the fp32 MFMA -- but WHICH of +-Inf / NaN is
//    not kept (the training step only asks "is the gradient norm finite", nnr_clip_adam);
//  * tiny values: the three images are exact while the third image's last bit (2^-24 |x|) is representable, i.e. |x| >= 2^-109; below that the
//    bits under bf16's smallest subnormal (2^-133) underflow -- absolute error <= 2^-133 per operand, five decades below fp32's smallest
//    normal value (fp32 denormals themselves are of that size).
__device__ __forceinline__ void split3_bf16(float x, __bf16& h1, __bf16& h2, __bf16& h3) {
  const unsigned u = __builtin_bit_cast(unsigned, x), mag = u & 0x7fffffffu;
  const bool big = mag >= 0x7f7f8000u, nonfinite = mag >= 0x7f800000u;
  h1 = (__bf16)x;                       // round-to-nearest-even
  if (big) h1 = __builtin_bit_cast(__bf16, (unsigned short)((u >> 16) | (mag > 0x7f800000u ? 0x40u : 0u)));
  const float r1 = nonfinite ? 0.f : x - (float)h1;       // exact: at most 16 significant bits remain
  h2 = (__bf16)r1;
  const float r2 = r1 - (float)h2;      // exact: at most 8 significant bits remain
  h3 = (__bf16)r2;                      // exact
}
__global__ void split_bf16x3_kernel(const float* __restrict__ w, int rows, int cols, int ld, int ldo, __bf16* __restrict__ out, long img_stride) {
  const long total = (long)rows * ldo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / ldo), c = (int)(i - (long)r * ldo);
    __bf16 a, b, d;
    split3_bf16(c < cols ? w[(long)r * ld + c] : 0.f, a, b, d);      // columns cols .. ldo-1 are zero padding (16-byte aligned bf16 rows)
    out[i] = a; out[img_stride + i] = b; out[2 * img_stride + i] = d;
  }
}
__device__ __forceinline__ int bx3_swzA(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 2); }      // 128-B fp32 rows, two b128 reads per lane (brute-force checked)
__device__ __forceinline__ int bx3_swzB(int r) { return (r >> 1) & 3; }                                  // 64-B bf16 rows, one b128 read per lane

// In-loop split of the ACTIVATION operand (round 6).  The round-5 form (three round-to-nearest conversions per value + the edge guards of split3_bf16)
// compiled to ~330 vector instructions per stage and wave beside 60 MFMAs: the kernel was bound by its VALU work (2 waves per SIMD x (960 MFMA
// cycles + ~1300 VALU cycles) = the measured ~3 100 cycles per stage), 120 TF-equivalent at the step's shapes where the micro-benchmark had shown 160.
// Here the images are taken by TRUNCATION, which needs no conversion and no guard: image 1 = the high 16 bits of the fp32 word (one v_perm_b32 packs
// two of them), remainder = x - image1 exactly (one v_pk_add_f32 for two values), image 2 = the high 16 bits of the remainder, image 3 = the high 16
// bits of the second remainder (<= 8 significant bits: exact).  x = a1 + a2 + a3 exactly as before (8 + 8 + 8 bits); |a1| <= |x|, so nothing ever
// rounds up to Inf (FLT_MAX included); +-Inf -> (Inf, NaN, NaN) and NaN -> non-finite images: non-finite in, non-finite out.  9 instructions per two
// values.  The dropped terms (a2 b3 + a3 b2 + a3 b3) stay below 2^-22 |a b| with the weights' images rounded to nearest (zero-mean in b2, b3).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned hi16_pair(float odd, float even) {     // {high 16 bits of odd : high 16 bits of even}
  return __builtin_amdgcn_perm(__float_as_uint(odd), __float_as_uint(even), 0x07060302u);
}
__device__ __forceinline__ void split3_trunc8(const f32x4& x0, const f32x4& x1, bf16x8_t& a1, bf16x8_t& a2, bf16x8_t& a3) {
  // (scalars only: with the pair held as an ext_vector and its elements bit-cast one by one, hipcc 7.0 fed the SAME register to both
  //  sources of v_perm_b32 -- every odd element became a copy of its even neighbour)
  const float xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
  unsigned w1[4], w2[4], w3[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float xe = xs[2 * p], xo = xs[2 * p + 1];
    const unsigned h1 = hi16_pair(xo, xe);
    const float re = xe - __uint_as_float(h1 << 16), ro = xo - __uint_as_float(h1 & 0xffff0000u);
    const unsigned h2 = hi16_pair(ro, re);
    const float se = re - __uint_as_float(h2 << 16), so = ro - __uint_as_float(h2 & 0xffff0000u);
    w1[p] = h1; w2[p] = h2; w3[p] = hi16_pair(so, se);
  }
  a1 = __builtin_bit_cast(bf16x8_t, u32x4_t{w1[0], w1[1], w1[2], w1[3]});
  a2 = __builtin_bit_cast(bf16x8_t, u32x4_t{w2[0], w2[1], w2[2], w2[3]});
  a3 = __builtin_bit_cast(bf16x8_t, u32x4_t{w3[0], w3[1], w3[2], w3[3]});
}

template <int TM, int TN, int NS>
__global__ __launch_bounds__(256, (NS * (64 * TM * 128 + 3 * 16 * TN * 64) > 80 * 1024 ? 1 : 2)) void gemm_nt_bx3_kernel(nnr_gemm_args g) {
  constexpr int BM = 64 * TM, BN = 16 * TN, BK = 32;
  constexpr int A_BYTES = BM * BK * 4, B_IMG_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + 3 * B_IMG_BYTES;
  constexpr int NIA = A_BYTES / 1024, NIB1 = B_IMG_BYTES / 1024;
  static_assert(A_BYTES % 1024 == 0 && B_IMG_BYTES % 1024 == 0 && NIA % 4 == 0, "tile shape");
  constexpr int E_LD = BN + 4;
  constexpr int LDS_BYTES = NS * STAGE_BYTES > 64 * E_LD * 4 ? NS * STAGE_BYTES : 64 * E_LD * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  int M = g.M;
  if (g.dyn_dim == 1) M = min(M, *g.dyn_dev);
  const int N = g.N, K = g.K;
  const int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN, nblk = nbm * nbn;
  if ((int)blockIdx.x >= nblk) return;
  int v;
  {
    const int b = blockIdx.x, qq = nblk >> 3, rem = nblk & 7, x = b & 7, slot = b >> 3;
    v = x * qq + min(x, rem) + slot;
  }
  const int bm = v / nbn, bn = v - bm * nbn, m0 = bm * BM, n0 = bn * BN;
  const int S = (K + BK - 1) / BK;
  const unsigned lds_base = (unsigned)(uintptr_t)lds_raw;
  const float* zero = nnr_zero_page;
  asm volatile("" : "+s"(zero));
  // wave w issues the A instructions w, w + 4, ... (8 tile rows of 128 B each) and the B instructions idx = w, w + 4, ... of the 3 NIB1
  // (16 rows of 64 B of one image each).  Per lane and instruction a CONSTANT byte offset from a wave-uniform stage base (SGPR), as in
  // gemm_nt_pipe2_kernel: a full stage is issued without any vector ALU work; only a k-tail stage (K % 32 != 0) selects the zero page per lane.
  constexpr int NA = NIA / 4, NBW = (3 * NIB1 + 3) / 4;
  static_assert(NA <= 8 && NBW <= 8, "tile shape");
  unsigned voffA[8], voffB[8];
  int akc[NA], bkc[NBW];
#pragma unroll
  for (int i = 0; i < 8; ++i) voffA[i] = voffB[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (w + 4 * i) * 8 + lane / 8, c = (lane % 8) ^ bx3_swzA(row & 15);
    akc[i] = 4 * c;
    voffA[i] = (unsigned)(((long)min(row, M - 1 - m0) * g.lda + 4 * c) * 4);          // rows past the edge: clamped, never stored
  }
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int idx = min(w + 4 * j, 3 * NIB1 - 1);
    const int img = idx / NIB1, jj = idx - img * NIB1, row = jj * 16 + lane / 4, c = (lane % 4) ^ bx3_swzB(row & 15);
    bkc[j] = 8 * c;
    voffB[j] = (unsigned)(((long)img * g.b3_stride + (long)min(row, N - 1 - n0) * g.ldb3 + 8 * c) * 2);
  }
  const float* Abase = g.A + (long)m0 * g.lda;                                          // wave-uniform stage bases
  const char* Bbase = reinterpret_cast<const char*>(reinterpret_cast<const __bf16*>(g.B3) + (long)n0 * g.ldb3);
  const bool nb_hi = ((3 * NIB1) % 4 == 0) || (w < (3 * NIB1) % 4);                     // this wave has NBW (else NBW - 1) B instructions
  const bool ktail = (K % BK) != 0;
  auto issue = [&](int s) __attribute__((always_inline)) {
    const int k0 = s * BK;
    const unsigned sb = lds_base + (unsigned)((s % NS) * STAGE_BYTES) + (unsigned)(w * 1024);
    if (!(ktail && s == S - 1)) {
      lds_dma16_block<NA>(Abase + k0, sb, voffA);
      const float* bb = reinterpret_cast<const float*>(Bbase + 2L * k0);
      if (nb_hi) lds_dma16_block<NBW>(bb, sb + A_BYTES, voffB);
      else if constexpr (NBW > 1) lds_dma16_block<NBW - 1>(bb, sb + A_BYTES, voffB);
      return;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i)
      lds_dma16((k0 + akc[i] < K) ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(Abase + k0) + voffA[i]) : zero, sb + (unsigned)(i * 4096));
#pragma unroll
    for (int j = 0; j < NBW; ++j)
      if (j < NBW - 1 || nb_hi)
        lds_dma16((k0 + bkc[j] < K) ? reinterpret_cast<const float*>(Bbase + 2L * k0 + voffB[j]) : zero, sb + A_BYTES + (unsigned)(j * 4096));
  };
  f32x4 acc_hi[TM][TN], acc_lo[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) acc_hi[m][n] = acc_lo[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < S) issue(s);
  for (int s = 0; s < S; ++s) {
    // this wave's DMAs of stage s have landed once at most min(NS - 2, stages issued beyond s) newer stages of its own are outstanding
    // (NS = 2: vmcnt(0); NS = 3: the loads of stage s + 1 stay in flight across the wait and the barrier)
    if (nb_hi) wait_stages<NA + NBW, NS - 2>(S - 1 - s);
    else wait_stages<NA + NBW - 1, NS - 2>(S - 1 - s);
    __builtin_amdgcn_s_barrier();
    if (s + NS - 1 < S) issue(s + NS - 1);          // into the buffer every wave finished reading before that barrier
    const unsigned char* st = lds_raw + (s % NS) * STAGE_BYTES;
    const float* As = reinterpret_cast<const float*>(st);
    const __bf16* Bi = reinterpret_cast<const __bf16*>(st + A_BYTES);
    bf16x8_t bf[3][TN];
#pragma unroll
    for (int img = 0; img < 3; ++img)
#pragma unroll
      for (int n = 0; n < TN; ++n)
        bf[img][n] = *reinterpret_cast<const bf16x8_t*>(Bi + img * (BN * BK) + (n * 16 + r) * BK + ((q ^ bx3_swzB(r)) * 8));
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      const float* arow = As + ((w * TM + m) * 16 + r) * BK;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(arow + 4 * ((2 * q) ^ bx3_swzA(r)));
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(arow + 4 * ((2 * q + 1) ^ bx3_swzA(r)));
      bf16x8_t a1, a2, a3;
      split3_trunc8(x0, x1, a1, a2, a3);
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        acc_hi[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf[0][n], acc_hi[m][n], 0, 0, 0);
        acc_lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, bf[0][n], acc_lo[m][n], 0, 0, 0);
        acc_lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf[2][n], acc_lo[m][n], 0, 0, 0);
        acc_lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bf[1][n], acc_lo[m][n], 0, 0, 0);
        acc_lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, bf[0][n], acc_lo[m][n], 0, 0, 0);
        acc_lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bf[1][n], acc_lo[m][n], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) acc_hi[m][n] += acc_lo[m][n];
  __syncthreads();
  gemm_epilogue<TM, TN>(g, acc_hi, reinterpret_cast<float*>(lds_raw), g.C, m0, n0, M, N, 0);
}

static bool bx3_ok(const nnr_gemm_args& g) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  return pipe_ok(g) && !g.a_idx && g.batch <= 1 && g.B3 && al(g.B3) && (g.ldb3 & 7) == 0 && g.ldb3 >= g.K && ((g.b3_stride * 2) & 15) == 0;
}
template <int TM, int TN, int NS>
int launch_bx3(const nnr_gemm_args& g, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 16 * TN;
  const int nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm_nt_bx3_kernel<TM, TN, NS>), dim3(nbm * nbn), dim3(256), 0, s, g);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

// ------------------------------------------------------------------------------------------------ pipelined TN GEMM (LDS-DMA)
// C[M,N] += A[K,M]^T . B[K,N] over a slice of the (device-side) reduction range -- the weight-gradient GEMMs: both operands
// are ACTIVATIONS stored row-major by token, i.e. K-major for this product, and K is the token count (10^5).
//  * a stage is 16 token rows of the A tile and of the B tile; a token row's tile columns are contiguous in memory, so one
//    LDS-DMA instruction (1 KiB) moves 256 / PA whole row segments: full cache lines, no register staging, no ds_write pass.
//  * LDS image [16][PA] / [16][PB] (PA = 64*TM, PB = 16*TN rounded up to 64 / 128 / 256 floats; columns past the tile edge
//    read the zero page).  Lane (r, kk) of MFMA step i reads row 4*kk + i: four rows per lane at immediate offsets.  Rows k
//    and k + 4 would hit the same banks, so position p of row k holds column p ^ 16*((k >> 2) & 1) -- applied to the DMA's
//    per-lane SOURCE address and to the ds_read_b32 address alike.
//  * B's token rows may be gathered (b_idx: the previous time step's packed row for dW_hh; negative = zero row).  The row
//    index of an instruction's segments is wave-uniform, so it is fetched with SCALAR loads one stage ahead (vector loads
//    would share vmcnt with the DMAs and drain the pipeline every stage).
//  * NS stage buffers, one counted vmcnt wait + one raw barrier per stage (as gemm_nt_pipe_kernel).
//  * split-K over blockIdx.z, f32 atomics in the epilogue; the fused bias gradient (column sums of A) of the column-block-0
//    workgroups is one extra MFMA per row tile and step against a fragment of ones -- no LDS traffic, no extra pass over A.
// Slices a token-reduction launch uses for a LIVE reduction length K (the host sizes split_k for the capacity of the token buffers):
// as many as keep a slice long enough to amortise its prologue and its epilogue (~96 stages of 16 rows) while still giving every CU a
// workgroup or two, rounded so that eff x nblk is a whole number of waves of resident workgroups (400 x 400 x 77 k tokens: 1 000
// workgroups on 768 slots = 379 us, 740 or 1 520 = 279 / 293 us).  Shared by the kernels and by the slab reduction (same arithmetic,
// same K: they agree on which slices exist).
__device__ __forceinline__ int tn_eff_slices(int split_k, int sched, int K, int nblk, int slots) {
  const int deal_mode = sched & 3;
  const int want_wg = ((sched >> 2) & 63) ? ((sched >> 2) & 63) * 64 : 512;      // workgroups wanted at least (tuning knob, default 512)
  const int slice_stages = (sched >> 8) ? (sched >> 8) : 96;                     // stages per slice aimed at (tuning knob)
  const int ktiles = (K + 15) / 16;
  const int want = (want_wg + nblk - 1) / nblk;
  int eff = max(1, min(min(split_k, max(want, ktiles / slice_stages)), max(1, ktiles / 12)));
  if (deal_mode != 2) {
    const int W = eff * nblk, wv = W / slots;
    const int target = W < slots ? slots : (((W - wv * slots) * 2 < slots) ? wv * slots : (wv + 1) * slots);
    eff = max(1, min(target / nblk, min(split_k, max(1, ktiles / 12))));
  }
  return eff;
}

template <int TM, int TN, int NS, int OCC, int PRIO = 0>
__global__ __launch_bounds__(256, OCC) void gemm_tn_pipe_kernel(nnr_gemm_args g) {
  if (PRIO == 1) set_prio_dyn((blockIdx.x + gridDim.x * blockIdx.z) >> 8);
  constexpr int BK = 16, BM = 64 * TM, BN = 16 * TN;
  constexpr int PA = BM, PB = BN <= 64 ? 64 : (BN <= 128 ? 128 : 256);
  constexpr int RA = 256 / PA, RB = 256 / PB;          // token rows per DMA instruction
  constexpr int NIA = BK / RA, NIB = BK / RB, NI = NIA + NIB;
  constexpr int NPW = (NI + 3) / 4;
  constexpr int STAGE = BK * (PA + PB);
  constexpr int E_LD = BN + 4;
  constexpr int LDS_FLOATS = (NS * STAGE > 64 * E_LD) ? NS * STAGE : 64 * E_LD;
  static_assert(BN <= 256 && NI % 4 == 0, "tile shape");
  __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kk = lane >> 4;
  const int M = g.M, N = g.N;
  int K = g.K;
  if (g.dyn_dim == 2) K = min(K, *g.dyn_dev);
  const int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN;
  const int nblk = nbm * nbn;
  // blockIdx -> (tile, reduction slice).  The host sizes split_k for the CAPACITY of the token buffers; the live reduction length
  // (device side) is typically a fifth of it: use only as many slices (eff) as keep a slice long enough to amortise its prologue
  // and its atomic epilogue (~96 stages) while still giving every CU a workgroup or two; the surplus workgroups exit.
  // With split-K the grid is 1-D and the eff x nblk workgroups, in slice-major order, are dealt to the XCDs in eight CONTIGUOUS
  // runs (XCD = blockIdx.x % 8 under round-robin placement; speed only): all tiles of a slice run on one XCD (a slice that
  // straddles a run boundary on two), so a token row of A and of B is pulled into one L2 instead of into every L2 whose
  // workgroups touch it -- 4-5x the operand bytes at the fabric counters for the 400 x 400 and 832 x 200 shapes when the tiles of
  // a slice are spread round-robin (profiles/pmc_traffic.json, round 2 first collection).  deal_mode 0 = that spread layout (A/B).
  int v, z = 0;
  int kbeg = 0, kend = K;
  if (g.split_k > 1) {
    const int deal_mode = g.sched & 3;                // set by the launcher
    const int ktiles = (K + BK - 1) / BK;
    const int eff = tn_eff_slices(g.split_k, g.sched, K, nblk, OCC * 256);
    if (deal_mode) {
      const int W = eff * nblk, per = (W + 7) >> 3;
      const int L = blockIdx.x, x = L & 7, slot = L >> 3;
      const int lin = x * per + slot;
      if (slot >= per || lin >= W) return;
      z = lin / nblk;
      v = lin - z * nblk;
    } else {
      z = blockIdx.x / nblk;
      v = blockIdx.x - z * nblk;
      if (z >= eff) return;
    }
    const int per_k = (ktiles + eff - 1) / eff;
    kbeg = z * per_k * BK;
    kend = min(K, kbeg + per_k * BK);
    if (kbeg >= kend) return;
  } else {
    const int b = blockIdx.x, q = nblk >> 3, rem = nblk & 7, x = b & 7, slot = b >> 3;
    v = x * q + min(x, rem) + slot;
  }
  const int bm = v / nbn, bn = v - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const float* __restrict__ A = g.A;
  const float* __restrict__ B = g.B;
  const int* __restrict__ bidx = g.b_idx;
  const float* zero = nnr_zero_page;
  asm volatile("" : "+s"(zero));

  // ---- per-lane DMA geometry.  Instruction q < NIA: A rows [q*RA, +RA); else B rows [(q-NIA)*RB, +RB)
  long coloff[NPW];          // column offset (floats) of this lane's 16-byte chunk, or -1 (beyond the matrix edge: zero page)
  int krow[NPW];             // token row inside the stage this lane feeds
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int q = w + 4 * i;
    if (q < NIA) {
      const int kl = q * RA + lane / (PA / 4), p4 = lane % (PA / 4);
      const int col = (4 * p4) ^ (16 * ((kl >> 2) & 1));
      krow[i] = kl;
      coloff[i] = (m0 + col < M) ? (long)(m0 + col) : -1;
    } else {
      const int kl = (q - NIA) * RB + lane / (PB / 4), p4 = lane % (PB / 4);
      const int col = (4 * p4) ^ (16 * ((kl >> 2) & 1));
      krow[i] = kl;
      coloff[i] = (col < BN && n0 + col < N) ? (long)(n0 + col) : -1;
    }
  }
  const unsigned lds_base = (unsigned)(uintptr_t)lds;
  const int S = (kend - kbeg + BK - 1) / BK;

  // gathered B rows: the RB row indices of each of this wave's B instructions for one stage, in SGPRs
  int bsrc[NPW][RB];
  auto fetch_idx = [&](int s) {
    if (bidx == nullptr) return;
    const int k0 = kbeg + s * BK;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int q = w + 4 * i;
      if (q >= NIA) {
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          const int k = min(k0 + (q - NIA) * RB + j, kend - 1);
          // SCALAR load (the compiler picks a vector load here: it cannot prove the table is never written by this kernel;
          // a vector load would share vmcnt with the DMAs and its wait would drain them).  Consumed one stage later, behind the
          // s_waitcnt lgkmcnt(0) in front of issue().
          asm volatile("s_load_dword %0, %1, %2" : "=s"(bsrc[i][j]) : "s"(bidx), "s"(k * 4) : "memory");
        }
      }
    }
  };
  auto issue = [&](int s) {
    const int k0 = kbeg + s * BK;
    const unsigned sb = lds_base + (unsigned)((s % NS) * STAGE * 4);
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int q = w + 4 * i;
      const int k = k0 + krow[i];
      const float* src = zero;
      if (q < NIA) {
        if (coloff[i] >= 0 && k < kend) src = A + (long)k * g.lda + coloff[i];
        lds_dma16(src, sb + (unsigned)(q * 1024));
      } else {
        long row = k;
        if (bidx != nullptr) {
          int sel = bsrc[i][0];
#pragma unroll
          for (int j = 1; j < RB; ++j) sel = (lane / (PB / 4) == j) ? bsrc[i][j] : sel;
          row = sel;
        }
        if (coloff[i] >= 0 && k < kend && row >= 0) src = B + row * g.ldb + coloff[i];
        lds_dma16(src, sb + (unsigned)(NIA * 1024 + (q - NIA) * 1024));
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = g.colsum_out != nullptr && bn == 0;
  f32x4 cs[TM];
#pragma unroll
  for (int m = 0; m < TM; ++m) cs[m] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < S) {
      fetch_idx(s);
      if (bidx != nullptr) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue(s);
    }
  if (NS - 1 < S) fetch_idx(NS - 1);

  const int xo = 16 * (kk & 1);                         // this lane's rows 4*kk + i all have ((k >> 2) & 1) == kk & 1
  for (int s = 0; s < S; ++s) {
    wait_stages<NPW, NS - 2>(S - 1 - s);
    __builtin_amdgcn_s_barrier();
    if (s + NS - 1 < S) {
      if (bidx != nullptr) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the gather indices fetched a stage ago
      issue(s + NS - 1);
    }
    const float* As = lds + (s % NS) * STAGE;
    const float* Bs = As + BK * PA;
    float af[4][TM];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float bf[TN];
#pragma unroll
      for (int m = 0; m < TM; ++m) af[i][m] = As[(4 * kk + i) * PA + (((w * TM + m) * 16 + r) ^ xo)];
#pragma unroll
      for (int n = 0; n < TN; ++n) bf[n] = Bs[(4 * kk + i) * PB + ((n * 16 + r) ^ xo)];
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][m], bf[n], acc[m][n], 0, 0, 0);
    }
    // the gather indices of the stage issued NEXT iteration: behind this stage's fragment reads, so the compiler's lgkmcnt
    // waits for those do not also wait for the scalar loads; they land while the wave sits in the next barrier
    if (s + NS < S) fetch_idx(s + NS);
    if (do_colsum) {      // (one branch per stage: inside the step loop it would cut the block the compiler pipelines reads over)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int m = 0; m < TM; ++m) cs[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][m], 1.f, cs[m], 0, 0, 0);
    }
  }
  if (do_colsum && r == 0) {      // every column of cs[m] holds the column sums of A for rows kk*4 + reg of row tile m
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = m0 + (w * TM + m) * 16 + kk * 4 + reg;
        if (row < M) {
          if (g.slab_mode) g.colsum_out[(long)z * M + row] = cs[m][reg];      // slice z's own row of the column-sum slab
          else atomicAdd(&g.colsum_out[row], cs[m][reg]);
        }
      }
  }
  __syncthreads();
  gemm_epilogue<TM, TN>(g, acc, lds, g.slab_mode ? g.C + (long)z * M * N : g.C, m0, n0, M, N, z);
}

// deal mode / slice-count tuning knobs of the split-K token reductions, as the kernels and the slab reduction read them from g.sched
static int tn_sched_bits() {
  static const int deal_mode = [] { const char* e = getenv("NNR_TN_DEAL"); return e ? atoi(e) : 1; }();      // A/B: 0 = tiles of a slice spread over the XCDs
  static const int want_code = [] { const char* e = getenv("NNR_TN_WANT"); return e ? atoi(e) / 64 : 0; }();    // tuning: minimum workgroup count (default 512)
  static const int stage_code = [] { const char* e = getenv("NNR_TN_STAGES"); return e ? atoi(e) : 0; }();       // tuning: stages per split-K slice (default 96)
  return (deal_mode & 3) | ((want_code & 63) << 2) | (stage_code << 8);
}

template <int TM, int TN, int NS, int OCC, int PRIO = 0>
int launch_tn_pipe(const nnr_gemm_args& g, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 16 * TN;
  const int nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
  dim3 grid(g.split_k > 1 ? nbm * nbn * ((g.split_k + 7) / 8) * 8 : nbm * nbn), block(256);      // split-K: slices are dealt to XCDs (see the kernel)
  nnr_gemm_args gg = g;
  gg.sched = tn_sched_bits();
  hipLaunchKernelGGL((gemm_tn_pipe_kernel<TM, TN, NS, OCC, PRIO>), grid, block, 0, s, gg);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

// ---- second-generation TN loop (see gemm_nt_pipe2_kernel): the four MFMA steps of a 16-row stage are software-pipelined -- the
// fragments of step i + 1 are read while step i's MFMAs run, the wait + barrier + DMA issue + the next stage's first reads sit
// between steps 2 and 3 -- and the DMA issue is lean: an SGPR row base per stage (or per gathered token row) plus per-lane
// offsets that never change.  Restrictions on top of gemm_tn_pipe_kernel's: a gathered B (b_idx) needs the 256-float pitch
// (TN > 8: one token row per instruction, so the row's base is a scalar).  A reduction tail (rows beyond kend in the last
// stage) is issued through per-lane pointers with the zero page.
template <int N>
__device__ __forceinline__ void lds_dma16_rows(const float* const (&sbase)[4], unsigned lds_dst, const unsigned (&voff)[8]) {
  unsigned keep;      // N loads, each with ITS OWN scalar base (one gathered token row per instruction) and lane offset (the swizzle depends on the row)
  static_assert(N >= 1 && N <= 4, "1..4 rows");
#define NNR_ROW_HEAD "s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\t"
#define NNR_ROW_LD(I) "s_nop 0\n\tglobal_load_lds_dwordx4 %[v" #I "], %[b" #I "]\n\ts_add_u32 m0, m0, 0x1000\n\t"
#define NNR_ROW_OPS : [keep] "=&s"(keep) : [dst] "s"(lds_dst), [v0] "v"(voff[0]), [v1] "v"(voff[1]), [v2] "v"(voff[2]), [v3] "v"(voff[3]), [b0] "s"(sbase[0]), [b1] "s"(sbase[1]), [b2] "s"(sbase[2]), [b3] "s"(sbase[3]) : "memory", "scc"
  if constexpr (N == 1) asm volatile(NNR_ROW_HEAD NNR_ROW_LD(0) "s_mov_b32 m0, %[keep]" NNR_ROW_OPS);
  if constexpr (N == 2) asm volatile(NNR_ROW_HEAD NNR_ROW_LD(0) NNR_ROW_LD(1) "s_mov_b32 m0, %[keep]" NNR_ROW_OPS);
  if constexpr (N == 3) asm volatile(NNR_ROW_HEAD NNR_ROW_LD(0) NNR_ROW_LD(1) NNR_ROW_LD(2) "s_mov_b32 m0, %[keep]" NNR_ROW_OPS);
  if constexpr (N == 4) asm volatile(NNR_ROW_HEAD NNR_ROW_LD(0) NNR_ROW_LD(1) NNR_ROW_LD(2) NNR_ROW_LD(3) "s_mov_b32 m0, %[keep]" NNR_ROW_OPS);
#undef NNR_ROW_HEAD
#undef NNR_ROW_LD
#undef NNR_ROW_OPS
}

template <int TM, int TN, int NS, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_tn_pipe2_kernel(nnr_gemm_args g) {
  constexpr int BK = 16, BM = 64 * TM, BN = 16 * TN;
  constexpr int PA = BM, PB = BN <= 64 ? 64 : (BN <= 128 ? 128 : 256);
  constexpr int RA = 256 / PA, RB = 256 / PB;
  constexpr int NIA = BK / RA, NIB = BK / RB, NA = NIA / 4, NB = NIB / 4, NPW = NA + NB;
  constexpr int STAGE = BK * (PA + PB);
  constexpr int E_LD = BN + 4;
  constexpr int LDS_FLOATS = (NS * STAGE > 64 * E_LD) ? NS * STAGE : 64 * E_LD;
  static_assert(BN <= 256 && NIA % 4 == 0 && NIB % 4 == 0 && NA <= 8 && NB <= 8 && NS >= 3, "tile shape");
  __shared__ __attribute__((aligned(1024))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, kk = lane >> 4;
  const int M = g.M, N = g.N;
  int K = g.K;
  if (g.dyn_dim == 2) K = min(K, *g.dyn_dev);
  const int nbm = (M + BM - 1) / BM, nbn = (N + BN - 1) / BN;
  const int nblk = nbm * nbn;
  // blockIdx -> (tile, reduction slice).  The host sizes split_k for the CAPACITY of the token buffers; the live reduction length
  // (device side) is typically a fifth of it: use only as many slices (eff) as keep a slice long enough to amortise its prologue
  // and its atomic epilogue (~96 stages) while still giving every CU a workgroup or two; the surplus workgroups exit.
  // With split-K the grid is 1-D and the eff x nblk workgroups, in slice-major order, are dealt to the XCDs in eight CONTIGUOUS
  // runs (XCD = blockIdx.x % 8 under round-robin placement; speed only): all tiles of a slice run on one XCD (a slice that
  // straddles a run boundary on two), so a token row of A and of B is pulled into one L2 instead of into every L2 whose
  // workgroups touch it -- 4-5x the operand bytes at the fabric counters for the 400 x 400 and 832 x 200 shapes when the tiles of
  // a slice are spread round-robin (profiles/pmc_traffic.json, round 2 first collection).  deal_mode 0 = that spread layout (A/B).
  int v, z = 0;
  int kbeg = 0, kend = K;
  if (g.split_k > 1) {
    const int deal_mode = g.sched & 3;                // set by the launcher
    const int ktiles = (K + BK - 1) / BK;
    const int eff = tn_eff_slices(g.split_k, g.sched, K, nblk, OCC * 256);
    if (deal_mode) {
      const int W = eff * nblk, per = (W + 7) >> 3;
      const int L = blockIdx.x, x = L & 7, slot = L >> 3;
      const int lin = x * per + slot;
      if (slot >= per || lin >= W) return;
      z = lin / nblk;
      v = lin - z * nblk;
    } else {
      z = blockIdx.x / nblk;
      v = blockIdx.x - z * nblk;
      if (z >= eff) return;
    }
    const int per_k = (ktiles + eff - 1) / eff;
    kbeg = z * per_k * BK;
    kend = min(K, kbeg + per_k * BK);
    if (kbeg >= kend) return;
  } else {
    const int b = blockIdx.x, q = nblk >> 3, rem = nblk & 7, x = b & 7, slot = b >> 3;
    v = x * q + min(x, rem) + slot;
  }
  const int bm = v / nbn, bn = v - bm * nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const float* __restrict__ A = g.A;
  const float* __restrict__ B = g.B;
  const int* __restrict__ bidx = g.b_idx;
  const bool gather = bidx != nullptr;                       // (RB == 1 guaranteed by the dispatcher)
  const float* zero = nnr_zero_page;
  asm volatile("" : "+s"(zero));
  const unsigned lds_base = (unsigned)(uintptr_t)lds;
  const int S = (kend - kbeg + BK - 1) / BK;
  const bool ktail = ((kend - kbeg) % BK) != 0;

  // ---- DMA geometry: wave w issues A instructions q = w + 4i (i < NA) and B instructions q = w + 4j (j < NB); instruction q
  // of A covers token rows [q*RA, +RA) of the stage.  Lane offsets (bytes) from the stage's row base; columns past the matrix
  // edge are CLAMPED to column 0 of the tile (they only feed outputs that are never stored).
  unsigned voffA[8], voffB[8];
  int krA[NA], colA[NA], krB[NB], colB[NB];                  // for the reduction-tail stage (per-lane pointers)
#pragma unroll
  for (int i = 0; i < 8; ++i) voffA[i] = voffB[i] = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int q = w + 4 * i;
    const int kl = q * RA + lane / (PA / 4), p4 = lane % (PA / 4);
    int col = (4 * p4) ^ (16 * ((kl >> 2) & 1));
    if (m0 + col >= M) col = 0;
    krA[i] = kl; colA[i] = col;
    voffA[i] = (unsigned)(((long)kl * g.lda + col) * 4);
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int q = w + 4 * j;
    const int kl = q * RB + lane / (PB / 4), p4 = lane % (PB / 4);
    int col = (4 * p4) ^ (16 * ((kl >> 2) & 1));
    if (col >= BN || n0 + col >= N) col = 0;
    krB[j] = kl; colB[j] = col;
    voffB[j] = gather ? (unsigned)(col * 4) : (unsigned)(((long)kl * g.ldb + col) * 4);
  }
  const float* Abase = A + (long)kbeg * g.lda + m0;          // + s * BK * lda per stage
  const float* Bbase = B + (long)kbeg * g.ldb + n0;

  // gathered B: the NB token-row indices of this wave's B instructions for one stage, fetched (scalar) one stage ahead
  int bsrc[NB > 0 ? NB : 1];
  auto fetch_idx = [&](int s) {
    const int k0 = kbeg + s * BK;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int k = min(k0 + (w + 4 * j) * RB, kend - 1);
      asm volatile("s_load_dword %0, %1, %2" : "=s"(bsrc[j]) : "s"(bidx), "s"(k * 4) : "memory");
    }
  };
  auto issue = [&](int s, bool tail) {
    const unsigned sb = lds_base + (unsigned)((s % NS) * STAGE * 4) + (unsigned)(w * 1024);
    const int k0 = kbeg + s * BK;
    if (tail) {                                              // rows >= kend of the last stage: zeros (per-lane source select)
#pragma unroll
      for (int i = 0; i < NA; ++i)
        lds_dma16((k0 + krA[i] < kend) ? A + (long)(k0 + krA[i]) * g.lda + m0 + colA[i] : zero, sb + i * 4096);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int k = k0 + krB[j];
        long row = k;
        if (gather) row = bsrc[j];
        lds_dma16((k < kend && row >= 0) ? B + row * g.ldb + n0 + colB[j] : zero, sb + NIA * 1024 + j * 4096);
      }
      return;
    }
    lds_dma16_block<NA>(Abase + (long)s * BK * g.lda, sb, voffA);
    if (!gather) {
      lds_dma16_block<NB>(Bbase + (long)s * BK * g.ldb, sb + NIA * 1024, voffB);
    } else {
      if constexpr (RB == 1 && NB <= 4) {
        const float* rows[4] = {zero, zero, zero, zero};
#pragma unroll
        for (int j = 0; j < NB; ++j) rows[j] = bsrc[j] >= 0 ? B + (long)bsrc[j] * g.ldb + n0 : zero;
        lds_dma16_rows<NB>(rows, sb + NIA * 1024, voffB);
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = g.colsum_out != nullptr && bn == 0;
  f32x4 cs[TM];
#pragma unroll
  for (int m = 0; m < TM; ++m) cs[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int xo = 16 * (kk & 1);
  float fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  auto rd = [&](int s, int i, float (&a)[TM], float (&b)[TN]) {
    const float* As = lds + (s % NS) * STAGE;
    const float* Bs = As + BK * PA;
#pragma unroll
    for (int m = 0; m < TM; ++m) a[m] = As[(4 * kk + i) * PA + (((w * TM + m) * 16 + r) ^ xo)];
#pragma unroll
    for (int n = 0; n < TN; ++n) b[n] = Bs[(4 * kk + i) * PB + ((n * 16 + r) ^ xo)];
  };
  auto wait_landed = [&](int ahead) { wait_stages<NPW, NS - 3>(ahead); };
#define NNR_LGKM0() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xC07F); } while (0)
#define NNR_PIN() __builtin_amdgcn_sched_barrier(0)

  // one pass over the stages; CS: also accumulate the column sums of A (the column-block-0 workgroups of a launch that asks
  // for the fused bias gradient) -- a separate instantiation so that the common loop carries no branch
  auto run = [&](auto cs_tag) {
    constexpr bool CS = decltype(cs_tag)::value;
    auto mm = [&](const float (&a)[TM], const float (&b)[TN]) {
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
      if constexpr (CS) {
#pragma unroll
        for (int m = 0; m < TM; ++m) cs[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], 1.f, cs[m], 0, 0, 0);
      }
    };
    // prologue
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < S) {
        if (gather) { fetch_idx(s); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        issue(s, ktail && s == S - 1);
      }
    if (gather && NS - 1 < S) fetch_idx(NS - 1);
    wait_stages<NPW, NS - 2>(S - 1);
    __builtin_amdgcn_s_barrier();
    rd(0, 0, fa0, fb0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    auto stage = [&](int s, int refill /* 0 none, 1 ordinary stage, 2 possibly the tail stage */) {
      rd(s, 1, fa1, fb1); NNR_PIN(); mm(fa0, fb0);
      rd(s, 2, fa0, fb0); NNR_PIN(); mm(fa1, fb1);
      rd(s, 3, fa1, fb1); NNR_PIN(); mm(fa0, fb0);
      NNR_LGKM0();
      wait_landed(S - 1 - (s + 1));
      __builtin_amdgcn_s_barrier();
      if (refill) {
        if (gather) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the row indices fetched a stage ago
        issue(s + NS - 1, refill == 2 && ktail && s + NS - 1 == S - 1);
        if (gather && s + NS < S) fetch_idx(s + NS);
      }
      rd(s + 1, 0, fa0, fb0); NNR_PIN(); mm(fa1, fb1);
      NNR_LGKM0();
    };
    int s = 0;
    for (; s + NS < S; ++s) stage(s, 1);                    // refill = stage s + NS - 1 <= S - 2: never the tail
    for (; s + 1 < S; ++s) stage(s, s + NS - 1 < S ? 2 : 0);
    rd(S - 1, 1, fa1, fb1); NNR_PIN(); mm(fa0, fb0);
    rd(S - 1, 2, fa0, fb0); NNR_PIN(); mm(fa1, fb1);
    rd(S - 1, 3, fa1, fb1); NNR_PIN(); mm(fa0, fb0);
    mm(fa1, fb1);
  };
  if (do_colsum) run(std::true_type{});
  else run(std::false_type{});
#undef NNR_LGKM0
#undef NNR_PIN

  if (do_colsum && r == 0) {
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = m0 + (w * TM + m) * 16 + kk * 4 + reg;
        if (row < M) {
          if (g.slab_mode) g.colsum_out[(long)z * M + row] = cs[m][reg];      // slice z's own row of the column-sum slab
          else atomicAdd(&g.colsum_out[row], cs[m][reg]);
        }
      }
  }
  __syncthreads();
  gemm_epilogue<TM, TN>(g, acc, lds, g.slab_mode ? g.C + (long)z * M * N : g.C, m0, n0, M, N, z);
}

template <int TM, int TN, int NS, int OCC>
int launch_tn_pipe2(const nnr_gemm_args& g, hipStream_t s) {
  constexpr int BM = 64 * TM, BN = 16 * TN;
  const int nbm = (g.M + BM - 1) / BM, nbn = (g.N + BN - 1) / BN;
  dim3 grid(g.split_k > 1 ? nbm * nbn * ((g.split_k + 7) / 8) * 8 : nbm * nbn), block(256);      // split-K: slices are dealt to XCDs (see the kernel)
  nnr_gemm_args gg = g;
  gg.sched = tn_sched_bits();
  hipLaunchKernelGGL((gemm_tn_pipe2_kernel<TM, TN, NS, OCC>), grid, block, 0, s, gg);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

static bool tn_pipe_ok(const nnr_gemm_args& g) {
  auto al = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
  return g.trans_a && g.trans_b && !g.a_idx && !g.c_idx && g.k_chunk <= 0 && !g.rowdot_w && g.drop_target == 0 && g.batch <= 1 &&
         (g.M & 3) == 0 && (g.N & 3) == 0 && (g.lda & 3) == 0 && (g.ldb & 3) == 0 && al(g.A) && al(g.B) && (g.dyn_dim == 0 || g.dyn_dim == 2) &&
         (g.atomic || g.split_k > 1 || g.accumulate == 1 || g.accumulate == 0);
}

// ------------------------------------------------------------------------------------------------ skinny GEMM (small launches)
// For launches that cannot fill the chip (M of a few hundred to a few thousand rows: per-news vectors, SUE heads) the tiled
// kernel is bound by its own dependent chain -- one workgroup walks K / 16 stages of (global load -> LDS -> barrier -> MFMA),
// ~0.8 us each with nothing else on the CU to hide them: 20-50 us for a few MFLOP.  Here a workgroup owns a 16 x 80 output
// tile and its four waves split the K range (wave w takes the 16-wide k-groups w, w+4, ...): 4x shorter chains, 4x more
// workgroups, MFMA fragments loaded straight from global memory (no LDS staging, no barrier inside the loop), one LDS
// reduction of the four partial tiles, then the same element-wise epilogue as the tiled kernel.
// NW waves split K (round 4: 8 / 16 waves for the few-tile, long-K launches of the dependent chain -- a wave's K share is a chain of
// global-load round trips, 14 of them at K = 900 with 4 waves and one group in flight: 20-28 us for 0.9 GFLOP; and TWO groups in flight)
template <bool TB, int NW>
__global__ __launch_bounds__(64 * NW) void skinny_gemm_kernel(nnr_gemm_args g) {
  constexpr int TN = 5, BN = 16 * TN, E_LD = BN + 4;
  __shared__ float red[NW][16 * E_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, kk = lane >> 4;
  const int nbn = (g.N + BN - 1) / BN;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x - bm * nbn;
  const int m0 = bm * 16, n0 = bn * BN;
  const int M = g.M, N = g.N, K = g.K;
  const float* __restrict__ A = g.A;
  const float* __restrict__ B = g.B;
  float* __restrict__ C = g.C;
  float* aux = g.aux_out;
  const float* res = g.resid;
  const int z = blockIdx.z;
  if (g.batch > 1) {
    A += (long)z * g.strideA; B += (long)z * g.strideB; C += (long)z * g.strideC;
    if (aux) aux += (long)z * g.stride_aux;
    if (res) res += (long)z * g.stride_res;
  }
  const bool vecA = ((g.lda & 3) == 0) && ((((uintptr_t)A) & 15) == 0);
  const bool vecB = ((g.ldb & 3) == 0) && ((((uintptr_t)B) & 15) == 0);
  const int arow = m0 + r;
  auto load_frags = [&](int k0, f32x4& af, f32x4 (&bf)[TN]) __attribute__((always_inline)) {
    const int k = k0 + 4 * kk;
    af = f32x4{0.f, 0.f, 0.f, 0.f};
    if (arow < M && k < K) {
      const float* p = A + (long)arow * g.lda + k;
      if (vecA && k + 3 < K) af = *reinterpret_cast<const f32x4*>(p);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e < K) af[e] = p[e];
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bf[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int col = n0 + 16 * j + r;
      if (col < N && k < K) {
        if (!TB) {
          const float* p = B + (long)col * g.ldb + k;
          if (vecB && k + 3 < K) bf[j] = *reinterpret_cast<const f32x4*>(p);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (k + e < K) bf[j][e] = p[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (k + e < K) bf[j][e] = B[(long)(k + e) * g.ldb + col];
        }
      }
    }
  };
  f32x4 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int KS = 16 * NW;                              // K covered by one round of the workgroup's waves
  f32x4 af, bf[TN], a1 = {0.f, 0.f, 0.f, 0.f}, b1[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) b1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int k0 = w * 16;
  if (k0 < K) load_frags(k0, af, bf);
  if (k0 + KS < K) load_frags(k0 + KS, a1, b1);
  for (; k0 < K; k0 += KS) {
    f32x4 an = {0.f, 0.f, 0.f, 0.f}, bn_[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bn_[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (k0 + 2 * KS < K) load_frags(k0 + 2 * KS, an, bn_);   // two groups in flight under this group's MFMAs
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j][i], acc[j], 0, 0, 0);
    af = a1;
    a1 = an;
#pragma unroll
    for (int j = 0; j < TN; ++j) { bf[j] = b1[j]; b1[j] = bn_[j]; }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) red[w][(kk * 4 + reg) * E_LD + 16 * j + r] = acc[j][reg];
  __syncthreads();
  const float* mulp = g.mul;
  for (int idx = tid; idx < 16 * BN; idx += 64 * NW) {
    const int lr = idx / BN, c = idx - lr * BN;
    const int row = m0 + lr, col = n0 + c;
    if (row >= M || col >= N) continue;
    float x = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) x += red[q][lr * E_LD + c];          // fixed order
    x *= g.alpha;
    if (g.accumulate == 2) x += C[(long)row * g.ldc + col];      // running sum BEFORE bias / activation
    if (g.bias) x += g.bias[col];
    if (g.rowvec) x += g.rowvec[(long)(g.rowvec_map ? g.rowvec_map[row] : row) * g.ldrv + col];
    if (g.act == 1) x = fmaxf(x, 0.f);
    else if (g.act == 2) x = fast_tanh(x);
    else if (g.act == 3) x = fast_sigmoid(x);
    if (aux) aux[(long)row * g.ldaux + col] = x;
    if (mulp) x *= mulp[(long)row * g.ldmul + col];
    if (res) x += res[(long)row * g.ldres + col];
    if (g.drop_target == 3)
      x = nnr_keep(g.drop_seed, (uint64_t)(row + (long)z * g.M) * g.drop_cols + col, g.drop_thresh) ? x * g.drop_scale : 0.f;
    if (C) {
      float* cp = C + (long)row * g.ldc + col;
      if (g.accumulate == 1) x += *cp;
      *cp = x;
    }
  }
}

int launch_skinny(const nnr_gemm_args& g, hipStream_t s) {
  const int tiles = ((g.M + 15) / 16) * ((g.N + 79) / 80) * (g.batch > 1 ? g.batch : 1);
  dim3 grid(((g.M + 15) / 16) * ((g.N + 79) / 80), 1, g.batch > 1 ? g.batch : 1);
  static const int force = [] { const char* e = getenv("NNR_SKINNY_WAVES"); return e ? atoi(e) : 0; }();      // A/B: 4 = the round 1-3 kernel everywhere
  int nw = 4;
  if (tiles <= 640 && g.K >= 384) nw = 8;              // every workgroup resident at once (43 KB of LDS): 8 waves.  (16 waves for the handful-of-tiles
                                                       // launches measured no better than 8 -- batch 8: 26.2 vs 23.6 us per launch incl. dispatch gaps, 28.7 with 4)
  if (force == 4 || force == 8 || force == 16) nw = force;
#define NNR_SKINNY(TBV, NWV) hipLaunchKernelGGL((skinny_gemm_kernel<TBV, NWV>), grid, dim3(64 * NWV), 0, s, g)
  if (g.trans_b) { if (nw == 16) NNR_SKINNY(true, 16); else if (nw == 8) NNR_SKINNY(true, 8); else NNR_SKINNY(true, 4); }
  else { if (nw == 16) NNR_SKINNY(false, 16); else if (nw == 8) NNR_SKINNY(false, 8); else NNR_SKINNY(false, 4); }
#undef NNR_SKINNY
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

// ---------------------------------------------------------------------------------------------- split-K slabs: fixed-order reduction
// Reproducible weight gradients (the reference's runs are seeded and deterministic: config.py:125-130): with `slab` set, slice z of a
// split-K launch STORES its M x N partial tile-by-tile into slab[z] (plain float4 stores at ~6 TB/s instead of f32 atomics at the
// memory-side units' ~1.3 TB/s, all of them at the end of a wave of workgroups) and this kernel adds the live slices in slice order:
// C[m][n] += sum_z slab[z][m][n], colsum_out[m] += sum_z colsum_slab[z][m].  Which slices exist for the live reduction length is
// decided by the same arithmetic as in the GEMM kernel (kind 1: the LDS-DMA TN tiles' device-side slice count; kind 0: the
// register-staged tiles' static split).  The final add is ONE f32 atomic per element and launch (a parameter gradient receives at
// most two such launches per step -- the two encoder calls of the plugin API -- and two addends into a zeroed buffer commute).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ cs_slab, float* __restrict__ C,
                                                            int ldc, float* __restrict__ colsum_out, int M, int N, int K, const int* __restrict__ dyn_dev,
                                                            int split_k, int sched, int nblk, int slots, int bk, int kind) {
  if (dyn_dev) K = min(K, *dyn_dev);
  int live;
  const int ktiles = (K + bk - 1) / bk;
  if (ktiles <= 0) {
    live = 0;
  } else if (kind == 1) {
    const int eff = tn_eff_slices(split_k, sched, K, nblk, slots);
    const int per_k = (ktiles + eff - 1) / eff;
    live = min(eff, (ktiles + per_k - 1) / per_k);
  } else {
    const int per = (ktiles + split_k - 1) / split_k;
    live = per > 0 ? min(split_k, (ktiles + per - 1) / per) : 0;
  }
  const long MN = (long)M * N, MN4 = MN >> 2;               // (N % 4 == 0 is checked by the dispatcher)
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < MN4; i += stride) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 4 <= live; z += 4) {                          // four slices in flight, added in slice order
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const f32x4*>(slab + (long)(z + u) * MN)[i];
#pragma unroll
      for (int u = 0; u < 4; ++u) t += v[u];
    }
    for (; z < live; ++z) t += reinterpret_cast<const f32x4*>(slab + (long)z * MN)[i];
    const long e = i << 2;
    const int m = (int)(e / N), n = (int)(e - (long)m * N);
    float* cp = C + (long)m * ldc + n;
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(cp + q, t[q]);
  }
  if (colsum_out != nullptr && cs_slab != nullptr) {
    for (long m = blockIdx.x * (long)blockDim.x + threadIdx.x; m < M; m += stride) {
      float t = 0.f;
      for (int z = 0; z < live; ++z) t += cs_slab[(long)z * M + m];
      atomicAdd(&colsum_out[m], t);
    }
  }
}

}  // namespace

// tile id -> (rows, columns, stage depth, workgroups per CU the kernel is built for, 1 = LDS-DMA TN tile with the device-side slice count)
static bool tile_shape(int tile, int* bm, int* bn, int* bk, int* occ, int* kind) {
  struct T { int tile, bm, bn, bk, occ, kind; };
  static const T tab[] = {{2, 64, 80, 16, 1, 0}, {3, 128, 208, 16, 1, 0}, {4, 128, 80, 16, 1, 0}, {5, 128, 80, 32, 1, 0}, {6, 64, 80, 64, 1, 0},
                          {20, 128, 80, 16, 3, 1}, {26, 128, 80, 16, 3, 1}, {27, 128, 208, 16, 2, 1}, {30, 128, 160, 16, 2, 1}, {32, 64, 208, 16, 2, 1}};
  for (const T& t : tab)
    if (t.tile == tile) { *bm = t.bm; *bn = t.bn; *bk = t.bk; *occ = t.occ; *kind = t.kind; return true; }
  return false;
}

static int dispatch_tile(int tile, const nnr_gemm_args& g, hipStream_t stream) {
  switch (tile) {
    case 2: return launch_cfg<1, 5, 16>(g, stream);    //  64 x 80
    case 3: return launch_cfg<2, 13, 16>(g, stream);   // 128 x 208 (whole rows in one wave: fused row-dot)
    case 4: return launch_cfg<2, 5, 16>(g, stream);    // 128 x 80 (4 waves/SIMD: more workgroups in flight per CU)
    case 5: return launch_cfg<2, 5, 32>(g, stream);   // 128 x 80, BK = 32: half the barriers per FLOP, 3 workgroups per CU
    case 6: return launch_cfg<1, 5, 64>(g, stream);   //  64 x 80, BK = 64: latency-bound small launches (few stages, 74 KB LDS)
    case 9:  if (!pipe_ok(g) || g.a_idx) return NNR_ERR_ARG; return launch_pipe2<2, 5, 3, 2>(g, stream);    // gen-2 loop, 128 x 80, 3 x 26 KB stages, 2 workgroups / CU
    case 15: if (!pipe_ok(g)) return NNR_ERR_ARG; return launch_pipe<2, 5, 16, 3, 4>(g, stream);   // 128 x 80, BK 16, 3 x 13 KB stages, 4 workgroups / CU
    case 16: if (!pipe_ok(g)) return NNR_ERR_ARG; return launch_pipe<2, 5, 16, 2, 5>(g, stream);   // 128 x 80, BK 16, 2 stages, 5-6 workgroups / CU
    case 20: if (!tn_pipe_ok(g)) return NNR_ERR_ARG; return launch_tn_pipe<2, 5, 3, 3>(g, stream);   // TN 128 x 80, 3 x 16 KB stages
    case 26: if (!tn_pipe_ok(g) || g.b_idx) return NNR_ERR_ARG; return launch_tn_pipe2<2, 5, 3, 3>(g, stream);    // gen-2 TN 128 x 80 (no gather)
    case 27: if (!tn_pipe_ok(g)) return NNR_ERR_ARG; return launch_tn_pipe2<2, 13, 3, 2>(g, stream);              // gen-2 TN 128 x 208
    case 30: if (!tn_pipe_ok(g)) return NNR_ERR_ARG; return launch_tn_pipe2<2, 10, 3, 2>(g, stream);              // gen-2 TN 128 x 160 (N = 300 in two column blocks)
    case 50: if (!bx3_ok(g)) return NNR_ERR_ARG; return launch_bx3<2, 5, 2>(g, stream);           // bf16x3 NT 128 x 80 (needs args.B3: pre-split weights)
    case 51: if (!bx3_ok(g)) return NNR_ERR_ARG; return launch_bx3<1, 5, 2>(g, stream);        // ... 64 x 80: 2 x 23 KB stages, 3 workgroups / CU
    case 32: if (!tn_pipe_ok(g)) return NNR_ERR_ARG; return launch_tn_pipe2<1, 13, 3, 2>(g, stream);              // gen-2 TN 64 x 208, 3 x 20 KB stages: row tiles of 64 fit M = 200 / 400 / 832
                                                                                                                  // (256 / 448 / 832 rows of MFMA work instead of 256 / 512 / 896)
    case 7:
      if (g.pre_add || g.gate_bwd) return NNR_ERR_ARG;
      if (g.trans_a || g.a_idx || g.b_idx || g.c_idx || g.dyn_dev || g.split_k > 1 || g.k_chunk > 0 || g.rowdot_w || g.colsum_out || g.atomic ||
          (g.drop_target != 0 && g.drop_target != 3)) return NNR_ERR_ARG;
      return launch_skinny(g, stream);
    default: return NNR_ERR_ARG;
  }
}

extern "C" int nnr_split_bf16x3(const float* w, int rows, int cols, int ld, int ldo, void* out3, long img_stride, hipStream_t stream) {
  if (!w || !out3 || rows <= 0 || cols <= 0 || ld < cols || ldo < cols || (ldo & 7) || img_stride < (long)rows * ldo) return NNR_ERR_ARG;
  const long total = (long)rows * ldo;
  const int blocks = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(blocks), dim3(256), 0, stream, w, rows, cols, ld, ldo, reinterpret_cast<__bf16*>(out3), img_stride);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}

extern "C" int nnr_gemm_f32(const nnr_gemm_args* a, hipStream_t stream) {
  if (!a || !a->A || !a->B) return NNR_ERR_ARG;
  if (!a->C && !a->rowdot_out && !a->aux_out) return NNR_ERR_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return NNR_OK;
  nnr_gemm_args g = *a;
  g.drop_thresh = nnr_drop_thresh(g.drop_target ? g.drop_p : 0.f);
  g.drop_scale = (g.drop_target && g.drop_p > 0.f) ? 1.f / (1.f - g.drop_p) : 1.f;
  if (g.drop_thresh == 0u) g.drop_target = 0;
  if (g.k_chunk > 0) { if ((g.k_chunk & 31) || g.batch > 1) return NNR_ERR_ARG; g.split_k = 2; }   // shares split-K's restrictions below
  if (g.split_k > 1 && (g.bias || g.rowvec || g.act || g.aux_out || g.mul || g.resid || g.batch > 1 || g.rowdot_w))
    return NNR_ERR_ARG;
  if ((g.a_idx && g.trans_a) || (g.b_idx && !g.trans_b)) return NNR_ERR_ARG;
  if (g.colsum_out && !g.trans_a) return NNR_ERR_ARG;
  if ((g.dyn_dim != 0) != (g.dyn_dev != nullptr)) return NNR_ERR_ARG;
  {
    auto ok = [](const void* p, int ld) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (ld & 3) == 0); };
    const bool strides = g.batch <= 1 || (((g.strideC | g.stride_aux | g.stride_res) & 3) == 0);
    g.vec_epi = (!g.c_idx && !g.atomic && g.split_k <= 1 && !g.rowdot_w && (g.N & 3) == 0 && ok(g.C, g.ldc) && ok(g.bias, 0) &&
                 ok(g.rowvec, g.ldrv) && ok(g.aux_out, g.ldaux) && ok(g.mul, g.ldmul) && ok(g.resid, g.ldres) && strides &&
                 (g.drop_target != 3 || (g.drop_cols & 3) == 0)) ? 1 : 0;
  }
  if (g.pre_add || g.gate_bwd) {
    auto al = [](const void* p, int ld) { return p != nullptr && (((uintptr_t)p) & 15) == 0 && (ld & 3) == 0; };
    if (!g.vec_epi || g.split_k > 1 || g.k_chunk > 0 || g.batch > 1 || (g.pre_add && !al(g.pre_add, g.ldpre))) return NNR_ERR_ARG;
    if (g.gate_bwd && (!g.C || !g.aux_out || !g.mul || !g.resid || g.bias || g.rowvec || g.act || g.drop_target || g.accumulate || g.rowdot_w)) return NNR_ERR_ARG;
  }
  // reproducible split-K: the slices store into the caller's slab, a second launch adds them in slice order (splitk_reduce_kernel)
  float* slab_C = nullptr;
  float* slab_cs = nullptr;
  int slab_ldc = 0;
  if (g.slab != nullptr) {
    if (!(g.trans_a && g.trans_b) || g.split_k <= 1 || g.k_chunk > 0 || g.c_idx || (g.N & 3) || (((uintptr_t)g.slab) & 15) || !g.C || g.accumulate == 2 ||
        g.slab_floats < (long)g.split_k * ((long)g.M * g.N + g.M))
      return NNR_ERR_ARG;
    slab_C = g.C; slab_ldc = g.ldc; slab_cs = g.colsum_out;
    g.C = g.slab; g.ldc = g.N; g.atomic = 0; g.accumulate = 0; g.slab_mode = 1; g.vec_epi = 1;
    if (g.colsum_out) g.colsum_out = g.slab + (long)g.split_k * g.M * g.N;
  } else {
    g.slab_mode = 0;
  }
  int tile = g.tile;
  if (tile == 0) {
    const long wg128 = (long)((g.M + 127) / 128) * ((g.N + 79) / 80) * (g.k_chunk > 0 ? (g.K + g.k_chunk - 1) / g.k_chunk : (g.split_k > 1 ? g.split_k : (g.batch > 1 ? g.batch : 1)));
    const long wg64 = (long)((g.M + 63) / 64) * ((g.N + 79) / 80) * (g.split_k > 1 ? g.split_k : (g.batch > 1 ? g.batch : 1));
    const bool plain = !g.trans_a && !g.a_idx && !g.b_idx && !g.c_idx && !g.dyn_dev && g.split_k <= 1 && g.k_chunk <= 0 && !g.rowdot_w &&
                       !g.colsum_out && !g.atomic && (g.drop_target == 0 || g.drop_target == 3) && !g.pre_add && !g.gate_bwd;
    static const bool use_t9 = [] { const char* e = getenv("NNR_NT9"); return !(e && atoi(e) == 0); }();   // A/B
    static const bool use_pipe = [] { const char* e = getenv("NNR_GEMM_PIPE"); return !(e && atoi(e) == 0); }();   // A/B switch
    if (g.rowdot_w) tile = 3;
    else if (plain && wg64 <= 512 && g.K >= 64) tile = 7;   // small row-parallel launch: 16 x 80 tiles, K split over the 4 waves
    else if (use_pipe && use_t9 && pipe_ok(g) && !g.a_idx && g.K >= 800 && wg64 > 512) {
      tile = 9;              // long reductions (dX: K = 1664, SUE: K = 900): the
                             // software-pipelined loop with the lean DMA issue, 2 workgroups / CU (130 vs 112 TF, 93 vs 79 TF)
    }
    else if (use_pipe && pipe_ok(g) && (g.dyn_dev || wg128 >= 640)) tile = 15;   // GPU-filling NT: LDS-DMA staged 128 x 80, BK 16, 4 workgroups / CU
                             // (112 vs 98 TF on the 131 072-row CNE shapes, tools/gemm_pipe_bench.py)
    else if (use_pipe && pipe_ok(g) && wg64 > 512 && g.K >= 128) tile = 16;      // mid-size NT (SUE: 4 352 x 900 x 900): same tile, two 13 KB stages,
                             // 5-6 workgroups / CU cover the round trips (79 vs 65-72 TF)
    else if (wg64 <= 512 && !g.dyn_dev && g.k_chunk <= 0 && g.K >= 128 && !g.trans_a) tile = 6;   // (not for TN: the small weight-gradient
                             // launches run on the leaf stream BESIDE the recurrence, whose workgroups hold 98 KB of LDS per CU; a 74 KB tile cannot
                             // move in next to them and waits for free CUs, the 19 KB tile can: 12.60 vs 12.70 ms/step)   // at most 2 workgroups per CU: nothing hides the
                             // memory round trip each k-stage pays with a one-stage prefetch -> BK = 64, 4x fewer stages
    else if (g.M <= 512 || (wg128 < 640 && !g.dyn_dev)) tile = 2;   // too few 128-row tiles to fill 256 CUs x 4: use 64-row tiles
    else if (g.trans_a) tile = 2;   // TN (token-reduction dW): callers pick the LDS-DMA tiles 20 / 24 explicitly (their split-K factor
                                    // depends on the tile); this is the register-staged fallback
    else if (!g.trans_a && !g.trans_b) tile = 5;   // NT that the pipelined kernel cannot take (unaligned operands)
    else tile = 4;           // NN with a K-major B (activations on both sides, or a weight the caller did not transpose)
  }
  const int rc = dispatch_tile(tile, g, stream);
  if (rc != NNR_OK || !g.slab_mode) return rc;
  int bm, bn, bk, occ, kind;
  if (!tile_shape(tile, &bm, &bn, &bk, &occ, &kind)) return NNR_ERR_ARG;
  const int nblk = ((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
  const long mn4 = ((long)g.M * g.N) >> 2;
  const int blocks = (int)(mn4 / 256 + 1 > 2048 ? 2048 : mn4 / 256 + 1);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)g.slab, (const float*)g.colsum_out, slab_C, slab_ldc, slab_cs,
                     g.M, g.N, g.K, g.dyn_dim == 2 ? g.dyn_dev : (const int*)nullptr, g.split_k, tn_sched_bits(), nblk, occ * 256, bk, kind);
  NNR_CHECK_LAUNCH();
  return NNR_OK;
}
