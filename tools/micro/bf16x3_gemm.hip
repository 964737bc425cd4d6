// Exploratory micro-benchmark (round-4 verdict, item 7; NOT part of libnnr_hip.so): an f32 NT GEMM  C[M,N] = A[M,K] . B[N,K]^T  computed on
// the BF16 matrix pipe without narrowing the arithmetic.  An fp32 value is EXACTLY the sum of three bf16 values
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)        (8 + 8 + 8 significant bits)
// so a . b = sum_{i,j} a_i b_j; the six terms with i + j <= 4 carry everything above 2^-26 |a b| (the dropped a2 b3, a3 b2, a3 b3 are below
// the rounding of an fp32 product, 2^-24), every bf16 x bf16 product is exact in fp32, and v_mfma_f32_16x16x32_bf16 accumulates in fp32.
// gfx950 issues bf16 MFMA at 16x the f32-MFMA rate (MI355X_MICROARCH.md, Matrix cores), so six bf16 MFMAs per k-chunk cost 6/16 of the
// eight v_mfma_f32_16x16x4_f32 they replace.
//
// What is measured, on the gate-GEMM shape of the CNE step (M = 450 560, N = 400, K = 400) and on the input-projection shape
// (N = 1 664, K = 300):
//   * native:  nnr_gemm_f32 of libnnr_hip.so (gemm_nt_pipe_kernel, v_mfma_f32_16x16x4_f32) -- TFLOP/s and error vs an fp64 product;
//   * bf16x3:  the kernel below -- A (activations, fp32 in HBM) is split on the way into LDS (global -> registers -> 3 bf16 images),
//              B (weights) is pre-split once by split3_kernel (3 bf16 copies in HBM, 1.5x the weight bytes, KBs); 6 MFMAs per
//              (16 x 16 x 32) block into TWO fp32 accumulators (a1 b1 | the five small terms); same LDS-staged coalesced epilogue.
// Kill criterion of the verdict: < 1.25x the native kernel, or an error above the native kernel's.
//
// Build (cross-compiles without a GPU):  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/bf16x3_gemm.hip -o tools/micro/bf16x3_gemm \
//                                              -Lnnr_amd -lnnr_hip -Wl,-rpath,$PWD/nnr_amd
// Run on the GPU box:  tools/micro/bf16x3_gemm [M N K]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../../include/nnr_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__device__ __forceinline__ void split3(float x, __bf16& h1, __bf16& h2, __bf16& h3) {
  h1 = (__bf16)x;                       // round-to-nearest-even (v_cvt_pk_bf16_f32)
  const float r1 = x - (float)h1;       // exact: at most 16 significant bits remain
  h2 = (__bf16)r1;
  const float r2 = r1 - (float)h2;      // exact: at most 8 significant bits remain
  h3 = (__bf16)r2;                      // exact
}

__global__ void split3_kernel(const float* __restrict__ x, long n, __bf16* __restrict__ o1, __bf16* __restrict__ o2, __bf16* __restrict__ o3) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) split3(x[i], o1[i], o2[i], o3[i]);
}

// chunk swizzle of the [row][32 k] bf16 images (64-B rows, four 16-B chunks): conflict-free ds_read_b128 fragments (brute-force checked
// against the lane groups of MI355X_MICROARCH.md, LDS)
__device__ __forceinline__ int swz(int r) { return (r >> 1) & 3; }

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_nt(const float* __restrict__ A, const __bf16* __restrict__ Bs, long Bstride,
                                                         float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 64 * TM, BN = 16 * TN, BK = 32;
  constexpr int A_IMG = BM * BK, B_IMG = BN * BK;                 // bf16 elements per image
  constexpr int STAGE = 3 * (A_IMG + B_IMG);                      // bf16 elements per stage
  constexpr int E_LD = BN + 4;
  constexpr int LDS_BYTES = (2 * STAGE * 2 > 64 * E_LD * 4) ? 2 * STAGE * 2 : 64 * E_LD * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
  __bf16* lds = reinterpret_cast<__bf16*>(lds_raw);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 15, q = lane >> 4;
  const int nbn = (N + BN - 1) / BN, nbm = (M + BM - 1) / BM, nblk = nbm * nbn;
  int v;
  {   // XCD-aware: the 8 XCDs each walk a contiguous range of tiles (column blocks fastest)
    const int b = blockIdx.x, qq = nblk >> 3, rem = nblk & 7, x = b & 7, slot = b >> 3;
    v = x * qq + min(x, rem) + slot;
  }
  const int bm = v / nbn, bn = v - bm * nbn, m0 = bm * BM, n0 = bn * BN;
  const int S = (K + BK - 1) / BK;

  // ---- staging geometry.  A: thread t covers row t >> 1 (rows 0..127: TM = 2), k-half t & 1 (16 floats = 4 float4).
  static_assert(BM == 128, "A staging assumes a 128-row tile");
  const int arow = tid >> 1, ah = tid & 1;
  const float* ap = A + (long)min(m0 + arow, M - 1) * K + ah * 16;      // rows past the edge are clamped: they only feed rows never stored
  // B: 3 images (Bs + img * Bstride) x BN rows x 4 chunks of 16 B; chunk id c = tid + 256 j < 3 * BN * 4
  constexpr int BCH = 3 * BN * 4, BJ = (BCH + 255) / 256;
  f32x4 ar[4];
  uint4 br[BJ];
  // per-thread B chunk geometry (constant over the stages): image, row, chunk -> global offset (bf16 elements) and LDS offset
  long boff[BJ];
  int bk8[BJ], blds[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int c = tid + 256 * j;
    const int img = c / (BN * 4), rc = c - img * BN * 4, row = rc >> 2, ch = rc & 3;
    boff[j] = (c < BCH) ? (long)img * Bstride + (long)min(n0 + row, N - 1) * K + ch * 8 : -1;
    bk8[j] = ch * 8;
    blds[j] = 3 * A_IMG + img * B_IMG + row * BK + ((ch ^ swz(row & 15)) * 8);
  }
#define GLOAD(s_)                                                                                                             \
  do {                                                                                                                        \
    const int k0_ = (s_) * BK;                                                                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                           \
      const int k = k0_ + ah * 16 + 4 * i;                                                                                    \
      ar[i] = (k < K) ? *reinterpret_cast<const f32x4*>(ap + k0_ + 4 * i) : f32x4{0.f, 0.f, 0.f, 0.f};                        \
    }                                                                                                                         \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j) {                                                                          \
      br[j] = uint4{0u, 0u, 0u, 0u};                                                                                          \
      if (boff[j] >= 0 && k0_ + bk8[j] < K) br[j] = *reinterpret_cast<const uint4*>(Bs + boff[j] + k0_);                      \
    }                                                                                                                         \
  } while (0)
#define LSTORE(buf_)                                                                                                          \
  do {                                                                                                                        \
    __bf16* st_ = lds + (buf_) * STAGE;                                                                                       \
    bf16x8 h0[2], h1[2], h2[2];                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int e = 0; e < 4; ++e) {                             \
      __bf16 a_, b_, c_;                                                                                                      \
      split3(ar[i][e], a_, b_, c_);                                                                                           \
      h0[i >> 1][(i & 1) * 4 + e] = a_; h1[i >> 1][(i & 1) * 4 + e] = b_; h2[i >> 1][(i & 1) * 4 + e] = c_;                    \
    }                                                                                                                         \
    _Pragma("unroll") for (int cc = 0; cc < 2; ++cc) {                                                                        \
      const int ch = (ah * 2 + cc) ^ swz(arow & 15);                                                                          \
      *reinterpret_cast<bf16x8*>(st_ + 0 * A_IMG + arow * BK + ch * 8) = h0[cc];                                              \
      *reinterpret_cast<bf16x8*>(st_ + 1 * A_IMG + arow * BK + ch * 8) = h1[cc];                                              \
      *reinterpret_cast<bf16x8*>(st_ + 2 * A_IMG + arow * BK + ch * 8) = h2[cc];                                              \
    }                                                                                                                         \
    _Pragma("unroll") for (int j = 0; j < BJ; ++j)                                                                            \
      if (boff[j] >= 0) *reinterpret_cast<uint4*>(st_ + blds[j]) = br[j];                                                     \
  } while (0)

  f32x4 acc_hi[TM][TN], acc_lo[TM][TN];
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int n = 0; n < TN; ++n) acc_hi[m][n] = acc_lo[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  GLOAD(0);
  LSTORE(0);
  __syncthreads();
  for (int s = 0; s < S; ++s) {
    if (s + 1 < S) GLOAD(s + 1);                                   // in flight under this stage's MFMAs
    const __bf16* st = lds + (s & 1) * STAGE;
    bf16x8 af[3][TM], bf[3][TN];
#pragma unroll
    for (int img = 0; img < 3; ++img) {
#pragma unroll
      for (int m = 0; m < TM; ++m)
        af[img][m] = *reinterpret_cast<const bf16x8*>(st + img * A_IMG + ((w * TM + m) * 16 + r) * BK + ((q ^ swz(r)) * 8));
#pragma unroll
      for (int n = 0; n < TN; ++n)
        bf[img][n] = *reinterpret_cast<const bf16x8*>(st + 3 * A_IMG + img * B_IMG + (n * 16 + r) * BK + ((q ^ swz(r)) * 8));
    }
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
      for (int n = 0; n < TN; ++n) {
        acc_hi[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][m], bf[0][n], acc_hi[m][n], 0, 0, 0);      // a1 b1
        acc_lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[2][m], bf[0][n], acc_lo[m][n], 0, 0, 0);      // a3 b1   (smallest first)
        acc_lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][m], bf[2][n], acc_lo[m][n], 0, 0, 0);      // a1 b3
        acc_lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][m], bf[1][n], acc_lo[m][n], 0, 0, 0);      // a2 b2
        acc_lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][m], bf[0][n], acc_lo[m][n], 0, 0, 0);      // a2 b1
        acc_lo[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][m], bf[1][n], acc_lo[m][n], 0, 0, 0);      // a1 b2
      }
    if (s + 1 < S) LSTORE((s + 1) & 1);                            // the other buffer: free since the barrier that closed stage s - 1
    __syncthreads();
  }
  // ---- epilogue: accumulators through LDS, coalesced float4 stores (as gemm_epilogue of csrc/gemm.hip, plain store)
  float* stage = reinterpret_cast<float*>(lds_raw);
#pragma unroll
  for (int m = 0; m < TM; ++m) {
    if (m > 0) __syncthreads();
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) stage[(w * 16 + q * 4 + reg) * E_LD + n * 16 + r] = acc_hi[m][n][reg] + acc_lo[m][n][reg];
    __syncthreads();
    constexpr int NV = BN / 4;
    for (int idx = tid; idx < 64 * NV; idx += 256) {
      const int lr = idx / NV, c4 = idx - lr * NV;
      const int row = m0 + (lr >> 4) * (TM * 16) + m * 16 + (lr & 15), col = n0 + 4 * c4;
      if (row < M && col < N) *reinterpret_cast<f32x4*>(C + (long)row * N + col) = *reinterpret_cast<const f32x4*>(&stage[lr * E_LD + 4 * c4]);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------- version 2
// The staging of libnnr_hip.so's gemm_nt_pipe kernels (LDS-DMA, `global_load_lds_dwordx4`: 1 KiB per wave-instruction straight into LDS, no
// staging registers, NS stage buffers, one counted vmcnt wait + one barrier per stage) with the bf16x3 arithmetic:
//   * A (activations) is DMA'd as fp32 and split IN REGISTERS, by the wave that owns the rows, right in front of its MFMAs (a wave's A
//     fragments are private: no redundant conversion, no LDS round trip of converted data);
//   * B (weights) is DMA'd from the three pre-split bf16 images.
__device__ __attribute__((aligned(1024))) float zero_page[512] = {};
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ int swzA(int r) { return ((r >> 1) & 1) | (((r >> 3) & 1) << 2); }      // 128-B fp32 rows, two b128 reads per lane: brute-force checked

template <int TM, int TN, int NS>
__global__ __launch_bounds__(256, 1) void gemm_bf16x3_v2(const float* __restrict__ A, const __bf16* __restrict__ Bs, long Bstride,
                                                         float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 64 * TM, BN = 16 * TN, BK = 32;
  constexpr int A_BYTES = BM * BK * 4, B_IMG_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + 3 * B_IMG_BYTES;
  constexpr int NIA = A_BYTES / 1024, NIB1 = B_IMG_BYTES / 1024, NI = NIA + 3 * NIB1;
  static_assert(A_BYTES % 1024 == 0 && B_IMG_BYTES % 1024 == 0 && NIA % 4 == 0, "tile shape");
  constexpr int E_LD = BN + 4;
  constexpr int LDS_BYTES = NS * STAGE_BYTES > 64 * E_LD * 4 ? NS * STAGE_BYTES : 64 * E_LD * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, q = lane >> 4;
  const int nbn = (N + BN - 1) / BN, nbm = (M + BM - 1) / BM, nblk = nbm * nbn;
  int v;
  {
    const int b = blockIdx.x, qq = nblk >> 3, rem = nblk & 7, x = b & 7, slot = b >> 3;
    v = x * qq + min(x, rem) + slot;
  }
  const int bm = v / nbn, bn = v - bm * nbn, m0 = bm * BM, n0 = bn * BN;
  const int S = (K + BK - 1) / BK;
  const unsigned lds_base = (unsigned)(uintptr_t)lds_raw;
  const float* zero = zero_page;
  asm volatile("" : "+s"(zero));
  // ---- DMA geometry: wave w issues A instructions w, w + 4, ... (NIA / 4 each) and the B instructions idx (0 .. 3 NIB1 - 1) with idx % 4 == w
  constexpr int NA = NIA / 4, NBW = (3 * NIB1 + 3) / 4;
  const char* asrc[NA];
  int akc[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (w + 4 * i) * 8 + lane / 8, c = (lane % 8) ^ swzA(row & 15);
    akc[i] = 4 * c;
    asrc[i] = reinterpret_cast<const char*>(A + (long)min(m0 + row, M - 1) * K + 4 * c);
  }
  const char* bsrc[NBW];
  int bkc[NBW];
  bool bon[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int idx = w + 4 * j;
    bon[j] = idx < 3 * NIB1;
    const int img = bon[j] ? idx / NIB1 : 0, jj = idx - img * NIB1, row = jj * 16 + lane / 4, c = (lane % 4) ^ swz(row & 15);
    bkc[j] = 8 * c;
    bsrc[j] = reinterpret_cast<const char*>(Bs + (long)img * Bstride + (long)min(n0 + row, N - 1) * K + 8 * c);
  }
  const int my_cnt = NA + ((3 * NIB1) / 4) + ((w < (3 * NIB1) % 4) ? 1 : 0);       // this wave's DMA instructions per stage
  auto issue = [&](int s) __attribute__((always_inline)) {
    const int k0 = s * BK;
    const unsigned sb = lds_base + (unsigned)((s % NS) * STAGE_BYTES);
#pragma unroll
    for (int i = 0; i < NA; ++i)
      lds_dma16((k0 + akc[i] < K) ? (const void*)(asrc[i] + (long)k0 * 4) : (const void*)zero, sb + (unsigned)((w + 4 * i) * 1024));
#pragma unroll
    for (int j = 0; j < NBW; ++j)
      if (bon[j]) lds_dma16((k0 + bkc[j] < K) ? (const void*)(bsrc[j] + (long)k0 * 2) : (const void*)zero, sb + A_BYTES + (unsigned)((w + 4 * j) * 1024));
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
    // stage s has landed once at most min(NS - 2, S - 1 - s) newer stages of this wave's own DMAs are outstanding
    const int ahead = min(NS - 2, S - 1 - s);
    if (ahead >= 2) { if (my_cnt == NA + 4) wait_vmcnt<2 * (NA + 4)>(); else wait_vmcnt<2 * (NA + 3)>(); }
    else if (ahead == 1) { if (my_cnt == NA + 4) wait_vmcnt<NA + 4>(); else wait_vmcnt<NA + 3>(); }
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (s + NS - 1 < S) issue(s + NS - 1);
    const unsigned char* st = lds_raw + (s % NS) * STAGE_BYTES;
    const float* As = reinterpret_cast<const float*>(st);
    const __bf16* Bi = reinterpret_cast<const __bf16*>(st + A_BYTES);
    bf16x8 bf[3][TN];
#pragma unroll
    for (int img = 0; img < 3; ++img)
#pragma unroll
      for (int n = 0; n < TN; ++n)
        bf[img][n] = *reinterpret_cast<const bf16x8*>(Bi + img * (BN * BK) + (n * 16 + r) * BK + ((q ^ swz(r)) * 8));
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      const float* arow = As + ((w * TM + m) * 16 + r) * BK;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(arow + 4 * ((2 * q) ^ swzA(r)));
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(arow + 4 * ((2 * q + 1) ^ swzA(r)));
      bf16x8 a1, a2, a3;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        __bf16 h1, h2, h3;
        split3(x0[e], h1, h2, h3); a1[e] = h1; a2[e] = h2; a3[e] = h3;
        split3(x1[e], h1, h2, h3); a1[4 + e] = h1; a2[4 + e] = h2; a3[4 + e] = h3;
      }
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
  __syncthreads();
  float* stage = reinterpret_cast<float*>(lds_raw);
#pragma unroll
  for (int m = 0; m < TM; ++m) {
    if (m > 0) __syncthreads();
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) stage[(w * 16 + q * 4 + reg) * E_LD + n * 16 + r] = acc_hi[m][n][reg] + acc_lo[m][n][reg];
    __syncthreads();
    constexpr int NV = BN / 4;
    for (int idx = tid; idx < 64 * NV; idx += 256) {
      const int lr = idx / NV, c4 = idx - lr * NV;
      const int row = m0 + (lr >> 4) * (TM * 16) + m * 16 + (lr & 15), col = n0 + 4 * c4;
      if (row < M && col < N) *reinterpret_cast<f32x4*>(C + (long)row * N + col) = *reinterpret_cast<const f32x4*>(&stage[lr * E_LD + 4 * c4]);
    }
  }
}

__global__ void ref64_kernel(const float* __restrict__ A, const float* __restrict__ B, double* __restrict__ C, int rows, int N, int K) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)rows * N) return;
  const int m = (int)(i / N), n = (int)(i - (long)m * N);
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += (double)A[(long)m * K + k] * (double)B[(long)n * K + k];
  C[i] = s;
}

struct Err { double max_abs, rel_l2, max_scaled; };
static Err compare(const std::vector<float>& got, const std::vector<double>& ref, const std::vector<double>& absref) {
  double ma = 0, num = 0, den = 0, ms = 0;
  for (size_t i = 0; i < ref.size(); ++i) {
    const double d = fabs((double)got[i] - ref[i]);
    ma = fmax(ma, d); num += d * d; den += ref[i] * ref[i];
    ms = fmax(ms, d / absref[i]);                  // error relative to sum_k |a_k b_k| (the natural scale of a dot product's rounding)
  }
  return Err{ma, sqrt(num / den), ms};
}

__global__ void absref_kernel(const float* __restrict__ A, const float* __restrict__ B, double* __restrict__ C, int rows, int N, int K) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)rows * N) return;
  const int m = (int)(i / N), n = (int)(i - (long)m * N);
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += fabs((double)A[(long)m * K + k] * (double)B[(long)n * K + k]);
  C[i] = s;
}

static void fill(std::vector<float>& v, uint64_t seed, float scale) {      // ~N(0,1) * scale (sum of 4 uniforms), deterministic
  uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
  for (auto& x : v) {
    float a = 0.f;
    for (int i = 0; i < 4; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a += (float)((s >> 40) & 0xFFFFFF) / 16777216.f - 0.5f; }
    x = a * 1.7320508f * scale;
  }
}

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 450560, N = argc > 3 ? atoi(argv[2]) : 400, K = argc > 3 ? atoi(argv[3]) : 400;
  const int ROWS = 2048;            // rows checked against fp64
  printf("bf16x3 GEMM micro-benchmark: C[%d,%d] = A[%d,%d] . B[%d,%d]^T\n", M, N, M, K, N, K);
  if (K % 8 || N % 4) { fprintf(stderr, "K %% 8 == 0 and N %% 4 == 0 required\n"); return 1; }
  std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
  fill(hA, 1, 1.0f); fill(hB, 2, 0.05f);
  float *A, *B, *C0, *C1;
  __bf16 *B1, *B2, *B3;
  double *R, *RA;
  CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4));
  CK(hipMalloc(&C0, (size_t)M * N * 4)); CK(hipMalloc(&C1, (size_t)M * N * 4));
  CK(hipMalloc(&B1, 3 * hB.size() * 2)); B2 = B1 + hB.size(); B3 = B2 + hB.size();
  CK(hipMalloc(&R, (size_t)ROWS * N * 8)); CK(hipMalloc(&RA, (size_t)ROWS * N * 8));
  CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  nnr_gemm_args g;
  memset(&g, 0, sizeof(g));
  g.A = A; g.B = B; g.C = C0; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.alpha = 1.f;
  constexpr int TM = 2, TN = 5;
  const int nblk = ((M + 64 * TM - 1) / (64 * TM)) * ((N + 16 * TN - 1) / (16 * TN));
  auto run_native = [&]() { if (nnr_gemm_f32(&g, st) != 0) { fprintf(stderr, "nnr_gemm_f32 failed\n"); exit(3); } };
  auto run_split = [&]() { hipLaunchKernelGGL(split3_kernel, dim3(64), dim3(256), 0, st, B, (long)N * K, B1, B2, B3); };
  auto run_x3 = [&]() { hipLaunchKernelGGL((gemm_bf16x3_nt<TM, TN>), dim3(nblk), dim3(256), 0, st, A, B1, (long)N * K, C1, M, N, K); };
  float* C2;
  CK(hipMalloc(&C2, (size_t)M * N * 4));
  const int nblk4 = ((M + 255) / 256) * ((N + 16 * TN - 1) / (16 * TN));
  auto run_v2 = [&]() { hipLaunchKernelGGL((gemm_bf16x3_v2<TM, TN, 3>), dim3(nblk), dim3(256), 0, st, A, B1, (long)N * K, C2, M, N, K); };
  auto run_v2c = [&]() { hipLaunchKernelGGL((gemm_bf16x3_v2<TM, TN, 2>), dim3(nblk), dim3(256), 0, st, A, B1, (long)N * K, C2, M, N, K); };
  auto run_v2b = [&]() { hipLaunchKernelGGL((gemm_bf16x3_v2<4, TN, 3>), dim3(nblk4), dim3(256), 0, st, A, B1, (long)N * K, C2, M, N, K); };
  run_split(); run_native(); run_x3(); run_v2();
  CK(hipStreamSynchronize(st));
  CK(hipGetLastError());
  // ---- error vs fp64 on the first ROWS rows
  hipLaunchKernelGGL(ref64_kernel, dim3((ROWS * N + 255) / 256), dim3(256), 0, st, A, B, R, ROWS, N, K);
  hipLaunchKernelGGL(absref_kernel, dim3((ROWS * N + 255) / 256), dim3(256), 0, st, A, B, RA, ROWS, N, K);
  CK(hipStreamSynchronize(st));
  std::vector<double> ref((size_t)ROWS * N), absref((size_t)ROWS * N);
  std::vector<float> c0((size_t)ROWS * N), c1((size_t)ROWS * N);
  CK(hipMemcpy(ref.data(), R, ref.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(absref.data(), RA, absref.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c0.data(), C0, c0.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(c1.data(), C1, c1.size() * 4, hipMemcpyDeviceToHost));
  std::vector<float> c2((size_t)ROWS * N);
  CK(hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost));
  const Err e0 = compare(c0, ref, absref), e1 = compare(c1, ref, absref), e2 = compare(c2, ref, absref);
  run_v2b();
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(c2.data(), C2, c2.size() * 4, hipMemcpyDeviceToHost));
  const Err e3 = compare(c2, ref, absref);
  // whole-matrix agreement of the two kernels (catches a tile that is wrong outside the checked rows)
  std::vector<float> f0((size_t)M * N), f1((size_t)M * N);
  CK(hipMemcpy(f0.data(), C0, f0.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(f1.data(), C1, f1.size() * 4, hipMemcpyDeviceToHost));
  double dmax = 0;
  for (size_t i = 0; i < f0.size(); ++i) dmax = fmax(dmax, fabs((double)f0[i] - (double)f1[i]));
  // ---- timing: interleaved rounds
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int IT = 10, ROUNDS = 5;
  double best[6] = {1e9, 1e9, 1e9, 1e9, 1e9, 1e9}, med[6][ROUNDS];
  for (int rd = 0; rd < ROUNDS; ++rd)
    for (int which = 0; which < 6; ++which) {
      CK(hipEventRecord(a, st));
      for (int i = 0; i < IT; ++i) { if (which == 0) run_native(); else if (which == 1) run_x3(); else if (which == 2) run_split(); else if (which == 3) run_v2(); else if (which == 4) run_v2b(); else run_v2c(); }
      CK(hipEventRecord(b, st));
      CK(hipEventSynchronize(b));
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      med[which][rd] = ms / IT;
      best[which] = fmin(best[which], ms / IT);
    }
  const double fl = 2.0 * M * N * K;
  printf("  native f32 MFMA (nnr_gemm_f32): best %.1f us = %.1f TFLOP/s   | max|err| %.3e  rel-L2 %.3e  max err / sum|ab| %.3e\n", 1e3 * best[0],
         fl / best[0] / 1e9, e0.max_abs, e0.rel_l2, e0.max_scaled);
  printf("  bf16x3 (6 bf16 MFMAs, 2 acc):   best %.1f us = %.1f TFLOP/s-equivalent | max|err| %.3e  rel-L2 %.3e  max err / sum|ab| %.3e\n", 1e3 * best[1],
         fl / best[1] / 1e9, e1.max_abs, e1.rel_l2, e1.max_scaled);
  printf("  bf16x3 v2 (LDS-DMA staging, A split in registers), 128 x 80: best %.1f us = %.1f TFLOP/s-equivalent = %.2fx | rel-L2 %.3e  max err / sum|ab| %.3e\n",
         1e3 * best[3], fl / best[3] / 1e9, best[0] / best[3], e2.rel_l2, e2.max_scaled);
  printf("  bf16x3 v2, 256 x 80 tile:                                          best %.1f us = %.1f TFLOP/s-equivalent = %.2fx | rel-L2 %.3e  max err / sum|ab| %.3e\n",
         1e3 * best[4], fl / best[4] / 1e9, best[0] / best[4], e3.rel_l2, e3.max_scaled);
  printf("  bf16x3 v2, 128 x 80, 2 stages (2 workgroups / CU):                 best %.1f us = %.1f TFLOP/s-equivalent = %.2fx\n", 1e3 * best[5], fl / best[5] / 1e9, best[0] / best[5]);
  printf("  weight pre-split (once per optimizer step): %.1f us;  max |native - bf16x3| over the whole matrix %.3e\n", 1e3 * best[2], dmax);
  const double bx = fmin(fmin(best[1], best[3]), fmin(best[4], best[5]));
  const double worst_l2 = fmax(e1.rel_l2, fmax(e2.rel_l2, e3.rel_l2)), worst_sc = fmax(e1.max_scaled, fmax(e2.max_scaled, e3.max_scaled));
  printf("  best bf16x3 variant: %.2fx the native kernel (kill criterion: < 1.25x, or error above the native kernel's)  -> %s\n", best[0] / bx,
         (best[0] / bx >= 1.25 && worst_l2 <= e0.rel_l2 * 1.05 && worst_sc <= e0.max_scaled * 1.05) ? "SURVIVES" : "KILLED");
  printf("JSON {\"M\": %d, \"N\": %d, \"K\": %d, \"native_us\": %.2f, \"native_tflops\": %.2f, \"bf16x3_us\": %.2f, \"bf16x3_tflops_equiv\": %.2f, \"speedup\": %.3f, "
         "\"native_rel_l2\": %.3e, \"bf16x3_rel_l2\": %.3e, \"native_max_err_over_sum_abs\": %.3e, \"bf16x3_max_err_over_sum_abs\": %.3e, \"split_us\": %.2f, \"v2_128x80_us\": %.2f, \"v2_128x80_speedup\": %.3f, \"v2_256x80_us\": %.2f, \"v2_256x80_speedup\": %.3f, \"v2_rel_l2\": %.3e}\n",
         M, N, K, 1e3 * best[0], fl / best[0] / 1e9, 1e3 * best[1], fl / best[1] / 1e9, best[0] / best[1], e0.rel_l2, e1.rel_l2, e0.max_scaled, e1.max_scaled, 1e3 * best[2], 1e3 * best[3], best[0] / best[3], 1e3 * best[4], best[0] / best[4], e2.rel_l2);
  return 0;
}
