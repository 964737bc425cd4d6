// Shared device helpers for libnnr_hip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nnr_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NNR_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) return NNR_ERR_LAUNCH;             \
  } while (0)

// ---------------------------------------------------------------- counter-based dropout
// keep(idx) is a pure function of (seed, idx): forward and backward recompute the same mask,
// nothing is stored.  lowbias32 finaliser (two rounds) -- statistical quality is ample for dropout.
__device__ __forceinline__ uint32_t nnr_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
// One 32-bit hash serves TWO neighbouring elements (16 bits each): keep(idx) tests bits [16*(idx&1), +16) of hash(idx>>1)
// against a 16-bit threshold, so P(drop) is p rounded to 1/65536 and a float4 of elements costs two hashes.
__device__ __forceinline__ uint32_t nnr_hash_pair(uint32_t seed, uint64_t pair) {
  return nnr_hash32((uint32_t)pair ^ nnr_hash32((uint32_t)(pair >> 32) + seed));
}
__device__ __forceinline__ bool nnr_keep(uint32_t seed, uint64_t idx, uint32_t thresh) {
  const uint32_t h = nnr_hash_pair(seed, idx >> 1);
  return ((h >> (16u * (uint32_t)(idx & 1))) & 0xFFFFu) >= thresh;   // P(keep) = 1 - thresh / 65536
}
// four consecutive elements starting at an index that is a multiple of 4 (the common float4 case): two hashes
__device__ __forceinline__ void nnr_keep4(uint32_t seed, uint64_t idx4, uint32_t thresh, bool (&k)[4]) {
  const uint32_t h0 = nnr_hash_pair(seed, idx4 >> 1), h1 = nnr_hash_pair(seed, (idx4 >> 1) + 1);
  k[0] = (h0 & 0xFFFFu) >= thresh; k[1] = (h0 >> 16) >= thresh; k[2] = (h1 & 0xFFFFu) >= thresh; k[3] = (h1 >> 16) >= thresh;
}
static inline uint32_t nnr_drop_thresh(float p) {
  if (p <= 0.f) return 0u;
  double t = (double)p * 65536.0 + 0.5;
  if (t > 65535.0) t = 65535.0;
  return (uint32_t)t;
}

// ---------------------------------------------------------------- wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the 16 lanes that share (lane >> 4)
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// Hardware-transcendental activations for the LSTM cell update, which sits on the critical path of the recurrence
// (128 dependent steps): v_exp_f32 + v_rcp_f32, ~4 instructions instead of the ~40-60 of OCML expf / tanhf / division.
// Absolute error ~2e-7 (1-2 ulp of the result), far inside the 1e-4 parity bar (tests/test_hip_ops_gpu.py: 2e-5).
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.f * fast_sigmoid(2.f * x) - 1.f; }
