// v_mfma_f32_4x4x1_16B_f32 on gfx950: operand layout check (16 blocks of 4x4, A broadcast from block 0 with CBSZ=4) and the issue
// rate of dependent accumulation chains.  hipcc --offload-arch=gfx950 -O3 mfma4.hip -o mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// D[b][i][j] = sum_k A[i][k] * B[k][4b + j]:  4 rows (lanes 0-3 hold A, broadcast to all 16 blocks), 64 columns (lane = column)
__global__ void layout(const float* A, const float* B, float* D, int K) {
  const int lane = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < K; ++k) {
    const float a = A[(lane & 3) * K + k];     // only lanes 0..3 matter
    const float b = B[k * 64 + lane];
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 4, 0, 0);
  }
  for (int i = 0; i < 4; ++i) D[i * 64 + lane] = acc[i];
}

// A[i][k0 + b] in lane (b, i); ABID = b broadcasts block b's A: 16 instructions consume one VGPR of A (16 k)
__global__ void layout_abid(const float* A, const float* B, float* D, int K) {
  const int lane = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float a = A[(lane & 3) * K + k0 + (lane >> 2)];
#define STEP(j) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, B[(k0 + j) * 64 + lane], acc, 4, j, 0);
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
  }
  for (int i = 0; i < 4; ++i) D[i * 64 + lane] = acc[i];
}
// no broadcast: block b multiplies ITS OWN A (lane (b, i) = row i, k index kq(b)): columns 4b..4b+3 see k-quarter b >> 2
__global__ void layout_own(const float* A, const float* B, float* D, int K) {      // D[i][lane] = sum_{k in quarter(lane>>4)} A[i][k] B[k][lane]
  const int lane = threadIdx.x, q = lane >> 4, KQ = K / 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < KQ; ++k)
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[(lane & 3) * K + q * KQ + k], B[(q * KQ + k) * 64 + lane], acc, 0, 0, 0);
  for (int i = 0; i < 4; ++i) D[i * 64 + lane] = acc[i];
}

template <int NACC>
__global__ void rate(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f;
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u % NACC], 4, 0, 0);
  }
  long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[4096] = (float)(t1 - t0) / (iters * 16.f);
  if ((threadIdx.x & 63) == 0) { ((long*)(out + 5000))[2 * (threadIdx.x >> 6)] = t0; ((long*)(out + 5000))[2 * (threadIdx.x >> 6) + 1] = t1; }
}

template <int NACC>
__global__ void rate16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f;
  __syncthreads();
  long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % NACC], 0, 0, 0);
  }
  long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[4096] = (float)(t1 - t0) / (iters * 16.f);
  if ((threadIdx.x & 63) == 0) { ((long*)(out + 5000))[2 * (threadIdx.x >> 6)] = t0; ((long*)(out + 5000))[2 * (threadIdx.x >> 6) + 1] = t1; }
}

static float span(float* dD, int nw, int iters) {
  long h[32];
  hipMemcpy(h, dD + 5000, nw * 16, hipMemcpyDeviceToHost);
  long lo = h[0], hi = h[1];
  for (int i = 0; i < nw; ++i) { if (h[2 * i] < lo) lo = h[2 * i]; if (h[2 * i + 1] > hi) hi = h[2 * i + 1]; }
  return (float)(hi - lo) / (iters * 16.f);
}

int main() {
  const int K = 32;
  std::vector<float> A(4 * K), B(K * 64), D(256), R(256, 0.f);
  for (int i = 0; i < 4 * K; ++i) A[i] = (float)((i * 7) % 11) - 5.f;
  for (int i = 0; i < K * 64; ++i) B[i] = (float)((i * 13) % 17) - 8.f;
  for (int i = 0; i < 4; ++i) for (int c = 0; c < 64; ++c) for (int k = 0; k < K; ++k) R[i * 64 + c] += A[i * K + k] * B[k * 64 + c];
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 8192 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
  hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
  printf("layout: D[i][lane] = sum_k A[i][k] B[k][lane]  mismatches %d / 256\n", bad);
  hipLaunchKernelGGL(layout_abid, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
  hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
  int bad2 = 0;
  for (int i = 0; i < 256; ++i) bad2 += D[i] != R[i];
  printf("abid: one A register (lane (b,i) = A[i][k0+b]) + ABID = 0..15   mismatches %d / 256\n", bad2);
  hipLaunchKernelGGL(layout_own, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
  hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
  int bad3 = 0;
  for (int i = 0; i < 4; ++i) for (int c = 0; c < 64; ++c) {
    float r = 0.f; const int q = c >> 4, KQ = K / 4;
    for (int k = q * KQ; k < (q + 1) * KQ; ++k) r += A[i * K + k] * B[k * 64 + c];
    bad3 += D[i * 64 + c] != r;
  }
  printf("own A per block (CBSZ = 0): column quarter q sums k-quarter q   mismatches %d / 256\n", bad3);
  bad += bad2 + bad3;
  float t;
  hipLaunchKernelGGL(rate<1>, dim3(1), dim3(64), 0, 0, dD, 2000); hipMemcpy(&t, dD + 4096, 4, hipMemcpyDeviceToHost); printf("1 accumulator : %.1f clk / mfma (1 wave)\n", t);
  hipLaunchKernelGGL(rate<2>, dim3(1), dim3(64), 0, 0, dD, 2000); hipMemcpy(&t, dD + 4096, 4, hipMemcpyDeviceToHost); printf("2 accumulators: %.1f clk / mfma\n", t);
  hipLaunchKernelGGL(rate<4>, dim3(1), dim3(64), 0, 0, dD, 2000); hipMemcpy(&t, dD + 4096, 4, hipMemcpyDeviceToHost); printf("4 accumulators: %.1f clk / mfma\n", t);
  hipLaunchKernelGGL(rate<4>, dim3(1), dim3(512), 0, 0, dD, 2000); hipMemcpy(&t, dD + 4096, 4, hipMemcpyDeviceToHost); printf("4 accumulators, 8 waves (2 / SIMD): %.1f clk / mfma / wave\n", t);
  for (int nt : {64, 256, 512, 1024}) {
    hipLaunchKernelGGL(rate<4>, dim3(1), dim3(nt), 0, 0, dD, 4000); hipMemcpy(&t, dD + 4096, 4, hipMemcpyDeviceToHost);
    float t16;
    hipLaunchKernelGGL(rate16<4>, dim3(1), dim3(nt), 0, 0, dD, 4000); hipMemcpy(&t16, dD + 4096, 4, hipMemcpyDeviceToHost);
    printf("%4d threads: 4x4x1 %.1f clk/mfma/wave   16x16x4 %.1f clk/mfma/wave (wave 0's view)\n", nt, t, t16);
    hipLaunchKernelGGL(rate<4>, dim3(1), dim3(nt), 0, 0, dD, 4000); hipDeviceSynchronize(); float s4 = span(dD, nt / 64, 4000);
    hipLaunchKernelGGL(rate16<4>, dim3(1), dim3(nt), 0, 0, dD, 4000); hipDeviceSynchronize(); float s16 = span(dD, nt / 64, 4000);
    printf("             whole workgroup: 4x4x1 %.1f clk per mfma-of-each-wave   16x16x4 %.1f\n", s4, s16);
  }
  return bad != 0;
}
