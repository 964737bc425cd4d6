// Latency of a per-step exchange between two workgroups on different CUs (release/acquire flags at agent scope +
// a 6.4 KB payload), the building block of a 2-CU weights-stationary LSTM.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int PAY = 1600;          // floats per exchange (16 seqs x 100 units)

template <int MODE>
__global__ __launch_bounds__(512) void pp_kernel(float* buf, unsigned* flags, int rounds, int stride, unsigned* err) {
    // pair p = blocks (b, b + stride); me = 0/1
    int b = blockIdx.x;
    int pair, me;
    if (stride == 1) { pair = b >> 1; me = b & 1; }
    else { pair = (b / (2 * stride)) * stride + (b % stride); me = (b / stride) & 1; }
    float* mine = buf + (size_t)(pair * 2 + me) * 2 * PAY;       // double buffered
    float* theirs = buf + (size_t)(pair * 2 + (1 - me)) * 2 * PAY;
    unsigned* myflag = flags + (pair * 2 + me) * 32;
    unsigned* thflag = flags + (pair * 2 + (1 - me)) * 32;
    __shared__ float sh[PAY];
    float acc = 0.f;
    for (int t = 1; t <= rounds; ++t) {
        float* dst = mine + (t & 1) * PAY;
        if (MODE == 2) {
            for (int i = threadIdx.x; i < PAY; i += blockDim.x) __hip_atomic_store(dst + i, (float)t + acc * 1e-9f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(myflag, (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
        for (int i = threadIdx.x; i < PAY; i += blockDim.x) dst[i] = (float)t + acc * 1e-9f;
        }
        if (MODE == 2) {
        } else if (MODE == 0) {
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(myflag, (unsigned)t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();
            if (threadIdx.x == 0) { __threadfence(); __hip_atomic_store(myflag, (unsigned)t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
        }
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(thflag, MODE == 2 ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)t) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 2000000) { atomicAdd(err, 1u); break; }
            }
        }
        __syncthreads();
        const float* src = theirs + (t & 1) * PAY;
        if (MODE == 2) { for (int i = threadIdx.x; i < PAY; i += blockDim.x) sh[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else for (int i = threadIdx.x; i < PAY; i += blockDim.x) sh[i] = __builtin_nontemporal_load(src + i);
        __syncthreads();
        acc += sh[(threadIdx.x * 7) % PAY];
        if (sh[threadIdx.x % PAY] != (float)t + 0.f && fabsf(sh[threadIdx.x % PAY] - (float)t) > 0.5f) atomicAdd(err + 1, 1u);
    }
    if (acc == 12345.f) buf[0] = acc;
}


template <int MODE>
__global__ __launch_bounds__(512) void tag_kernel(unsigned long long* buf, int rounds, int stride, unsigned* err) {
    int b = blockIdx.x;
    int pair, me;
    if (stride == 1) { pair = b >> 1; me = b & 1; }
    else { pair = (b / (2 * stride)) * stride + (b % stride); me = (b / stride) & 1; }
    constexpr int W = 1792;                                       // words per exchange (16 rows x 112 units)
    unsigned long long* mine = buf + (size_t)(pair * 2 + me) * 2 * W;
    const unsigned long long* theirs = buf + (size_t)(pair * 2 + (1 - me)) * 2 * W;
    __shared__ float sh[W];
    float acc = 0.f;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = 1; t <= rounds; ++t) {
        unsigned long long* dst = mine + (t & 1) * W;
        if (w < 7) {                                              // 7 "compute" waves write 4 words per lane
            for (int e = 0; e < 4; ++e) {
                int i = (w * 4 + e) * 64 + lane;
                __hip_atomic_store(dst + i, ((unsigned long long)t << 32) | __float_as_uint((float)t + acc * 1e-9f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const unsigned long long* src = theirs + (t & 1) * W;
        if (MODE == 3) {
            for (int i = threadIdx.x; i < W; i += 512) {
                unsigned long long v = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while ((unsigned)(v >> 32) != (unsigned)t) { __builtin_amdgcn_s_sleep(1); v = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (++spins > 2000000) { atomicAdd(err, 1u); break; } }
                sh[i] = __uint_as_float((unsigned)v);
            }
        } else if (w == 7) {
            unsigned long long v[28];
#pragma unroll
            for (int j = 0; j < 28; ++j) v[j] = __hip_atomic_load(src + lane + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int j = 0; j < 28; ++j) {
                int spins = 0;
                while ((unsigned)(v[j] >> 32) != (unsigned)t) { __builtin_amdgcn_s_sleep(1); v[j] = __hip_atomic_load(src + lane + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (++spins > 2000000) { atomicAdd(err, 1u); break; } }
                sh[lane + 64 * j] = __uint_as_float((unsigned)v[j]);
            }
        }
        __syncthreads();
        acc += sh[(threadIdx.x * 7) % W];
        if (fabsf(sh[threadIdx.x] - (float)t) > 0.5f) atomicAdd(err + 1, 1u);
        __syncthreads();
    }
    if (acc == 12345.f) buf[0] = (unsigned long long)acc;
}

int main() {
    int rounds = 2000;
    for (int mode = 2; mode < 3; ++mode)
    for (int stride : {1, 8}) {
        for (int pairs : {1, 8, 64, 128}) {
            int blocks = pairs * 2;
            if (stride == 8 && blocks % 16) continue;
            float* buf; unsigned* flags; unsigned* err;
            CK(hipMalloc(&buf, (size_t)blocks * 2 * PAY * 4));
            CK(hipMalloc(&flags, blocks * 32 * 4));
            CK(hipMalloc(&err, 8));
            CK(hipMemset(flags, 0, blocks * 32 * 4));
            CK(hipMemset(err, 0, 8));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            if (mode == 0) pp_kernel<0><<<blocks, 512>>>(buf, flags, rounds, stride, err);
            else if (mode == 1) pp_kernel<1><<<blocks, 512>>>(buf, flags, rounds, stride, err);
            else pp_kernel<2><<<blocks, 512>>>(buf, flags, rounds, stride, err);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h[2]; CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
            printf("mode %d stride %d pairs %3d: %.2f us per exchange round (timeouts %u, bad payload %u)\n", mode, stride, pairs, ms * 1e3 / rounds, h[0], h[1]);
            CK(hipFree(buf)); CK(hipFree(flags)); CK(hipFree(err));
        }
    }
    for (int mode = 3; mode < 5; ++mode)
    for (int stride : {1, 8}) {
        for (int pairs : {1, 8, 128}) {
            int blocks = pairs * 2;
            if (stride == 8 && blocks % 16) continue;
            unsigned long long* buf; unsigned* err;
            CK(hipMalloc(&buf, (size_t)blocks * 2 * 1792 * 8));
            CK(hipMalloc(&err, 8));
            CK(hipMemset(buf, 0, (size_t)blocks * 2 * 1792 * 8));
            CK(hipMemset(err, 0, 8));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            if (mode == 3) tag_kernel<3><<<blocks, 512>>>(buf, rounds, stride, err);
            else tag_kernel<4><<<blocks, 512>>>(buf, rounds, stride, err);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned h[2]; CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
            printf("tagged mode %d (%s) stride %d pairs %3d: %.2f us per round (timeouts %u, bad payload %u)\n", mode, mode == 3 ? "all waves load" : "one wave loads", stride, pairs, ms * 1e3 / rounds, h[0], h[1]);
            CK(hipFree(buf)); CK(hipFree(err));
        }
    }
    return 0;
}
