// Shared helpers for the gfx950 kernels of libdgmr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dgmr_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void dgmr_set_error(const char* fmt, ...);
extern int g_deterministic;  // dgmr_set_deterministic (ops.hip): fixed-order cross-workgroup sums

#define DGMR_CHECK_ARG(cond, ...)        \
    do {                                 \
        if (!(cond)) {                   \
            dgmr_set_error(__VA_ARGS__); \
            return -1;                   \
        }                                \
    } while (0)

#define DGMR_CHECK_LAUNCH()                                                \
    do {                                                                   \
        hipError_t e__ = hipGetLastError();                                \
        if (e__ != hipSuccess) {                                           \
            dgmr_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return -2;                                                     \
        }                                                                  \
    } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blocks of up to 1024 threads; result valid in every thread.
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 17 floats */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += smem[i];
    return t;
}
