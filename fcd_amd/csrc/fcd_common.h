// Internal helpers shared by the gfx950 kernels of libfcd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/fcd_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

void fcd_set_error(const char* fmt, ...);

// norm.hip: second stage of the InstanceNorm statistics on partial[N][chunks][C][2] doubles
// (used by the convolution's fused-statistics entry point in conv_igemm.hip).
int fcd_finalize_stats(const double* partial, float* mean, float* rstd, float* var_unbiased, int N, long HW, int C,
                       int chunks, float eps, hipStream_t st);
// norm.hip: the same, followed by normalise + affine + activation (+ residual) of z into y; one launch for 32 x 32 images
int fcd_finalize_apply(const double* partial, const float* z, float* mean, float* rstd, float* var_unbiased, int N, long HW,
                       int C, int chunks, float eps, const float* gamma, const float* beta, const float* residual, float* y,
                       float act_slope, float* y_amax, hipStream_t st);

// norm.hip: resolves the current device's copy of the built-in scratch array of
// fcd_instnorm_stats_noalloc (hipGetSymbolAddress, once per device; `device` must be current).
int fcd_noalloc_scratch_init(int device);

// runtime.hip: the current device's overflow word of FCD_PREC_F16X2 (fcd_set_overflow_flag), or NULL
unsigned* fcd_overflow_flag_current(void);

#define FCD_REQUIRE(cond, ...)            \
  do {                                    \
    if (!(cond)) {                        \
      fcd_set_error(__VA_ARGS__);         \
      return FCD_EINVAL;                  \
    }                                     \
  } while (0)

// Launch errors surface as a positive hipError_t.
static inline int fcd_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fcd_set_error("%s: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return FCD_OK;
}

static inline hipStream_t fcd_stream(void* s) { return (hipStream_t)s; }

static inline int fcd_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// 64-lane wavefront sum (gfx950: wave = 64).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block sum for blockDim.x = 256 (4 waves); result valid in every thread.
__device__ __forceinline__ float block_sum_256(float v, float* smem4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem4[w] = v;
  __syncthreads();
  return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}
__device__ __forceinline__ double block_sum_256_d(double v, double* smem4) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem4[w] = v;
  __syncthreads();
  return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

// Activations.  A code in [0, 1] is the leaky family (0 = ReLU, 0.01 = LeakyReLU, 1 = none);
// FCD_ACT_SILU / FCD_ACT_MISH (fcd_hip.h) select nn.SiLU / nn.Mish (fixed_point_gan.py:13-16).
// Formulas as in torch: silu = u * sigmoid(u); mish = u * tanh(softplus(u)), softplus(u) = u for u > 20.
// act_slope_at / act_curv_at are the first / second derivative at the pre-activation value u.
__device__ __forceinline__ float act_value(float u, float code) {
  if (code <= 1.f) return u > 0.f ? u : u * code;
  if (code == FCD_ACT_SILU) return u / (1.f + expf(-u));
  const float sp = u > 20.f ? u : log1pf(expf(u));
  return u * tanhf(sp);
}

__device__ __forceinline__ float act_slope_at(float u, float code) {
  if (code <= 1.f) return u > 0.f ? 1.f : code;
  const float sg = 1.f / (1.f + expf(-u));
  if (code == FCD_ACT_SILU) return sg * (1.f + u * (1.f - sg));
  const float sp = u > 20.f ? u : log1pf(expf(u));
  const float t = tanhf(sp);
  return t + u * sg * (1.f - t * t);
}

__device__ __forceinline__ float act_curv_at(float u, float code) {
  if (code <= 1.f) return 0.f;
  const float sg = 1.f / (1.f + expf(-u));
  if (code == FCD_ACT_SILU) return sg * (1.f - sg) * (2.f + u * (1.f - 2.f * sg));
  const float sp = u > 20.f ? u : log1pf(expf(u));
  const float t = tanhf(sp);
  const float w = (1.f - t * t) * sg;          // d tanh(softplus(u)) / du
  return 2.f * w + u * w * ((1.f - sg) - 2.f * t * sg);
}
