// HBM-bound pieces of the FCD loss graph: module-boundary layout changes, the Generator
// prologue/epilogue, LeakyReLU backward, the loss reductions, Adam, and the fused
// difference-map + threshold of mask generation.  Scalars stay on the device.
#include "fcd_common.h"

namespace {

constexpr int RED_BLOCKS = 256;   // partial sums per two-stage reduction

// ---------------------------------------------------------------- layout ---------------
// NCHW [N][C][H][W] (+ tiled labels) -> NHWC [N][H][W][CP]
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ c,
                                                           float* __restrict__ out, int N, int C, int CD, int CP, long HW) {
  const long total = (long)N * HW;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long n = idx / HW, px = idx - n * HW;
    float* dst = out + idx * CP;
    for (int ch = 0; ch < CP; ch += 4) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = ch + e;
        v[e] = k < C ? x[(n * C + k) * HW + px] : (k < C + CD ? c[n * CD + (k - C)] : 0.f);
      }
      *reinterpret_cast<float4*>(dst + ch) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           int accumulate, int N, int C, int CP, long HW) {
  const long total = (long)N * HW;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long n = idx / HW, px = idx - n * HW;
    const float* src = x + idx * CP;
    for (int k = 0; k < C; ++k) {
      const long o = (n * C + k) * HW + px;
      out[o] = accumulate ? out[o] + src[k] : src[k];
    }
  }
}

// many channels, few pixels (the discriminator's feature map): one thread per output element
__global__ __launch_bounds__(256) void nhwc_to_nchw_wide_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                                int accumulate, int N, int C, int CP, long HW) {
  const long total = (long)N * C * HW;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long px = idx % HW, nk = idx / HW;
    const long k = nk % C, n = nk / C;
    const float v = x[(n * HW + px) * CP + k];
    out[idx] = accumulate ? out[idx] + v : v;
  }
}

// ---------------------------------------------------------------- G epilogue -----------
__global__ __launch_bounds__(256) void g_epilogue_fwd_kernel(const float* __restrict__ delta, const float* __restrict__ x,
                                                             float* __restrict__ xf, int N, int C, int CP, long HW) {
  const long total = (long)N * HW;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long n = idx / HW, px = idx - n * HW;
    const float* d = delta + idx * CP;
    if ((CP & 3) == 0) {
      // a pixel's channels as 16-byte loads (a lane's 4-byte reads of its own 64-byte run do not coalesce)
      for (int k4 = 0; k4 < C; k4 += 4) {
        const float4 dv = *reinterpret_cast<const float4*>(d + k4);
        const float dk[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (k4 + e < C) {
            const long o = (n * C + k4 + e) * HW + px;
            const float xv = x[o];
            const float t = tanhf(dk[e] + xv);
            xf[o] = (xv == -1.f) ? -1.f : t;
          }
        }
      }
      continue;
    }
    for (int k = 0; k < C; ++k) {
      const long o = (n * C + k) * HW + px;
      const float xv = x[o];
      const float t = tanhf(d[k] + xv);
      xf[o] = (xv == -1.f) ? -1.f : t;
    }
  }
}

__global__ __launch_bounds__(256) void g_epilogue_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                             const float* __restrict__ xf, float* __restrict__ gdelta,
                                                             float* __restrict__ gx, int accumulate_gx, int N, int C,
                                                             int CP, long HW) {
  const long total = (long)N * HW;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long n = idx / HW, px = idx - n * HW;
    float* d = gdelta + idx * CP;
    if ((CP & 3) == 0) {
      // the pixel's 4 channels' plane reads in flight together, one 16-byte store (same values as the loop below)
      for (int k4 = 0; k4 < CP; k4 += 4) {
        float gv[4] = {0.f, 0.f, 0.f, 0.f}, tv[4], xv[4], ov[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (k4 + e < C) {
            const long o = (n * C + k4 + e) * HW + px;
            tv[e] = xf[o];
            xv[e] = x[o];
            gv[e] = gout[o];
            if (gx && accumulate_gx) ov[e] = gx[o];
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (k4 + e < C) {
            const long o = (n * C + k4 + e) * HW + px;
            gv[e] = (xv[e] == -1.f) ? 0.f : gv[e] * (1.f - tv[e] * tv[e]);
            if (gx) gx[o] = accumulate_gx ? ov[e] + gv[e] : gv[e];
          }
        }
        *reinterpret_cast<float4*>(d + k4) = make_float4(gv[0], gv[1], gv[2], gv[3]);
      }
      continue;
    }
    for (int k = 0; k < CP; ++k) {
      float g = 0.f;
      if (k < C) {
        const long o = (n * C + k) * HW + px;
        const float t = xf[o];
        g = (x[o] == -1.f) ? 0.f : gout[o] * (1.f - t * t);
        if (gx) gx[o] = accumulate_gx ? gx[o] + g : g;
      }
      d[k] = g;
    }
  }
}

// ---------------------------------------------------------------- elementwise ----------
__global__ __launch_bounds__(256) void lrelu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                                        float* __restrict__ out, size_t n, float slope) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    out[i] = g[i] * (y[i] > 0.f ? 1.f : slope);
}

// Smooth activations of the discriminator (SiLU / Mish), kept un-fused: a = act(z); first backward
// g * act'(z); backward of that backward: u_out = ubar * act'(z), q = ubar * ga * act''(z).
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ z, float* __restrict__ a, size_t n,
                                                      float code) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = act_value(z[i], code);
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ z,
                                                      float* __restrict__ out, size_t n, float code, int accumulate) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = g[i] * act_slope_at(z[i], code);
    out[i] = accumulate ? out[i] + v : v;
  }
}

__global__ __launch_bounds__(256) void act_bwd2_kernel(const float* __restrict__ ubar, const float* __restrict__ ga,
                                                       const float* __restrict__ z, float* __restrict__ u_out,
                                                       float* __restrict__ q_out, size_t n, float code) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float ub = ubar[i], zz = z[i];
    u_out[i] = ub * act_slope_at(zz, code);
    q_out[i] = ub * ga[i] * act_curv_at(zz, code);
  }
}

__global__ __launch_bounds__(256) void fill_scaled_kernel(float* __restrict__ out, size_t n, float scale,
                                                          const float* __restrict__ gscale) {
  const float v = scale * gscale[0];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = v;
}

__global__ __launch_bounds__(256) void lerp_rows_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ alpha, float* __restrict__ out,
                                                        size_t per_row, size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const float al = alpha[i / per_row];
    out[i] = al * a[i] + (1.f - al) * b[i];
  }
}

// ---------------------------------------------------------------- reductions -----------
// stage 1: RED_BLOCKS double partials; stage 2: one block.  mode 0: sum x; 1: sum |a-b|; 2: BCE
__global__ __launch_bounds__(256) void reduce_stage1_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            double* __restrict__ partial, size_t n, int mode) {
  __shared__ double sm[4];
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float v;
    if (mode == 0) {
      v = a[i];
    } else if (mode == 1) {
      v = fabsf(a[i] - b[i]);
    } else {
      const float x = a[i], t = b[i];
      v = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    }
    s += (double)v;
  }
  s = block_sum_256_d(s, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void reduce_stage2_kernel(const double* __restrict__ partial, float* __restrict__ out,
                                                            int nb, double scale) {
  __shared__ double sm[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  s = block_sum_256_d(s, sm);
  if (threadIdx.x == 0) out[0] = (float)(s * scale);
}

__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ gscale, float* __restrict__ ga,
                                                     float* __restrict__ gb, int accumulate, size_t n) {
  const float gs = gscale[0] / (float)n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float d = a[i] - b[i];
    const float s = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    if (ga) ga[i] = accumulate ? ga[i] + s : s;
    if (gb) gb[i] = accumulate ? gb[i] - s : -s;
  }
}

__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                      const float* __restrict__ gscale, float* __restrict__ gx, size_t n) {
  const float gs = gscale[0] / (float)n;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float sig = 1.f / (1.f + expf(-x[i]));
    gx[i] = (sig - t[i]) * gs;
  }
}

// one block per row: norm[n] = sqrt(sum g^2).  Eight 16-byte loads in flight per thread and four independent double
// sums (one dependent load-add per iteration walked 160 steps at the baseline's size: 47 us on 16 CUs)
__global__ __launch_bounds__(256) void gp_norm_kernel(const float* __restrict__ g, float* __restrict__ norm, size_t per_row) {
  __shared__ double sm[4];
  const float* row = g + (size_t)blockIdx.x * per_row;
  double s = 0.0;
  if ((per_row & 3) == 0 && ((uintptr_t)row & 15) == 0) {
    const size_t n4 = per_row / 4;
    double q[4] = {0.0, 0.0, 0.0, 0.0};
    size_t i = threadIdx.x;
    for (; i + 7 * 256 < n4; i += 8 * 256) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(row + (i + u * 256) * 4);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        q[u & 3] += (double)(v[u].x * v[u].x) + (double)(v[u].y * v[u].y) + (double)(v[u].z * v[u].z) + (double)(v[u].w * v[u].w);
    }
    for (; i < n4; i += 256) {
      const float4 v = *reinterpret_cast<const float4*>(row + i * 4);
      q[0] += (double)(v.x * v.x) + (double)(v.y * v.y) + (double)(v.z * v.z) + (double)(v.w * v.w);
    }
    s = (q[0] + q[1]) + (q[2] + q[3]);
  } else {
    for (size_t i = threadIdx.x; i < per_row; i += 256) s += (double)(row[i] * row[i]);
  }
  s = block_sum_256_d(s, sm);
  if (threadIdx.x == 0) norm[blockIdx.x] = (float)sqrt(s);
}

__global__ __launch_bounds__(64) void gp_loss_kernel(const float* __restrict__ norm, float* __restrict__ out, int N) {
  double s = 0.0;
  for (int i = threadIdx.x; i < N; i += 64) {
    const float d = norm[i] - 1.f;
    s += (double)(d * d);
  }
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[0] = (float)(s / (double)N);
}

__global__ __launch_bounds__(256) void gp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ norm,
                                                     const float* __restrict__ gscale, float* __restrict__ gg, int N,
                                                     size_t per_row, size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const float nr = norm[i / per_row];
    const float coef = gscale[0] * 2.f * (nr - 1.f) / ((float)N * nr);
    gg[i] = coef * g[i];
  }
}

// ---------------------------------------------------------------- Adam -----------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float step_size, float beta1,
                                                   float beta2, float eps, float bc2_sqrt) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

// ---------------------------------------------------------------- masks ----------------
__global__ __launch_bounds__(256) void diffmap_kernel(const float* __restrict__ xf, const float* __restrict__ x,
                                                      float* __restrict__ diff, uint8_t* __restrict__ mask, float thr, int N,
                                                      int C, long HW) {
  const long total = (long)N * HW;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long n = idx / HW, px = idx - n * HW;
    float s = 0.f;
    for (int k = 0; k < C; ++k) {
      const long o = (n * C + k) * HW + px;
      s += fabsf(xf[o] - x[o]) / 2.f;
    }
    const float d = s / (float)C;
    if (diff) diff[idx] = d;
    if (mask) mask[idx] = d > thr ? 1 : 0;
  }
}

// ---------------------------------------------------------------- D heads --------------
// one block per output pixel; wp packed [O][KH][KW][IP]
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ in, const float* __restrict__ wp,
                                                       float* __restrict__ out, int IH, int IW, int IP, int O, int OH,
                                                       int OW, int KH, int KW, int pad) {
  __shared__ float sm[4];
  const int ow = blockIdx.x % OW, oh = (blockIdx.x / OW) % OH, n = blockIdx.x / (OW * OH);
  const int ip4 = IP / 4;
  for (int o = 0; o < O; ++o) {
    float s = 0.f;
    for (int t = threadIdx.x; t < KH * KW * ip4; t += 256) {
      const int tap = t / ip4, i4 = t - tap * ip4;
      const int kh = tap / KW, kw = tap - kh * KW;
      const int ih = oh - pad + kh, iw = ow - pad + kw;
      if (ih < 0 || ih >= IH || iw < 0 || iw >= IW) continue;
      const float4 a = *reinterpret_cast<const float4*>(in + (((long)n * IH + ih) * IW + iw) * IP + i4 * 4);
      const float4 w = *reinterpret_cast<const float4*>(wp + ((long)o * KH * KW + tap) * IP + i4 * 4);
      s += a.x * w.x + a.y * w.y + a.z * w.z + a.w * w.w;
    }
    s = block_sum_256(s, sm);
    if (threadIdx.x == 0) out[(((long)n * O + o) * OH + oh) * OW + ow] = s;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void head_dgrad_kernel(const float* __restrict__ gout, const float* __restrict__ wp,
                                                         float* __restrict__ din, const float* __restrict__ mask,
                                                         float mask_slope, int accumulate, int N, int IH, int IW, int IP,
                                                         int O, int OH, int OW, int KH, int KW, int pad) {
  const int ip4 = IP / 4;
  const long total = (long)N * IH * IW * ip4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int i4 = (int)(idx % ip4);
    long t = idx / ip4;
    const int iw = (int)(t % IW); t /= IW;
    const int ih = (int)(t % IH);
    const int n = (int)(t / IH);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < KH; ++kh) {
      const int oh = ih + pad - kh;
      if (oh < 0 || oh >= OH) continue;
      for (int kw = 0; kw < KW; ++kw) {
        const int ow = iw + pad - kw;
        if (ow < 0 || ow >= OW) continue;
        for (int o = 0; o < O; ++o) {
          const float g = gout[(((long)n * O + o) * OH + oh) * OW + ow];
          const float4 w = *reinterpret_cast<const float4*>(wp + ((long)o * KH * KW + kh * KW + kw) * IP + i4 * 4);
          s.x += g * w.x; s.y += g * w.y; s.z += g * w.z; s.w += g * w.w;
        }
      }
    }
    float4* dst = reinterpret_cast<float4*>(din + idx * 4);
    if (mask) {
      const float4 mk = *reinterpret_cast<const float4*>(mask + idx * 4);
      s.x *= mk.x > 0.f ? 1.f : mask_slope;
      s.y *= mk.y > 0.f ? 1.f : mask_slope;
      s.z *= mk.z > 0.f ? 1.f : mask_slope;
      s.w *= mk.w > 0.f ? 1.f : mask_slope;
    }
    if (accumulate) {
      const float4 o = *dst;
      s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *dst = s;
  }
}

// dw in parameter layout [O][I][KH][KW]
__global__ __launch_bounds__(256) void head_wgrad_kernel(const float* __restrict__ gout, const float* __restrict__ in,
                                                         float* __restrict__ dw, int accumulate, int N, int IH, int IW,
                                                         int I, int IP, int O, int OH, int OW, int KH, int KW, int pad) {
  const long total = (long)O * KH * KW * I;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int i = (int)(idx % I);
    long t = idx / I;
    const int kw = (int)(t % KW); t /= KW;
    const int kh = (int)(t % KH);
    const int o = (int)(t / KH);
    float s = 0.f;
    for (int n = 0; n < N; ++n)
      for (int oh = 0; oh < OH; ++oh) {
        const int ih = oh - pad + kh;
        if (ih < 0 || ih >= IH) continue;
        for (int ow = 0; ow < OW; ++ow) {
          const int iw = ow - pad + kw;
          if (iw < 0 || iw >= IW) continue;
          s += gout[(((long)n * O + o) * OH + oh) * OW + ow] * in[(((long)n * IH + ih) * IW + iw) * IP + i];
        }
      }
    const long d = (((long)o * I + i) * KH + kh) * KW + kw;
    dw[d] = accumulate ? dw[d] + s : s;
  }
}

inline int grid_for(size_t n, int cap = 8192) {
  size_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b < (size_t)cap ? b : (size_t)cap);
}

int two_stage(const float* a, const float* b, float* out, size_t n, int mode, double scale, void* ws, size_t ws_bytes,
              hipStream_t st, const char* what) {
  if (!ws || ws_bytes < RED_BLOCKS * sizeof(double)) {
    fcd_set_error("%s: workspace must hold %zu bytes", what, (size_t)RED_BLOCKS * sizeof(double));
    return FCD_EWORKSPACE;
  }
  const int nb = grid_for(n, RED_BLOCKS);
  hipLaunchKernelGGL(reduce_stage1_kernel, dim3(nb), dim3(256), 0, st, a, b, (double*)ws, n, mode);
  int rc = fcd_check_launch(what);
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_stage2_kernel, dim3(1), dim3(256), 0, st, (const double*)ws, out, nb, scale);
  return fcd_check_launch(what);
}

}  // namespace

extern "C" size_t fcd_reduce_workspace_bytes(void) { return RED_BLOCKS * sizeof(double); }

extern "C" int fcd_g_prologue(const float* x, const float* c, float* out, int N, int C, int CD, int CP, int H, int W,
                              void* stream) {
  FCD_REQUIRE(x && c && out && N > 0 && C > 0 && CD >= 0 && CP >= C + CD && CP % 4 == 0 && H > 0 && W > 0,
              "g_prologue: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, fcd_stream(stream), x, c, out,
                     N, C, CD, CP, (long)H * W);
  return fcd_check_launch("g_prologue");
}

extern "C" int fcd_nchw_to_nhwc(const float* x, float* out, int N, int C, int CP, int H, int W, void* stream) {
  FCD_REQUIRE(x && out && N > 0 && C > 0 && CP >= C && CP % 4 == 0 && H > 0 && W > 0, "nchw_to_nhwc: bad arguments");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, fcd_stream(stream), x,
                     (const float*)nullptr, out, N, C, 0, CP, (long)H * W);
  return fcd_check_launch("nchw_to_nhwc");
}

extern "C" int fcd_nhwc_to_nchw(const float* x, float* out, int accumulate, int N, int C, int CP, int H, int W,
                                void* stream) {
  FCD_REQUIRE(x && out && N > 0 && C > 0 && CP >= C && H > 0 && W > 0, "nhwc_to_nchw: bad arguments");
  if (C > 16)
    hipLaunchKernelGGL(nhwc_to_nchw_wide_kernel, dim3(grid_for((size_t)N * C * H * W)), dim3(256), 0, fcd_stream(stream), x,
                       out, accumulate, N, C, CP, (long)H * W);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, fcd_stream(stream), x, out,
                       accumulate, N, C, CP, (long)H * W);
  return fcd_check_launch("nhwc_to_nchw");
}

extern "C" int fcd_g_epilogue_fwd(const float* delta, const float* x, float* x_fake, int N, int C, int CP, int H, int W,
                                  void* stream) {
  FCD_REQUIRE(delta && x && x_fake && N > 0 && C > 0 && CP >= C && H > 0 && W > 0, "g_epilogue_fwd: bad arguments");
  hipLaunchKernelGGL(g_epilogue_fwd_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, fcd_stream(stream), delta, x,
                     x_fake, N, C, CP, (long)H * W);
  return fcd_check_launch("g_epilogue_fwd");
}

extern "C" int fcd_g_epilogue_bwd(const float* gout, const float* x, const float* x_fake, float* gdelta, float* gx,
                                  int accumulate_gx, int N, int C, int CP, int H, int W, void* stream) {
  FCD_REQUIRE(gout && x && x_fake && gdelta && N > 0 && C > 0 && CP >= C && H > 0 && W > 0, "g_epilogue_bwd: bad arguments");
  hipLaunchKernelGGL(g_epilogue_bwd_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, fcd_stream(stream), gout, x,
                     x_fake, gdelta, gx, accumulate_gx, N, C, CP, (long)H * W);
  return fcd_check_launch("g_epilogue_bwd");
}

extern "C" int fcd_lrelu_bwd(const float* g, const float* y, float* out, size_t n, float slope, void* stream) {
  FCD_REQUIRE(g && y && out && n > 0, "lrelu_bwd: bad arguments");
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, fcd_stream(stream), g, y, out, n, slope);
  return fcd_check_launch("lrelu_bwd");
}

static int act_code_ok(float code) {
  return (code >= 0.f && code <= 1.f) || code == FCD_ACT_SILU || code == FCD_ACT_MISH;
}

extern "C" int fcd_act_fwd(const float* z, float* a, size_t n, float act, void* stream) {
  FCD_REQUIRE(z && a && n > 0 && act_code_ok(act), "act_fwd: bad arguments");
  hipLaunchKernelGGL(act_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, fcd_stream(stream), z, a, n, act);
  return fcd_check_launch("act_fwd");
}

extern "C" int fcd_act_bwd(const float* g, const float* z, float* out, size_t n, float act, int accumulate, void* stream) {
  FCD_REQUIRE(g && z && out && n > 0 && act_code_ok(act), "act_bwd: bad arguments");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, fcd_stream(stream), g, z, out, n, act, accumulate);
  return fcd_check_launch("act_bwd");
}

extern "C" int fcd_act_bwd2(const float* ubar, const float* ga, const float* z, float* u_out, float* q_out, size_t n,
                            float act, void* stream) {
  FCD_REQUIRE(ubar && ga && z && u_out && q_out && n > 0 && act_code_ok(act), "act_bwd2: bad arguments");
  hipLaunchKernelGGL(act_bwd2_kernel, dim3(grid_for(n)), dim3(256), 0, fcd_stream(stream), ubar, ga, z, u_out, q_out, n, act);
  return fcd_check_launch("act_bwd2");
}

extern "C" int fcd_fill_scaled(float* out, size_t n, float scale, const float* gscale, void* stream) {
  FCD_REQUIRE(out && gscale && n > 0, "fill_scaled: bad arguments");
  hipLaunchKernelGGL(fill_scaled_kernel, dim3(grid_for(n)), dim3(256), 0, fcd_stream(stream), out, n, scale, gscale);
  return fcd_check_launch("fill_scaled");
}

extern "C" int fcd_lerp_rows(const float* a, const float* b, const float* alpha, float* out, int N, size_t per_row,
                             void* stream) {
  FCD_REQUIRE(a && b && alpha && out && N > 0 && per_row > 0, "lerp_rows: bad arguments");
  const size_t total = (size_t)N * per_row;
  hipLaunchKernelGGL(lerp_rows_kernel, dim3(grid_for(total)), dim3(256), 0, fcd_stream(stream), a, b, alpha, out, per_row, total);
  return fcd_check_launch("lerp_rows");
}

extern "C" int fcd_sum_scaled(const float* x, float* out, size_t n, float scale, void* workspace, size_t workspace_bytes,
                              void* stream) {
  FCD_REQUIRE(x && out && n > 0, "sum_scaled: bad arguments");
  return two_stage(x, nullptr, out, n, 0, (double)scale, workspace, workspace_bytes, fcd_stream(stream), "sum_scaled");
}

extern "C" int fcd_l1_mean(const float* a, const float* b, float* out, size_t n, void* workspace, size_t workspace_bytes,
                           void* stream) {
  FCD_REQUIRE(a && b && out && n > 0, "l1_mean: bad arguments");
  return two_stage(a, b, out, n, 1, 1.0 / (double)n, workspace, workspace_bytes, fcd_stream(stream), "l1_mean");
}

extern "C" int fcd_l1_mean_bwd(const float* a, const float* b, const float* gscale, float* ga, float* gb, int accumulate,
                               size_t n, void* stream) {
  FCD_REQUIRE(a && b && gscale && (ga || gb) && n > 0, "l1_mean_bwd: bad arguments");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, fcd_stream(stream), a, b, gscale, ga, gb, accumulate, n);
  return fcd_check_launch("l1_mean_bwd");
}

extern "C" int fcd_bce_logits_mean(const float* logit, const float* target, float* out, size_t n, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  FCD_REQUIRE(logit && target && out && n > 0, "bce_logits_mean: bad arguments");
  return two_stage(logit, target, out, n, 2, 1.0 / (double)n, workspace, workspace_bytes, fcd_stream(stream), "bce_logits_mean");
}

extern "C" int fcd_bce_logits_mean_bwd(const float* logit, const float* target, const float* gscale, float* glogit,
                                       size_t n, void* stream) {
  FCD_REQUIRE(logit && target && gscale && glogit && n > 0, "bce_logits_mean_bwd: bad arguments");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, fcd_stream(stream), logit, target, gscale, glogit, n);
  return fcd_check_launch("bce_logits_mean_bwd");
}

extern "C" int fcd_gp_fwd(const float* g, float* norm, float* out, int N, size_t per_row, void* stream) {
  FCD_REQUIRE(g && norm && out && N > 0 && per_row > 0, "gp_fwd: bad arguments");
  hipStream_t st = fcd_stream(stream);
  hipLaunchKernelGGL(gp_norm_kernel, dim3(N), dim3(256), 0, st, g, norm, per_row);
  int rc = fcd_check_launch("gp_fwd/norm");
  if (rc) return rc;
  hipLaunchKernelGGL(gp_loss_kernel, dim3(1), dim3(64), 0, st, norm, out, N);
  return fcd_check_launch("gp_fwd/loss");
}

extern "C" int fcd_gp_bwd(const float* g, const float* norm, const float* gscale, float* gg, int N, size_t per_row,
                          void* stream) {
  FCD_REQUIRE(g && norm && gscale && gg && N > 0 && per_row > 0, "gp_bwd: bad arguments");
  const size_t total = (size_t)N * per_row;
  hipLaunchKernelGGL(gp_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, fcd_stream(stream), g, norm, gscale, gg, N, per_row, total);
  return fcd_check_launch("gp_bwd");
}

extern "C" int fcd_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                             float eps, int step, void* stream) {
  FCD_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, fcd_stream(stream), p, g, m, v, n, step_size, beta1, beta2,
                     eps, bc2_sqrt);
  return fcd_check_launch("adam_step");
}

extern "C" int fcd_diffmap_threshold(const float* x_fake, const float* x, float* diff, uint8_t* mask, float thr, int N,
                                     int C, int H, int W, void* stream) {
  FCD_REQUIRE(x_fake && x && (diff || mask) && N > 0 && C > 0 && H > 0 && W > 0, "diffmap_threshold: bad arguments");
  hipLaunchKernelGGL(diffmap_kernel, dim3(grid_for((size_t)N * H * W)), dim3(256), 0, fcd_stream(stream), x_fake, x, diff,
                     mask, thr, N, C, (long)H * W);
  return fcd_check_launch("diffmap_threshold");
}

extern "C" int fcd_head_fwd(const float* in, const float* w_packed, float* out, int N, int IH, int IW, int IP, int O,
                            int KH, int KW, int pad, void* stream) {
  FCD_REQUIRE(in && w_packed && out && N > 0 && IP > 0 && IP % 4 == 0 && O > 0, "head_fwd: bad arguments");
  const int OH = IH + 2 * pad - KH + 1, OW = IW + 2 * pad - KW + 1;
  FCD_REQUIRE(OH > 0 && OW > 0, "head_fwd: empty output");
  hipLaunchKernelGGL(head_fwd_kernel, dim3(N * OH * OW), dim3(256), 0, fcd_stream(stream), in, w_packed, out, IH, IW, IP, O,
                     OH, OW, KH, KW, pad);
  return fcd_check_launch("head_fwd");
}

extern "C" int fcd_head_dgrad(const float* gout, const float* w_packed, float* din, const float* mask_src,
                              float mask_slope, int accumulate, int N, int IH, int IW, int IP, int O, int KH, int KW,
                              int pad, void* stream) {
  FCD_REQUIRE(gout && w_packed && din && N > 0 && IP > 0 && IP % 4 == 0 && O > 0, "head_dgrad: bad arguments");
  const int OH = IH + 2 * pad - KH + 1, OW = IW + 2 * pad - KW + 1;
  FCD_REQUIRE(OH > 0 && OW > 0, "head_dgrad: empty output");
  hipLaunchKernelGGL(head_dgrad_kernel, dim3(grid_for((size_t)N * IH * IW * (IP / 4))), dim3(256), 0, fcd_stream(stream),
                     gout, w_packed, din, mask_src, mask_slope, accumulate, N, IH, IW, IP, O, OH, OW, KH, KW, pad);
  return fcd_check_launch("head_dgrad");
}

extern "C" int fcd_head_wgrad(const float* gout, const float* in, float* dw, int accumulate, int N, int IH, int IW, int I,
                              int IP, int O, int KH, int KW, int pad, void* stream) {
  FCD_REQUIRE(gout && in && dw && N > 0 && I > 0 && IP >= I && O > 0, "head_wgrad: bad arguments");
  const int OH = IH + 2 * pad - KH + 1, OW = IW + 2 * pad - KW + 1;
  FCD_REQUIRE(OH > 0 && OW > 0, "head_wgrad: empty output");
  hipLaunchKernelGGL(head_wgrad_kernel, dim3(grid_for((size_t)O * KH * KW * I)), dim3(256), 0, fcd_stream(stream), gout, in,
                     dw, accumulate, N, IH, IW, I, IP, O, OH, OW, KH, KW, pad);
  return fcd_check_launch("head_wgrad");
}
