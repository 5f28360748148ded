// Saliency losses (kldiv / cc / similarity, loss.py:13-99), fused Adam and the
// AViNet bilinear fusion.  Loss reductions run one workgroup (1024 lanes) per
// sample with wave-shuffle + LDS tree reductions and fp64 accumulators, which
// covers both the fp32 DHF1K path and the silently-fp64 DIEM path.
#include "common.h"

#define LOSS_EPS 2.2204e-16

template <bool G64> VN_DEV double ldg(const void* gt, long i) {
  if (G64) return ((const double*)gt)[i];
  return (double)((const float*)gt)[i];
}

struct MinIdx { double v; int i; };

VN_DEV double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  double r = 0.0;
  for (int k = 0; k < nw; ++k) r += sh[k];
  return r;
}
VN_DEV MinIdx block_min_d(double v, int idx, double* sh, int* shi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (ov < v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) { sh[wv] = v; shi[wv] = idx; }
  __syncthreads();
  MinIdx r = {sh[0], shi[0]};
  for (int k = 1; k < nw; ++k)
    if (sh[k] < r.v || (sh[k] == r.v && shi[k] < r.i)) { r.v = sh[k]; r.i = shi[k]; }
  return r;
}
VN_DEV double block_max_d(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  double r = sh[0];
  for (int k = 1; k < nw; ++k) r = fmax(r, sh[k]);
  return r;
}

// saved[b][0..7]:
//  kldiv: S_s, S_g, loss_b
//  cc:    mu_s, mu_g, Sxx, Syy, Sxy, r
//  sim:   lo_s, D_s, lo_g, D_g, argmin_s, loss_b
//  nss:   mu_s, std_s (unbiased), sum((s-mu)/(std+eps)*g), sum(g), -, nss_b      (loss.py:101-120, forward only)
template <bool G64>
__global__ __launch_bounds__(1024) void loss_fwd_kernel(int which, const float* __restrict__ s, const void* gt, int n,
                                                        double* __restrict__ saved) {
  __shared__ double sh[16];
  __shared__ int shi[16];
  const int b = blockIdx.x;
  const float* sp = s + (long)b * n;
  const long gb = (long)b * n;
  double* sv = saved + b * 8;
  if (which == 0) {
    double a = 0, c = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { a += (double)sp[i]; c += ldg<G64>(gt, gb + i); }
    const double Ss = block_sum_d(a, sh), Sg = block_sum_d(c, sh);
    double l = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double p = (double)sp[i] / Ss, q = ldg<G64>(gt, gb + i) / Sg;
      l += q * log(LOSS_EPS + q / (p + LOSS_EPS));
    }
    l = block_sum_d(l, sh);
    if (threadIdx.x == 0) { sv[0] = Ss; sv[1] = Sg; sv[2] = l; }
  } else if (which == 1) {
    double a = 0, c = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { a += (double)sp[i]; c += ldg<G64>(gt, gb + i); }
    const double ms = block_sum_d(a, sh) / n, mg = block_sum_d(c, sh) / n;
    double xx = 0, yy = 0, xy = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double x = (double)sp[i] - ms, y = ldg<G64>(gt, gb + i) - mg;
      xx += x * x; yy += y * y; xy += x * y;
    }
    xx = block_sum_d(xx, sh); yy = block_sum_d(yy, sh); xy = block_sum_d(xy, sh);
    if (threadIdx.x == 0) { sv[0] = ms; sv[1] = mg; sv[2] = xx; sv[3] = yy; sv[4] = xy; sv[5] = xy / sqrt(xx * yy); }
  } else if (which == 3) {
    double a = 0, c = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { a += (double)sp[i]; c += ldg<G64>(gt, gb + i); }
    const double ms = block_sum_d(a, sh) / n, cnt = block_sum_d(c, sh);
    double xx = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const double x = (double)sp[i] - ms; xx += x * x; }
    const double sd = sqrt(block_sum_d(xx, sh) / (n - 1));     // torch.std: unbiased
    double l = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) l += ((double)sp[i] - ms) / (sd + LOSS_EPS) * ldg<G64>(gt, gb + i);
    l = block_sum_d(l, sh);
    if (threadIdx.x == 0) { sv[0] = ms; sv[1] = sd; sv[2] = l; sv[3] = cnt; sv[5] = l / cnt; }
  } else {
    double mn = INFINITY, mg = INFINITY;
    int mi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double v = (double)sp[i];
      if (v < mn) { mn = v; mi = i; }
      mg = fmin(mg, ldg<G64>(gt, gb + i));
    }
    const MinIdx m = block_min_d(mn, mi, sh, shi);
    const double lo_g = -block_max_d(-mg, sh);
    double ds = 0, dg = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { ds += (double)sp[i] - m.v; dg += ldg<G64>(gt, gb + i) - lo_g; }
    const double Ds = block_sum_d(ds, sh), Dg = block_sum_d(dg, sh);
    double l = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
      l += fmin(((double)sp[i] - m.v) / Ds, (ldg<G64>(gt, gb + i) - lo_g) / Dg);
    l = block_sum_d(l, sh);
    if (threadIdx.x == 0) { sv[0] = m.v; sv[1] = Ds; sv[2] = lo_g; sv[3] = Dg; sv[4] = (double)m.i; sv[5] = l; }
  }
}

__global__ void loss_mean_kernel(int which, const double* saved, int B, float* loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double t = 0;
    const int slot = which == 0 ? 2 : 5;
    for (int b = 0; b < B; ++b) t += saved[b * 8 + slot];
    *loss = (float)(t / B);
  }
}

extern "C" int vinet_loss_fwd(int32_t which, const float* s, const void* gt, int32_t gt_is_f64, int32_t B, int32_t n,
                              double* saved, float* loss, void* stream) {
  VN_CHECK_ARG(which >= 0 && which <= 3 && s && gt && saved && loss && B > 0 && n > 0, "loss_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (gt_is_f64) hipLaunchKernelGGL(loss_fwd_kernel<true>, dim3(B), dim3(1024), 0, st, which, s, gt, n, saved);
  else hipLaunchKernelGGL(loss_fwd_kernel<false>, dim3(B), dim3(1024), 0, st, which, s, gt, n, saved);
  hipLaunchKernelGGL(loss_mean_kernel, dim3(1), dim3(64), 0, st, which, saved, B, loss);
  return vn_launch_status("loss_fwd");
}

template <bool G64>
__global__ __launch_bounds__(1024) void loss_bwd_kernel(int which, const float* __restrict__ s, const void* gt, int n,
                                                        int B, const double* __restrict__ saved, const float* gscale,
                                                        float coeff, int accumulate, float* __restrict__ ds) {
  __shared__ double sh[16];
  const int b = blockIdx.x;
  const float* sp = s + (long)b * n;
  float* dp = ds + (long)b * n;
  const long gb = (long)b * n;
  const double* sv = saved + b * 8;
  const double gsc = (double)(gscale ? *gscale : 1.f) * (double)coeff / (double)B;
  if (which == 0) {
    // L = sum q log(eps + q/(p+eps)), p = s/S:  dL/ds_j = (A_j - sum_i A_i p_i)/S,
    // A_i = -q_i^2 / ((eps + q_i/(p_i+eps)) (p_i+eps)^2)
    const double Ss = sv[0], Sg = sv[1];
    double ap = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double p = (double)sp[i] / Ss, q = ldg<G64>(gt, gb + i) / Sg;
      const double pe = p + LOSS_EPS;
      ap += (-q * q / ((LOSS_EPS + q / pe) * pe * pe)) * p;
    }
    ap = block_sum_d(ap, sh);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double p = (double)sp[i] / Ss, q = ldg<G64>(gt, gb + i) / Sg;
      const double pe = p + LOSS_EPS;
      const double A = -q * q / ((LOSS_EPS + q / pe) * pe * pe);
      const float g = (float)(gsc * (A - ap) / Ss);
      dp[i] = accumulate ? dp[i] + g : g;
    }
  } else if (which == 1) {
    // r = Sxy / sqrt(Sxx Syy):  dr/ds_j = y_j / sqrt(Sxx Syy) - r x_j / Sxx
    const double ms = sv[0], mg = sv[1], xx = sv[2], yy = sv[3], r = sv[5];
    const double inv = 1.0 / sqrt(xx * yy);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double x = (double)sp[i] - ms, y = ldg<G64>(gt, gb + i) - mg;
      const float g = (float)(gsc * (y * inv - r * x / xx));
      dp[i] = accumulate ? dp[i] + g : g;
    }
  } else {
    // p_i = u_i / D, u = s - lo, D = sum u;  M_i = [p_i < q_i] (0.5 on ties)
    const double lo = sv[0], D = sv[1], lo_g = sv[2], Dg = sv[3];
    const int am = (int)sv[4];
    double sm = 0, smp = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double p = ((double)sp[i] - lo) / D, q = (ldg<G64>(gt, gb + i) - lo_g) / Dg;
      const double M = p < q ? 1.0 : (p == q ? 0.5 : 0.0);
      sm += M; smp += M * p;
    }
    sm = block_sum_d(sm, sh); smp = block_sum_d(smp, sh);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const double p = ((double)sp[i] - lo) / D, q = (ldg<G64>(gt, gb + i) - lo_g) / Dg;
      const double M = p < q ? 1.0 : (p == q ? 0.5 : 0.0);
      double g = (M - smp) / D;
      if (i == am) g += (-sm + (double)n * smp) / D;
      const float gf = (float)(gsc * g);
      dp[i] = accumulate ? dp[i] + gf : gf;
    }
  }
}

extern "C" int vinet_loss_bwd(int32_t which, const float* s, const void* gt, int32_t gt_is_f64, int32_t B, int32_t n,
                              const double* saved, const float* gscale, float coeff, int32_t accumulate, float* ds,
                              void* stream) {
  VN_CHECK_ARG(which >= 0 && which <= 2 && s && gt && saved && ds && B > 0 && n > 0, "loss_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (gt_is_f64) hipLaunchKernelGGL(loss_bwd_kernel<true>, dim3(B), dim3(1024), 0, st, which, s, gt, n, B, saved, gscale, coeff, accumulate, ds);
  else hipLaunchKernelGGL(loss_bwd_kernel<false>, dim3(B), dim3(1024), 0, st, which, s, gt, n, B, saved, gscale, coeff, accumulate, ds);
  return vn_launch_status("loss_bwd");
}

// ---- Adam ---------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2, float gs) {
  const float step = lr / bc1;
  const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      const float4 gv = *(const float4*)(g + i);
      float4 mv = *(const float4*)(m + i), vv = *(const float4*)(v + i), pv = *(const float4*)(p + i);
      const float ge[4] = {gv.x * gs, gv.y * gs, gv.z * gs, gv.w * gs};
      float me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w}, pe[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        me[e] = b1 * me[e] + (1.f - b1) * ge[e];
        ve[e] = b2 * ve[e] + (1.f - b2) * ge[e] * ge[e];
        const float denom = sqrtf(ve[e]) * inv_sqrt_bc2 + eps;
        pe[e] -= step * (me[e] / denom);
      }
      *(float4*)(m + i) = make_float4(me[0], me[1], me[2], me[3]);
      *(float4*)(v + i) = make_float4(ve[0], ve[1], ve[2], ve[3]);
      *(float4*)(p + i) = make_float4(pe[0], pe[1], pe[2], pe[3]);
    } else {
      for (long k = i; k < n; ++k) {
        const float ge = g[k] * gs;
        const float me = b1 * m[k] + (1.f - b1) * ge;
        const float ve = b2 * v[k] + (1.f - b2) * ge * ge;
        m[k] = me; v[k] = ve;
        p[k] -= step * (me / (sqrtf(ve) * inv_sqrt_bc2 + eps));
      }
    }
  }
}

extern "C" int vinet_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                               float beta2, float eps, float bias_c1, float bias_c2, float grad_scale, void* stream) {
  VN_CHECK_ARG(p && g && m && v && n > 0, "adam_step: bad arguments");
  VN_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: buffers must be 16-byte aligned");
  long grid = (n / 4 + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr, beta1,
                     beta2, eps, bias_c1, bias_c2, grad_scale);
  return vn_launch_status("adam_step");
}

// ---- nn.Bilinear fusion (model.py:230,236) -------------------------------------
#define BIL_MAX_I 48
#define BIL_MAX_J 4

// lane <-> channel c (C is the fastest axis of x1 / x2 / out: every access is a coalesced row); the I + J inputs of
// the lane's channel live in registers (fully unrolled, predicated loops: runtime-indexed arrays would go to
// scratch), the weights are wave-uniform (scalar loads), blockIdx.z splits the O outputs.
template <typename T>
__global__ __launch_bounds__(64) void bilinear_fwd_kernel(const T* __restrict__ x1, const T* __restrict__ x2, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int C, int I, int J, int O, int o_per, T* __restrict__ out) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int b = blockIdx.y;
  const bool ok = c < C;
  const int cc = ok ? c : 0;
  float a[BIL_MAX_I], d[BIL_MAX_J];
#pragma unroll
  for (int i = 0; i < BIL_MAX_I; ++i) a[i] = i < I ? load1<T>(x1 + ((long)b * I + i) * C + cc) : 0.f;
#pragma unroll
  for (int j = 0; j < BIL_MAX_J; ++j) d[j] = j < J ? load1<T>(x2 + ((long)b * J + j) * C + cc) : 0.f;
  const int o0 = blockIdx.z * o_per;
  const int o1 = o0 + o_per < O ? o0 + o_per : O;
  for (int o = o0; o < o1; ++o) {
    const float* wo = w + (long)o * I * J;
    float acc = bias ? bias[o] : 0.f;
#pragma unroll
    for (int i = 0; i < BIL_MAX_I; ++i) {
      if (i < I) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < BIL_MAX_J; ++j)
          if (j < J) t = fmaf(wo[i * J + j], d[j], t);
        acc = fmaf(a[i], t, acc);
      }
    }
    if (ok) store1<T>(out + ((long)b * O + o) * C + c, acc);
  }
}

extern "C" int vinet_bilinear_fwd(const void* x1, const void* x2, int32_t dtype, const float* w, const float* bias,
                                  int32_t B, int32_t C, int32_t I, int32_t J, int32_t O, void* out, void* stream) {
  VN_CHECK_ARG(x1 && x2 && w && out && B > 0 && C > 0 && I > 0 && I <= BIL_MAX_I && J > 0 && J <= BIL_MAX_J && O > 0,
               "bilinear_fwd: bad arguments");
  const int o_per = 16;
  const dim3 grid((C + 63) / 64, B, (O + o_per - 1) / o_per), blk(64);
  if (dtype == VINET_F32) hipLaunchKernelGGL(bilinear_fwd_kernel<float>, grid, blk, 0, (hipStream_t)stream, (const float*)x1, (const float*)x2, w, bias, C, I, J, O, o_per, (float*)out);
  else hipLaunchKernelGGL(bilinear_fwd_kernel<bf16_t>, grid, blk, 0, (hipStream_t)stream, (const bf16_t*)x1, (const bf16_t*)x2, w, bias, C, I, J, O, o_per, (bf16_t*)out);
  return vn_launch_status("bilinear_fwd");
}

// dx1[b,i,c] = sum_{o,j} dout[b,o,c] w[o,i,j] x2[b,j,c];  dx2[b,j,c] = sum_{o,i} dout[b,o,c] w[o,i,j] x1[b,i,c].
// One workgroup per (b, 64-channel tile): lane <-> channel, the four waves split the I inputs (register
// accumulators, fully unrolled), weights go through LDS 16 outputs at a time (wave-uniform broadcast reads), the
// dx2 partial sums of the four waves meet in LDS at the end.
#define BILX_IPW ((BIL_MAX_I + 3) / 4)
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_x_kernel(const T* __restrict__ x1, const T* __restrict__ x2, const T* __restrict__ dout,
                                                             const float* __restrict__ w, int C, int I, int J, int O, T* __restrict__ dx1,
                                                             T* __restrict__ dx2) {
  __shared__ float Ws[16][BIL_MAX_I * BIL_MAX_J];
  __shared__ float Gd[4][BIL_MAX_J][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int b = blockIdx.y;
  const bool ok = c < C;
  const int cc = ok ? c : 0;
  const int ipw = (I + 3) / 4, i0 = wave * ipw;
  const int IJ = I * J;
  float a[BILX_IPW], ga[BILX_IPW], d[BIL_MAX_J], gd[BIL_MAX_J];
#pragma unroll
  for (int k = 0; k < BILX_IPW; ++k) { a[k] = (k < ipw && i0 + k < I) ? load1<T>(x1 + ((long)b * I + i0 + k) * C + cc) : 0.f; ga[k] = 0.f; }
#pragma unroll
  for (int j = 0; j < BIL_MAX_J; ++j) { d[j] = j < J ? load1<T>(x2 + ((long)b * J + j) * C + cc) : 0.f; gd[j] = 0.f; }
  for (int o0 = 0; o0 < O; o0 += 16) {
    __syncthreads();
    for (int e = tid; e < 16 * IJ; e += 256) {
      const int ol = e / IJ, ij = e - ol * IJ;
      Ws[ol][ij] = o0 + ol < O ? w[(long)(o0 + ol) * IJ + ij] : 0.f;
    }
    __syncthreads();
    for (int ol = 0; ol < 16 && o0 + ol < O; ++ol) {
      const float g = load1<T>(dout + ((long)b * O + o0 + ol) * C + cc);
      const float* wo = &Ws[ol][i0 * J];
#pragma unroll
      for (int k = 0; k < BILX_IPW; ++k) {
        if (k < ipw && i0 + k < I) {
#pragma unroll
          for (int j = 0; j < BIL_MAX_J; ++j)
            if (j < J) {
              const float wv = wo[k * J + j] * g;
              ga[k] = fmaf(wv, d[j], ga[k]);
              gd[j] = fmaf(wv, a[k], gd[j]);
            }
        }
      }
    }
  }
  if (dx1 && ok) {
#pragma unroll
    for (int k = 0; k < BILX_IPW; ++k)
      if (k < ipw && i0 + k < I) store1<T>(dx1 + ((long)b * I + i0 + k) * C + c, ga[k]);
  }
  if (dx2) {
#pragma unroll
    for (int j = 0; j < BIL_MAX_J; ++j) Gd[wave][j][lane] = gd[j];
    __syncthreads();
    if (wave == 0 && ok) {
#pragma unroll
      for (int j = 0; j < BIL_MAX_J; ++j)
        if (j < J) store1<T>(dx2 + ((long)b * J + j) * C + c, Gd[0][j][lane] + Gd[1][j][lane] + Gd[2][j][lane] + Gd[3][j][lane]);
    }
  }
}

// dw[o][i][j] += sum_{b,c} dout[b,o,c] x1[b,i,c] x2[b,j,c];  dbias[o] += sum dout.
// A workgroup owns 16 outputs o x all (i,j) (+ one bias column) and walks a strided share of the (b, 64-channel
// tile) pairs: the x1 / x2 / dout tiles of a pair are staged in LDS as fp32, every lane keeps its 8 (o, ij)
// accumulators in registers and sweeps the 64 channels starting at a lane-dependent offset (all three arrays are
// [row][64]: a common c would put the whole wave on one bank).  One atomicAdd per (o, ij) and workgroup at the end;
// the per-(o,ij)-block version read every input 127 x 336 times from global memory (6 ms of AViNet's step).
#define BILW_OT 16
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_w_kernel(const T* __restrict__ x1, const T* __restrict__ x2,
                                                             const T* __restrict__ dout, int B, int C, int I, int J,
                                                             int O, float* __restrict__ dw, float* __restrict__ dbias) {
  __shared__ float A[BIL_MAX_I][64], D[BIL_MAX_J][64], G[BILW_OT][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int o0 = blockIdx.x * BILW_OT;
  const int IJ = I * J, cols = IJ + 1;                 // last column = bias
  const int nout = BILW_OT * cols;
  const int ctiles = (C + 63) / 64, npairs = B * ctiles;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int p = blockIdx.y; p < npairs; p += gridDim.y) {
    const int b = p / ctiles, c0 = (p - b * ctiles) * 64;
    __syncthreads();
    for (int e = tid; e < I * 64; e += 256) { const int i = e >> 6, c = c0 + (e & 63); A[i][e & 63] = c < C ? load1<T>(x1 + ((long)b * I + i) * C + c) : 0.f; }
    for (int e = tid; e < J * 64; e += 256) { const int j = e >> 6, c = c0 + (e & 63); D[j][e & 63] = c < C ? load1<T>(x2 + ((long)b * J + j) * C + c) : 0.f; }
    for (int e = tid; e < BILW_OT * 64; e += 256) {
      const int o = o0 + (e >> 6), c = c0 + (e & 63);
      G[e >> 6][e & 63] = (o < O && c < C) ? load1<T>(dout + ((long)b * O + o) * C + c) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {          // (static k: acc[] stays in registers; the sweeps below must NOT be unrolled
      const int idx = tid + 256 * k;       //  further -- fully unrolled this kernel needs 2048 live LDS reads and spills 4 KB/lane)
      if (idx < nout) {
        const int ol = idx / cols, ij = idx - ol * cols;
        const float* gp = &G[ol][0];
        float a0 = 0.f, a1 = 0.f;
        if (ij < IJ) {
          const int i = ij / J, j = ij - i * J;
          const float* ap = &A[i][0];
          const float* dp = &D[j][0];
#pragma unroll 2
          for (int cc = 0; cc < 64; cc += 2) {
            const int c = (cc + lane) & 63, c2 = (cc + 1 + lane) & 63;
            a0 = fmaf(gp[c] * ap[c], dp[c], a0);
            a1 = fmaf(gp[c2] * ap[c2], dp[c2], a1);
          }
        } else {
#pragma unroll 4
          for (int cc = 0; cc < 64; ++cc) a0 += gp[(cc + lane) & 63];
        }
        acc[k] += a0 + a1;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int idx = tid + 256 * k;
    if (idx < nout) {
      const int ol = idx / cols, ij = idx - ol * cols, o = o0 + ol;
      if (o < O) {
        if (ij < IJ) atomicAdd(dw + (long)o * IJ + ij, acc[k]);
        else if (dbias) atomicAdd(dbias + o, acc[k]);
      }
    }
  }
}

extern "C" int vinet_bilinear_bwd(const void* x1, const void* x2, const void* dout, int32_t dtype, const float* w,
                                  int32_t B, int32_t C, int32_t I, int32_t J, int32_t O, void* dx1, void* dx2,
                                  float* dw, float* dbias, void* stream) {
  VN_CHECK_ARG(x1 && x2 && dout && w && B > 0 && C > 0 && I > 0 && I <= BIL_MAX_I && J > 0 && J <= BIL_MAX_J && O > 0,
               "bilinear_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const dim3 gx((C + 63) / 64, B), bx(256);
  VN_CHECK_ARG(BILW_OT * (I * J + 1) <= 8 * 256, "bilinear_bwd: I*J too large");
  int splits = B * ((C + 63) / 64);
  if (splits > 32) splits = 32;
  const dim3 gw((O + BILW_OT - 1) / BILW_OT, splits), bw(256);
  if (dtype == VINET_F32) {
    if (dx1 || dx2) hipLaunchKernelGGL(bilinear_bwd_x_kernel<float>, gx, bx, 0, s, (const float*)x1, (const float*)x2, (const float*)dout, w, C, I, J, O, (float*)dx1, (float*)dx2);
    if (dw) hipLaunchKernelGGL(bilinear_bwd_w_kernel<float>, gw, bw, 0, s, (const float*)x1, (const float*)x2, (const float*)dout, B, C, I, J, O, dw, dbias);
  } else {
    if (dx1 || dx2) hipLaunchKernelGGL(bilinear_bwd_x_kernel<bf16_t>, gx, bx, 0, s, (const bf16_t*)x1, (const bf16_t*)x2, (const bf16_t*)dout, w, C, I, J, O, (bf16_t*)dx1, (bf16_t*)dx2);
    if (dw) hipLaunchKernelGGL(bilinear_bwd_w_kernel<bf16_t>, gw, bw, 0, s, (const bf16_t*)x1, (const bf16_t*)x2, (const bf16_t*)dout, B, C, I, J, O, dw, dbias);
  }
  return vn_launch_status("bilinear_bwd");
}
