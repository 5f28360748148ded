// MaxPool3d forward / backward kernels (model.py:696-714, model_utils.py:178): generic gathers, the packed-key 8-channel
// forward, LDS halo-tile and T-walking 3x3x3/s1 kernels, the 2x2-block 1x3x3/s2 backward.
#include "elementwise.h"

// ============================================================================
// MaxPool3d
// ============================================================================
struct PoolP { int kT, kH, kW, sT, sH, sW, pT, pH, pW; FastDiv dsT, dsH, dsW; };
static inline PoolP make_poolp(const VinetPoolDesc* d) {
  PoolP p = {d->kT, d->kH, d->kW, d->sT, d->sH, d->sW, d->pT, d->pH, d->pW, make_fastdiv((uint32_t)d->sT), make_fastdiv((uint32_t)d->sH), make_fastdiv((uint32_t)d->sW)};
  return p;
}

// Pooling semantics: a pending affine (+ReLU) is applied AND rounded to the activation dtype before the comparison --
// what a bf16 pipeline that had stored the BN+ReLU output would pool over, and what every other consumer of a
// pending activation sees (the conv kernels round at fragment time).  It also makes the packed 16-bit kernel below
// agree with these fp32-compare kernels on every tie.
template <typename T> VN_DEV float pool_round(float v) { return v; }
template <> VN_DEV float pool_round<bf16_t>(float v) { return bf2f(f2bf(v)); }

template <typename T>
__global__ void maxpool_fwd_kernel(PoolP p, TView x, Affine pre, TView y, uint8_t* __restrict__ argmax, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, y.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(y.C / 4));
  const long vox = (long)vox_u;
  int b, to, ho, wo;
  decode_vox(y, vox, b, to, ho, wo);
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {0, 0, 0, 0};
  for (int kt = 0; kt < p.kT; ++kt) {
    const int t = to * p.sT - p.pT + kt;
    if ((unsigned)t >= (unsigned)x.T) continue;
    for (int kh = 0; kh < p.kH; ++kh) {
      const int h = ho * p.sH - p.pH + kh;
      if ((unsigned)h >= (unsigned)x.H) continue;
      for (int kw = 0; kw < p.kW; ++kw) {
        const int w = wo * p.sW - p.pW + kw;
        if ((unsigned)w >= (unsigned)x.W) continue;
        float4 v = ldq<T>((const T*)x.p + vox_off(x, b, t, h, w) + q * 4);
        v = affine4(v, pre, q * 4);
        if (pre.scale) { v.x = pool_round<T>(v.x); v.y = pool_round<T>(v.y); v.z = pool_round<T>(v.z); v.w = pool_round<T>(v.w); }
        const int tap = (kt * p.kH + kh) * p.kW + kw;
        const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (f[e] > best[e] || (f[e] != f[e] && best[e] == best[e])) { best[e] = f[e]; bi[e] = tap; }
      }
    }
  }
  stq<T>((T*)y.p + vox_off(y, b, to, ho, wo) + q * 4, make_float4(best[0], best[1], best[2], best[3]));
  if (argmax) *(uint32_t*)(argmax + vox * y.C + q * 4) = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
}

// kT == 3, sT == 1, pT == 1 (the Inception branch-3 pools, model_utils.py:178): one thread
// walks T for a fixed output (h,w), keeps the maxima of the last three (kH x kW) planes and
// so reads kH*kW instead of 3*kH*kW inputs per output.  Same first-max tie rule: planes in
// t order, (h,w) scan order inside a plane, strict comparisons.
template <typename T>
__global__ void maxpool_tslide_kernel(PoolP p, TView x, Affine pre, TView y, uint8_t* __restrict__ argmax, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t col = fdiv((uint32_t)i, y.dQ);
  const int q = (int)((uint32_t)i - col * (uint32_t)(y.C / 4));
  const uint32_t r1 = fdiv(col, y.dW);
  const int wo = (int)(col - r1 * (uint32_t)y.W);
  const uint32_t r2 = fdiv(r1, y.dH);
  const int ho = (int)(r1 - r2 * (uint32_t)y.H);
  const int b = (int)r2;
  float pm[3][4];
  int pa[3][4];
  const int khw = p.kH * p.kW;
  // plane tp feeds outputs tp-1, tp, tp+1; output t is complete once plane t+1 is in
  for (int tp = 0; tp <= x.T; ++tp) {
    const int slot = tp % 3;
    if (tp < x.T) {
      float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int bi[4] = {0, 0, 0, 0};
      for (int kh = 0; kh < p.kH; ++kh) {
        const int h = ho * p.sH - p.pH + kh;
        if ((unsigned)h >= (unsigned)x.H) continue;
        for (int kw = 0; kw < p.kW; ++kw) {
          const int w = wo * p.sW - p.pW + kw;
          if ((unsigned)w >= (unsigned)x.W) continue;
          float4 v = ldq<T>((const T*)x.p + vox_off(x, b, tp, h, w) + q * 4);
          v = affine4(v, pre, q * 4);
          if (pre.scale) { v.x = pool_round<T>(v.x); v.y = pool_round<T>(v.y); v.z = pool_round<T>(v.z); v.w = pool_round<T>(v.w); }
        if (pre.scale) { v.x = pool_round<T>(v.x); v.y = pool_round<T>(v.y); v.z = pool_round<T>(v.z); v.w = pool_round<T>(v.w); }
          const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (f[e] > best[e] || (f[e] != f[e] && best[e] == best[e])) { best[e] = f[e]; bi[e] = kh * p.kW + kw; }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { pm[slot][e] = best[e]; pa[slot][e] = bi[e]; }
    }
    const int to = tp - 1;
    if (to < 0) continue;
    float o[4];
    int oi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = -INFINITY; oi[e] = 0; }
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int t = to - 1 + kt;
      if (t < 0 || t >= x.T) continue;
      const int sl = t % 3;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (pm[sl][e] > o[e] || (pm[sl][e] != pm[sl][e] && o[e] == o[e])) { o[e] = pm[sl][e]; oi[e] = kt * khw + pa[sl][e]; }
    }
    stq<T>((T*)y.p + vox_off(y, b, to, ho, wo) + q * 4, make_float4(o[0], o[1], o[2], o[3]));
    if (argmax) {
      const long ovox = (((long)b * y.T + to) * y.H + ho) * y.W + wo;
      *(uint32_t*)(argmax + ovox * y.C + q * 4) = (uint32_t)oi[0] | ((uint32_t)oi[1] << 8) | ((uint32_t)oi[2] << 16) | ((uint32_t)oi[3] << 24);
    }
  }
}

// 8 channels per lane forms of the two forward kernels above (same scan order and tie rule)
VN_DEV void affine8(float* v, const Affine& a, int c) {
  if (a.scale) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], a.scale[c + e], a.shift[c + e]);
  }
  if (a.relu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd8_kernel(PoolP p, TView x, Affine pre, TView y, uint8_t* __restrict__ argmax, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = y.C >> 3;
  const uint32_t vox_u = (uint32_t)(i / G);
  const int g = (int)(i - (long)vox_u * G);
  int b, to, ho, wo;
  decode_vox(y, (long)vox_u, b, to, ho, wo);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = pre.scale ? pre.scale[g * 8 + e] : 1.f; sh[e] = pre.scale ? pre.shift[g * 8 + e] : 0.f; }
  float best[8];
  unsigned long long bi = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
  for (int kt = 0; kt < p.kT; ++kt) {
    const int t = to * p.sT - p.pT + kt;
    if ((unsigned)t >= (unsigned)x.T) continue;
    for (int kh = 0; kh < p.kH; ++kh) {
      const int h = ho * p.sH - p.pH + kh;
      if ((unsigned)h >= (unsigned)x.H) continue;
      for (int kw = 0; kw < p.kW; ++kw) {
        const int w = wo * p.sW - p.pW + kw;
        if ((unsigned)w >= (unsigned)x.W) continue;
        float f[8];
        ld8<T>((const T*)x.p + vox_off(x, b, t, h, w) + g * 8, f);
        const unsigned long long tap = (unsigned long long)((kt * p.kH + kh) * p.kW + kw);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = fmaf(f[e], sc[e], sh[e]);
          if (pre.relu) v = fmaxf(v, 0.f);
          if (pre.scale) v = pool_round<T>(v);
          if (v > best[e] || (v != v && best[e] == best[e])) { best[e] = v; bi = (bi & ~(0xffull << (8 * e))) | (tap << (8 * e)); }
        }
      }
    }
  }
  st8<T>((T*)y.p + vox_off(y, b, to, ho, wo) + g * 8, best);
  if (argmax) *(unsigned long long*)(argmax + (long)vox_u * y.C + g * 8) = bi;
}

// 8-channel T-walking forward for kT == 3, sT == 1, pT == 1 (any in-plane window): one lane owns an output
// column (b, ho, wo, 8 channels), computes each input plane's (kH x kW) window maximum ONCE and keeps the
// last three in named registers (explicit rotation: a runtime-indexed ring would live in scratch).  Tie rule
// as everywhere: planes in t order, (h, w) scan order inside a plane, strict comparisons.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_tslide8_kernel(PoolP p, TView x, Affine pre, TView y, uint8_t* __restrict__ argmax, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = y.C >> 3;
  const uint32_t col = (uint32_t)(i / G);
  const int g = (int)(i - (long)col * G);
  const uint32_t r1 = fdiv(col, y.dW);
  const int wo = (int)(col - r1 * (uint32_t)y.W);
  const uint32_t r2 = fdiv(r1, y.dH);
  const int ho = (int)(r1 - r2 * (uint32_t)y.H);
  const int b = (int)r2;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = pre.scale ? pre.scale[g * 8 + e] : 1.f; sh[e] = pre.scale ? pre.shift[g * 8 + e] : 0.f; }
  const int khw = p.kH * p.kW;
  float m_a[8], m_b[8], m_c[8];                 // plane maxima of planes tp-2, tp-1, tp
  unsigned long long i_a = 0, i_b = 0, i_c = 0;   // ... and their in-plane argmax codes (8 x 8 bit)
#pragma unroll
  for (int e = 0; e < 8; ++e) { m_a[e] = -INFINITY; m_b[e] = -INFINITY; m_c[e] = -INFINITY; }
  for (int tp = 0; tp <= x.T; ++tp) {
    // rotate: (a, b, c) <- (b, c, new plane tp)
#pragma unroll
    for (int e = 0; e < 8; ++e) { m_a[e] = m_b[e]; m_b[e] = m_c[e]; m_c[e] = -INFINITY; }
    i_a = i_b; i_b = i_c; i_c = 0;
    if (tp < x.T) {
      for (int kh = 0; kh < p.kH; ++kh) {
        const int h = ho * p.sH - p.pH + kh;
        if ((unsigned)h >= (unsigned)x.H) continue;
        for (int kw = 0; kw < p.kW; ++kw) {
          const int w = wo * p.sW - p.pW + kw;
          if ((unsigned)w >= (unsigned)x.W) continue;
          float f[8];
          ld8<T>((const T*)x.p + vox_off(x, b, tp, h, w) + g * 8, f);
          const unsigned long long code = (unsigned long long)(kh * p.kW + kw);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v = fmaf(f[e], sc[e], sh[e]);
            if (pre.relu) v = fmaxf(v, 0.f);
            if (pre.scale) v = pool_round<T>(v);
            if (v > m_c[e] || (v != v && m_c[e] == m_c[e])) { m_c[e] = v; i_c = (i_c & ~(0xffull << (8 * e))) | (code << (8 * e)); }
          }
        }
      }
    }
    const int to = tp - 1;       // complete once plane tp = to + 1 is in: window planes (a, b, c) = (to-1, to, to+1)
    if (to < 0) continue;
    float o[8];
    unsigned long long oi = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float best = -INFINITY;
      unsigned long long bi = 0;
      // kt = 0: plane to-1 (exists iff to >= 1; otherwise m_a is -inf and never wins against a real plane)
      if (to >= 1 && (m_a[e] > best || (m_a[e] != m_a[e] && best == best))) { best = m_a[e]; bi = (i_a >> (8 * e)) & 0xffull; }
      if (m_b[e] > best || (m_b[e] != m_b[e] && best == best)) { best = m_b[e]; bi = (unsigned long long)khw + ((i_b >> (8 * e)) & 0xffull); }
      if (to + 1 < x.T && (m_c[e] > best || (m_c[e] != m_c[e] && best == best))) { best = m_c[e]; bi = 2ull * khw + ((i_c >> (8 * e)) & 0xffull); }
      o[e] = best;
      oi |= bi << (8 * e);
    }
    st8<T>((T*)y.p + vox_off(y, b, to, ho, wo) + g * 8, o);
    if (argmax) {
      const long ovox = (((long)b * y.T + to) * y.H + ho) * y.W + wo;
      *(unsigned long long*)(argmax + ovox * y.C + g * 8) = oi;
    }
  }
}

// 3x3x3 / s1 / p1 forward through an LDS halo tile: a 512-thread workgroup owns 8 x 8 outputs x 64 channels of
// one clip and walks T.  Per input plane the 10 x 10 halo (affine + ReLU applied once per element, fp32) is staged
// in LDS (double buffered: one barrier per plane), every lane takes its 3 x 3 window maximum from LDS, and the
// last three plane maxima live in named registers as in maxpool_tslide8_kernel.  One global read per input
// element (plus halo) instead of nine cached ones: the T-walking kernel is bound by the CU's load path (0.8 TB/s).
// Out-of-range halo positions hold -inf: strict comparisons never select them, so the tie rule is unchanged.
template <typename T>
__global__ __launch_bounds__(512) void maxpool_k3s1_lds_kernel(TView x, Affine pre, TView y, uint8_t* __restrict__ argmax,
                                                               int tilesH, int tilesW) {
  __shared__ __attribute__((aligned(16))) float P[2][100][64];
  const int tid = threadIdx.x;
  const int oct = tid & 7, pos = tid >> 3;          // 8 channel octets x 64 positions
  const int ph = pos >> 3, pw = pos & 7;
  int bid = blockIdx.x;
  const int ncg = (x.C + 63) >> 6;                  // channel groups of 64 (the last may be partial)
  const int cg = bid % ncg; bid /= ncg;
  const int tw = bid % tilesW; bid /= tilesW;
  const int th = bid % tilesH; bid /= tilesH;
  const int b = bid;
  const int h0 = th * 8, w0 = tw * 8, c0 = cg * 64 + oct * 8;
  const int ho = h0 + ph, wo = w0 + pw;
  const bool out_ok = ho < y.H && wo < y.W && c0 < x.C;
  // scale / shift of the tile's 64 channels in LDS (16 more live registers per lane would cost a workgroup of
  // occupancy, which this latency-bound walk cannot afford)
  __shared__ __attribute__((aligned(16))) float S[2][64];
  if (tid < 64) {
    const bool ok = pre.scale != nullptr && cg * 64 + tid < x.C;
    S[0][tid] = ok ? pre.scale[cg * 64 + tid] : 1.f;
    S[1][tid] = ok ? pre.shift[cg * 64 + tid] : 0.f;
  }
  __syncthreads();
  float m_a[8], m_b[8], m_c[8];
  unsigned long long i_a = 0, i_b = 0, i_c = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) { m_a[e] = -INFINITY; m_b[e] = -INFINITY; m_c[e] = -INFINITY; }
  const int T_ = x.T;
  for (int tp = 0; tp <= T_; ++tp) {
    float (*buf)[64] = P[tp & 1];
    if (tp < T_) {
      // stage the halo of plane tp: 100 positions x 8 octets = 800 items over 512 threads
      for (int it = tid; it < 800; it += 512) {
        const int o8 = oct, hp = it >> 3;           // (it & 7) == oct
        const int hh = h0 - 1 + hp / 10, ww = w0 - 1 + hp % 10;
        float v[8];
        if (cg * 64 + o8 * 8 >= x.C) continue;        // partial last channel group
        if ((unsigned)hh < (unsigned)x.H && (unsigned)ww < (unsigned)x.W) {
          ld8<T>((const T*)x.p + vox_off(x, b, tp, hh, ww) + cg * 64 + o8 * 8, v);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = fmaf(v[e], S[0][oct * 8 + e], S[1][oct * 8 + e]);
            if (pre.relu) v[e] = fmaxf(v[e], 0.f);
            if (pre.scale) v[e] = pool_round<T>(v[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = -INFINITY;
        }
        *(float4*)&buf[hp][o8 * 8] = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)&buf[hp][o8 * 8 + 4] = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
    __syncthreads();     // plane tp staged; every lane finished reading the buffer staged two planes ago
#pragma unroll
    for (int e = 0; e < 8; ++e) { m_a[e] = m_b[e]; m_b[e] = m_c[e]; m_c[e] = -INFINITY; }
    i_a = i_b; i_b = i_c; i_c = 0;
    if (tp < T_) {
      // window maximum of the plane, 4 channels at a time: the max by four 3-input maxima, then the FIRST tap
      // that equals it (scan from the last tap down, so the smallest index is the one left standing) -- 20 VALU
      // per element instead of ~45 for compare-and-track.  v_max3 drops NaNs where aten propagates them: a NaN
      // anywhere in the window (sum test) takes the compare-and-track path.
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float f[9][4];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float4 q = *(const float4*)&buf[(ph + kh) * 10 + pw + kw][oct * 8 + half * 4];
            f[kh * 3 + kw][0] = q.x; f[kh * 3 + kw][1] = q.y; f[kh * 3 + kw][2] = q.z; f[kh * 3 + kw][3] = q.w;
          }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int e = half * 4 + c;
          const float sum = ((f[0][c] + f[1][c]) + (f[2][c] + f[3][c])) + ((f[4][c] + f[5][c]) + (f[6][c] + f[7][c])) + f[8][c];
          float m;
          unsigned idx;
          if (sum == sum) {
            m = fmaxf(fmaxf(fmaxf(f[0][c], f[1][c]), f[2][c]), fmaxf(fmaxf(fmaxf(f[3][c], f[4][c]), f[5][c]), fmaxf(fmaxf(f[6][c], f[7][c]), f[8][c])));
            idx = 8;
#pragma unroll
            for (int k = 7; k >= 0; --k) idx = (f[k][c] == m) ? (unsigned)k : idx;
          } else {
            m = -INFINITY; idx = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k)
              if (f[k][c] > m || (f[k][c] != f[k][c] && m == m)) { m = f[k][c]; idx = (unsigned)k; }
          }
          m_c[e] = m;
          i_c |= (unsigned long long)idx << (8 * e);
        }
      }
    }
    const int to = tp - 1;
    if (to < 0 || !out_ok) continue;
    float o[8];
    unsigned long long oi = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float best = -INFINITY;
      unsigned long long bi = 0;
      if (to >= 1 && (m_a[e] > best || (m_a[e] != m_a[e] && best == best))) { best = m_a[e]; bi = (i_a >> (8 * e)) & 0xffull; }
      if (m_b[e] > best || (m_b[e] != m_b[e] && best == best)) { best = m_b[e]; bi = 9ull + ((i_b >> (8 * e)) & 0xffull); }
      if (to + 1 < T_ && (m_c[e] > best || (m_c[e] != m_c[e] && best == best))) { best = m_c[e]; bi = 18ull + ((i_c >> (8 * e)) & 0xffull); }
      o[e] = best;
      oi |= bi << (8 * e);
    }
    st8<T>((T*)y.p + vox_off(y, b, to, ho, wo) + c0, o);
    if (argmax) {
      const long ovox = (((long)b * y.T + to) * y.H + ho) * y.W + wo;
      *(unsigned long long*)(argmax + ovox * y.C + c0) = oi;
    }
  }
}


// order-preserving 16-bit codes of two packed bf16 (see maxpool_k3s1_pk_kernel) and their inverse
VN_DEV uint32_t pool_code2(uint32_t u) {
  const uint32_t m = ((u >> 15) & 0x00010001u) * 0x7fffu;
  return u ^ (m | 0x80008000u);
}
VN_DEV uint32_t pool_decode2(uint32_t k) {
  const uint32_t m = (((k >> 15) & 0x00010001u) ^ 0x00010001u) * 0x7fffu;
  return k ^ (m | 0x80008000u);
}

// Generic window (any k / stride / padding, up to 255 taps), bf16, on packed keys: one lane = one output voxel x 8
// channels; every in-range tap is loaded (16 B), transformed, coded and folded with key = code << 16 | (ntaps-1-tap).
// 7 VALU per element and tap instead of ~12 for affine + round + compare-and-track in fp32.
__global__ __launch_bounds__(256) void maxpool_fwd8_pk_kernel(PoolP p, TView x, Affine pre, TView y, uint8_t* __restrict__ argmax, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = y.C >> 3;
  const uint32_t vox_u = (uint32_t)(i / G);
  const int g = (int)(i - (long)vox_u * G);
  int b, to, ho, wo;
  decode_vox(y, (long)vox_u, b, to, ho, wo);
  f32x2_v sc2[4], sh2[4];
  const bool aff = pre.scale != nullptr;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sc2[e] = aff ? (f32x2_v){pre.scale[g * 8 + 2 * e], pre.scale[g * 8 + 2 * e + 1]} : (f32x2_v){1.f, 1.f};
    sh2[e] = aff ? (f32x2_v){pre.shift[g * 8 + 2 * e], pre.shift[g * 8 + 2 * e + 1]} : (f32x2_v){0.f, 0.f};
  }
  uint32_t best[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) best[e] = 0;
  const uint32_t last = (uint32_t)(p.kT * p.kH * p.kW - 1);
  for (int kt = 0; kt < p.kT; ++kt) {
    const int t = to * p.sT - p.pT + kt;
    if ((unsigned)t >= (unsigned)x.T) continue;
    for (int kh = 0; kh < p.kH; ++kh) {
      const int h = ho * p.sH - p.pH + kh;
      if ((unsigned)h >= (unsigned)x.H) continue;
      for (int kw = 0; kw < p.kW; ++kw) {
        const int w = wo * p.sW - p.pW + kw;
        if ((unsigned)w >= (unsigned)x.W) continue;
        const uint4 q = *(const uint4*)((const bf16_t*)x.p + vox_off(x, b, t, h, w) + g * 8);
        uint32_t w4[4] = {q.x, q.y, q.z, q.w};
        const uint32_t ck = last - (uint32_t)((kt * p.kH + kh) * p.kW + kw);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (aff) {
            if (pre.relu) w4[e] = pre_relu_pair(w4[e], sc2[e], sh2[e]);
            else w4[e] = pack2bf(fmaf(__uint_as_float(w4[e] << 16), sc2[e].x, sh2[e].x), fmaf(__uint_as_float(w4[e] & 0xffff0000u), sc2[e].y, sh2[e].y));
          } else if (pre.relu) {
            asm("v_pk_max_i16 %0, %1, 0" : "=v"(w4[e]) : "v"(w4[e]));
          }
          const uint32_t c = pool_code2(w4[e]);
          const uint32_t klo = (c << 16) | ck, khi = (c & 0xffff0000u) | ck;
          best[2 * e] = best[2 * e] > klo ? best[2 * e] : klo;
          best[2 * e + 1] = best[2 * e + 1] > khi ? best[2 * e + 1] : khi;
        }
      }
    }
  }
  uint32_t ov[4];
  unsigned long long bi = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const uint32_t k = best[e];
    if (e & 1) ov[e >> 1] |= k & 0xffff0000u; else ov[e >> 1] = k >> 16;
    bi |= (unsigned long long)(last - (k & 0xffu)) << (8 * e);
  }
  uint4 o;
  o.x = pool_decode2(ov[0]); o.y = pool_decode2(ov[1]); o.z = pool_decode2(ov[2]); o.w = pool_decode2(ov[3]);
  *(uint4*)((bf16_t*)y.p + vox_off(y, b, to, ho, wo) + g * 8) = o;
  if (argmax) *(unsigned long long*)(argmax + (long)vox_u * y.C + g * 8) = bi;
}

// bf16 form of the kernel above on PACKED 32-bit keys.  The halo holds order-preserving 16-bit codes of the
// (affine + ReLU'd, bf16-rounded) activations: code = bits ^ 0x8000 for non-negative values, ~bits for negative
// ones, so unsigned integer order = numeric order and 0 is below everything (out-of-range taps).  A window
// candidate becomes key = code << 16 | (8 - tap): v_max3_u32 over the nine keys yields the maximum AND, in its low
// bits, the first tap that attains it (ties: the larger low field = the smaller tap) -- 1 op to build a key, half an
// op to fold it, instead of ~20 for max-then-find-first in fp32.  The three plane keys are folded the same way with
// +18 / +9 / +0 (earlier plane wins ties), so tap = 26 - (key & 0xff).  Half the LDS bytes per plane, 9 instead of 18
// ds_read_b128 per lane and plane.  (A sign-bit NaN would order lowest; aten's NaN-wins only holds for positive NaNs.)
__global__ __launch_bounds__(512) void maxpool_k3s1_pk_kernel(TView x, Affine pre, TView y, uint8_t* __restrict__ argmax,
                                                              int tilesH, int tilesW) {
  __shared__ __attribute__((aligned(16))) uint16_t P[2][100][64];
  __shared__ __attribute__((aligned(16))) float S[2][64];
  const int tid = threadIdx.x;
  const int oct = tid & 7, pos = tid >> 3;
  const int ph = pos >> 3, pw = pos & 7;
  int bid = blockIdx.x;
  const int ncg = (x.C + 63) >> 6;
  const int cg = bid % ncg; bid /= ncg;
  const int tw = bid % tilesW; bid /= tilesW;
  const int th = bid % tilesH; bid /= tilesH;
  const int b = bid;
  const int h0 = th * 8, w0 = tw * 8, c0 = cg * 64 + oct * 8;
  const int ho = h0 + ph, wo = w0 + pw;
  const bool out_ok = ho < y.H && wo < y.W && c0 < x.C;
  const bool aff = pre.scale != nullptr;
  if (tid < 64) {
    const bool ok = aff && cg * 64 + tid < x.C;
    S[0][tid] = ok ? pre.scale[cg * 64 + tid] : 1.f;
    S[1][tid] = ok ? pre.shift[cg * 64 + tid] : 0.f;
  }
  __syncthreads();
  uint32_t m_a[8], m_b[8], m_c[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { m_a[e] = 0; m_b[e] = 0; m_c[e] = 0; }
  const int T_ = x.T;
  for (int tp = 0; tp <= T_; ++tp) {
    uint16_t (*buf)[64] = P[tp & 1];
    if (tp < T_) {
      for (int it = tid; it < 800; it += 512) {
        const int hp = it >> 3;                      // (it & 7) == oct
        const int hh = h0 - 1 + hp / 10, ww = w0 - 1 + hp % 10;
        if (cg * 64 + oct * 8 >= x.C) continue;
        uint4 q = make_uint4(0, 0, 0, 0);
        if ((unsigned)hh < (unsigned)x.H && (unsigned)ww < (unsigned)x.W) {
          q = *(const uint4*)((const bf16_t*)x.p + vox_off(x, b, tp, hh, ww) + cg * 64 + oct * 8);
          if (aff) {
            const float2* sp = (const float2*)&S[0][oct * 8];
            const float2* hp2 = (const float2*)&S[1][oct * 8];
            uint32_t* w4 = (uint32_t*)&q;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float2 s2 = sp[e], h2 = hp2[e];
              if (pre.relu) w4[e] = pre_relu_pair(w4[e], (f32x2_v){s2.x, s2.y}, (f32x2_v){h2.x, h2.y});
              else w4[e] = pack2bf(fmaf(__uint_as_float(w4[e] << 16), s2.x, h2.x), fmaf(__uint_as_float(w4[e] & 0xffff0000u), s2.y, h2.y));
            }
          } else if (pre.relu) {
            uint32_t* w4 = (uint32_t*)&q;
#pragma unroll
            for (int e = 0; e < 4; ++e) asm("v_pk_max_i16 %0, %1, 0" : "=v"(w4[e]) : "v"(w4[e]));
          }
          q.x = pool_code2(q.x); q.y = pool_code2(q.y); q.z = pool_code2(q.z); q.w = pool_code2(q.w);
        }
        *(uint4*)&buf[hp][oct * 8] = q;
      }
    }
    __syncthreads();     // plane tp staged; every lane finished reading the buffer staged two planes ago
#pragma unroll
    for (int e = 0; e < 8; ++e) { m_a[e] = m_b[e]; m_b[e] = m_c[e]; m_c[e] = 0; }
    if (tp < T_) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const uint4 q = *(const uint4*)&buf[(ph + kh) * 10 + pw + kw][oct * 8];
          const uint32_t ck = 8u - (uint32_t)(kh * 3 + kw);
          const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t klo = (w4[e] << 16) | ck, khi = (w4[e] & 0xffff0000u) | ck;
            m_c[2 * e] = m_c[2 * e] > klo ? m_c[2 * e] : klo;
            m_c[2 * e + 1] = m_c[2 * e + 1] > khi ? m_c[2 * e + 1] : khi;
          }
        }
    }
    const int to = tp - 1;
    if (to < 0 || !out_ok) continue;
    uint32_t ov[4];
    unsigned long long oi = 0;
    const bool va = to >= 1, vc = to + 1 < T_;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const uint32_t ka = va ? m_a[e] + 18u : 0u, kb = m_b[e] + 9u, kc = vc ? m_c[e] : 0u;
      uint32_t k = ka > kb ? ka : kb;
      k = k > kc ? k : kc;
      if (e & 1) ov[e >> 1] |= k & 0xffff0000u; else ov[e >> 1] = k >> 16;
      oi |= (unsigned long long)(26u - (k & 0xffu)) << (8 * e);
    }
    uint4 o;
    o.x = pool_decode2(ov[0]); o.y = pool_decode2(ov[1]); o.z = pool_decode2(ov[2]); o.w = pool_decode2(ov[3]);
    *(uint4*)((bf16_t*)y.p + vox_off(y, b, to, ho, wo) + c0) = o;
    if (argmax) {
      const long ovox = (((long)b * y.T + to) * y.H + ho) * y.W + wo;
      *(unsigned long long*)(argmax + ovox * y.C + c0) = oi;
    }
  }
}

extern "C" int vinet_maxpool3d(const VinetPoolDesc* d, const VinetTensor* x, VinetAffine pre, const VinetTensor* y,
                               uint8_t* argmax, void* stream) {
  VN_CHECK_ARG(d && x && y && quad_ok(*x, esize(d->dtype)) && quad_ok(*y, esize(d->dtype)) && x->C == y->C && x->B == y->B,
               "maxpool3d: bad views");
  VN_CHECK_ARG(d->kT * d->kH * d->kW <= 255 && d->kT > 0 && d->kH > 0 && d->kW > 0, "maxpool3d: window too large");
  const PoolP p = make_poolp(d);
  const bool k3s1 = d->kT == 3 && d->kH == 3 && d->kW == 3 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 1 && d->pH == 1 &&
                    d->pW == 1 && y->T == x->T && y->H == x->H && y->W == x->W;
  if (k3s1 && g_vinet_opt_pool_lds && x->T >= 2 && oct_ok(*x) && oct_ok(*y) && (!argmax || ((uintptr_t)argmax % 8) == 0) &&
      (g_vinet_opt_pool_lds >= 2 || (long)y->B * y->H * y->W * (y->C / 8) >= 65536)) {
    const int tilesH = (y->H + 7) / 8, tilesW = (y->W + 7) / 8;
    const long blocks = (long)y->B * tilesH * tilesW * ((y->C + 63) / 64);
    const bool pre_ok = !pre.scale || pre.shift;
    if (d->dtype == VINET_BF16 && g_vinet_opt_pool_pk && pre_ok && x->ld % 8 == 0 && ((uintptr_t)x->ptr % 16) == 0 && ((uintptr_t)y->ptr % 16) == 0 &&
        x->sB % 8 == 0 && y->sB % 8 == 0) {
      hipLaunchKernelGGL(maxpool_k3s1_pk_kernel, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, make_view(*x),
                         make_affine(pre), make_view(*y), argmax, tilesH, tilesW);
      return vn_launch_status("maxpool3d(k3s1 packed keys)");
    }
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_k3s1_lds_kernel<T>, dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream,
                                               make_view(*x), make_affine(pre), make_view(*y), argmax, tilesH, tilesW);)
    return vn_launch_status("maxpool3d(k3s1 lds)");
  }
  if (d->kT == 3 && d->sT == 1 && d->pT == 1 && y->T == x->T && x->T >= 2 && oct_ok(*x) && oct_ok(*y) &&
      (!argmax || ((uintptr_t)argmax % 8) == 0) &&
      (g_vinet_opt_pool_twalk >= 2 || (long)y->B * y->H * y->W * (y->C / 8) >= 65536)) {
    // (fewer columns than that cannot fill the chip while each lane walks T serially: batch-1 inference
    //  takes the one-thread-per-output kernel below)
    const long cols8 = (long)y->B * y->H * y->W * (y->C / 8);
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_tslide8_kernel<T>, dim3(ew_grid(cols8)), dim3(256), 0,
                                               (hipStream_t)stream, p, make_view(*x), make_affine(pre), make_view(*y), argmax, cols8);)
    return vn_launch_status("maxpool3d(tslide8)");
  }
  if (d->kT == 3 && d->sT == 1 && d->pT == 1 && y->T == x->T && x->T >= 2 && !(oct_ok(*x) && oct_ok(*y))) {
    const long cols = (long)y->B * y->H * y->W * (y->C / 4);
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_tslide_kernel<T>, dim3(ew_grid(cols)), dim3(256), 0,
                                               (hipStream_t)stream, p, make_view(*x), make_affine(pre), make_view(*y), argmax, cols);)
    return vn_launch_status("maxpool3d(tslide)");
  }
  if (oct_ok(*x) && oct_ok(*y) && (!argmax || ((uintptr_t)argmax % 8) == 0)) {
    const long total8 = view_voxels(*y) * (y->C / 8);
    if (d->dtype == VINET_BF16 && g_vinet_opt_pool_pk && (!pre.scale || pre.shift) && ((uintptr_t)x->ptr % 16) == 0 && ((uintptr_t)y->ptr % 16) == 0 &&
        x->sB % 8 == 0 && y->sB % 8 == 0) {
      hipLaunchKernelGGL(maxpool_fwd8_pk_kernel, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream, p, make_view(*x),
                         make_affine(pre), make_view(*y), argmax, total8);
      return vn_launch_status("maxpool3d(8, packed keys)");
    }
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_fwd8_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0,
                                               (hipStream_t)stream, p, make_view(*x), make_affine(pre), make_view(*y), argmax, total8);)
    return vn_launch_status("maxpool3d(8)");
  }
  const long total = view_voxels(*y) * (y->C / 4);
  DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0,
                                             (hipStream_t)stream, p, make_view(*x), make_affine(pre), make_view(*y), argmax, total);)
  return vn_launch_status("maxpool3d");
}

// g += v in the lanes whose code byte (byte BYTE of `word`) equals `code`: v_cmpx (SDWA byte select) -> add under EXEC -> EXEC restored
#define POOL_ADD_IF_X(g, word, BYTE, code, v, exec0)                                                                                  \
  asm volatile("v_cmpx_eq_u32_sdwa vcc, %1, %2 src0_sel:BYTE_" #BYTE " src1_sel:DWORD\n\tv_add_f32_e32 %0, %0, %3\n\ts_mov_b64 exec, %4" \
               : "+v"(g) : "v"(word), "s"(code), "v"(v), "s"(exec0) : "vcc")
#define POOL_ADD_IF(g, word, BYTE, code, v) POOL_ADD_IF_X(g, word, BYTE, code, v, exec0)
// byte e (0..7) of a 64-bit code word
#define POOL_ADD_IF8(g, lo, hi, e, code, v, ex)                                           \
  do {                                                                                      \
    const uint32_t w_ = (e) < 4 ? (lo) : (hi);                                              \
    switch ((e) & 3) {                                                                      \
      case 0: POOL_ADD_IF_X(g, w_, 0, code, v, ex); break;                                  \
      case 1: POOL_ADD_IF_X(g, w_, 1, code, v, ex); break;                                  \
      case 2: POOL_ADD_IF_X(g, w_, 2, code, v, ex); break;                                  \
      default: POOL_ADD_IF_X(g, w_, 3, code, v, ex); break;                                 \
    }                                                                                       \
  } while (0)

// backward as a gather over the (at most ceil(k/s)^3) windows covering each input voxel
template <typename T>
__global__ void maxpool_bwd_kernel(PoolP p, TView dy, const uint8_t* __restrict__ argmax, TView dx, int accumulate,
                                   long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, dx.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(dx.C / 4));
  const long vox = (long)vox_u;
  int b, t, h, w;
  decode_vox(dx, vox, b, t, h, w);
  float g[4] = {0, 0, 0, 0};
  // windows: o*s - pad <= pos <= o*s - pad + k - 1
  // (all numerators are >= 0 after the max: fast unsigned division)
  const int to1 = min((int)fdiv((uint32_t)(t + p.pT), p.dsT), dy.T - 1), ho1 = min((int)fdiv((uint32_t)(h + p.pH), p.dsH), dy.H - 1),
            wo1 = min((int)fdiv((uint32_t)(w + p.pW), p.dsW), dy.W - 1);
  const int to0 = (int)fdiv((uint32_t)max(0, t + p.pT - p.kT + p.sT), p.dsT), ho0 = (int)fdiv((uint32_t)max(0, h + p.pH - p.kH + p.sH), p.dsH),
            wo0 = (int)fdiv((uint32_t)max(0, w + p.pW - p.kW + p.sW), p.dsW);
  for (int to = to0; to <= to1; ++to) {
    const int kt = t + p.pT - to * p.sT;
    if (kt < 0 || kt >= p.kT) continue;
    for (int ho = ho0; ho <= ho1; ++ho) {
      const int kh = h + p.pH - ho * p.sH;
      if (kh < 0 || kh >= p.kH) continue;
      for (int wo = wo0; wo <= wo1; ++wo) {
        const int kw = w + p.pW - wo * p.sW;
        if (kw < 0 || kw >= p.kW) continue;
        const uint32_t tap = (uint32_t)((kt * p.kH + kh) * p.kW + kw);
        const long ovox = (((long)b * dy.T + to) * dy.H + ho) * dy.W + wo;
        const uint32_t am = *(const uint32_t*)(argmax + ovox * dy.C + q * 4);
        // a window routes its gradient to exactly one of its taps: most candidates do not match,
        // and then dy is not read at all (zero-byte test on am ^ tap-in-every-byte)
        const uint32_t xr = am ^ (tap * 0x01010101u);
        if (!((xr - 0x01010101u) & ~xr & 0x80808080u)) continue;
        const float4 d = ldq<T>((const T*)dy.p + vox_off(dy, b, to, ho, wo) + q * 4);
        if ((xr & 0xffu) == 0) g[0] += d.x;
        if ((xr & 0xff00u) == 0) g[1] += d.y;
        if ((xr & 0xff0000u) == 0) g[2] += d.z;
        if ((xr & 0xff000000u) == 0) g[3] += d.w;
      }
    }
  }
  T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + q * 4;
  if (accumulate) { const float4 o = ldq<T>(dst); g[0] += o.x; g[1] += o.y; g[2] += o.z; g[3] += o.w; }
  stq<T>(dst, make_float4(g[0], g[1], g[2], g[3]));
}

// 3x3x3 / stride 1 / pad 1 (the Inception branch-3 pools, model_utils.py:178): every input voxel
// is covered by up to 27 windows.  One thread owns 8 channels of one voxel, issues all 27
// argmax loads (8 codes = 8 bytes each) before looking at any of them, and reads dy only for
// the windows that route a gradient here.  The generic kernel walks the same 27 windows as a
// dependent load -> compare -> branch chain and is latency bound (0.5 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_k3s1_kernel(TView dy, const uint8_t* __restrict__ argmax, TView dx,
                                                               int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  const uint32_t vox_u = (uint32_t)(i / G);          // G is small and launch-invariant; one division per thread
  const int g = (int)(i - (long)vox_u * G);
  int b, t, h, w;
  decode_vox(dx, (long)vox_u, b, t, h, w);
  const int T_ = dx.T, H = dx.H, W = dx.W;
  const long am_c = (long)vox_u * dx.C + g * 8;
  const long dy_c = vox_off(dy, b, t, h, w) + g * 8;
  unsigned long long am[27];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        // window (t+1-kt, h+1-kh, w+1-kw) holds this voxel as its tap (kt,kh,kw)
        const int dt = 1 - kt, dh = 1 - kh, dw = 1 - kw;
        const bool ok = (unsigned)(t + dt) < (unsigned)T_ && (unsigned)(h + dh) < (unsigned)H && (unsigned)(w + dw) < (unsigned)W;
        const long d = ((long)(dt * H + dh) * W + dw) * (long)dx.C;
        am[(kt * 3 + kh) * 3 + kw] = ok ? *(const unsigned long long*)(argmax + am_c + d) : ~0ull;
      }
  float gr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tap = (kt * 3 + kh) * 3 + kw;
        const unsigned long long xr = am[tap] ^ (0x0101010101010101ull * (unsigned long long)tap);
        if (!((xr - 0x0101010101010101ull) & ~xr & 0x8080808080808080ull)) continue;   // no zero byte: no match
        const int dt = 1 - kt, dh = 1 - kh, dw = 1 - kw;
        const T* src = (const T*)dy.p + dy_c + ((long)(dt * H + dh) * W + dw) * (long)dy.ld;
        const float4 d0 = ldq<T>(src), d1 = ldq<T>(src + 4);
        const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (((xr >> (8 * e)) & 0xffull) == 0) gr[e] += dv[e];
      }
  T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
  if (accumulate) {
    const float4 o0 = ldq<T>(dst), o1 = ldq<T>(dst + 4);
    gr[0] += o0.x; gr[1] += o0.y; gr[2] += o0.z; gr[3] += o0.w; gr[4] += o1.x; gr[5] += o1.y; gr[6] += o1.z; gr[7] += o1.w;
  }
  stq<T>(dst, make_float4(gr[0], gr[1], gr[2], gr[3]));
  stq<T>(dst + 4, make_float4(gr[4], gr[5], gr[6], gr[7]));
}

// generic backward, 8 channels per lane (same gather as maxpool_bwd_kernel)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd8_kernel(PoolP p, TView dy, const uint8_t* __restrict__ argmax, TView dx, int accumulate,
                                                           long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  const uint32_t vox_u = (uint32_t)(i / G);
  const int g = (int)(i - (long)vox_u * G);
  int b, t, h, w;
  decode_vox(dx, (long)vox_u, b, t, h, w);
  float gr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int to1 = min((int)fdiv((uint32_t)(t + p.pT), p.dsT), dy.T - 1), ho1 = min((int)fdiv((uint32_t)(h + p.pH), p.dsH), dy.H - 1),
            wo1 = min((int)fdiv((uint32_t)(w + p.pW), p.dsW), dy.W - 1);
  const int to0 = (int)fdiv((uint32_t)max(0, t + p.pT - p.kT + p.sT), p.dsT), ho0 = (int)fdiv((uint32_t)max(0, h + p.pH - p.kH + p.sH), p.dsH),
            wo0 = (int)fdiv((uint32_t)max(0, w + p.pW - p.kW + p.sW), p.dsW);
  for (int to = to0; to <= to1; ++to) {
    const int kt = t + p.pT - to * p.sT;
    if (kt < 0 || kt >= p.kT) continue;
    for (int ho = ho0; ho <= ho1; ++ho) {
      const int kh = h + p.pH - ho * p.sH;
      if (kh < 0 || kh >= p.kH) continue;
      for (int wo = wo0; wo <= wo1; ++wo) {
        const int kw = w + p.pW - wo * p.sW;
        if (kw < 0 || kw >= p.kW) continue;
        const unsigned long long tap = (unsigned long long)((kt * p.kH + kh) * p.kW + kw);
        const long ovox = (((long)b * dy.T + to) * dy.H + ho) * dy.W + wo;
        const unsigned long long am = *(const unsigned long long*)(argmax + ovox * dy.C + g * 8);
        const unsigned long long xr = am ^ (tap * 0x0101010101010101ull);
        if (!((xr - 0x0101010101010101ull) & ~xr & 0x8080808080808080ull)) continue;
        float dv[8];
        ld8<T>((const T*)dy.p + vox_off(dy, b, to, ho, wo) + g * 8, dv);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (((xr >> (8 * e)) & 0xffull) == 0) gr[e] += dv[e];
      }
    }
  }
  T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
  if (accumulate) {
    float o[8];
    ld8<T>(dst, o);
#pragma unroll
    for (int e = 0; e < 8; ++e) gr[e] += o[e];
  }
  st8<T>(dst, gr);
}

// 1x3x3 / s(1,2,2) / p(0,1,1) backward (the two big spatial pools, model.py:696,700): one lane owns the 2 x 2 input
// block {2ho, 2ho+1} x {2wo, 2wo+1} x 8 channels.  Only the four windows (ho..ho+1, wo..wo+1) reach it -- (ho,wo) all
// four inputs, (ho,wo+1) and (ho+1,wo) two each, (ho+1,wo+1) one -- so 4 argmax words and at most 4 dy rows serve 4
// outputs, and the voxel decode and window arithmetic are paid once per 64 bytes written instead of once per 16:
// the generic gather is bound by exactly that integer work (2.1 TB/s of tensors on the 112 x 192 pool).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_k133s2_kernel(TView dy, const uint8_t* __restrict__ argmax, TView dx, int accumulate,
                                                                 int HB, int WB, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  long r = i / G;
  const int g = (int)(i - r * G);
  const int wb = (int)(r % WB); r /= WB;
  const int hb = (int)(r % HB); r /= HB;
  const int t = (int)(r % dx.T);
  const int b = (int)(r / dx.T);
  const int h0 = 2 * hb, w0 = 2 * wb;
  // windows q = dh*2 + dw at (hb + dh, wb + dw)
  unsigned long long am[4];
  bool wok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ho = hb + (q >> 1), wo = wb + (q & 1);
    wok[q] = ho < dy.H && wo < dy.W;
    const long ovox = (((long)b * dy.T + t) * dy.H + (wok[q] ? ho : 0)) * dy.W + (wok[q] ? wo : 0);
    am[q] = wok[q] ? *(const unsigned long long*)(argmax + ovox * dy.C + g * 8) : ~0ull;
  }
  float dv[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // taps of window q that land in the block: rows kh in {1,2} (dh = 0) or {0} (dh = 1); same for columns
    bool any = false;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const bool in = ((q >> 1) ? kh == 0 : kh >= 1) && ((q & 1) ? kw == 0 : kw >= 1);
        if (!in) continue;
        const unsigned long long x = am[q] ^ ((unsigned long long)(kh * 3 + kw) * 0x0101010101010101ull);
        any |= ((x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull) != 0;
      }
    if (any) ld8<T>((const T*)dy.p + vox_off(dy, b, t, hb + (q >> 1), wb + (q & 1)) + g * 8, dv[q]);
    else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dv[q][e] = 0.f;
    }
  }
#pragma unroll
  for (int ih = 0; ih < 2; ++ih)
#pragma unroll
    for (int iw = 0; iw < 2; ++iw) {
      const int h = h0 + ih, w = w0 + iw;
      if (h >= dx.H || w >= dx.W) continue;
      float gr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      unsigned long long ex;                                  // (the lanes that are inside the image: POOL_ADD_IF restores this mask)
      asm volatile("s_mov_b64 %0, exec" : "=s"(ex));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int dh = q >> 1, dw = q & 1;
        // input (h, w) as tap (kh, kw) of window (hb + dh, wb + dw): kh = h + 1 - 2*(hb + dh) = ih + 1 - 2*dh
        const int kh = ih + 1 - 2 * dh, kw = iw + 1 - 2 * dw;
        if (kh < 0 || kw < 0) continue;                       // (compile-time: the window does not reach this input)
        const uint32_t alo = (uint32_t)am[q], ahi = (uint32_t)(am[q] >> 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) POOL_ADD_IF8(gr[e], alo, ahi, e, (uint32_t)(kh * 3 + kw), dv[q][e], ex);
      }
      T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
      if (accumulate) {
        float o[8];
        ld8<T>(dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) gr[e] += o[e];
      }
      st8<T>(dst, gr);
    }
}

// 3x3x3 / s2 / p1 backward (the pool between the 28 x 48 and 14 x 24 stages, model.py:705): one lane owns the 2 x 2 x 2 input block
// {2a, 2a+1} x {2b, 2b+1} x {2c, 2c+1} x 8 channels.  Per dimension block voxel i (0 / 1) is tap i + 1 of window a and, for i = 1,
// tap 0 of window a + 1: eight windows reach the block through 27 (window, voxel) pairs, each with ONE fixed tap code -- so a pair
// costs a byte compare and a predicated add per channel, with no code decode and no division (the generic gather walks up to 8
// windows per voxel with their index arithmetic: 1.4 TB/s of tensors on the 480-channel pool).
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_k3s2_kernel(TView dy, const uint8_t* __restrict__ argmax, TView dx, int accumulate,
                                                               int TB, int HB, int WB, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  long r = i / G;
  const int g = (int)(i - r * G);
  const int wb = (int)(r % WB); r /= WB;
  const int hb = (int)(r % HB); r /= HB;
  const int tb = (int)(r % TB);
  const int b = (int)(r / TB);
  // windows w = (dt * 2 + dh) * 2 + dw at (tb + dt, hb + dh, wb + dw)
  unsigned long long am[8];
  float dv[8][8];
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const int to = tb + (w >> 2), ho = hb + ((w >> 1) & 1), wo = wb + (w & 1);
    const bool ok = to < dy.T && ho < dy.H && wo < dy.W;
    am[w] = ~0ull;
    if (ok) {
      const long ovox = (((long)b * dy.T + to) * dy.H + ho) * dy.W + wo;
      am[w] = *(const unsigned long long*)(argmax + ovox * dy.C + g * 8);
      ld8<T>((const T*)dy.p + vox_off(dy, b, to, ho, wo) + g * 8, dv[w]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) dv[w][e] = 0.f;
    }
  }
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const int it = v >> 2, ih = (v >> 1) & 1, iw = v & 1;
    const int t = 2 * tb + it, h = 2 * hb + ih, w_ = 2 * wb + iw;
    if (t >= dx.T || h >= dx.H || w_ >= dx.W) continue;
    float gr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ex;
    asm volatile("s_mov_b64 %0, exec" : "=s"(ex));
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const int dt = w >> 2, dh = (w >> 1) & 1, dw = w & 1;
      if (dt > it || dh > ih || dw > iw) continue;            // (compile time: the window does not reach this voxel)
      const int kt = dt ? 0 : it + 1, kh = dh ? 0 : ih + 1, kw = dw ? 0 : iw + 1;
      const uint32_t alo = (uint32_t)am[w], ahi = (uint32_t)(am[w] >> 32);
#pragma unroll
      for (int e = 0; e < 8; ++e) POOL_ADD_IF8(gr[e], alo, ahi, e, (uint32_t)((kt * 3 + kh) * 3 + kw), dv[w][e], ex);
    }
    T* dst = (T*)dx.p + vox_off(dx, b, t, h, w_) + g * 8;
    if (accumulate) {
      float o[8];
      ld8<T>(dst, o);
#pragma unroll
      for (int e = 0; e < 8; ++e) gr[e] += o[e];
    }
    st8<T>(dst, gr);
  }
}

// 3x3x3 / s1 / p1 backward, T-walking form: one lane owns an input column (b, h, w, 8 channels) and walks the
// output planes; the argmax word of each of the 9 in-plane neighbour windows is read ONCE per plane and tested
// against the three temporal taps it could route to, accumulating into three named accumulators (inputs
// to-1, to, to+1).  9 argmax reads per voxel instead of 27.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_k3s1_twalk_kernel(TView dy, const uint8_t* __restrict__ argmax, TView dx,
                                                                     int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  const uint32_t col = (uint32_t)(i / G);
  const int g = (int)(i - (long)col * G);
  const uint32_t r1 = fdiv(col, dx.dW);
  const int w = (int)(col - r1 * (uint32_t)dx.W);
  const uint32_t r2 = fdiv(r1, dx.dH);
  const int h = (int)(r1 - r2 * (uint32_t)dx.H);
  const int b = (int)r2;
  const int T_ = dx.T, H = dx.H, W = dx.W;
  float g_m[8], g_0[8], g_p[8];      // gradients of inputs to-1, to, to+1 while output plane `to` is processed
#pragma unroll
  for (int e = 0; e < 8; ++e) { g_m[e] = 0.f; g_0[e] = 0.f; g_p[e] = 0.f; }
  for (int to = 0; to <= T_; ++to) {
    if (to < T_) {
      unsigned long long am[9];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ho = h + 1 - kh, wo = w + 1 - kw;     // the window that holds (h, w) as its in-plane tap (kh, kw)
          const bool ok = (unsigned)ho < (unsigned)H && (unsigned)wo < (unsigned)W;
          const long ovox = (((long)b * T_ + to) * H + (ok ? ho : 0)) * W + (ok ? wo : 0);
          am[kh * 3 + kw] = ok ? *(const unsigned long long*)(argmax + ovox * dx.C + g * 8) : ~0ull;
        }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const unsigned long long s = (unsigned long long)(kh * 3 + kw) * 0x0101010101010101ull;
          // codes kt*9 + s for kt = 0 (input to-1), 1 (input to), 2 (input to+1)
          const unsigned long long x0 = am[kh * 3 + kw] ^ s, x1 = am[kh * 3 + kw] ^ (s + 9ull * 0x0101010101010101ull),
                                   x2 = am[kh * 3 + kw] ^ (s + 18ull * 0x0101010101010101ull);
          const unsigned long long z0 = (x0 - 0x0101010101010101ull) & ~x0, z1 = (x1 - 0x0101010101010101ull) & ~x1,
                                   z2 = (x2 - 0x0101010101010101ull) & ~x2;
          if (!((z0 | z1 | z2) & 0x8080808080808080ull)) continue;
          float dv[8];
          ld8<T>((const T*)dy.p + vox_off(dy, b, to, h + 1 - kh, w + 1 - kw) + g * 8, dv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (((x0 >> (8 * e)) & 0xffull) == 0) g_m[e] += dv[e];
            if (((x1 >> (8 * e)) & 0xffull) == 0) g_0[e] += dv[e];
            if (((x2 >> (8 * e)) & 0xffull) == 0) g_p[e] += dv[e];
          }
        }
    }
    const int t = to - 1;          // input plane to-1 has now seen all of its windows (planes to-2, to-1, to)
    if (t >= 0) {
      T* dst = (T*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
      if (accumulate) {
        float o[8];
        ld8<T>(dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) g_m[e] += o[e];
      }
      st8<T>(dst, g_m);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { g_m[e] = g_0[e]; g_0[e] = g_p[e]; g_p[e] = 0.f; }
  }
}

// Same walk for bf16 with every load of a plane issued up front: the 9 argmax words AND the 9 gradient vectors of the
// in-plane neighbour windows are fetched unconditionally (they are L1 / L2 hits for 8 of 9 lanes), 18 independent loads
// per lane and plane, so a lane pays one memory latency per plane instead of two dependent ones
// (the conditional gradient loads of the form above left the kernel latency-bound at 1.4 TB/s).  Routing is branch-free:
// byte code - (kh*3+kw) is 0 / 9 / 18 for the temporal taps to-1 / to / to+1.
__global__ __launch_bounds__(256) void maxpool_bwd_k3s1_tw2_kernel(TView dy, const uint8_t* __restrict__ argmax, TView dx,
                                                                   int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int G = dx.C >> 3;
  const uint32_t col = (uint32_t)(i / G);
  const int g = (int)(i - (long)col * G);
  const uint32_t r1 = fdiv(col, dx.dW);
  const int w = (int)(col - r1 * (uint32_t)dx.W);
  const uint32_t r2 = fdiv(r1, dx.dH);
  const int h = (int)(r1 - r2 * (uint32_t)dx.H);
  const int b = (int)r2;
  const int T_ = dx.T, H = dx.H, W = dx.W;
  // lane-relative addresses of the 9 windows (the lane's own voxel where the window does not exist: any valid address)
  const uint8_t* amp = argmax + ((((long)b * T_) * H + h) * W + w) * (long)dx.C + g * 8;
  const unsigned short* dyp = (const unsigned short*)dy.p + vox_off(dy, b, 0, h, w) + g * 8;
  const long am_plane = (long)H * W * dx.C, dy_plane = (long)H * W * dy.ld;
  uint32_t okmask = 0;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ho = h + 1 - kh, wo = w + 1 - kw;
      okmask |= (((unsigned)ho < (unsigned)H && (unsigned)wo < (unsigned)W) ? 1u : 0u) << (kh * 3 + kw);
    }
  float g_m[8], g_0[8], g_p[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { g_m[e] = 0.f; g_0[e] = 0.f; g_p[e] = 0.f; }
  for (int to = 0; to <= T_; ++to) {
    if (to < T_) {
      unsigned long long am[9];
      uint4 dv[9];
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const int dvox = (1 - s / 3) * W + (1 - s % 3);
        const bool ok = (okmask >> s) & 1u;
        am[s] = *(const unsigned long long*)(amp + (ok ? dvox * dx.C : 0));
        dv[s] = *(const uint4*)(dyp + (ok ? dvox * dy.ld : 0));
      }
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const unsigned long long a = ((okmask >> s) & 1u) ? am[s] : ~0ull;
        const uint32_t alo = (uint32_t)a, ahi = (uint32_t)(a >> 32);
        const uint32_t q[4] = {dv[s].x, dv[s].y, dv[s].z, dv[s].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t c = (((e < 4) ? alo : ahi) >> (8 * (e & 3))) & 0xffu;
          const int dlt = (int)c - s;
          const float v = (e & 1) ? __uint_as_float(q[e >> 1] & 0xffff0000u) : __uint_as_float(q[e >> 1] << 16);
          g_m[e] += (dlt == 0) ? v : 0.f;
          g_0[e] += (dlt == 9) ? v : 0.f;
          g_p[e] += (dlt == 18) ? v : 0.f;
        }
      }
      amp += am_plane;
      dyp += dy_plane;
    }
    const int t = to - 1;
    if (t >= 0) {
      unsigned short* dst = (unsigned short*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
      if (accumulate) {
        float o[8];
        ld8<unsigned short>(dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) g_m[e] += o[e];
      }
      st8<unsigned short>(dst, g_m);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { g_m[e] = g_0[e]; g_0[e] = g_p[e]; g_p[e] = 0.f; }
  }
}

// the same walk with the routing done on the EXEC mask: 1127 -> 836 VALU instructions per plane and lane, 3.01 -> 2.37 ms on the
// 256-channel pool at 28 x 48 (the default; pool_twalk = 4 selects the form above)
__global__ __launch_bounds__(256) void maxpool_bwd_k3s1_tw3_kernel(TView dy, const uint8_t* __restrict__ argmax, TView dx,
                                                                   int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  unsigned long long exec0;
  asm volatile("s_mov_b64 %0, exec" : "=s"(exec0));
  const int G = dx.C >> 3;
  const uint32_t col = (uint32_t)(i / G);
  const int g = (int)(i - (long)col * G);
  const uint32_t r1 = fdiv(col, dx.dW);
  const int w = (int)(col - r1 * (uint32_t)dx.W);
  const uint32_t r2 = fdiv(r1, dx.dH);
  const int h = (int)(r1 - r2 * (uint32_t)dx.H);
  const int b = (int)r2;
  const int T_ = dx.T, H = dx.H, W = dx.W;
  // lane-relative addresses of the 9 windows (the lane's own voxel where the window does not exist: any valid address)
  const uint8_t* amp = argmax + ((((long)b * T_) * H + h) * W + w) * (long)dx.C + g * 8;
  const unsigned short* dyp = (const unsigned short*)dy.p + vox_off(dy, b, 0, h, w) + g * 8;
  const long am_plane = (long)H * W * dx.C, dy_plane = (long)H * W * dy.ld;
  uint32_t okmask = 0;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int ho = h + 1 - kh, wo = w + 1 - kw;
      okmask |= (((unsigned)ho < (unsigned)H && (unsigned)wo < (unsigned)W) ? 1u : 0u) << (kh * 3 + kw);
    }
  float g_m[8], g_0[8], g_p[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { g_m[e] = 0.f; g_0[e] = 0.f; g_p[e] = 0.f; }
  for (int to = 0; to <= T_; ++to) {
    if (to < T_) {
      unsigned long long am[9];
      uint4 dv[9];
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const int dvox = (1 - s / 3) * W + (1 - s % 3);
        const bool ok = (okmask >> s) & 1u;
        am[s] = *(const unsigned long long*)(amp + (ok ? dvox * dx.C : 0));
        dv[s] = *(const uint4*)(dyp + (ok ? dvox * dy.ld : 0));
      }
#pragma unroll
      for (int s = 0; s < 9; ++s) {
        const unsigned long long a = ((okmask >> s) & 1u) ? am[s] : ~0ull;
        const uint32_t alo = (uint32_t)a, ahi = (uint32_t)(a >> 32);
        const uint32_t q[4] = {dv[s].x, dv[s].y, dv[s].z, dv[s].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t word = (e < 4) ? alo : ahi;
          const float v = (e & 1) ? __uint_as_float(q[e >> 1] & 0xffff0000u) : __uint_as_float(q[e >> 1] << 16);
          // code byte == tap code -> EXEC, add under EXEC, EXEC back: 2 VALU + 1 SALU per (window, channel, temporal tap) instead
          // of compare + select + add (+ the byte extraction)
          switch (e & 3) {
            case 0: POOL_ADD_IF(g_m[e], word, 0, (uint32_t)s, v); POOL_ADD_IF(g_0[e], word, 0, (uint32_t)(s + 9), v); POOL_ADD_IF(g_p[e], word, 0, (uint32_t)(s + 18), v); break;
            case 1: POOL_ADD_IF(g_m[e], word, 1, (uint32_t)s, v); POOL_ADD_IF(g_0[e], word, 1, (uint32_t)(s + 9), v); POOL_ADD_IF(g_p[e], word, 1, (uint32_t)(s + 18), v); break;
            case 2: POOL_ADD_IF(g_m[e], word, 2, (uint32_t)s, v); POOL_ADD_IF(g_0[e], word, 2, (uint32_t)(s + 9), v); POOL_ADD_IF(g_p[e], word, 2, (uint32_t)(s + 18), v); break;
            default: POOL_ADD_IF(g_m[e], word, 3, (uint32_t)s, v); POOL_ADD_IF(g_0[e], word, 3, (uint32_t)(s + 9), v); POOL_ADD_IF(g_p[e], word, 3, (uint32_t)(s + 18), v); break;
          }
        }
      }
      amp += am_plane;
      dyp += dy_plane;
    }
    const int t = to - 1;
    if (t >= 0) {
      unsigned short* dst = (unsigned short*)dx.p + vox_off(dx, b, t, h, w) + g * 8;
      if (accumulate) {
        float o[8];
        ld8<unsigned short>(dst, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) g_m[e] += o[e];
      }
      st8<unsigned short>(dst, g_m);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { g_m[e] = g_0[e]; g_0[e] = g_p[e]; g_p[e] = 0.f; }
  }
}

extern "C" int vinet_maxpool3d_bwd(const VinetPoolDesc* d, const VinetTensor* dy, const uint8_t* argmax,
                                   const VinetTensor* dx, int32_t accumulate, void* stream) {
  VN_CHECK_ARG(d && dy && dx && argmax && quad_ok(*dy, esize(d->dtype)) && quad_ok(*dx, esize(d->dtype)) &&
                   dx->C == dy->C && dx->B == dy->B, "maxpool3d_bwd: bad views");
  const PoolP p = make_poolp(d);
  const bool k3s1 = d->kT == 3 && d->kH == 3 && d->kW == 3 && d->sT == 1 && d->sH == 1 && d->sW == 1 && d->pT == 1 && d->pH == 1 &&
                    d->pW == 1 && dy->T == dx->T && dy->H == dx->H && dy->W == dx->W;
  if (k3s1 && dx->C % 8 == 0 && dx->ld % 8 == 0 && dy->ld % 8 == 0 && dx->sB % 8 == 0 && dy->sB % 8 == 0 &&
      ((uintptr_t)dx->ptr % 16) == 0 && ((uintptr_t)dy->ptr % 16) == 0 && ((uintptr_t)argmax % 8) == 0) {
    const long cols8 = (long)dx->B * dx->H * dx->W * (dx->C / 8);
    if (g_vinet_opt_pool_twalk != 3 && d->dtype == VINET_BF16 && (g_vinet_opt_pool_twalk >= 2 || (g_vinet_opt_pool_twalk && cols8 >= 65536)) &&
        (long)dx->H * dx->W * dx->C < (1l << 30) && (long)dx->H * dx->W * dy->ld < (1l << 30)) {   // 3: the conditional-load form (A/B)
      if (g_vinet_opt_pool_twalk != 4)      // (4: the compare / select / add form, for A/B and tests)
        hipLaunchKernelGGL(maxpool_bwd_k3s1_tw3_kernel, dim3(ew_grid(cols8)), dim3(256), 0, (hipStream_t)stream, make_view(*dy), argmax,
                           make_view(*dx), accumulate, cols8);
      else
        hipLaunchKernelGGL(maxpool_bwd_k3s1_tw2_kernel, dim3(ew_grid(cols8)), dim3(256), 0, (hipStream_t)stream, make_view(*dy), argmax,
                           make_view(*dx), accumulate, cols8);
      return vn_launch_status("maxpool3d_bwd(k3s1 tw2)");
    }
    if (g_vinet_opt_pool_twalk >= 2 || (g_vinet_opt_pool_twalk && cols8 >= 65536)) {   // 2: force (tests)   // enough columns to fill the chip
      DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_bwd_k3s1_twalk_kernel<T>, dim3(ew_grid(cols8)), dim3(256), 0,
                                                 (hipStream_t)stream, make_view(*dy), argmax, make_view(*dx), accumulate, cols8);)
      return vn_launch_status("maxpool3d_bwd(k3s1 twalk)");
    }
    const long total8 = view_voxels(*dx) * (dx->C / 8);
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_bwd_k3s1_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream,
                                               make_view(*dy), argmax, make_view(*dx), accumulate, total8);)
    return vn_launch_status("maxpool3d_bwd(k3s1)");
  }
  if (oct_ok(*dx) && oct_ok(*dy) && ((uintptr_t)argmax % 8) == 0) {
    const long total8 = view_voxels(*dx) * (dx->C / 8);
    if (g_vinet_opt_pool_blk && d->kT == 1 && d->sT == 1 && d->pT == 0 && d->kH == 3 && d->kW == 3 && d->sH == 2 && d->sW == 2 && d->pH == 1 &&
        d->pW == 1 && dy->T == dx->T && dy->H == (dx->H - 1) / 2 + 1 && dy->W == (dx->W - 1) / 2 + 1) {
      const int HB = (dx->H + 1) / 2, WB = (dx->W + 1) / 2;
      const long nthr = (long)dx->B * dx->T * HB * WB * (dx->C / 8);
      DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_bwd_k133s2_kernel<T>, dim3(ew_grid(nthr)), dim3(256), 0, (hipStream_t)stream,
                                                 make_view(*dy), argmax, make_view(*dx), accumulate, HB, WB, nthr);)
      return vn_launch_status("maxpool3d_bwd(k133s2)");
    }
    if (g_vinet_opt_pool_blk && d->kT == 3 && d->kH == 3 && d->kW == 3 && d->sT == 2 && d->sH == 2 && d->sW == 2 && d->pT == 1 && d->pH == 1 &&
        d->pW == 1 && dy->T == (dx->T - 1) / 2 + 1 && dy->H == (dx->H - 1) / 2 + 1 && dy->W == (dx->W - 1) / 2 + 1) {
      const int TB = (dx->T + 1) / 2, HB = (dx->H + 1) / 2, WB = (dx->W + 1) / 2;
      const long nthr = (long)dx->B * TB * HB * WB * (dx->C / 8);
      DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_bwd_k3s2_kernel<T>, dim3(ew_grid(nthr)), dim3(256), 0, (hipStream_t)stream,
                                                 make_view(*dy), argmax, make_view(*dx), accumulate, TB, HB, WB, nthr);)
      return vn_launch_status("maxpool3d_bwd(k3s2)");
    }
    DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_bwd8_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream,
                                               p, make_view(*dy), argmax, make_view(*dx), accumulate, total8);)
    return vn_launch_status("maxpool3d_bwd8");
  }
  const long total = view_voxels(*dx) * (dx->C / 4);
  DISPATCH_T(d->dtype, T, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(ew_grid(total)), dim3(256), 0,
                                             (hipStream_t)stream, p, make_view(*dy), argmax, make_view(*dx), accumulate, total);)
  return vn_launch_status("maxpool3d_bwd");
}
