// Layout and packing kernels: conv weights -> packed bf16 / fp32 tiles (single and multi-tensor), packed fp32 weight
// gradients -> torch layout, NCDHW <-> channels-last import / export, copy with the pending affine (concat, materialise),
// the skinny (<= 8 output channels) pointwise weight gradient, SoundNet's 1-D unfold, fp32 fill.
#include "elementwise.h"

// ============================================================================
// weight packing
// ============================================================================
// Destination formats.  VINET_F32 / VINET_BF16: element L of the logical [slice][row][Kp] array at out[L].  VINET_F32S (fp32
// tensors, split-bf16 arithmetic): every 32-wide K chunk of a row becomes 128 bytes = [32 bf16 hi | 32 bf16 lo] with hi = bf16(v),
// lo = bf16(v - hi) -- the same bytes per row as fp32 -- and the chunk's channels PERMUTED so that MFMA lane group q (K
// positions 8q .. 8q+7) holds channels {4q .. 4q+3, 16+4q .. 16+4q+3}: the activation side of the split kernels reads its fp32
// rows in 16-byte pieces (4 channels), piece q and piece q + 4 per lane (conv_dma3.h, conv_igemm.h), and both sides must agree
// on which channel sits in which K position.
template <int DT> VN_DEV void pack_store(void* out, long L, float v) {
  if constexpr (DT == VINET_F32) ((float*)out)[L] = v;
  else if constexpr (DT == VINET_BF16) ((bf16_t*)out)[L] = f2bf(v);
  else {
    const int c = (int)(L & 31);
    const int pos = c < 16 ? ((c >> 2) * 8 + (c & 3)) : (((c - 16) >> 2) * 8 + 4 + (c & 3));
    bf16_t* o = (bf16_t*)out + (L >> 5) * 64 + pos;
    const bf16_t hi = f2bf(v);
    o[0] = hi;
    o[32] = f2bf(v - bf2f(hi));
  }
}
#define DISPATCH_PACK(dt, DT, ...)                                   \
  if ((dt) == VINET_F32) { constexpr int DT = VINET_F32; __VA_ARGS__ }      \
  else if ((dt) == VINET_F32S) { constexpr int DT = VINET_F32S; __VA_ARGS__ } \
  else { constexpr int DT = VINET_BF16; __VA_ARGS__ }

template <int DT>
__global__ void pack_weights_kernel(const float* __restrict__ w, int N, int Cin, int ntaps, int transpose, int stem,
                                    int rows, int Kp, int nslices, void* __restrict__ out) {
  const long total = (long)nslices * rows * Kp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kp);
    const int r = (int)((i / Kp) % rows);
    const int s = (int)(i / ((long)Kp * rows));
    float v = 0.f;
    if (stem) {  // out[kh][n][kw*4+c], w[n][c][kh*7+kw]
      const int kw = k >> 2, c = k & 3;
      if (kw < 7 && c < Cin) v = w[((long)r * Cin + c) * ntaps + s * 7 + kw];
    } else if (!transpose) {  // out[t][n][c]
      if (k < Cin) v = w[((long)r * Cin + k) * ntaps + s];
    } else {  // out[t][c][n]
      if (k < N) v = w[((long)k * Cin + r) * ntaps + s];
    }
    pack_store<DT>(out, i, v);
  }
}

extern "C" int vinet_pack_weights(const float* w, int32_t N, int32_t Cin, int32_t ntaps, int32_t transpose,
                                  int32_t stem, int32_t dtype, void* out, void* stream) {
  VN_CHECK_ARG(w && out && N > 0 && Cin > 0 && ntaps > 0, "pack_weights: bad arguments");
  VN_CHECK_ARG(dtype == VINET_F32 || dtype == VINET_BF16 || dtype == VINET_F32S, "pack_weights: bad dtype %d", dtype);
  int rows, Kp, nslices;
  if (stem) {
    VN_CHECK_ARG(ntaps == 49 && Cin <= 4 && !transpose, "pack_weights stem: need 1x7x7, Cin<=4");
    rows = N; Kp = 32; nslices = 7;
  } else if (!transpose) { rows = N; Kp = (Cin + 31) / 32 * 32; nslices = ntaps; }
  else { rows = Cin; Kp = (N + 31) / 32 * 32; nslices = ntaps; }
  const long total = (long)nslices * rows * Kp;
  int grid = ew_grid(total); if (grid > 8192) grid = 8192;
  DISPATCH_PACK(dtype, DT, hipLaunchKernelGGL(pack_weights_kernel<DT>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, N,
                                             Cin, ntaps, transpose, stem, rows, Kp, nslices, out);)
  return vn_launch_status("pack_weights");
}

// Multi-tensor form: every weight of the model is re-packed after each optimizer step, 170 launches of a few
// microseconds each when done one by one.  `table` (device memory) holds 8 int64 per job:
//   { w pointer, out pointer, N, Cin, ntaps, transpose | stem << 1, first output index (prefix sum), ld | col << 32 }
// ld != 0 (transposed jobs only): rows of the destination are `ld` elements apart and this job owns columns
// [col, col + N) of them -- several convs that share an input, packed side by side along K for ONE dgrad.
// plus one trailing row whose prefix field is the total; one thread per output element, job by binary search.
template <int DT>
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const long* __restrict__ table, int njobs, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int lo = 0, hi = njobs;               // last job with prefix <= i
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (table[mid * 8 + 6] <= i) lo = mid; else hi = mid;
    }
    const long* J = table + lo * 8;
    const float* w = (const float*)J[0];
    void* out = (void*)J[1];
    const int N = (int)J[2], Cin = (int)J[3], ntaps = (int)J[4];
    const int transpose = (int)(J[5] & 1), stem = (int)((J[5] >> 1) & 1);
    const long e = i - J[6];
    const int rows = (stem || !transpose) ? N : Cin;
    const int Kp = stem ? 32 : ((transpose ? N : Cin) + 31) / 32 * 32;
    const int k = (int)(e % Kp);
    const int r = (int)((e / Kp) % rows);
    const int sl = (int)(e / ((long)Kp * rows));
    float v = 0.f;
    if (stem) {
      const int kw = k >> 2, c = k & 3;
      if (kw < 7 && c < Cin) v = w[((long)r * Cin + c) * ntaps + sl * 7 + kw];
    } else if (!transpose) {
      if (k < Cin) v = w[((long)r * Cin + k) * ntaps + sl];
    } else {
      if (k < N) v = w[((long)k * Cin + r) * ntaps + sl];
      const long ld = J[7] & 0xffffffffl;
      if (ld) {          // side-by-side destination: only the job's own columns are written
        if (k < N) pack_store<DT>(out, ((long)sl * rows + r) * ld + (J[7] >> 32) + k, v);
        continue;
      }
    }
    pack_store<DT>(out, e, v);
  }
}


// ---- LDS-tiled forms of the two multi-tensor kernels (round 4) -----------------------------------------------------------
// The element-wise kernels above walk the PACKED order: the torch side ([n][c][tap], taps fastest) is then touched with a
// stride of `ntaps` floats, every 128-byte line of a 27- or 45-tap weight is fetched once per tap (0.73 ms for 62 M packed
// elements, 0.63 ms on the way back: both are per-step costs that do not shrink with the batch).  Here a workgroup owns a tile
// of (n, c) pairs with ALL their taps: the torch side moves as contiguous rows of TC * ntaps floats, the packed side as
// runs of TC (forward / unpack: 8 n x 32 c) or TR (transposed: 32 n x 8 c) consecutive elements per (tap, row), and the
// permutation happens in LDS (row stride odd: conflict-free in both directions).  Work units are tiles; the job of a unit is
// found by binary search in a prefix table every workgroup builds in LDS from the 8-word job rows (<= 1024 jobs; more fall
// back to the element-wise kernels).  The stem's 7 x 7 jobs (one per model) run element-wise, 256 packed elements per unit.
constexpr int PT_MAXJOBS = 1024, PT_LDS_FLOATS = 8448;     // 256 pairs x 32 taps + the odd row strides

struct PtShape { int TRn, TCc, tilesR, tilesC, rs; long units; };
// tile shape of a job: (rows of n) x (columns of c); `cols` = extent tiled along c (Kp forward, Cin transposed), `rowsn` along n
__device__ __forceinline__ PtShape pt_shape(int N, int Cin, int ntaps, bool transpose, bool stem, bool has_ld) {
  PtShape h;
  if (stem) { h.TRn = h.TCc = h.tilesR = h.tilesC = h.rs = 0; h.units = (7L * N * 32 + 255) / 256; return h; }
  if (ntaps > 256) {       // (no such layer in the nets: a tile of all taps would not fit) element-wise, 256 packed elements per unit
    const long kp = ((transpose ? N : Cin) + 31) / 32 * 32;
    h.TRn = h.TCc = h.tilesR = h.tilesC = h.rs = 0;
    h.units = ((long)ntaps * (transpose ? Cin : N) * kp + 255) / 256;
    return h;
  }
  const int pairs = ntaps <= 32 ? 256 : (ntaps <= 64 ? 128 : (ntaps <= 128 ? 64 : 32));
  if (!transpose) { h.TCc = 32; h.TRn = pairs / 32; }
  else { h.TCc = 8; h.TRn = pairs / 8; }
  const int Kp = ((transpose ? N : Cin) + 31) / 32 * 32;
  const int extC = transpose ? Cin : Kp;                       // forward: the padding columns are written (zeros) too
  const int extN = transpose ? (has_ld ? N : Kp) : N;          // transposed without ld: the padding columns n >= N too
  h.tilesC = (extC + h.TCc - 1) / h.TCc;
  h.tilesR = (extN + h.TRn - 1) / h.TRn;
  h.rs = h.TCc * ntaps + 1;
  h.units = (long)h.tilesC * h.tilesR;
  return h;
}

// prefix[j] = units of jobs 0 .. j-1 (prefix[njobs] = total) in LDS; returns the total.  `unpack`: rows are
// {dw, grad, N, Cin, ntaps, stem, ...}, else {w, out, N, Cin, ntaps, transpose | stem << 1, prefix, ld | col << 32}
__device__ __forceinline__ long pt_build_prefix(const long* __restrict__ table, int njobs, bool unpack, long* prefix, long* seg) {
  const int tid = threadIdx.x;
  const int per = (njobs + 255) / 256;
  long sum = 0;
  for (int q = 0; q < per; ++q) {
    const int j = tid * per + q;
    if (j < njobs) {
      const long* J = table + j * 8;
      const bool tr = !unpack && (J[5] & 1), st = unpack ? (J[5] != 0) : ((J[5] >> 1) & 1);
      const bool ld = !unpack && (J[7] & 0xffffffffl) != 0;
      prefix[j] = sum;
      sum += pt_shape((int)J[2], (int)J[3], (int)J[4], tr, st, ld).units;
    }
  }
  seg[tid] = sum;
  __syncthreads();
  if (tid == 0) {
    long run = 0;
    for (int t = 0; t < 256; ++t) { const long v = seg[t]; seg[t] = run; run += v; }
    prefix[njobs] = run;
  }
  __syncthreads();
  const long base = seg[tid];
  for (int q = 0; q < per; ++q) {
    const int j = tid * per + q;
    if (j < njobs) prefix[j] += base;
  }
  __syncthreads();
  return prefix[njobs];
}
__device__ __forceinline__ int pt_find(const long* prefix, int njobs, long u) {
  int lo = 0, hi = njobs;                 // last job with prefix <= u
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= u) lo = mid; else hi = mid;
  }
  return lo;
}

template <int DT>
__global__ __launch_bounds__(256) void pack_weights_tiled_kernel(const long* __restrict__ table, int njobs) {
  __shared__ long prefix[PT_MAXJOBS + 1];
  __shared__ long seg[256];
  __shared__ float tile[PT_LDS_FLOATS];
  const int tid = threadIdx.x;
  const long units = pt_build_prefix(table, njobs, false, prefix, seg);
  for (long u = blockIdx.x; u < units; u += gridDim.x) {
    const int j = pt_find(prefix, njobs, u);
    const long* J = table + j * 8;
    const float* __restrict__ w = (const float*)J[0];
    void* out = (void*)J[1];
    const int N = (int)J[2], Cin = (int)J[3], ntaps = (int)J[4];
    const bool transpose = J[5] & 1, stem = (J[5] >> 1) & 1;
    const long ld = J[7] & 0xffffffffl;
    const int col = (int)(J[7] >> 32);
    const long lu = u - prefix[j];
    if (stem) {        // out[kh][n][kw*4 + c] = w[n][c][kh*7 + kw]: element-wise, 256 packed elements per unit
      const long e = lu * 256 + tid;
      if (e < 7L * N * 32) {
        const int k = (int)(e & 31), r = (int)((e >> 5) % N), sl = (int)(e / (32L * N));
        const int kw = k >> 2, c = k & 3;
        pack_store<DT>(out, e, (kw < 7 && c < Cin) ? w[((long)r * Cin + c) * ntaps + sl * 7 + kw] : 0.f);
      }
      continue;
    }
    const PtShape h = pt_shape(N, Cin, ntaps, transpose, false, ld != 0);
    const int Kp = ((transpose ? N : Cin) + 31) / 32 * 32;
    if (h.TRn == 0) {
      const int rows = transpose ? Cin : N;
      const long e = lu * 256 + tid;
      if (e < (long)ntaps * rows * Kp) {
        const int k = (int)(e % Kp), r = (int)((e / Kp) % rows), sl = (int)(e / ((long)Kp * rows));
        if (!transpose) pack_store<DT>(out, e, k < Cin ? w[((long)r * Cin + k) * ntaps + sl] : 0.f);
        else if (ld) { if (k < N) pack_store<DT>(out, ((long)sl * rows + r) * ld + col + k, w[((long)k * Cin + r) * ntaps + sl]); }
        else pack_store<DT>(out, e, k < N ? w[((long)k * Cin + r) * ntaps + sl] : 0.f);
      }
      continue;
    }
    const int tc = (int)(lu % h.tilesC), tr = (int)(lu / h.tilesC);
    const int n0 = tr * h.TRn, c0 = tc * h.TCc;
    const int rowlen = h.TCc * ntaps;
    int live = Cin - c0; live = live < 0 ? 0 : (live > h.TCc ? h.TCc : live);
    const int livelen = live * ntaps;
    __syncthreads();                                   // the previous unit's tile has been read
    for (int idx = tid; idx < h.TRn * rowlen; idx += 256) {
      const int rr = idx / rowlen, off = idx - rr * rowlen;
      const int n = n0 + rr;
      tile[rr * h.rs + off] = (n < N && off < livelen) ? w[((long)n * Cin + c0) * ntaps + off] : 0.f;
    }
    __syncthreads();
    const int total = h.TRn * h.TCc * ntaps;
    if (!transpose) {              // out[t][n][c]: runs of TCc consecutive c
      for (int idx = tid; idx < total; idx += 256) {
        const int cc = idx % h.TCc, q = idx / h.TCc, rr = q % h.TRn, t = q / h.TRn;
        const int n = n0 + rr;
        if (n < N) pack_store<DT>(out, ((long)t * N + n) * Kp + c0 + cc, tile[rr * h.rs + cc * ntaps + t]);
      }
    } else {                       // out[t][c][n]: runs of TRn consecutive n
      for (int idx = tid; idx < total; idx += 256) {
        const int rr = idx % h.TRn, q = idx / h.TRn, cc = q % h.TCc, t = q / h.TCc;
        const int n = n0 + rr, c = c0 + cc;
        if (c >= Cin) continue;
        const float v = tile[rr * h.rs + cc * ntaps + t];
        if (ld) { if (n < N) pack_store<DT>(out, ((long)t * Cin + c) * ld + col + n, v); }
        else if (n < Kp) pack_store<DT>(out, ((long)t * Cin + c) * Kp + n, v);
      }
    }
  }
}

__global__ __launch_bounds__(256) void unpack_wgrad_tiled_kernel(const long* __restrict__ table, int njobs, int flags) {
  __shared__ long prefix[PT_MAXJOBS + 1];
  __shared__ long seg[256];
  __shared__ float tile[PT_LDS_FLOATS];
  const int tid = threadIdx.x;
  const int accumulate = flags & 1, clear = flags & 2;
  const long units = pt_build_prefix(table, njobs, true, prefix, seg);
  for (long u = blockIdx.x; u < units; u += gridDim.x) {
    const int j = pt_find(prefix, njobs, u);
    const long* J = table + j * 8;
    float* __restrict__ dw = (float*)J[0];
    float* __restrict__ grad = (float*)J[1];
    const int N = (int)J[2], Cin = (int)J[3], ntaps = (int)J[4], stem = (int)J[5];
    const long lu = u - prefix[j];
    if (stem) {
      const long e = lu * 256 + tid;
      if (e < 7L * N * 32) {
        const int k = (int)(e & 31), n = (int)((e >> 5) % N), sl = (int)(e / (32L * N));
        const int kw = k >> 2, c = k & 3;
        if (kw < 7 && c < Cin) {
          const long dst = ((long)n * Cin + c) * ntaps + sl * 7 + kw;
          const float v = dw[e];
          grad[dst] = accumulate ? grad[dst] + v : v;
        }
        if (clear) dw[e] = 0.f;
      }
      continue;
    }
    const PtShape h = pt_shape(N, Cin, ntaps, false, false, false);
    const int Kp = (Cin + 31) / 32 * 32;
    if (h.TRn == 0) {
      const long e = lu * 256 + tid;
      if (e < (long)ntaps * N * Kp) {
        const int k = (int)(e % Kp), n = (int)((e / Kp) % N), sl = (int)(e / ((long)Kp * N));
        if (k < Cin) {
          const long dst = ((long)n * Cin + k) * ntaps + sl;
          const float v = dw[e];
          grad[dst] = accumulate ? grad[dst] + v : v;
        }
        if (clear) dw[e] = 0.f;
      }
      continue;
    }
    const int tc = (int)(lu % h.tilesC), tr = (int)(lu / h.tilesC);
    const int n0 = tr * h.TRn, c0 = tc * h.TCc;
    const int rowlen = h.TCc * ntaps;
    int live = Cin - c0; live = live < 0 ? 0 : (live > h.TCc ? h.TCc : live);
    const int livelen = live * ntaps;
    const int total = h.TRn * h.TCc * ntaps;
    __syncthreads();
    for (int idx = tid; idx < total; idx += 256) {      // dw[t][n][c]: runs of TCc consecutive c (the padding columns too)
      const int cc = idx % h.TCc, q = idx / h.TCc, rr = q % h.TRn, t = q / h.TRn;
      const int n = n0 + rr;
      if (n < N) {
        const long src = ((long)t * N + n) * Kp + c0 + cc;
        tile[rr * h.rs + cc * ntaps + t] = dw[src];
        if (clear) dw[src] = 0.f;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < h.TRn * rowlen; idx += 256) {
      const int rr = idx / rowlen, off = idx - rr * rowlen;
      const int n = n0 + rr;
      if (n < N && off < livelen) {
        const long dst = ((long)n * Cin + c0) * ntaps + off;
        const float v = tile[rr * h.rs + off];
        grad[dst] = accumulate ? grad[dst] + v : v;
      }
    }
  }
}
int g_vinet_opt_pack_tiled = 1;     // LDS-tiled multi-tensor pack / unpack (0 = the element-wise kernels)

extern "C" int vinet_pack_weights_multi(const int64_t* table, int32_t njobs, int64_t total, int32_t dtype, void* stream) {
  VN_CHECK_ARG(table && njobs > 0 && total > 0, "pack_weights_multi: bad arguments");
  if (g_vinet_opt_pack_tiled && njobs <= PT_MAXJOBS) {
    // units are not known on the host (the table lives in device memory): a grid of 8 workgroups per CU walks them
    DISPATCH_PACK(dtype, DT, hipLaunchKernelGGL(pack_weights_tiled_kernel<DT>, dim3(2048), dim3(256), 0, (hipStream_t)stream,
                                               (const long*)table, njobs);)
    return vn_launch_status("pack_weights_tiled");
  }
  int grid = ew_grid(total); if (grid > 16384) grid = 16384;
  DISPATCH_PACK(dtype, DT, hipLaunchKernelGGL(pack_weights_multi_kernel<DT>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                                             (const long*)table, njobs, (long)total);)
  return vn_launch_status("pack_weights_multi");
}

__global__ void unpack_wgrad_kernel(float* __restrict__ dw, int N, int Cin, int ntaps, int stem, int Kp,
                                    int flags, float* __restrict__ grad) {
  const int accumulate = flags & 1, clear = flags & 2;
  const long total = (long)N * Cin * ntaps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % ntaps);
    const int c = (int)((i / ntaps) % Cin);
    const int n = (int)(i / ((long)ntaps * Cin));
    long src;
    if (stem) { const int kh = t / 7, kw = t % 7; src = ((long)kh * N + n) * 32 + kw * 4 + c; }
    else src = ((long)t * N + n) * Kp + c;
    const float v = dw[src];
    grad[i] = accumulate ? grad[i] + v : v;
    if (clear) dw[src] = 0.f;      // every packed element is read by exactly one thread: hand the buffer back zeroed
  }
  if (clear) {
    // columns no torch element maps to (channel padding; split-K atomics may have touched them)
    const int nsl = stem ? 7 : ntaps, kp = stem ? 32 : Kp;
    const long ptotal = (long)nsl * N * kp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ptotal; i += (long)gridDim.x * blockDim.x) {
      const int col = (int)(i % kp);
      const bool valid = stem ? (col < 28 && (col & 3) < Cin) : (col < Cin);
      if (!valid) dw[i] = 0.f;
    }
  }
}

extern "C" int vinet_unpack_wgrad(float* dw, int32_t N, int32_t Cin, int32_t ntaps, int32_t stem,
                                  int32_t flags, float* grad, void* stream) {
  VN_CHECK_ARG(dw && grad && N > 0 && Cin > 0 && ntaps > 0, "unpack_wgrad: bad arguments");
  const int Kp = stem ? 32 : (Cin + 31) / 32 * 32;
  const long total = (long)N * Cin * ntaps;
  int grid = ew_grid(total); if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dw, N, Cin, ntaps, stem, Kp,
                     flags, grad);
  return vn_launch_status("unpack_wgrad");
}

// every weight gradient of a backward pass in ONE launch (84 vinet_unpack_wgrad launches per ViNet-32 step otherwise): the
// threads walk the PACKED elements of all jobs, so transfer and hand-back-zeroed are one loop
__global__ __launch_bounds__(256) void unpack_wgrad_multi_kernel(const long* __restrict__ table, int njobs, long total, int flags) {
  const int accumulate = flags & 1, clear = flags & 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int lo = 0, hi = njobs;               // last job with prefix <= i
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (table[mid * 8 + 6] <= i) lo = mid; else hi = mid;
    }
    const long* J = table + lo * 8;
    float* dw = (float*)J[0];
    float* grad = (float*)J[1];
    const int N = (int)J[2], Cin = (int)J[3], ntaps = (int)J[4], stem = (int)J[5];
    const long e = i - J[6];
    const int Kp = stem ? 32 : (Cin + 31) / 32 * 32;
    const int k = (int)(e % Kp);
    const int n = (int)((e / Kp) % N);
    const int sl = (int)(e / ((long)Kp * N));
    long dst = -1;
    if (stem) {
      const int kw = k >> 2, c = k & 3;
      if (kw < 7 && c < Cin) dst = ((long)n * Cin + c) * ntaps + sl * 7 + kw;
    } else if (k < Cin) {
      dst = ((long)n * Cin + k) * ntaps + sl;
    }
    if (dst >= 0) {
      const float v = dw[e];
      grad[dst] = accumulate ? grad[dst] + v : v;
    }
    if (clear) dw[e] = 0.f;
  }
}

extern "C" int vinet_unpack_wgrad_multi(const int64_t* table, int32_t njobs, int64_t total, int32_t flags, void* stream) {
  VN_CHECK_ARG(table && njobs > 0 && total > 0, "unpack_wgrad_multi: bad arguments");
  if (g_vinet_opt_pack_tiled && njobs <= PT_MAXJOBS) {
    hipLaunchKernelGGL(unpack_wgrad_tiled_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const long*)table, njobs, flags);
    return vn_launch_status("unpack_wgrad_tiled");
  }
  int grid = ew_grid(total); if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(unpack_wgrad_multi_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const long*)table, njobs, (long)total, flags);
  return vn_launch_status("unpack_wgrad_multi");
}

// ============================================================================
// NCDHW <-> channels-last
// ============================================================================
template <typename T>
__global__ void import_ncdhw_kernel(const float* __restrict__ src, long sb, long sc, long st, long sh, long sw, int C,
                                    TView dst, long nvox) {
  const long vox = (long)blockIdx.x * blockDim.x + threadIdx.x;   // voxels fastest: coalesced planar reads
  if (vox >= nvox) return;
  const int q = blockIdx.y;
  int b, t, h, w;
  decode_vox(dst, vox, b, t, h, w);
  const float* s = src + b * sb + t * st + h * sh + w * sw;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { const int c = q * 4 + e; v[e] = c < C ? s[c * sc] : 0.f; }
  stq<T>((T*)dst.p + vox_off(dst, b, t, h, w) + q * 4, make_float4(v[0], v[1], v[2], v[3]));
}

extern "C" int vinet_import_ncdhw(const float* src, int64_t sb, int64_t sc, int64_t st, int64_t sh, int64_t sw,
                                  int32_t C, const VinetTensor* dst, int32_t dst_dtype, void* stream) {
  VN_CHECK_ARG(src && dst && quad_ok(*dst, esize(dst_dtype)) && C > 0 && C <= dst->C, "import_ncdhw: bad arguments");
  const long nvox = view_voxels(*dst);
  DISPATCH_T(dst_dtype, T, hipLaunchKernelGGL(import_ncdhw_kernel<T>, dim3(ew_grid(nvox), dst->C / 4), dim3(256), 0,
                                              (hipStream_t)stream, src, sb, sc, st, sh, sw, C, make_view(*dst), nvox);)
  return vn_launch_status("import_ncdhw");
}

// import into a zero-padded buffer: dst voxel (h, w) <- src(h - pad_top, w - pad_left), zero outside
template <typename T>
__global__ void import_pad_kernel(const float* __restrict__ src, long sb, long sc, long st, long sh, long sw, int C,
                                  int Hs, int Ws, int pad_top, int pad_left, TView dst, long nvox) {
  const long vox = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vox >= nvox) return;
  const int q = blockIdx.y;
  int b, t, h, w;
  decode_vox(dst, vox, b, t, h, w);
  const int hs = h - pad_top, ws = w - pad_left;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if ((unsigned)hs < (unsigned)Hs && (unsigned)ws < (unsigned)Ws) {
    const float* s = src + b * sb + t * st + hs * sh + ws * sw;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int c = q * 4 + e; if (c < C) v[e] = s[c * sc]; }
  }
  stq<T>((T*)dst.p + vox_off(dst, b, t, h, w) + q * 4, make_float4(v[0], v[1], v[2], v[3]));
}

extern "C" int vinet_import_ncdhw_pad(const float* src, int64_t sb, int64_t sc, int64_t st, int64_t sh, int64_t sw,
                                      int32_t C, int32_t Hs, int32_t Ws, int32_t pad_top, int32_t pad_left,
                                      const VinetTensor* dst, int32_t dst_dtype, void* stream) {
  VN_CHECK_ARG(src && dst && quad_ok(*dst, esize(dst_dtype)) && C > 0 && C <= dst->C && Hs > 0 && Ws > 0 &&
                   pad_top >= 0 && pad_left >= 0 && pad_top + Hs <= dst->H && pad_left + Ws <= dst->W,
               "import_ncdhw_pad: bad arguments");
  const long nvox = view_voxels(*dst);
  DISPATCH_T(dst_dtype, T, hipLaunchKernelGGL(import_pad_kernel<T>, dim3(ew_grid(nvox), dst->C / 4), dim3(256), 0,
                                              (hipStream_t)stream, src, sb, sc, st, sh, sw, C, Hs, Ws, pad_top, pad_left,
                                              make_view(*dst), nvox);)
  return vn_launch_status("import_ncdhw_pad");
}

template <typename T>
__global__ void export_ncdhw_kernel(TView src, Affine pre, float* __restrict__ dst, long sb, long sc, long st, long sh,
                                    long sw, int accumulate, long nvox) {
  const long vox = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (vox >= nvox) return;
  const int q = blockIdx.y;
  int b, t, h, w;
  decode_vox(src, vox, b, t, h, w);
  float4 v = ldq<T>((const T*)src.p + vox_off(src, b, t, h, w) + q * 4);
  v = affine4(v, pre, q * 4);
  float* d = dst + b * sb + t * st + h * sh + w * sw + (long)(q * 4) * sc;
  const float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) d[e * sc] = accumulate ? d[e * sc] + o[e] : o[e];
}

extern "C" int vinet_export_ncdhw(const VinetTensor* src, int32_t src_dtype, VinetAffine pre, float* dst, int64_t sb,
                                  int64_t sc, int64_t st, int64_t sh, int64_t sw, int32_t accumulate, void* stream) {
  VN_CHECK_ARG(src && dst && quad_ok(*src, esize(src_dtype)), "export_ncdhw: bad arguments");
  const long nvox = view_voxels(*src);
  DISPATCH_T(src_dtype, T, hipLaunchKernelGGL(export_ncdhw_kernel<T>, dim3(ew_grid(nvox), src->C / 4), dim3(256), 0,
                                              (hipStream_t)stream, make_view(*src), make_affine(pre), dst, sb, sc, st,
                                              sh, sw, accumulate, nvox);)
  return vn_launch_status("export_ncdhw");
}

template <typename TI, typename TO>
__global__ void copy_affine_kernel(TView src, Affine pre, TView dst, int accumulate, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, src.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(src.C / 4));
  const long vox = (long)vox_u;
  float4 v = ldq<TI>((const TI*)src.p + vox_lin(src, vox) + q * 4);
  v = affine4(v, pre, q * 4);
  TO* d = (TO*)dst.p + vox_lin(dst, vox) + q * 4;
  if (accumulate) { const float4 o = ldq<TO>(d); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
  stq<TO>(d, v);
}

// 8-channel form (structure of bn_bwd_apply8_kernel): a lane keeps the scale / shift of its 8 channels in registers
// and streams voxels, 16-byte loads and stores, 4 voxels in flight -- the quad kernel above pays a voxel decode and
// two coefficient loads per 8 bytes.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void copy_affine8_kernel(TView src, Affine pre, TView dst, int accumulate, long nvox, long vb) {
  const int G = src.C / 8;
  const int Gb = G < 256 ? G : 256;
  const int R = 256 / Gb;
  const int r = threadIdx.x / Gb;
  const int g0 = threadIdx.x % Gb;
  if (r >= R) return;
  const long v0 = (long)blockIdx.x * vb;
  long v1 = v0 + vb; if (v1 > nvox) v1 = nvox;
  for (int g = g0; g < G; g += Gb) {
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = pre.scale ? pre.scale[g * 8 + e] : 1.f; sh[e] = pre.scale ? pre.shift[g * 8 + e] : 0.f; }
    constexpr int U = 4;
    for (long vq = v0 + r; vq < v1; vq += (long)R * U) {
      float xv[U][8], ov[U][8];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long v = vq + (long)u * R;
        ok[u] = v < v1;
        if (ok[u]) {
          ld8<TI>((const TI*)src.p + vox_lin(src, v) + g * 8, xv[u]);
          if (accumulate) ld8<TO>((const TO*)dst.p + vox_lin(dst, v) + g * 8, ov[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!ok[u]) continue;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = pre.scale ? fmaf(xv[u][e], sc[e], sh[e]) : xv[u][e];
          if (pre.relu) t = fmaxf(t, 0.f);
          o[e] = accumulate ? t + ov[u][e] : t;
        }
        st8<TO>((TO*)dst.p + vox_lin(dst, vq + (long)u * R) + g * 8, o);
      }
    }
  }
}

extern "C" int vinet_copy_affine(const VinetTensor* src, int32_t src_dtype, VinetAffine pre, const VinetTensor* dst,
                                 int32_t dst_dtype, int32_t accumulate, void* stream) {
  VN_CHECK_ARG(src && dst && quad_ok(*src, esize(src_dtype)) && quad_ok(*dst, esize(dst_dtype)) && same_dims(*src, *dst),
               "copy_affine: bad views");
  const long total = view_voxels(*src) * (src->C / 4);
  hipStream_t s = (hipStream_t)stream;
  const TView sv = make_view(*src), dv = make_view(*dst);
  const Affine a = make_affine(pre);
  if (src_dtype == VINET_BF16 && dst_dtype == VINET_BF16 && oct_ok(*src) && oct_ok(*dst) && view_voxels(*src) >= 65536) {
    const long nvox = view_voxels(*src);
    const int G = src->C / 8, R = 256 / (G < 256 ? G : 256);
    long vb = R * 16;
    while ((nvox + vb - 1) / vb > 16384) vb *= 2;
    hipLaunchKernelGGL((copy_affine8_kernel<bf16_t, bf16_t>), dim3((unsigned)((nvox + vb - 1) / vb)), dim3(256), 0, s, sv, a, dv, accumulate, nvox, vb);
    return vn_launch_status("copy_affine8");
  }
  const dim3 g(ew_grid(total)), blk(256);
  if (src_dtype == VINET_F32 && dst_dtype == VINET_F32) hipLaunchKernelGGL((copy_affine_kernel<float, float>), g, blk, 0, s, sv, a, dv, accumulate, total);
  else if (src_dtype == VINET_F32) hipLaunchKernelGGL((copy_affine_kernel<float, bf16_t>), g, blk, 0, s, sv, a, dv, accumulate, total);
  else if (dst_dtype == VINET_F32) hipLaunchKernelGGL((copy_affine_kernel<bf16_t, float>), g, blk, 0, s, sv, a, dv, accumulate, total);
  else hipLaunchKernelGGL((copy_affine_kernel<bf16_t, bf16_t>), g, blk, 0, s, sv, a, dv, accumulate, total);
  return vn_launch_status("copy_affine");
}


// fp32 view (+ pending affine) -> hi and lo bf16 planes, hi = bf16(v), lo = bf16(v - hi): the operands of the bf16 kernels when
// they serve the split-bf16 form (the weight gradient of VINET_F32S runs as three bf16 launches hi*hi + lo*hi + hi*lo that
// accumulate into one fp32 workspace: engine.py)
__global__ void split_bf16_kernel(TView src, Affine pre, TView hi, TView lo, long total) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const uint32_t vox_u = fdiv((uint32_t)i, src.dQ);
  const int q = (int)((uint32_t)i - vox_u * (uint32_t)(src.C / 4));
  const long vox = (long)vox_u;
  float4 v = ldq<float>((const float*)src.p + vox_lin(src, vox) + q * 4);
  v = affine4(v, pre, q * 4);
  const uint32_t h01 = pack2bf(v.x, v.y), h23 = pack2bf(v.z, v.w);
  const uint32_t l01 = pack2bf(v.x - __uint_as_float(h01 << 16), v.y - __uint_as_float(h01 & 0xffff0000u));
  const uint32_t l23 = pack2bf(v.z - __uint_as_float(h23 << 16), v.w - __uint_as_float(h23 & 0xffff0000u));
  *(uint2*)((bf16_t*)hi.p + vox_lin(hi, vox) + q * 4) = make_uint2(h01, h23);
  *(uint2*)((bf16_t*)lo.p + vox_lin(lo, vox) + q * 4) = make_uint2(l01, l23);
}
extern "C" int vinet_split_bf16(const VinetTensor* src, VinetAffine pre, const VinetTensor* hi, const VinetTensor* lo, void* stream) {
  VN_CHECK_ARG(src && hi && lo && quad_ok(*src, 4) && quad_ok(*hi, 2) && quad_ok(*lo, 2) && same_dims(*src, *hi) && same_dims(*src, *lo),
               "split_bf16: bad views");
  const long total = view_voxels(*src) * (src->C / 4);
  hipLaunchKernelGGL(split_bf16_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, make_view(*src), make_affine(pre),
                     make_view(*hi), make_view(*lo), total);
  return vn_launch_status("split_bf16");
}

// ---- "skinny" weight gradient: a pointwise conv with at most 8 output channels (ViNet's 32 -> 1 head, model.py:279: the
// channel-padded dy has 8 columns, 7 of them exactly zero) over tens of millions of voxels is a per-channel reduction, not a
// GEMM: dw[n][c] = sum_v dy[v][n] * x[v][c].  The 64 x 64 MFMA tile spent 0.9 ms (0.8 TF/s) on it; here a lane owns 8 input
// channels of a strided share of the voxels with an 8 x 8 block of fp32 accumulators, lanes of equal channel group meet by
// wave shuffles, waves in LDS, and a workgroup adds its 8 x Cin block to dw with one atomic per element.
__global__ __launch_bounds__(256) void wgrad_skinny_kernel(TView x, TView dy, long nvox, int Kp, float* __restrict__ dw) {
  __shared__ float red[4][8][64];                  // [wave][group][n * 8 + e]
  const int G = x.C >> 3;                          // 1, 2, 4 or 8 groups of 8 input channels
  const int tid = threadIdx.x, g = tid % G, r = tid / G, R = 256 / G;
  float acc[8][8];
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[n][e] = 0.f;
  for (long v = (long)blockIdx.x * R + r; v < nvox; v += (long)gridDim.x * R) {
    float xv[8], gv[8];
    ld8<bf16_t>((const bf16_t*)x.p + vox_lin(x, v) + g * 8, xv);
    ld8<bf16_t>((const bf16_t*)dy.p + vox_lin(dy, v), gv);
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[n][e] = fmaf(gv[n], xv[e], acc[n][e]);
  }
  // lanes g, g + G, g + 2G, ... of a wave hold the same channel group
#pragma unroll
  for (int n = 0; n < 8; ++n)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = acc[n][e];
      for (int o = 32; o >= G; o >>= 1) a += __shfl_xor(a, o);
      acc[n][e] = a;
    }
  const int lane = tid & 63, wave = tid >> 6;
  if (lane < G) {
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave][lane][n * 8 + e] = acc[n][e];
  }
  __syncthreads();
  for (int i = tid; i < G * 64; i += 256) {
    const int gg = i >> 6, ne = i & 63, n = ne >> 3, e = ne & 7;
    const float a = red[0][gg][ne] + red[1][gg][ne] + red[2][gg][ne] + red[3][gg][ne];
    atomicAdd(dw + (long)n * Kp + gg * 8 + e, a);
  }
}

int g_vinet_opt_wgrad_skinny = 1;   // 0 = off, 2 = every eligible shape (tests)

bool vinet_wgrad_use_skinny(const VinetWgradDesc* d) {
  if (!g_vinet_opt_wgrad_skinny || d->dtype != VINET_BF16 || d->mode != VINET_CONV_GENERIC || d->ntaps != 1 || d->pre.scale || d->pre.relu ||
      d->bnb_z)
    return false;
  const int Cin = d->x.C;
  const bool shape = d->dy.C == 8 && (Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64) && d->Kp >= Cin && d->sT == 1 && d->sH == 1 && d->sW == 1 &&
                     d->x.T == d->dy.T && d->x.H == d->dy.H && d->x.W == d->dy.W && oct_ok(d->x) && oct_ok(d->dy);
  if (!shape) return false;
  return g_vinet_opt_wgrad_skinny >= 2 || (long)d->dy.B * d->dy.T * d->dy.H * d->dy.W >= (1L << 20);
}

int vinet_launch_wgrad_skinny(const VinetWgradDesc* d, hipStream_t s) {
  // taps: a pointwise conv has one tap, (0, 0, 0, slice 0) -- nothing to read from the device-side table
  const long nvox = view_voxels(d->dy);
  const int R = 256 / (d->x.C / 8);
  long blocks = (nvox + R - 1) / R;
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(wgrad_skinny_kernel, dim3((unsigned)blocks), dim3(256), 0, s, make_view(d->x), make_view(d->dy), nvox, d->Kp, d->dw);
  return vn_launch_status("wgrad_skinny");
}

// ---- 1-D unfold (im2col along T) of a single-channel signal: SoundNet's first conv (model.py:751: Conv2d(1, 16, (64, 1),
// stride 2, padding 32)) has one input channel and 64 taps -- as a (k,1,1) conv its K axis would be 64 taps x 32 padded
// channels with one real column in 32.  Unfolded, y[b, m, c] = x[b, s*m - p + c, channel 0] (zero outside), it is a pointwise
// conv with 64 input channels: 0.58 GB written once per step instead of a 32x padded K loop in forward and weight gradient.
template <typename T>
__global__ __launch_bounds__(256) void unfold1d_kernel(TView x, TView y, int stride, int pad, long total8) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total8) return;
  const int G = y.C >> 3;
  const long row = i / G;                          // b * To + m
  const int g = (int)(i - row * G);
  const long b = row / y.T;
  const int m = (int)(row - b * y.T);
  const T* src = (const T*)x.p + b * x.sB;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const long pos = (long)m * stride - pad + g * 8 + e;
    v[e] = (pos >= 0 && pos < x.T) ? load1<T>(src + pos * x.ld) : 0.f;
  }
  st8<T>((T*)y.p + b * y.sB + (long)m * y.ld + g * 8, v);
}

extern "C" int vinet_unfold1d(const VinetTensor* x, const VinetTensor* y, int32_t dtype, int32_t stride, int32_t pad, void* stream) {
  VN_CHECK_ARG(x && y && (dtype == VINET_F32 || dtype == VINET_BF16) && x->ptr && y->ptr && x->B == y->B && x->H == 1 && x->W == 1 &&
                   y->H == 1 && y->W == 1 && x->C >= 1 && y->C % 8 == 0 && y->ld % 8 == 0 && y->sB % 8 == 0 &&
                   ((uintptr_t)y->ptr % 16) == 0 && stride >= 1 && pad >= 0 && y->T == (x->T + 2 * pad - y->C) / stride + 1,
               "unfold1d: bad views");
  const long total8 = (long)y->B * y->T * (y->C / 8);
  DISPATCH_T(dtype, T, hipLaunchKernelGGL(unfold1d_kernel<T>, dim3(ew_grid(total8)), dim3(256), 0, (hipStream_t)stream, make_view(*x),
                                          make_view(*y), stride, pad, total8);)
  return vn_launch_status("unfold1d");
}

__global__ void fill_f32_kernel(float* p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
extern "C" int vinet_fill_f32(float* p, int64_t n, float value, void* stream) {
  VN_CHECK_ARG(p && n >= 0, "fill_f32: bad arguments");
  if (n == 0) return 0;
  int grid = ew_grid(n); if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, (long)n, value);
  return vn_launch_status("fill_f32");
}

// Schedule-stress tool (tests / tools only): ONE wave that idles on its stream for ~`cycles` shader clocks.  Put in front of a
// kernel it delays everything behind it on that stream without touching memory: the way to move a cross-stream schedule around
// in EAGER mode and see whether results depend on it (tools/dbg_defer.py; the weight-gradient stream's ordering tests).
__global__ void debug_spin_kernel(long cycles) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long)(__builtin_amdgcn_s_memtime() - t0) < cycles) __builtin_amdgcn_s_sleep(32);
}
extern "C" int vinet_debug_spin(int64_t cycles, void* stream) {
  VN_CHECK_ARG(cycles >= 0 && cycles <= (1L << 33), "debug_spin: cycles out of range (at most ~3.5 s)");
  hipLaunchKernelGGL(debug_spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long)cycles);
  return vn_launch_status("debug_spin");
}
