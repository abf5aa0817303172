// fp32 heads and gallery matching on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
//   frmap_linear_f32        nn.Linear (+ folded BatchNorm1d) (+ReLU)   face_models.py:32-33,75,467-468,488,678
//   frmap_l2_normalize_f32  F.normalize(p=2, dim=1, eps)               face_models.py:179,525,590
//   frmap_match_top1        compare_faces' arg-min over the gallery    app.py:58-63
//   frmap_cosine_logits     class-centre cosine logits + arg-max       hyperparameter_tuning.py:1038-1046
//   frmap_arcmargin_eval    ArcMarginProduct.forward, eval mode        face_models.py:351-429
//
// All four GEMM-shaped ops share one NT kernel  C[b][n] = sum_k A[b][k] * W[n][k]  (both operands
// K-contiguous, like nn.Linear): 256 threads, tile 64 rows x 128 cols, K staged 32 at a time in LDS
// (pitch 33 floats -> conflict-free column reads), wave w owns columns [32w, 32w+32) and both
// 32-row halves.  fp32 in, fp32 accumulate: bitwise an fmaf chain, so arg-min / arg-max decisions
// are taken on true fp32 scores.
#include "frmap_common.h"
#include <type_traits>

enum { MODE_LINEAR = 0, MODE_COS = 1, MODE_ARC = 2, MODE_DIST = 3 };

struct GemmEpi {
  // LINEAR
  const float* scale;
  const float* shift;
  int relu;
  // COS / ARC
  const float* inv_a;  // [B] 1/max(||a||,eps)
  const float* inv_w;  // [N]
  float s;
  float m;
  int easy;
  const int64_t* label;
  unsigned int* minmax_key;  // [2] ordered-uint keys (max, min) or null
  unsigned long long* argkey;  // [B] packed (score, idx) keys or null
  // DIST
  const float* stat_a;  // [B][2] = (sum x^2, sum x)
  const float* stat_w;  // [N][2]
  MatchRec* recs;       // [4 * gridDim.x][B] candidate records, slot = 32 gallery rows (one wave's columns)
  float* out;           // [B][N] or null
};

static inline int waves_blocks(int rows) { return (rows + 3) / 4; }

__device__ __forceinline__ unsigned int f32_ordered(float f) {
  const unsigned int b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b ^ 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(unsigned int k) {
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// min (or max) of `v` over the 32 lanes of this lane's half-wave, and the LOWEST lane of the half holding it.
// Four DPP steps inside each 16-lane row (quad xor 1, xor 2, half-mirror, mirror), one cross-row exchange,
// then a ballot + find-first: ~10 instructions per reduced row where a 64-bit (value, index) butterfly took ~30.
template <bool MAX>
__device__ __forceinline__ float half_wave_best(float v, int lk, int& first_li) {
  auto comb = [](float a, float b) { return MAX ? fmaxf(a, b) : fminf(a, b); };
  auto dpp = [](float x, auto ctrl) {
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, decltype(ctrl)::value, 0xF, 0xF, false));
  };
  float m = v;
  m = comb(m, dpp(m, std::integral_constant<int, 0xB1>{}));   // quad_perm [1,0,3,2]
  m = comb(m, dpp(m, std::integral_constant<int, 0x4E>{}));   // quad_perm [2,3,0,1]
  m = comb(m, dpp(m, std::integral_constant<int, 0x141>{}));  // row_half_mirror
  m = comb(m, dpp(m, std::integral_constant<int, 0x140>{}));  // row_mirror
  m = comb(m, __shfl_xor(m, 16, 64));
  const unsigned long long hit = __ballot(v == m);
  const unsigned half = lk ? (unsigned)(hit >> 32) : (unsigned)hit;
  first_li = __ffs(half) - 1;  // >= 0: the lane(s) that contributed m are in the mask
  return m;
}

// plain minimum over the 32 lanes of this lane's half-wave (same DPP ladder, no index)
__device__ __forceinline__ float half_wave_min(float v) {
  auto dpp = [](float x, auto ctrl) {
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, decltype(ctrl)::value, 0xF, 0xF, false));
  };
  float m = v;
  m = fminf(m, dpp(m, std::integral_constant<int, 0xB1>{}));
  m = fminf(m, dpp(m, std::integral_constant<int, 0x4E>{}));
  m = fminf(m, dpp(m, std::integral_constant<int, 0x141>{}));
  m = fminf(m, dpp(m, std::integral_constant<int, 0x140>{}));
  return fminf(m, __shfl_xor(m, 16, 64));
}

template <int MODE>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                          int B, int N, int K, GemmEpi ep) {
  constexpr int BMr = 64, BNc = 128, KC = 32, PITCH = KC + 1;
  __shared__ float As[BMr * PITCH];
  __shared__ float Ws[BNc * PITCH];
  __shared__ unsigned long long s_key[4 * BMr];  // arg-min / arg-max keys per (wave, row) before the one atomic per row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.y * BMr, n0 = blockIdx.x * BNc;
  const int li = lane & 31, lk = lane >> 5;

  f32x16_t acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  const int srow = tid >> 3, skq = (tid & 7) * 4;  // staging: 32 rows x 8 float4 per pass
  for (int k0 = 0; k0 < K; k0 += KC) {
    float4 va[2], vw[4];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = b0 + srow + r * 32;
      va[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < B && k0 + skq < K) va[r] = *(const float4*)(A + (size_t)row * K + k0 + skq);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = n0 + srow + r * 32;
      vw[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < N && k0 + skq < K) vw[r] = *(const float4*)(W + (size_t)row * K + k0 + skq);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float* d = As + (srow + r * 32) * PITCH + skq;
      d[0] = va[r].x; d[1] = va[r].y; d[2] = va[r].z; d[3] = va[r].w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* d = Ws + (srow + r * 32) * PITCH + skq;
      d[0] = vw[r].x; d[1] = vw[r].y; d[2] = vw[r].z; d[3] = vw[r].w;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < KC; kk += 2) {
      const float a0 = As[li * PITCH + kk + lk];
      const float a1 = As[(32 + li) * PITCH + kk + lk];
      const float w = Ws[(wave * 32 + li) * PITCH + kk + lk];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w, acc1, 0, 0, 0);
    }
  }

  // epilogue: lane holds column n = n0 + 32*wave + (lane&31), rows (reg&3)+8*(reg>>2)+4*(lane>>5) (+32 for acc1)
  const int n = n0 + wave * 32 + li;
  const bool nvalid = n < N;
  float cmax = -INFINITY, cmin = INFINITY;
  float wscale = 1.f, wshift = 0.f, winv = 0.f, wn2 = 0.f, wsum = 0.f, wband = 0.f;
  if (nvalid) {
    if (MODE == MODE_LINEAR) {
      wscale = ep.scale ? ep.scale[n] : 1.f;
      wshift = ep.shift ? ep.shift[n] : 0.f;
    } else if (MODE == MODE_DIST) {
      wn2 = ep.stat_w[2 * n];
      wsum = ep.stat_w[2 * n + 1];
      wband = match_band(wn2, (float)K);
    } else {
      winv = ep.inv_w[n];
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int b = b0 + h * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      const float dot = h ? acc1[r] : acc0[r];
      const bool valid = nvalid && b < B;
      if (MODE == MODE_LINEAR) {
        if (valid) {
          float v = dot * wscale + wshift;
          if (ep.relu) v = fmaxf(v, 0.f);
          ep.out[(size_t)b * N + n] = v;
        }
      } else if (MODE == MODE_COS) {
        float v = -INFINITY;
        if (valid) {
          v = dot * ep.inv_a[b] * winv * ep.s;
          if (ep.out) ep.out[(size_t)b * N + n] = v;
        }
        if (ep.argkey) {
          // arg-max over n, first index wins ties: maximise (score, ~n); -inf marks columns / rows outside the problem
          int fl;
          const float best = half_wave_best<true>(v, lk, fl);
          if (li == 0) {
            const int nb = n0 + wave * 32 + fl;
            s_key[wave * BMr + h * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk] =
                (b < B && best > -INFINITY) ? (((unsigned long long)f32_ordered(best) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)nb)) : 0ull;
          }
        }
      } else if (MODE == MODE_ARC) {
        if (valid) {
          const float cosr = dot * ep.inv_a[b] * winv;
          cmax = fmaxf(cmax, cosr);
          cmin = fminf(cmin, cosr);
          const float lo = (float)(-1.0 + 1e-7), hi = (float)(1.0 - 1e-7);
          const float c = fminf(fmaxf(cosr, lo), hi);
          float v = c;
          if (ep.label[b] == (int64_t)n) {
            const float theta = acosf(c);
            if (ep.easy) v = c > 0.f ? cosf(theta + ep.m) : c;
            else v = cosf(fminf((float)(3.14159265358979323846 - 1e-4), theta + ep.m));
          }
          v *= ep.s;
          if (isnan(v) || isinf(v)) v = 0.f;
          ep.out[(size_t)b * N + n] = v;
        }
      } else {  // MODE_DIST: ||a - g + eps||^2 = |a|^2 + |g|^2 - 2 a.g + 2 eps (sum a - sum g) + K eps^2, with its error band
        float L = INFINITY, U = INFINITY;  // columns / rows outside the problem never become candidates
        if (valid) {
          const float eps = 1e-6f, kf = (float)K, keps = kf * eps * eps;
          const float sa2 = ep.stat_a[2 * b];
          const float d2 = sa2 + wn2 - 2.f * dot + 2.f * eps * (ep.stat_a[2 * b + 1] - wsum) + keps;
          const float dl = match_kappa(K) * (match_band(sa2, kf) + wband + keps);
          L = d2 - dl; U = d2 + dl;
        }
        int fl;
        const float l1 = half_wave_best<false>(L, lk, fl);
        const float l2 = half_wave_min(li == fl ? INFINITY : L);
        const float up = half_wave_min(U);
        if (li == 0 && b < B) {
          MatchRec r; r.lo1 = l1; r.idx = l1 < INFINITY ? n0 + wave * 32 + fl : -1; r.lo2 = l2; r.up = up;
          ep.recs[(size_t)(blockIdx.x * 4 + wave) * B + b] = r;
        }
      }
    }
  }
  if (MODE == MODE_COS && ep.argkey) {
    // the four waves' candidates per row meet in LDS: one atomic per row and workgroup instead of one per row and wave
    __syncthreads();
    if (tid < BMr && b0 + tid < B) {
      unsigned long long key = s_key[tid];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const unsigned long long o = s_key[w * BMr + tid];
        key = o > key ? o : key;
      }
      if (key) atomicMax(ep.argkey + b0 + tid, key);
    }
  }
  if (MODE == MODE_ARC && ep.minmax_key) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
      cmin = fminf(cmin, __shfl_xor(cmin, o, 64));
    }
    if (lane == 0 && cmax >= cmin) {
      atomicMax(ep.minmax_key, f32_ordered(cmax));
      atomicMin(ep.minmax_key + 1, f32_ordered(cmin));
    }
  }
}

template <int MODE>
static int launch_gemm(const float* A, const float* W, int B, int N, int K, const GemmEpi& ep, hipStream_t st) {
  dim3 grid((N + 127) / 128, (B + 63) / 64);
  hipLaunchKernelGGL(gemm_nt_f32_kernel<MODE>, grid, dim3(256), 0, st, A, W, B, N, K, ep);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// ---- row statistics: one wave per row ------------------------------------------------------------
// mode 0: out[r] = 1 / max(||x_r||, eps)      mode 1: out[2r] = sum x^2, out[2r+1] = sum x
__global__ void row_stats_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int D, int mode, float eps) {
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (r >= R) return;
  float s2 = 0.f, s1 = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float v = x[(size_t)r * D + k];
    s2 += v * v;
    s1 += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s2 += __shfl_xor(s2, o, 64);
    s1 += __shfl_xor(s1, o, 64);
  }
  if (lane == 0) {
    if (mode == 0) out[r] = 1.0f / fmaxf(sqrtf(s2), eps);
    else { out[2 * r] = s2; out[2 * r + 1] = s1; }
  }
}

__global__ void l2_normalize_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int D, float eps) {
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (r >= R) return;
  float s2 = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float v = x[(size_t)r * D + k];
    s2 += v * v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
  const float denom = fmaxf(sqrtf(s2), eps);
  for (int k = lane; k < D; k += 64) out[(size_t)r * D + k] = x[(size_t)r * D + k] / denom;
}

__global__ void fill_u64_kernel(unsigned long long* p, int n, unsigned long long v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ||(a - g) + eps||_2^2 the way F.pairwise_distance forms its elements (fp32 subtract, fp32 add of eps), squares summed in
// float64 by the whole wave: the result does not depend on a summation order, identical rows give identical values, and it
// is within 2^-24 of what any fp32 summation of the same 512 squares returns.  Every lane gets the sum.
__device__ __forceinline__ double match_exact_d2(const float* __restrict__ a, const float* __restrict__ g, int D, int lane) {
  double s2 = 0.0;
  for (int k = lane * 4; k < D; k += 256) {
    const f32x4_t av = *(const f32x4_t*)(a + k), gv = *(const f32x4_t*)(g + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = (av[j] - gv[j]) + 1e-6f;
      s2 += (double)d * (double)d;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
  return s2;
}

// One wave per probe.  Reads the probe's candidate records (frmap_common.h: MatchRec [nslots][B], slot = slot_w consecutive
// gallery rows), takes ug = min over slots of `up`, and re-scores with the exact distance every slot whose lo1 <= ug: the
// slot's single candidate row when its second-best bound lies outside the band, every row of the slot otherwise.  Slots and
// rows are visited in ascending order and only a strictly smaller distance replaces the best one, so the result is the
// FIRST row attaining the minimum of the exact distance: compare_faces' loop (/root/reference/src/app.py:58-63).
__global__ void match_finalize_rec_kernel(const float* __restrict__ emb, const float* __restrict__ gal,
                                          const MatchRec* __restrict__ recs, int nslots, int slot_w,
                                          int32_t* __restrict__ idx_out, float* __restrict__ dist_out,
                                          int32_t* __restrict__ id_thr_out, int32_t* __restrict__ packed_out, float thresh,
                                          int B, int G, int D) {
  const int b = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (b >= B) return;
  const float* a = emb + (size_t)b * D;
  float ug = INFINITY;
  for (int s = lane; s < nslots; s += 64) ug = fminf(ug, recs[(size_t)s * B + b].up);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ug = fminf(ug, __shfl_xor(ug, o, 64));
  double best = INFINITY;
  int besti = -1;
  for (int s0 = 0; s0 < nslots; s0 += 64) {
    MatchRec r; r.lo1 = INFINITY; r.idx = -1; r.lo2 = INFINITY; r.up = INFINITY;
    if (s0 + lane < nslots) r = recs[(size_t)(s0 + lane) * B + b];
    unsigned long long mask = __ballot(r.lo1 <= ug && r.idx >= 0);
    while (mask) {
      const int l = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const int ci = __shfl(r.idx, l, 64);
      const float l2 = __shfl(r.lo2, l, 64);
      if (l2 <= ug) {
        const int g0 = (s0 + l) * slot_w, g1 = min(g0 + slot_w, G);
        for (int g = g0; g < g1; ++g) {
          const double d = match_exact_d2(a, gal + (size_t)g * D, D, lane);
          if (d < best) { best = d; besti = g; }
        }
      } else {
        const double d = match_exact_d2(a, gal + (size_t)ci * D, D, lane);
        if (d < best) { best = d; besti = ci; }
      }
    }
  }
  if (lane == 0) {
    const float d = besti >= 0 ? (float)sqrt(best) : INFINITY;
    idx_out[b] = besti; dist_out[b] = d;
    const int idt = (besti >= 0 && d <= thresh) ? besti : -1;
    if (id_thr_out) id_thr_out[b] = idt;
    if (packed_out) { packed_out[2 * b] = idt; packed_out[2 * b + 1] = __float_as_int(d); }
  }
}

// Small galleries (the demo's handful of enrolled faces, the 36-ID benchmark gallery): one wave per
// probe walks the gallery directly with the exact F.pairwise_distance arithmetic — no GEMM
// expansion, no atomics, one launch.  lane owns dims lane, lane+64, ...
__global__ __launch_bounds__(256) void match_small_kernel(const float* __restrict__ emb, const float* __restrict__ gal,
                                                          int32_t* __restrict__ idx_out, float* __restrict__ dist_out,
                                                          int32_t* __restrict__ id_thr_out, int32_t* __restrict__ packed_out,
                                                          float thresh, int B, int G, int D) {
  // one workgroup per probe; wave w scores gallery rows 8w..8w+7, 8(w+4).. (8 rows per pass so their
  // loads are independent); the four waves' (best, index) pairs meet in LDS, lowest index wins ties
  __shared__ float s_best[4];
  __shared__ int s_idx[4];
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float best = INFINITY;
  int besti = 0x7FFFFFFF;
  for (int g0 = wave * 8; g0 < G; g0 += 32) {
    float s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s2[j] = 0.f;
    for (int k = lane; k < D; k += 64) {
      const float e = emb[(size_t)b * D + k];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int gi = min(g0 + j, G - 1);
        const float d = (e - gal[(size_t)gi * D + k]) + 1e-6f;
        s2[j] += d * d;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = s2[j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (g0 + j < G && v < best) { best = v; besti = g0 + j; }  // strict <: first minimum within this wave's rows
    }
  }
  if (lane == 0) { s_best[wave] = best; s_idx[wave] = besti; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (s_best[w] < best || (s_best[w] == best && s_idx[w] < besti)) { best = s_best[w]; besti = s_idx[w]; }
    const bool any = besti != 0x7FFFFFFF;
    const float d = any ? sqrtf(best) : INFINITY;
    const int bi = any ? besti : -1;
    idx_out[b] = bi;
    dist_out[b] = d;
    const int idt = (any && d <= thresh) ? bi : -1;
    if (id_thr_out) id_thr_out[b] = idt;
    if (packed_out) { packed_out[2 * b] = idt; packed_out[2 * b + 1] = __float_as_int(d); }
  }
}

// The whole tail of the ResNet-18 ('cnn') embed-and-match step for small galleries in one launch, one workgroup
// per face: AdaptiveAvgPool2d(1) of the trunk map (face_models.py:100) -> (optional) F.normalize -> the
// compare_faces scan of match_small_kernel.  Replaces avgpool_global + l2_normalize + match_small (3 launches
// of 5-12 us that are pure latency at 256 faces); the embedding only passes through LDS.
template <typename TT>
__global__ __launch_bounds__(256) void gap_norm_match_kernel(const typename TT::elem* __restrict__ map, const float* __restrict__ gal,
                                                             float* __restrict__ emb_out, int32_t* __restrict__ idx_out,
                                                             float* __restrict__ dist_out, int32_t* __restrict__ id_thr_out,
                                                             int32_t* __restrict__ packed_out, float thresh, int normalize,
                                                             float eps, int HW, int C, int G) {
  extern __shared__ float s_e[];  // [C] embedding, then 4 + 4 reduction slots
  float* s_red = s_e + C;
  int* s_idx = (int*)(s_red + 4);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const typename TT::elem* src = map + (size_t)b * HW * C;
  // pooling: the C/8 8-channel groups x `nparts` interleaved row subsets over the 256 threads, partial sums meet in LDS
  const int C8 = C >> 3, nparts = C8 < 256 ? 256 / C8 : 1;
  float* s_part = s_e + C + 8;  // [nparts][C]
  auto pool_group = [&](int c8, int part) {
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int s = part;
    for (; s + 15 * nparts < HW; s += 16 * nparts) {  // sixteen independent 16-byte loads in flight
      u32x4_t r[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = *(const u32x4_t*)(src + (size_t)(s + q * nparts) * C + c8 * 8);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float f[8];
        unpack8<TT>(r[q], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += f[j];
      }
    }
    {  // the last (< 16) rows of this subset: loads first (rows past the end re-read the last one with weight 0)
      u32x4_t r[16];
      const int n = s < HW ? (HW - 1 - s) / nparts + 1 : 0;
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = *(const u32x4_t*)(src + (size_t)min(s + q * nparts, HW - 1) * C + c8 * 8);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float f[8];
        unpack8<TT>(r[q], f);
        const float wq = q < n ? 1.0f : 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaf(wq, f[j], a[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s_part[part * C + c8 * 8 + j] = a[j];
  };
  if (C8 < 256) {
    if (tid < C8 * nparts) pool_group(tid % C8, tid / C8);
  } else {
    for (int c8 = tid; c8 < C8; c8 += 256) pool_group(c8, 0);
  }
  __syncthreads();
  float ss = 0.f;
  {
    const float inv = 1.0f / (float)HW;
    for (int k = tid; k < C; k += 256) {
      float v = 0.f;
      for (int q = 0; q < nparts; ++q) v += s_part[q * C + k];
      v *= inv;
      s_e[k] = v;
      ss += v * v;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
  if (lane == 0) s_red[wave] = ss;
  __syncthreads();
  if (normalize) {
    const float denom = fmaxf(sqrtf(s_red[0] + s_red[1] + s_red[2] + s_red[3]), eps);
    for (int k = tid; k < C; k += 256) s_e[k] = s_e[k] / denom;
    __syncthreads();
  }
  if (emb_out)
    for (int k = tid; k < C; k += 256) emb_out[(size_t)b * C + k] = s_e[k];
  // compare_faces scan (as match_small_kernel): wave w scores rows 8w..8w+7, 8(w+4)..; first strict minimum wins
  float best = INFINITY;
  int besti = 0x7FFFFFFF;
  constexpr int RPW = 10;   // gallery rows per wave and pass: 40 rows per pass (the 36-ID gallery in one)
  for (int g0 = 0; g0 < G; g0 += 4 * RPW) {  // this pass: rows g0 + wave, g0 + wave + 4, ...
    // lane l owns dims 8 l .. 8 l + 7 of every 512-dim block: all of a pass's gallery reads (10 rows x 2 x 16 bytes per block)
    // are issued before the first use - one L2 round trip per pass instead of one per 64 dims
    float s2[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) s2[j] = 0.f;
    for (int kk = lane * 8; kk < C; kk += 512) {
      f32x4_t r[RPW][2];
#pragma unroll
      for (int j = 0; j < RPW; ++j) {
        const float* gp = gal + (size_t)min(g0 + 4 * j + wave, G - 1) * C + kk;
        r[j][0] = *(const f32x4_t*)gp;
        r[j][1] = *(const f32x4_t*)(gp + 4);
      }
      const f32x4_t e0 = *(const f32x4_t*)(s_e + kk), e1 = *(const f32x4_t*)(s_e + kk + 4);
#pragma unroll
      for (int j = 0; j < RPW; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float d0 = (e0[c] - r[j][0][c]) + 1e-6f, d1 = (e1[c] - r[j][1][c]) + 1e-6f;
          s2[j] = fmaf(d0, d0, s2[j]);
          s2[j] = fmaf(d1, d1, s2[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {  // ascending row index within the wave: strict < keeps the first minimum
      float v = s2[j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      const int gi = g0 + 4 * j + wave;
      if (gi < G && v < best) { best = v; besti = gi; }
    }
  }
  __syncthreads();  // (s_red is reused)
  if (lane == 0) { s_red[wave] = best; s_idx[wave] = besti; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (s_red[w] < best || (s_red[w] == best && s_idx[w] < besti)) { best = s_red[w]; besti = s_idx[w]; }
    const bool any = besti != 0x7FFFFFFF;
    const float d = any ? sqrtf(best) : INFINITY;
    const int bi = any ? besti : -1;
    idx_out[b] = bi;
    dist_out[b] = d;
    const int idt = (any && d <= thresh) ? bi : -1;
    if (id_thr_out) id_thr_out[b] = idt;
    if (packed_out) { packed_out[2 * b] = idt; packed_out[2 * b + 1] = __float_as_int(d); }
  }
}

extern "C" int frmap_gap_norm_match(const void* map, const float* gallery, float* emb_out, int32_t* idx_out, float* dist_out,
                                    int32_t* id_or_unknown_out, int32_t* packed_out, float thresh, int normalize, float eps,
                                    int B, int HW, int C, int G, int dtype, void* stream) {
  FRMAP_REQUIRE(map && idx_out && dist_out, "gap_norm_match: null pointer");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "gap_norm_match: bad dtype");
  FRMAP_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 8 == 0 && C <= 4096, "gap_norm_match: bad shape B=%d HW=%d C=%d", B, HW, C);
  FRMAP_REQUIRE(G >= 0 && G <= 64 && (G == 0 || gallery), "gap_norm_match: gallery of 0..64 rows expected (got %d)", G);
  const int c8 = C / 8, nparts = c8 < 256 ? 256 / c8 : 1;
  const size_t lds = (size_t)(C + 8 + nparts * C) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(gap_norm_match_kernel<BF16>, dim3(B), dim3(256), lds, st, (const __bf16*)map, gallery, emb_out, idx_out,
                       dist_out, id_or_unknown_out, packed_out, thresh, normalize, eps, HW, C, G);
  else
    hipLaunchKernelGGL(gap_norm_match_kernel<F16>, dim3(B), dim3(256), lds, st, (const _Float16*)map, gallery, emb_out, idx_out,
                       dist_out, id_or_unknown_out, packed_out, thresh, normalize, eps, HW, C, G);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// ArcFaceNet head in one launch (face_models.py:573-590: `features(x).view(B,-1)` -> `self.embedding` (Linear, no bias)
// -> `self.bn` (BatchNorm1d, eval) -> F.normalize): global average pool of the NHWC trunk map, y = (x . Wt) * scale +
// shift, e = y / max(||y||, eps).  Replaces avgpool_global + linear_f32 + l2_normalize (three launches, 85 us at 1024
// faces, the pooled features and the un-normalised embedding round-tripping through HBM).
// One workgroup = FB faces: their pooled features sit in LDS as [K][FB] (one broadcast 16-byte read feeds 4 faces), thread t
// owns outputs t, t + 256, ..: the K x N weight matrix is read TRANSPOSED ([K][N], row k = 4 N bytes, coalesced) once
// per workgroup out of L2.
// ------------------------------------------------------------------------------------------------
template <typename TT, int FB, int NPT>
__global__ __launch_bounds__(256) void gap_linear_norm_kernel(const typename TT::elem* __restrict__ map, const float* __restrict__ wt,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              float* __restrict__ pre_out, float* __restrict__ emb_out, float eps,
                                                              int B, int HW, int K, int relu, int stagger) {
  constexpr int N = NPT * 256;
  extern __shared__ float s_all[];
  float* s_x = s_all;                 // [K][FB]
  float* s_red = s_all + K * FB;      // [FB][4] sums of squares
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b0 = blockIdx.x * FB;
  const int C8 = K >> 3;
  const float inv = 1.0f / (float)HW;
  // pooling: wave w takes faces w and w + 4 side by side (two independent accumulator sets: 16 loads of 16 bytes in flight
  // per lane).  K >= 512: lane l owns the 8-channel groups l, l + 64, .. and walks all HW rows, nothing meets in LDS.
  // K < 512 (BaselineNet: 128 channels x 784 pixels): the 64 lanes are K/8 groups x nsub row subsets, whose partial sums
  // meet in LDS (s_part) in a fixed order.
  static_assert(FB == 8, "two faces per wave, four waves");
  const int nsub = C8 >= 64 ? 1 : 64 / C8;
  float* s_part = s_red + FB * 4;   // [FB][nsub][K] (nsub > 1 only)
  {
    const int fa = wave, fb = wave + 4;
    const typename TT::elem* sa = map + (size_t)min(b0 + fa, B - 1) * HW * K;
    const typename TT::elem* sb = map + (size_t)min(b0 + fb, B - 1) * HW * K;
    const int sub = nsub > 1 ? lane / C8 : 0;
    for (int c8 = nsub > 1 ? lane % C8 : lane; c8 < C8 && sub < nsub; c8 += 64) {
      float a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int s0 = sub; s0 < HW; s0 += 8 * nsub) {
        u32x4_t ra[8], rb[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const size_t o = (size_t)min(s0 + q * nsub, HW - 1) * K + c8 * 8;
          ra[q] = *(const u32x4_t*)(sa + o);
          rb[q] = *(const u32x4_t*)(sb + o);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float va[8], vb[8];
          unpack8<TT>(ra[q], va);
          unpack8<TT>(rb[q], vb);
          const float wq = s0 + q * nsub < HW ? 1.0f : 0.0f;
#pragma unroll
          for (int j = 0; j < 8; ++j) { a[j] = fmaf(wq, va[j], a[j]); c[j] = fmaf(wq, vb[j], c[j]); }
        }
      }
      if (nsub == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          s_x[(c8 * 8 + j) * FB + fa] = a[j] * inv;
          s_x[(c8 * 8 + j) * FB + fb] = c[j] * inv;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          s_part[((size_t)fa * nsub + sub) * K + c8 * 8 + j] = a[j];
          s_part[((size_t)fb * nsub + sub) * K + c8 * 8 + j] = c[j];
        }
      }
      if (nsub > 1) break;   // (one group per lane)
    }
  }
  __syncthreads();
  if (nsub > 1) {
    for (int i = tid; i < K * FB; i += 256) {
      const int k = i / FB, f = i - k * FB;
      float v = 0.f;
      for (int q = 0; q < nsub; ++q) v += s_part[((size_t)f * nsub + q) * K + k];
      s_x[i] = v * inv;
    }
    __syncthreads();
  }
  // ---- y[f][n] = sum_k x[f][k] * wt[k][n]
  float acc[NPT][FB];
#pragma unroll
  for (int j = 0; j < NPT; ++j)
#pragma unroll
    for (int f = 0; f < FB; ++f) acc[j][f] = 0.f;
  // every workgroup reads the same K x N matrix: each starts at its own row block so that they do not all pull the same
  // L2 lines at the same time (stagger = 0 under batch-invariant planning: the summation order must not depend on where in
  // the batch a face sits)
  const int kstart = (int)(((long long)blockIdx.x * stagger) % (K / 8)) * 8;
  auto krow = [&](int it) { return it + kstart < K ? it + kstart : it + kstart - K; };
  float wc[8][NPT], wn[8][NPT];
  auto load_w = [&](float (&w)[8][NPT], int k0) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < NPT; ++j) w[u][j] = wt[(size_t)(k0 + u) * N + j * 256 + tid];
  };
  load_w(wc, krow(0));
  for (int it = 0; it < K; it += 8) {   // the next 8 rows are requested before this block's FMAs (one L2 round trip per block otherwise)
    const int k0 = krow(it);
    load_w(wn, krow(it + 8 < K ? it + 8 : it));
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float x[FB];
#pragma unroll
      for (int f4 = 0; f4 < FB; f4 += 4) {
        const f32x4_t v = *(const f32x4_t*)(s_x + (k0 + u) * FB + f4);
        x[f4] = v[0]; x[f4 + 1] = v[1]; x[f4 + 2] = v[2]; x[f4 + 3] = v[3];
      }
#pragma unroll
      for (int j = 0; j < NPT; ++j)
#pragma unroll
        for (int f = 0; f < FB; ++f) acc[j][f] = fmaf(x[f], wc[u][j], acc[j][f]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < NPT; ++j) wc[u][j] = wn[u][j];
  }
  // ---- BatchNorm1d (folded to scale / shift), sum of squares per face, normalise
  float ss[FB];
#pragma unroll
  for (int f = 0; f < FB; ++f) ss[f] = 0.f;
#pragma unroll
  for (int j = 0; j < NPT; ++j) {
    const int n = j * 256 + tid;
    const float sc = scale ? scale[n] : 1.f, sh = shift ? shift[n] : 0.f;
#pragma unroll
    for (int f = 0; f < FB; ++f) {
      acc[j][f] = acc[j][f] * sc + sh;
      if (relu) acc[j][f] = fmaxf(acc[j][f], 0.f);
      ss[f] += acc[j][f] * acc[j][f];
    }
  }
#pragma unroll
  for (int f = 0; f < FB; ++f) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss[f] += __shfl_xor(ss[f], o, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int f = 0; f < FB; ++f) s_red[f * 4 + wave] = ss[f];
  }
  __syncthreads();
#pragma unroll
  for (int f = 0; f < FB; ++f) {
    if (b0 + f >= B) break;
    const float denom = fmaxf(sqrtf(s_red[f * 4] + s_red[f * 4 + 1] + s_red[f * 4 + 2] + s_red[f * 4 + 3]), eps);
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
      const size_t o = (size_t)(b0 + f) * N + j * 256 + tid;
      if (pre_out) pre_out[o] = acc[j][f];
      if (emb_out) emb_out[o] = acc[j][f] / denom;
    }
  }
}

// wt: the Linear weight TRANSPOSED, fp32 [K][N] (N in {256, 512}); scale / shift: the folded BatchNorm1d or the Linear bias ([N], may
// be NULL); relu = 1: ReLU before the normalisation (BaselineNet: F.relu(self.fc1(pooled)), face_models.py:43-46);
// pre_out / emb_out: [B][N] fp32 un-normalised / unit-norm embeddings (either may be NULL)
extern "C" int frmap_gap_linear_norm(const void* map, const float* wt, const float* scale, const float* shift, float* pre_out,
                                     float* emb_out, float eps, int B, int HW, int K, int N, int relu, int dtype, void* stream) {
  FRMAP_REQUIRE(map && wt && (pre_out || emb_out), "gap_linear_norm: null pointer");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "gap_linear_norm: bad dtype");
  FRMAP_REQUIRE(B > 0 && HW > 0 && K > 0 && K % 8 == 0 && K <= 2048, "gap_linear_norm: bad shape B=%d HW=%d K=%d", B, HW, K);
  FRMAP_REQUIRE(N == 256 || N == 512, "gap_linear_norm: N=%d not supported (256 or 512)", N);
  constexpr int FB = 8;
  const int c8n = K / 8, nsubh = c8n >= 64 ? 1 : 64 / c8n;
  const size_t lds = (size_t)(K * FB + FB * 4 + (nsubh > 1 ? FB * nsubh * K : 0)) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((B + FB - 1) / FB);
  {
    const void* kerns[4] = {(const void*)gap_linear_norm_kernel<BF16, FB, 2>, (const void*)gap_linear_norm_kernel<BF16, FB, 1>,
                            (const void*)gap_linear_norm_kernel<F16, FB, 2>, (const void*)gap_linear_norm_kernel<F16, FB, 1>};
    if (frmap_big_lds(kerns[(dtype == FRMAP_BF16 ? 0 : 2) + (N == 512 ? 0 : 1)], 160 * 1024)) return -2;
  }
#define GLN_GO(TT, ET, NPT)                                                                                                  \
  hipLaunchKernelGGL((gap_linear_norm_kernel<TT, FB, NPT>), grid, dim3(256), lds, st, (const ET*)map, wt, scale, shift, pre_out, \
                     emb_out, eps, B, HW, K, relu, stagger)
  const int stagger = frmap_batch_invariant() ? 0 : 40;
  if (dtype == FRMAP_BF16) { if (N == 512) GLN_GO(BF16, __bf16, 2); else GLN_GO(BF16, __bf16, 1); }
  else { if (N == 512) GLN_GO(F16, _Float16, 2); else GLN_GO(F16, _Float16, 1); }
#undef GLN_GO
  FRMAP_LAUNCH_CHECK();
  return 0;
}

__global__ void argkey_finalize_kernel(const unsigned long long* __restrict__ keys, int32_t* __restrict__ out, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) out[b] = keys[b] ? (int32_t)(0xFFFFFFFFu - (unsigned)(keys[b] & 0xFFFFFFFFull)) : -1;
}

__global__ void minmax_init_kernel(unsigned int* k) {
  k[0] = 0u;           // running max key
  k[1] = 0xFFFFFFFFu;  // running min key
}
__global__ void minmax_finalize_kernel(const unsigned int* k, float* out) {
  out[0] = f32_unordered(k[0]);
  out[1] = f32_unordered(k[1]);
}

// row softmax + arg-max (first maximum), one wave per row: `probs = F.softmax(outputs, dim=1)`,
// `_, predicted = torch.max(outputs, 1)` of the evaluation loop (src/testing.py:278-279)
__global__ void softmax_argmax_kernel(const float* __restrict__ x, float* __restrict__ probs, int32_t* __restrict__ pred,
                                      int R, int C) {
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (r >= R) return;
  float mx = -INFINITY;
  int mi = 0x7FFFFFFF;
  for (int c = lane; c < C; c += 64) {
    const float v = x[(size_t)r * C + c];
    if (v > mx || (v == mx && c < mi)) { mx = v; mi = c; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 64);
    const int oi = __shfl_xor(mi, o, 64);
    if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
  }
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) sum += __expf(x[(size_t)r * C + c] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  if (probs)
    for (int c = lane; c < C; c += 64) probs[(size_t)r * C + c] = __expf(x[(size_t)r * C + c] - mx) / sum;
  if (pred && lane == 0) pred[r] = mi;
}

// F.pairwise_distance(a, b) row-wise (eps = 1e-6 on the difference) and the Siamese decision
// `pred = dist < thresh` (src/testing.py:175-177)
__global__ void pairwise_distance_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                         float* __restrict__ dist, int32_t* __restrict__ same, float thresh, int R, int D) {
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (r >= R) return;
  float s2 = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float d = (a[(size_t)r * D + k] - b[(size_t)r * D + k]) + 1e-6f;
    s2 += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
  if (lane == 0) {
    const float d = sqrtf(s2);
    dist[r] = d;
    if (same) same[r] = d < thresh ? 1 : 0;
  }
}

extern "C" size_t frmap_head_workspace_bytes(int B, int C) {
  return (size_t)16 * ((size_t)(B > 0 ? B : 0) + (size_t)(C > 0 ? C : 0)) + 256;
}


extern "C" int frmap_linear_f32(const float* x, const float* w, const float* scale, const float* shift, float* out,
                                int B, int K, int N, int relu, void* stream) {
  FRMAP_REQUIRE(x && w && out, "linear_f32: null pointer");
  FRMAP_REQUIRE(B > 0 && K > 0 && N > 0 && K % 4 == 0, "linear_f32: bad shape B=%d K=%d N=%d (K %% 4 == 0)", B, K, N);
  GemmEpi ep = {};
  ep.scale = scale; ep.shift = shift; ep.relu = relu; ep.out = out;
  return launch_gemm<MODE_LINEAR>(x, w, B, N, K, ep, (hipStream_t)stream);
}

extern "C" int frmap_l2_normalize_f32(const float* x, float* out, int B, int D, float eps, void* stream) {
  FRMAP_REQUIRE(x && out, "l2_normalize: null pointer");
  FRMAP_REQUIRE(B > 0 && D > 0, "l2_normalize: bad shape");
  hipLaunchKernelGGL(l2_normalize_kernel, dim3(waves_blocks(B)), dim3(256), 0, (hipStream_t)stream, x, out, B, D, eps);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

extern "C" int frmap_softmax_argmax(const float* logits, float* probs_out, int32_t* pred_out, int B, int C, void* stream) {
  FRMAP_REQUIRE(logits && (probs_out || pred_out), "softmax_argmax: null pointer");
  FRMAP_REQUIRE(B > 0 && C > 0, "softmax_argmax: bad shape");
  hipLaunchKernelGGL(softmax_argmax_kernel, dim3(waves_blocks(B)), dim3(256), 0, (hipStream_t)stream, logits, probs_out, pred_out, B, C);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

extern "C" int frmap_pairwise_distance(const float* a, const float* b, float* dist_out, int32_t* same_out, float thresh,
                                       int B, int D, void* stream) {
  FRMAP_REQUIRE(a && b && dist_out, "pairwise_distance: null pointer");
  FRMAP_REQUIRE(B > 0 && D > 0, "pairwise_distance: bad shape");
  hipLaunchKernelGGL(pairwise_distance_kernel, dim3(waves_blocks(B)), dim3(256), 0, (hipStream_t)stream, a, b, dist_out, same_out, thresh, B, D);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

extern "C" int frmap_match_top1(const float* emb, const float* gallery, int32_t* idx_out, float* dist_out,
                                int32_t* id_or_unknown_out, int32_t* packed_out, float thresh, void* workspace,
                                int B, int G, int D, void* stream) {
  FRMAP_REQUIRE(emb && idx_out && dist_out && workspace, "match_top1: null pointer");
  FRMAP_REQUIRE(B > 0 && D > 0 && D % 4 == 0 && G >= 0, "match_top1: bad shape B=%d G=%d D=%d", B, G, D);
  FRMAP_REQUIRE(G == 0 || gallery, "match_top1: null gallery");
  hipStream_t st = (hipStream_t)stream;
  if (G > 0 && G <= 64) {
    hipLaunchKernelGGL(match_small_kernel, dim3(B), dim3(256), 0, st, emb, gallery, idx_out, dist_out,
                       id_or_unknown_out, packed_out, thresh, B, G, D);
    FRMAP_LAUNCH_CHECK();
    return 0;
  }
  if (G <= 0) {
    hipLaunchKernelGGL(match_finalize_rec_kernel, dim3(waves_blocks(B)), dim3(256), 0, st, emb, gallery, (const MatchRec*)nullptr, 0, 32,
                       idx_out, dist_out, id_or_unknown_out, packed_out, thresh, B, 0, D);
    FRMAP_LAUNCH_CHECK();
    return 0;
  }
  // workspace (frmap_match_workspace_bytes): records [4 * ceil(G / 128)][B] | probe statistics [B][2] | gallery statistics [G][2]
  const int nslots = 4 * ((G + 127) / 128);
  MatchRec* recs = (MatchRec*)workspace;
  float* stat_a = (float*)(recs + (size_t)nslots * B);
  float* stat_w = stat_a + 2 * (size_t)B;
  hipLaunchKernelGGL(row_stats_kernel, dim3(waves_blocks(B)), dim3(256), 0, st, emb, stat_a, B, D, 1, 0.f);
  hipLaunchKernelGGL(row_stats_kernel, dim3(waves_blocks(G)), dim3(256), 0, st, gallery, stat_w, G, D, 1, 0.f);
  GemmEpi ep = {};
  ep.stat_a = stat_a; ep.stat_w = stat_w; ep.recs = recs;
  int rc = launch_gemm<MODE_DIST>(emb, gallery, B, G, D, ep, st);
  if (rc) return rc;
  hipLaunchKernelGGL(match_finalize_rec_kernel, dim3(waves_blocks(B)), dim3(256), 0, st, emb, gallery, (const MatchRec*)recs, nslots, 32,
                     idx_out, dist_out, id_or_unknown_out, packed_out, thresh, B, G, D);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Large galleries on the MFMA pipe.  The fp32 GEMM above runs at ~60 TFLOP/s (fp32 MFMA peaks at 1/16 of the 16-bit
// rate): at 1024 probes x 10 000 identities it is 175 us of a 4 ms step.  Here every fp32 operand x is split into two
// fp16 numbers, hi = fp16(S x), lo = fp16(S x - hi) (S = 256 keeps lo out of the fp16 subnormals for |x| >= 2^-11), and
// a.g ~ (a_hi.g_hi + a_hi.g_lo + a_lo.g_hi) / S^2, accumulated in fp32 by ONE K = 3 D GEMM on the 16x16x32 fp16 MFMA
// (relative error ~2^-22, the size of the fp32 GEMM's own rounding).  The epilogue and the exact finalize step are those
// of the fp32 path, so the result has the same contract: first minimum of the expanded distance, exact
// F.pairwise_distance of the winner.  The gallery is split and laid out in the kernel's weight order ONCE per gallery.
// ------------------------------------------------------------------------------------------------
// Each row gets its own power-of-two scale S (largest |x| of the row lands in [2^13, 2^14): nothing overflows fp16, lo stays
// out of the fp16 subnormals for every element within 2^-11 of the row maximum, and scaling by a power of two is exact);
// statistics record of a row: (sum x^2, sum x, 1 / S, band(x) = the row's share of the expanded distance's error bound, frmap_common.h).
__device__ __forceinline__ float match_row_scale(float amax) {
  if (!(amax > 0.f) || isinf(amax)) return 1.f;
  int e;
  frexpf(amax, &e);                 // amax = m * 2^e, m in [0.5, 1)
  return ldexpf(1.f, 14 - e);       // amax * S in [2^13, 2^14)
}

// one wave per row: statistics + (probes) the split row (a_hi | a_hi | a_lo)
__global__ void match_row_prep_kernel(const float* __restrict__ x, float* __restrict__ stat4, _Float16* __restrict__ split3,
                                      int R, int D) {
  const int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* row = x + (size_t)r * D;
  float s2 = 0.f, s1 = 0.f, amax = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float v = row[k];
    s2 += v * v; s1 += v; amax = fmaxf(amax, fabsf(v));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s2 += __shfl_xor(s2, o, 64); s1 += __shfl_xor(s1, o, 64); amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  }
  const float S = match_row_scale(amax);
  if (lane == 0) {
    stat4[4 * (size_t)r] = s2; stat4[4 * (size_t)r + 1] = s1; stat4[4 * (size_t)r + 2] = 1.0f / S;
    stat4[4 * (size_t)r + 3] = match_band(s2, (float)D);
  }
  if (split3) {
    _Float16* o = split3 + (size_t)r * 3 * D;
    for (int k = lane; k < D; k += 64) {
      const float v = row[k] * S;
      const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
      o[k] = hi; o[D + k] = hi; o[2 * D + k] = lo;
    }
  }
}

__global__ void match_pack_gallery_kernel(const float* __restrict__ gal, const float* __restrict__ stat4, _Float16* __restrict__ out,
                                          size_t e_lo, size_t e_hi, int G, int D) {
  const int nchunks = 3 * D / 32;
  size_t e = e_lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; e < e_hi; e += stride) {   // element order of frmap_pack_conv_weight for a 1x1 layer: [row/64][k/32][row%64][slot][8]
    const int j = (int)(e & 7), slot = (int)((e >> 3) & 3), cl = (int)((e >> 5) & 63);
    const size_t rest = e >> 11;
    const int chunk = (int)(rest % nchunks), ntile = (int)(rest / nchunks);
    const int cg = slot ^ (((cl >> 2) & 1) << 1);
    const int k3 = chunk * 32 + cg * 8 + j, row = ntile * 64 + cl;
    const int part = k3 / D, k = k3 - part * D;        // gallery side: (g_hi | g_lo | g_hi)
    float v = 0.f;
    if (row < G) v = gal[(size_t)row * D + k] * (1.0f / stat4[4 * (size_t)row + 2]);   // (S is a power of two: 1 / (1 / S) is exact)
    const _Float16 hi = (_Float16)v;
    out[e] = part == 1 ? (_Float16)(v - (float)hi) : hi;
  }
}

extern "C" size_t frmap_match_gallery_pack_bytes(int G, int D) {
  if (G <= 0 || D <= 0) return 0;
  return (size_t)((G + 255) / 256 * 256) * 3 * D * sizeof(_Float16);
}

// (re)pack gallery rows [row_lo, row_hi) of a gallery that now holds G rows: statistics of those rows + every 64-row tile they touch.
static int match_pack_rows(const float* gallery, void* packed_out, float* stat_w_out, int row_lo, int row_hi, int G, int D, hipStream_t st) {
  hipLaunchKernelGGL(match_row_prep_kernel, dim3(waves_blocks(row_hi - row_lo)), dim3(256), 0, st, gallery + (size_t)row_lo * D,
                     stat_w_out + 4 * (size_t)row_lo, (_Float16*)nullptr, row_hi - row_lo, D);
  const size_t per_tile = (size_t)(3 * D / 32) * 2048;
  const size_t e_lo = (size_t)(row_lo / 64) * per_tile;
  // through the end of the last tile's 256-row padding group, so that rows G .. Gpad - 1 are (re)written as zeros
  const size_t t_hi = row_hi >= G ? (size_t)((G + 255) / 256 * 4) : (size_t)((row_hi + 63) / 64);
  const size_t e_hi = t_hi * per_tile;
  const size_t n = e_hi - e_lo;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(match_pack_gallery_kernel, dim3(blocks), dim3(256), 0, st, gallery, (const float*)stat_w_out,
                     (_Float16*)packed_out, e_lo, e_hi, G, D);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// one-time preparation of a gallery for frmap_match_top1_packed: packed_out (frmap_match_gallery_pack_bytes), stat_w_out [G][4]
extern "C" int frmap_match_pack_gallery(const float* gallery, void* packed_out, float* stat_w_out, int G, int D, void* stream) {
  FRMAP_REQUIRE(gallery && packed_out && stat_w_out, "match_pack_gallery: null pointer");
  FRMAP_REQUIRE(G > 0 && D > 0 && D % 32 == 0, "match_pack_gallery: bad shape G=%d D=%d (D %% 32 == 0)", G, D);
  return match_pack_rows(gallery, packed_out, stat_w_out, 0, G, G, D, (hipStream_t)stream);
}

// incremental enrolment (/root/reference/src/app.py:428-436 appends one identity): rows [row_lo, row_hi) of the gallery were
// written (appended: row_hi == G, the new row count; or edited in place); only their statistics and 64-row tiles are re-packed.
// packed_out / stat_w_out must have been sized for at least G rows (frmap_match_gallery_pack_bytes(capacity, D), [capacity][4]).
extern "C" int frmap_match_pack_gallery_rows(const float* gallery, void* packed_out, float* stat_w_out, int row_lo, int row_hi,
                                             int G, int D, void* stream) {
  FRMAP_REQUIRE(gallery && packed_out && stat_w_out, "match_pack_gallery_rows: null pointer");
  FRMAP_REQUIRE(G > 0 && D > 0 && D % 32 == 0 && row_lo >= 0 && row_lo < row_hi && row_hi <= G,
                "match_pack_gallery_rows: bad range [%d, %d) of G=%d D=%d", row_lo, row_hi, G, D);
  return match_pack_rows(gallery, packed_out, stat_w_out, row_lo, row_hi, G, D, (hipStream_t)stream);
}

// frmap_match_top1 for a prepared gallery (same outputs, same contract).  workspace: frmap_match_workspace_bytes(B, G)
// bytes; probe_split: B * 3 * D fp16 scratch.
extern "C" int frmap_match_top1_packed(const float* emb, const float* gallery, const void* gallery_packed, const float* stat_w,
                                       int32_t* idx_out, float* dist_out, int32_t* id_or_unknown_out, int32_t* packed_out,
                                       float thresh, void* workspace, void* probe_split, int B, int G, int D, void* stream) {
  FRMAP_REQUIRE(emb && gallery && gallery_packed && stat_w && idx_out && dist_out && workspace && probe_split, "match_top1_packed: null pointer");
  FRMAP_REQUIRE(B > 0 && G > 0 && D > 0 && D % 32 == 0, "match_top1_packed: bad shape B=%d G=%d D=%d", B, G, D);
  hipStream_t st = (hipStream_t)stream;
  // workspace: records [Gpad / 64][B] (Gpad = G rounded up to 256: the GEMM's padded rows write never-candidate records) | statistics [B][4]
  const int nslots = (G + 255) / 256 * 4;
  MatchRec* recs = (MatchRec*)workspace;
  float* stat_a = (float*)(recs + (size_t)nslots * B);
  hipLaunchKernelGGL(match_row_prep_kernel, dim3(waves_blocks(B)), dim3(256), 0, st, emb, stat_a, (_Float16*)probe_split, B, D);
  const int rc = frmap_match_gemm_f16x3(probe_split, gallery_packed, stat_a, stat_w, recs, B, G, D, st);
  if (rc) return rc;
  hipLaunchKernelGGL(match_finalize_rec_kernel, dim3(waves_blocks(B)), dim3(256), 0, st, emb, gallery, (const MatchRec*)recs, nslots, 64,
                     idx_out, dist_out, id_or_unknown_out, packed_out, thresh, B, G, D);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t frmap_match_workspace_bytes(int B, int G) {
  const size_t b = B > 0 ? (size_t)B : 0, g = G > 0 ? (size_t)G : 0;
  return 16 * b * (4 * ((g + 127) / 128)) + 16 * b + 8 * g + 256;
}

extern "C" int frmap_cosine_logits(const float* x, const float* w, float* logits_out, int32_t* argmax_out,
                                   void* workspace, int B, int C, int D, float s, void* stream) {
  FRMAP_REQUIRE(x && w && workspace && (logits_out || argmax_out), "cosine_logits: null pointer");
  FRMAP_REQUIRE(B > 0 && C > 0 && D > 0 && D % 4 == 0, "cosine_logits: bad shape");
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* keys = (unsigned long long*)workspace;
  float* inv_a = (float*)(keys + B);
  float* inv_w = inv_a + B;
  hipLaunchKernelGGL(row_stats_kernel, dim3(waves_blocks(B)), dim3(256), 0, st, x, inv_a, B, D, 0, 1e-12f);
  hipLaunchKernelGGL(row_stats_kernel, dim3(waves_blocks(C)), dim3(256), 0, st, w, inv_w, C, D, 0, 1e-12f);
  GemmEpi ep = {};
  ep.inv_a = inv_a; ep.inv_w = inv_w; ep.s = s; ep.out = logits_out;
  if (argmax_out) {
    hipLaunchKernelGGL(fill_u64_kernel, dim3((B + 255) / 256), dim3(256), 0, st, keys, B, 0ull);
    ep.argkey = keys;
  }
  int rc = launch_gemm<MODE_COS>(x, w, B, C, D, ep, st);
  if (rc) return rc;
  if (argmax_out) hipLaunchKernelGGL(argkey_finalize_kernel, dim3((B + 255) / 256), dim3(256), 0, st, keys, argmax_out, B);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

extern "C" int frmap_arcmargin_eval(const float* x, const float* w, const int64_t* label, float* logits_out,
                                    float* minmax_out, void* workspace, int B, int C, int D, float s, float m,
                                    int easy_margin, void* stream) {
  FRMAP_REQUIRE(x && w && label && logits_out && workspace, "arcmargin_eval: null pointer");
  FRMAP_REQUIRE(B > 0 && C > 0 && D > 0 && D % 4 == 0, "arcmargin_eval: bad shape");
  hipStream_t st = (hipStream_t)stream;
  unsigned int* mm = (unsigned int*)workspace;
  float* inv_a = (float*)workspace + 4;
  float* inv_w = inv_a + B;
  hipLaunchKernelGGL(row_stats_kernel, dim3(waves_blocks(B)), dim3(256), 0, st, x, inv_a, B, D, 0, 1e-12f);
  hipLaunchKernelGGL(row_stats_kernel, dim3(waves_blocks(C)), dim3(256), 0, st, w, inv_w, C, D, 0, 1e-12f);
  GemmEpi ep = {};
  ep.inv_a = inv_a; ep.inv_w = inv_w; ep.s = s < 24.0f ? s : 24.0f; ep.m = m; ep.easy = easy_margin;
  ep.label = label; ep.out = logits_out;
  if (minmax_out) {
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, st, mm);
    ep.minmax_key = mm;
  }
  int rc = launch_gemm<MODE_ARC>(x, w, B, C, D, ep, st);
  if (rc) return rc;
  if (minmax_out) hipLaunchKernelGGL(minmax_finalize_kernel, dim3(1), dim3(1), 0, st, mm, minmax_out);
  FRMAP_LAUNCH_CHECK();
  return 0;
}
