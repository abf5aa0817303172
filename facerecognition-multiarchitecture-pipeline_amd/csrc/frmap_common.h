// Shared device/host helpers for the gfx950 kernels.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/frmap_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

struct BF16 {
  using elem = __bf16;
  using vec8 = bf16x8_t;
  static __device__ __forceinline__ f32x4_t mfma(vec8 a, vec8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(elem v) { return (float)v; }
  static __device__ __forceinline__ elem from_f32(float v) { return (elem)v; }
};
struct F16 {
  using elem = _Float16;
  using vec8 = f16x8_t;
  static __device__ __forceinline__ f32x4_t mfma(vec8 a, vec8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(elem v) { return (float)v; }
  static __device__ __forceinline__ elem from_f32(float v) { return (elem)v; }
};

// pack 4 fp32 -> 4 T (8 bytes)
template <typename TT>
__device__ __forceinline__ u32x2_t pack4(float a, float b, float c, float d) {
  typename TT::elem e[4] = {TT::from_f32(a), TT::from_f32(b), TT::from_f32(c), TT::from_f32(d)};
  u32x2_t r;
  __builtin_memcpy(&r, e, 8);
  return r;
}
template <typename TT>
__device__ __forceinline__ void unpack4(u32x2_t v, float* o) {
  typename TT::elem e[4];
  __builtin_memcpy(e, &v, 8);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = TT::to_f32(e[i]);
}
template <typename TT>
__device__ __forceinline__ void unpack8(u32x4_t v, float* o) {
  typename TT::elem e[8];
  __builtin_memcpy(e, &v, 16);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = TT::to_f32(e[i]);
}
template <typename TT>
__device__ __forceinline__ u32x4_t pack8(const float* f) {
  typename TT::elem e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = TT::from_f32(f[i]);
  u32x4_t r;
  __builtin_memcpy(&r, e, 16);
  return r;
}

// ------------------------------------------------------------------------------------------------
// Top-1 gallery match (compare_faces, /root/reference/src/app.py:58-63) behind a GEMM: candidate records.
//
// The GEMM kernels score a (probe, gallery row) pair by the EXPANDED squared distance
//     d2e = |a|^2 + |g|^2 - 2 a.g + 2 eps (sum a - sum g) + K eps^2        (eps = 1e-6, F.pairwise_distance's)
// whose rounding error is bounded by  delta(a, g) = kappa * (band(a) + band(g) + K eps^2),
//     band(x) = |x|^2 + 2 eps sqrt(K |x|^2)  (>= |x|^2 + 2 eps sum |x_i|),  kappa = (2 T + 64) * 2^-24,
// T = the number of products the dot product accumulates (K for the fp32 GEMM, 3 K for the split-fp16 GEMM whose
// operands carry another 3 * 2^-22 of relative error): a worst-case bound (gamma_T * sum |a_i g_i| <= T u (|a|^2 + |g|^2) / 2
// for the dot product, the same for the two squared norms, a few u for the combination), not a typical-case one.
// So with L = d2e - delta and U = d2e + delta, the row that minimises the EXACT distance has L <= min over all rows of U.
// An epilogue therefore writes, per probe and per SLOT of consecutive gallery rows, one record
//     (lo1, idx) = smallest L of the slot and its row,  lo2 = second smallest L,  up = smallest U
// (no atomics: every record has exactly one writer), and match_finalize_rec_kernel (head_match.hip) re-scores with the exact
// ||(a - g) + eps||_2 every slot whose lo1 <= min up: its single row when lo2 is outside the band, all of its rows otherwise,
// and keeps the first strict minimum - the reference loop's answer, not the expanded form's.
// ------------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) MatchRec {
  float lo1;
  int idx;
  float lo2;
  float up;
};
__device__ __forceinline__ float match_kappa(int terms) { return (float)(2 * terms + 64) * 5.9604644775390625e-8f; }
__device__ __forceinline__ float match_band(float s2, float kf) { return s2 + 2e-6f * sqrtf(kf * s2); }

// Epilogue of the split-fp16 match GEMMs (conv1x1_kernel<.., MATCH>, conv1x1_pp_kernel<.., MATCH>): after the K loop lane
// (lr, g) holds, per MFMA tile (mi, ni), the scaled dot products of probe b_base + mi * 16 + lr with gallery rows
// n0 + ni * 16 + 4 g + j.  The wave's 64 gallery rows are one slot (n0 / 64); records are laid out [slot][M].
// Row statistics (match_row_prep_kernel): (sum x^2, sum x, 1 / row scale, band(x)).
template <int MI>
__device__ __forceinline__ void match_epilogue_records(const f32x4_t (&acc)[MI][4], int b_base, int b_end, int n0, int G, int D,
                                                       int M, const float* __restrict__ stat_a,
                                                       const float* __restrict__ stat_w, MatchRec* __restrict__ recs,
                                                       int lane) {
  const int lr = lane & 15, g = lane >> 4;
  const float eps = 1e-6f, kf = (float)D, keps = kf * eps * eps, kap = match_kappa(3 * D);
  // One probe row (mi) at a time; the gallery-row statistics are re-read per (mi, ni) block (L1 hits) instead of being held for
  // the whole epilogue, so the live set stays at acc + ~30 registers: the kernels around this epilogue count their LDS-DMA with
  // s_waitcnt vmcnt(n) and must not spill (csrc/build.sh checks)
  MatchRec* out = recs + (size_t)(n0 >> 6) * M;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int b = b_base + mi * 16 + lr;
    const f32x4_t sa = *(const f32x4_t*)(stat_a + 4 * (size_t)min(b, M - 1));
    const float a2 = sa[0], as = sa[1], ai = sa[2], ab = sa[3] + keps;
    float l1 = INFINITY, l2 = INFINITY, up = INFINITY;
    int i1 = -1;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + ni * 16 + 4 * g + j;
        const f32x4_t sw = *(const f32x4_t*)(stat_w + 4 * (size_t)min(n, G - 1));
        const float d2 = a2 + sw[0] - 2.f * (acc[mi][ni][j] * ai * sw[2]) + 2.f * eps * (as - sw[1]) + keps;
        const float dl = kap * (ab + sw[3]);
        const float L = n < G ? d2 - dl : INFINITY, U = n < G ? d2 + dl : INFINITY;
        if (L < l1) { l2 = l1; l1 = L; i1 = n; }   // rows ascend inside the lane: the first of equal L keeps the index,
        else if (L < l2) l2 = L;                   // the second lands in lo2 (= lo1: the whole slot is re-scored)
        up = fminf(up, U);
      }
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const float ol1 = __shfl_xor(l1, o, 64), ol2 = __shfl_xor(l2, o, 64), ou = __shfl_xor(up, o, 64);
      const int oi1 = __shfl_xor(i1, o, 64);
      const float nl2 = fminf(fminf(l2, ol2), fmaxf(l1, ol1));
      if (ol1 < l1) { l1 = ol1; i1 = oi1; }
      l2 = nl2; up = fminf(up, ou);
    }
    if (g == 0 && b < b_end) {
      MatchRec r; r.lo1 = l1; r.idx = i1; r.lo2 = l2; r.up = up;
      out[b] = r;
    }
    __builtin_amdgcn_sched_barrier(0);   // one probe row's loads at a time: no hoisting of the next row's statistics loads
  }
}

// ------------------------------------------------------------------------------------------------
// Shared conv epilogue.  After the K loop a lane holds, per 16x16 MFMA tile (mi, ni), 4 consecutive
// output channels (ni*16 + g*4 ..+3) of ONE pixel (mi*16 + lr).  Storing that directly is 8 bytes
// per lane scattered over 16 pixel rows per instruction (32-byte fragments of 128-byte lines).
// Instead each wave transposes 16 pixels at a time through its own LDS scratch (fp32, row pitch
// NI*64+16 bytes) so that 8 consecutive lanes own one pixel's channel run and every global access
// (residual load, output store) is 16 bytes per lane and a whole line per 8 (or 4) lanes.
//   v = acc + shift[c] (+ residual) ; activation (relu: 0 none, 1 ReLU, 2 GELU) ; round once to the storage dtype.
// Caller guarantees: all waves are past their last read of the LDS tiles (a barrier), `scratch`
// is this wave's private 16*(NI*64+16)-byte region, 16-byte aligned.
// ------------------------------------------------------------------------------------------------
// `rv_pre` (optional): the residual pieces already in registers, rv_pre[mi * PER_LANE + j] = the 16 bytes
// at (pixel m_wave0 + mi*16 + (j*64 + lane) / PARTS, channels co0 + ((j*64 + lane) % PARTS) * 8 ..+7);
// `res` is then only a flag (non-null = add them).
template <typename TT, int MI, int NI>
__device__ __forceinline__ void conv_epilogue(const f32x4_t (&acc)[MI][NI], char* scratch, int m_wave0, int M,
                                              int Cout, int co0, const float* __restrict__ shift,
                                              const typename TT::elem* __restrict__ res,
                                              typename TT::elem* __restrict__ out, int relu, int lane,
                                              const u32x4_t* rv_pre = nullptr) {
  constexpr int PITCH = NI * 64 + 16;      // bytes per pixel row in scratch
  constexpr int PARTS = NI * 2;            // 8-channel runs per pixel
  constexpr int PER_LANE = PARTS / 4;      // (16 px * PARTS) / 64 lanes
  const int lr = lane & 15, g = lane >> 4;
  // this lane's channel run is the same in every pass
  float sh[PER_LANE][8];
  int part[PER_LANE], prow[PER_LANE];
#pragma unroll
  for (int j = 0; j < PER_LANE; ++j) {
    const int idx = j * 64 + lane;
    prow[j] = idx / PARTS;
    part[j] = idx % PARTS;
    const f32x4_t s0 = *(const f32x4_t*)(shift + co0 + part[j] * 8), s1 = *(const f32x4_t*)(shift + co0 + part[j] * 8 + 4);
    sh[j][0] = s0[0]; sh[j][1] = s0[1]; sh[j][2] = s0[2]; sh[j][3] = s0[3];
    sh[j][4] = s1[0]; sh[j][5] = s1[1]; sh[j][6] = s1[2]; sh[j][7] = s1[3];
  }
  // residual: all of this lane's 16-byte pieces are requested up front (one global round trip,
  // not one per pass); they land while the LDS transposes run
  u32x4_t rv[MI][PER_LANE];
  if (res) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int j = 0; j < PER_LANE; ++j) {
        const int m = m_wave0 + mi * 16 + prow[j];
        rv[mi][j] = (u32x4_t){0u, 0u, 0u, 0u};
        if (rv_pre) rv[mi][j] = rv_pre[mi * PER_LANE + j];
        else if (m < M) rv[mi][j] = *(const u32x4_t*)(res + (size_t)m * Cout + co0 + part[j] * 8);
      }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) *(f32x4_t*)(scratch + lr * PITCH + ni * 64 + g * 16) = acc[mi][ni];
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) {
      const f32x4_t a = *(const f32x4_t*)(scratch + prow[j] * PITCH + part[j] * 32);
      const f32x4_t b = *(const f32x4_t*)(scratch + prow[j] * PITCH + part[j] * 32 + 16);
      const int m = m_wave0 + mi * 16 + prow[j];
      if (m < M) {
        const size_t o = (size_t)m * Cout + co0 + part[j] * 8;
        float v[8] = {a[0] + sh[j][0], a[1] + sh[j][1], a[2] + sh[j][2], a[3] + sh[j][3],
                      b[0] + sh[j][4], b[1] + sh[j][5], b[2] + sh[j][6], b[3] + sh[j][7]};
        if (res) {
          float r[8];
          unpack8<TT>(rv[mi][j], r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
        if (relu == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (relu == 2) {  // exact (erf) GELU, nn.GELU() default
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
        }
        *(u32x4_t*)(out + o) = pack8<TT>(v);
      }
    }
  }
}

// 2x2 / stride-2 max-pool fused into the epilogue (`self.pool(F.relu(self.bn(self.conv(x))))`, face_models.py:38-40,
// and SiameseNet's conv -> BN -> ReLU -> MaxPool2d(2) runs, :121-141).  The kernel enumerates its output pixels in
// POOL-MAJOR order - index m = 4 * window + (dy * 2 + dx) - so the 16 pixels of an MFMA column group are 4 whole
// pooling windows and a window's 4 pixels sit in 4 consecutive scratch rows: the pooled value is the max over those
// rows (max and the monotonic shift + ReLU commute: shift + act are applied once, after the max).  Writes pooled
// pixel (mp_wave0 + mi * 4 + window) as whole 16-byte runs like conv_epilogue.
template <typename TT, int MI, int NI>
__device__ __forceinline__ void conv_epilogue_pool2(const f32x4_t (&acc)[MI][NI], char* scratch, int mp_wave0, int MP,
                                                    int Cout, int co0, const float* __restrict__ shift,
                                                    typename TT::elem* __restrict__ out, int relu, int lane) {
  constexpr int PITCH = NI * 64 + 16, PARTS = NI * 2;
  const int lr = lane & 15, g = lane >> 4;
  // every lane reads (lanes >= 4 * PARTS redo windows 0..3 and store nothing): a lane that only ever wrote the scratch
  // would let the compiler drop all but its last write (its own view has no read in between) - seen in the ISA
  const bool active = lane < 4 * PARTS;
  const int win = (lane / PARTS) & 3, part = lane % PARTS;
  float sh[8];
  {
    const f32x4_t s0 = *(const f32x4_t*)(shift + co0 + part * 8), s1 = *(const f32x4_t*)(shift + co0 + part * 8 + 4);
    sh[0] = s0[0]; sh[1] = s0[1]; sh[2] = s0[2]; sh[3] = s0[3];
    sh[4] = s1[0]; sh[5] = s1[1]; sh[6] = s1[2]; sh[7] = s1[3];
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) *(f32x4_t*)(scratch + lr * PITCH + ni * 64 + g * 16) = acc[mi][ni];
    __builtin_amdgcn_wave_barrier();
    const char* src = scratch + (win * 4) * PITCH + part * 32;
    f32x4_t a = *(const f32x4_t*)src, b = *(const f32x4_t*)(src + 16);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
      const f32x4_t a2 = *(const f32x4_t*)(src + q * PITCH), b2 = *(const f32x4_t*)(src + q * PITCH + 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) { a[e] = fmaxf(a[e], a2[e]); b[e] = fmaxf(b[e], b2[e]); }
    }
    const int mp = mp_wave0 + mi * 4 + win;
    if (active && mp < MP) {
      float v[8] = {a[0] + sh[0], a[1] + sh[1], a[2] + sh[2], a[3] + sh[3], b[0] + sh[4], b[1] + sh[5], b[2] + sh[6], b[3] + sh[7]};
      if (relu == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      *(u32x4_t*)(out + (size_t)mp * Cout + co0 + part * 8) = pack8<TT>(v);
    }
  }
}

// split-K variant of the epilogue: the raw fp32 accumulator tile goes to this K-slice's slab
// ([M][Cout] fp32), transposed through LDS the same way so every store is 16 bytes per lane.
template <int MI, int NI>
__device__ __forceinline__ void conv_epilogue_partial(const f32x4_t (&acc)[MI][NI], char* scratch, int m_wave0, int M,
                                                      int Cout, int co0, float* __restrict__ slab, int lane) {
  constexpr int PITCH = NI * 64 + 16, PARTS = NI * 2, PER_LANE = PARTS / 4;
  const int lr = lane & 15, g = lane >> 4;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) *(f32x4_t*)(scratch + lr * PITCH + ni * 64 + g * 16) = acc[mi][ni];
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) {
      const int idx = j * 64 + lane, prow = idx / PARTS, part = idx % PARTS;
      const f32x4_t a = *(const f32x4_t*)(scratch + prow * PITCH + part * 32);
      const f32x4_t b = *(const f32x4_t*)(scratch + prow * PITCH + part * 32 + 16);
      const int m = m_wave0 + mi * 16 + prow;
      if (m < M) {
        float* o = slab + (size_t)m * Cout + co0 + part * 8;
        *(f32x4_t*)o = a;
        *(f32x4_t*)(o + 4) = b;
      }
    }
  }
}

// exact floor(n / d) for n, d < 65536 with magic = ceil(2^32 / d)
__host__ __device__ inline uint32_t frmap_magic(uint32_t d) {
  return (uint32_t)(((1ull << 32) + d - 1) / d);
}
__device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t magic) { return __umulhi(n, magic); }

// exact floor(n / d) for 0 <= n < 2^31 and a run-time d >= 1 prepared on the host: a hardware-less integer division
// costs ~35 VALU instructions, this one a mul_hi and a shift.  d >= 2: k = ceil(log2 d), m = ceil(2^(31+k) / d) < 2^32,
// q = (n * m) >> (31 + k) = mul_hi(n, m) >> (k - 1)  (error term m*d - 2^(31+k) < d <= 2^k keeps it exact for n < 2^31).
struct FrmapDiv {
  uint32_t m;  // 0: d == 1
  uint32_t sh;
};
__host__ inline FrmapDiv frmap_div_make(uint32_t d) {
  FrmapDiv r = {0u, 0u};
  if (d <= 1) return r;
  uint32_t k = 0;
  while ((1ull << k) < d) ++k;
  r.m = (uint32_t)((((unsigned long long)1 << (31 + k)) + d - 1) / d);
  r.sh = k - 1;
  return r;
}
__device__ __forceinline__ int frmap_div(int n, FrmapDiv d) { return d.m ? (int)(__umulhi((uint32_t)n, d.m) >> d.sh) : n; }

// pool-major pixel order (conv_epilogue_pool2): m = 4 * w + q, window w = (n * Ho/2 + py) * Wo/2 + px row-major over the
// POOLED map, q = dy * 2 + dx inside the 2x2 window  ->  output pixel (n, 2 py + dy, 2 px + dx)
struct FrmapPoolOrder {
  FrmapDiv dWin, dWo2;  // divisions by (Ho/2 * Wo/2) and Wo/2
  int Win, Wo2;
};
__device__ __forceinline__ void frmap_pool_coords(int m, const FrmapPoolOrder& o, int& n, int& oy, int& ox) {
  const int w = m >> 2, q = m & 3;
  n = frmap_div(w, o.dWin);
  const int rem = w - n * o.Win;
  const int py = frmap_div(rem, o.dWo2);
  oy = 2 * py + (q >> 1);
  ox = 2 * (rem - py * o.Wo2) + (q & 1);
}

// host-side error plumbing
void frmap_set_error(const char* fmt, ...);
// raise a kernel's dynamic-LDS limit on the CURRENT device (once per (kernel, device)); 0 or -2 with the error set
int frmap_big_lds(const void* kern, int bytes);
// second-generation fused ResNet stem (stem_s2d.hip): 1 = launched, 0 = shape not taken, < 0 = error
int frmap_stem_s2d(const float* x_nchw, const unsigned char* x_u8, const float* mean3, const float* std3, const void* w_packed_c3,
                   const float* shift, void* out, int B, int Hi, int Wi, int dtype, hipStream_t st);
int frmap_batch_invariant();   // 1: planners must not look at the batch size (c_api.cpp)
// second-generation 3x3 stride-1 kernel (conv_pp.hip): 1 = launched, 0 = shape not taken, < 0 = error
struct FrmapPPShortcut {  // fused 1x1 projection shortcut: out += W . x[n, oy * stride, ox * stride, :]
  const void* in;
  const void* w;  // packed as a 1x1 conv
  int Hi, Wi, Cin, stride;
};
int frmap_conv3x3_pp(const void* in, const void* w_packed, const float* shift, const void* residual, void* out, int B, int Hi,
                     int Wi, int Cin, int Cout, int relu, int dtype, hipStream_t st, const FrmapPPShortcut* ds);
// the same for 3x3 stride-2 pad-1 layers with even input sizes (conv3x3s2_pp_kernel)
int frmap_conv3x3s2_pp(const void* in, const void* w_packed, const float* shift, const void* residual, void* out, int B, int Hi,
                       int Wi, int Cin, int Cout, int relu, int dtype, hipStream_t st);
// conv3x3 s1 p1 + MaxPool2d(2, 2) on the same pipeline (conv3x3_pp_kernel<..., PL = true>)
int frmap_conv3x3_pp_pool(const void* in, const void* w_packed, const float* shift, void* out, int B, int Hi, int Wi, int Cin,
                          int Cout, int relu, int dtype, hipStream_t st);
// 1x1 conv / Linear on the LDS-DMA ping-pong pipeline (conv1x1_pp_kernel): 1 = launched, 0 = not taken, < 0 = error
int frmap_conv1x1_pp(const void* in, const void* w_packed, const float* shift, const void* residual, void* out, int B, int Hi,
                     int Wi, int Cin, int Cout, int stride, int relu, int dtype, hipStream_t st);
int frmap_match_gemm_pp(const void* probes3, const void* gallery_packed, const float* stat_a, const float* stat_w,
                        MatchRec* recs, int B, int G, int Gpad, int D, hipStream_t st);
// top-1 match GEMM on the 1x1 MFMA kernel (conv_igemm.hip), see frmap_match_top1_packed
int frmap_match_gemm_f16x3(const void* probes3, const void* gallery_packed, const float* stat_a, const float* stat_w,
                           MatchRec* recs, int B, int G, int D, hipStream_t st);
#define FRMAP_REQUIRE(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      frmap_set_error(__VA_ARGS__);     \
      return -1;                        \
    }                                   \
  } while (0)
#define FRMAP_LAUNCH_CHECK()                                           \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      frmap_set_error("launch failed: %s", hipGetErrorString(e__));    \
      return -2;                                                       \
    }                                                                  \
  } while (0)
