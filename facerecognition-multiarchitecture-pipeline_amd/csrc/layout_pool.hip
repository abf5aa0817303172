// Layout packing, pooling and cast kernels (HBM-bound, 16-byte vectorised), gfx950.
#include "frmap_common.h"

// ------------------------------------------------------------------------------------------------
// fp32 NCHW (C=3) -> NHWC4 dtype.  One thread = 2 pixels (16 B out); reads are coalesced per plane.
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ void pack_input_kernel(const float* __restrict__ x, typename TT::elem* __restrict__ out, size_t npairs,
                                  size_t HW) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < npairs; i += stride) {
    const size_t pix = i * 2;
    const size_t n = pix / HW, s = pix - n * HW;
    const float* b = x + n * 3 * HW + s;
    const float2 r = *(const float2*)(b), gch = *(const float2*)(b + HW), bl = *(const float2*)(b + 2 * HW);
    const float f[8] = {r.x, gch.x, bl.x, 0.f, r.y, gch.y, bl.y, 0.f};
    *(u32x4_t*)(out + pix * 4) = pack8<TT>(f);
  }
}

// odd pixel counts per image: one pixel per thread (the float2 path needs 8-byte aligned planes)
template <typename TT>
__global__ void pack_input_scalar_kernel(const float* __restrict__ x, typename TT::elem* __restrict__ out, size_t npix,
                                         size_t HW) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < npix; i += stride) {
    const size_t n = i / HW, s = i - n * HW;
    const float* b = x + n * 3 * HW + s;
    typename TT::elem e[4] = {TT::from_f32(b[0]), TT::from_f32(b[HW]), TT::from_f32(b[2 * HW]), TT::from_f32(0.f)};
    u32x2_t w;
    __builtin_memcpy(&w, e, 8);
    *(u32x2_t*)(out + i * 4) = w;
  }
}

extern "C" int frmap_pack_input_nchw_f32(const float* x, void* out, int B, int H, int W, int dtype, void* stream) {
  FRMAP_REQUIRE(x && out, "pack_input: null pointer");
  FRMAP_REQUIRE(B > 0 && H > 0 && W > 0, "pack_input: empty input");
  if ((H * W) % 2 != 0) {
    FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "pack_input: bad dtype");
    const size_t HW1 = (size_t)H * W, npix = (size_t)B * HW1;
    const int blocks1 = (int)((npix + 255) / 256 < 16384 ? (npix + 255) / 256 : 16384);
    if (dtype == FRMAP_BF16)
      hipLaunchKernelGGL(pack_input_scalar_kernel<BF16>, dim3(blocks1), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)out, npix, HW1);
    else
      hipLaunchKernelGGL(pack_input_scalar_kernel<F16>, dim3(blocks1), dim3(256), 0, (hipStream_t)stream, x, (_Float16*)out, npix, HW1);
    FRMAP_LAUNCH_CHECK();
    return 0;
  }
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "pack_input: bad dtype");
  const size_t HW = (size_t)H * W, npairs = (size_t)B * HW / 2;
  const int blocks = (int)((npairs + 255) / 256 < 8192 ? (npairs + 255) / 256 : 8192);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(pack_input_kernel<BF16>, dim3(blocks), dim3(256), 0, st, x, (__bf16*)out, npairs, HW);
  else
    hipLaunchKernelGGL(pack_input_kernel<F16>, dim3(blocks), dim3(256), 0, st, x, (_Float16*)out, npairs, HW);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// conv weight packing into the LDS image order of conv_igemm.hip:
//   [ntile=Cout/64][chunk=Cin/32][tap][cout_l 0..63][slot 0..3][8]   slot = cg ^ (((cout_l>>2)&1)<<1)
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, typename TT::elem* __restrict__ out, size_t total,
                                        int Cin, int KH, int KW) {
  const int taps = KH * KW, nchunks = Cin / 32;
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; e < total; e += stride) {
    const int j = (int)(e & 7), slot = (int)((e >> 3) & 3), cl = (int)((e >> 5) & 63);
    size_t rest = e >> 11;
    const int tap = (int)(rest % taps);
    rest /= taps;
    const int chunk = (int)(rest % nchunks);
    const int ntile = (int)(rest / nchunks);
    const int cg = slot ^ (((cl >> 2) & 1) << 1);
    const int cin = chunk * 32 + cg * 8 + j, cout = ntile * 64 + cl;
    const int kh = tap / KW, kw = tap - kh * KW;
    out[e] = TT::from_f32(w[(((size_t)cout * Cin + cin) * KH + kh) * KW + kw]);
  }
}

extern "C" int frmap_pack_conv_weight(const float* w, void* out, int Cout, int Cin, int KH, int KW, int dtype,
                                      void* stream) {
  FRMAP_REQUIRE(w && out, "pack_conv_weight: null pointer");
  FRMAP_REQUIRE(Cin % 32 == 0 && Cout % 64 == 0 && KH == KW && (KH == 1 || KH == 3),
                "pack_conv_weight: unsupported shape Cout=%d Cin=%d K=%dx%d", Cout, Cin, KH, KW);
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "pack_conv_weight: bad dtype");
  const size_t total = (size_t)Cout * Cin * KH * KW;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(pack_conv_weight_kernel<BF16>, dim3(blocks), dim3(256), 0, st, w, (__bf16*)out, total, Cin, KH, KW);
  else
    hipLaunchKernelGGL(pack_conv_weight_kernel<F16>, dim3(blocks), dim3(256), 0, st, w, (_Float16*)out, total, Cin, KH, KW);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// first-layer (Cin=3) weights: [Cout][KPAD+16], k = kh*KR + kw*4 + c, zeros elsewhere
template <typename TT>
__global__ void pack_conv_weight_c3_kernel(const float* __restrict__ w, typename TT::elem* __restrict__ out, int Cout,
                                           int KH, int KW, int KR, int pitch) {
  const int total = Cout * pitch;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int cout = e / pitch, k = e - cout * pitch;
    const int kh = k / KR, r = k - kh * KR;
    const int kw = r >> 2, c = r & 3;
    float v = 0.f;
    if (kh < KH && kw < KW && c < 3) v = w[(((size_t)cout * 3 + c) * KH + kh) * KW + kw];
    out[e] = TT::from_f32(v);
  }
}

extern "C" int frmap_pack_conv_weight_c3(const float* w, void* out, int Cout, int KH, int KW, int dtype, void* stream) {
  FRMAP_REQUIRE(w && out, "pack_conv_weight_c3: null pointer");
  FRMAP_REQUIRE((KH == 7 && KW == 7) || (KH == 3 && KW == 3), "pack_conv_weight_c3: unsupported kernel %dx%d", KH, KW);
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "pack_conv_weight_c3: bad dtype");
  const int KR = (KW * 4 > 16) ? 32 : 16;
  const int pitch = frmap_small_cin_kpad(KH, KW);
  const int blocks = (Cout * pitch + 255) / 256;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(pack_conv_weight_c3_kernel<BF16>, dim3(blocks), dim3(256), 0, st, w, (__bf16*)out, Cout, KH, KW, KR, pitch);
  else
    hipLaunchKernelGGL(pack_conv_weight_c3_kernel<F16>, dim3(blocks), dim3(256), 0, st, w, (_Float16*)out, Cout, KH, KW, KR, pitch);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// max pool, NHWC, 8 channels (16 B) per thread; padding behaves as -inf (nn.MaxPool2d)
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ void maxpool_kernel(const typename TT::elem* __restrict__ in, typename TT::elem* __restrict__ out, int B,
                               int H, int W, int C, int Ho, int Wo, int k, int s, int pad) {
  const int C8 = C >> 3;
  const size_t total = (size_t)B * Ho * Wo * C8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int c8 = (int)(i % C8);
    size_t r = i / C8;
    const int ox = (int)(r % Wo);
    r /= Wo;
    const int oy = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
    for (int ky = 0; ky < k; ++ky) {
      const int iy = oy * s - pad + ky;
      if ((unsigned)iy >= (unsigned)H) continue;
      for (int kx = 0; kx < k; ++kx) {
        const int ix = ox * s - pad + kx;
        if ((unsigned)ix >= (unsigned)W) continue;
        float f[8];
        unpack8<TT>(*(const u32x4_t*)(in + (((size_t)n * H + iy) * W + ix) * C + c8 * 8), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], f[j]);
      }
    }
    *(u32x4_t*)(out + i * 8) = pack8<TT>(m);
  }
}

extern "C" int frmap_maxpool(const void* in, void* out, int B, int H, int W, int C, int k, int stride, int pad,
                             int dtype, void* stream) {
  FRMAP_REQUIRE(in && out, "maxpool: null pointer");
  FRMAP_REQUIRE(C % 8 == 0 && k >= 1 && stride >= 1 && pad >= 0 && pad * 2 <= k, "maxpool: bad geometry");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "maxpool: bad dtype");
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  FRMAP_REQUIRE(B > 0 && Ho > 0 && Wo > 0, "maxpool: empty output");
  const size_t total = (size_t)B * Ho * Wo * (C / 8);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(maxpool_kernel<BF16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)in, (__bf16*)out, B, H, W, C, Ho, Wo, k, stride, pad);
  else
    hipLaunchKernelGGL(maxpool_kernel<F16>, dim3(blocks), dim3(256), 0, st, (const _Float16*)in, (_Float16*)out, B, H, W, C, Ho, Wo, k, stride, pad);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// global average pool: [B][HW][C] dtype -> fp32 [B][C].  One wave per (n, 8-channel group) slice
// of HW, lanes stride over pixels, wave-reduce.
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ void avgpool_global_kernel(const typename TT::elem* __restrict__ in, float* __restrict__ out, int B, int HW,
                                      int C) {
  const int C8 = C >> 3;
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wid >= B * C8) return;
  const int n = wid / C8, c8 = wid - n * C8;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = lane; s < HW; s += 64) {
    float f[8];
    unpack8<TT>(*(const u32x4_t*)(in + ((size_t)n * HW + s) * C + c8 * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += f[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = a[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    a[j] = v;
  }
  if (lane == 0) {
    const float inv = 1.0f / (float)HW;
#pragma unroll
    for (int j = 0; j < 8; ++j) out[(size_t)n * C + c8 * 8 + j] = a[j] * inv;
  }
}

extern "C" int frmap_avgpool_global(const void* in, float* out, int B, int HW, int C, int dtype, void* stream) {
  FRMAP_REQUIRE(in && out, "avgpool_global: null pointer");
  FRMAP_REQUIRE(B > 0 && HW > 0 && C > 0 && C % 8 == 0, "avgpool_global: bad shape");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "avgpool_global: bad dtype");
  const int waves = B * (C / 8);
  const int blocks = (waves + 3) / 4;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(avgpool_global_kernel<BF16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)in, out, B, HW, C);
  else
    hipLaunchKernelGGL(avgpool_global_kernel<F16>, dim3(blocks), dim3(256), 0, st, (const _Float16*)in, out, B, HW, C);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// adaptive average pool (nn.AdaptiveAvgPool2d((OH,OW))): window [floor(i*H/OH), ceil((i+1)*H/OH))
template <typename TT>
__global__ void avgpool_adaptive_kernel(const typename TT::elem* __restrict__ in, typename TT::elem* __restrict__ out,
                                        int B, int H, int W, int C, int OH, int OW) {
  const int C8 = C >> 3;
  const size_t total = (size_t)B * OH * OW * C8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c8 = (int)(i % C8);
  size_t r = i / C8;
  const int ox = (int)(r % OW);
  r /= OW;
  const int oy = (int)(r % OH);
  const int n = (int)(r / OH);
  const int y0 = (oy * H) / OH, y1 = ((oy + 1) * H + OH - 1) / OH;
  const int x0 = (ox * W) / OW, x1 = ((ox + 1) * W + OW - 1) / OW;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      float f[8];
      unpack8<TT>(*(const u32x4_t*)(in + (((size_t)n * H + y) * W + x) * C + c8 * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += f[j];
    }
  const float inv = 1.0f / (float)((y1 - y0) * (x1 - x0));
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] *= inv;
  *(u32x4_t*)(out + i * 8) = pack8<TT>(a);
}

extern "C" int frmap_avgpool_adaptive(const void* in, void* out, int B, int H, int W, int C, int OH, int OW,
                                      int dtype, void* stream) {
  FRMAP_REQUIRE(in && out, "avgpool_adaptive: null pointer");
  FRMAP_REQUIRE(B > 0 && H > 0 && W > 0 && C % 8 == 0 && OH > 0 && OW > 0, "avgpool_adaptive: bad shape");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "avgpool_adaptive: bad dtype");
  const size_t total = (size_t)B * OH * OW * (C / 8);
  const int blocks = (int)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(avgpool_adaptive_kernel<BF16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)in, (__bf16*)out, B, H, W, C, OH, OW);
  else
    hipLaunchKernelGGL(avgpool_adaptive_kernel<F16>, dim3(blocks), dim3(256), 0, st, (const _Float16*)in, (_Float16*)out, B, H, W, C, OH, OW);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// casts
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ void cast_to_f32_kernel(const typename TT::elem* __restrict__ in, float* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = TT::to_f32(in[i]);
}
template <typename TT>
__global__ void cast_from_f32_kernel(const float* __restrict__ in, typename TT::elem* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = TT::from_f32(in[i]);
}

extern "C" int frmap_cast_to_f32(const void* in, float* out, size_t n, int dtype, void* stream) {
  FRMAP_REQUIRE(in && out, "cast_to_f32: null pointer");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "cast_to_f32: bad dtype");
  if (n == 0) return 0;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(cast_to_f32_kernel<BF16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)in, out, n);
  else
    hipLaunchKernelGGL(cast_to_f32_kernel<F16>, dim3(blocks), dim3(256), 0, st, (const _Float16*)in, out, n);
  FRMAP_LAUNCH_CHECK();
  return 0;
}
extern "C" int frmap_cast_from_f32(const float* in, void* out, size_t n, int dtype, void* stream) {
  FRMAP_REQUIRE(in && out, "cast_from_f32: null pointer");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "cast_from_f32: bad dtype");
  if (n == 0) return 0;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(cast_from_f32_kernel<BF16>, dim3(blocks), dim3(256), 0, st, in, (__bf16*)out, n);
  else
    hipLaunchKernelGGL(cast_from_f32_kernel<F16>, dim3(blocks), dim3(256), 0, st, in, (_Float16*)out, n);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// uint8 HWC RGB -> normalised tensor: transforms.ToTensor() (u/255) + transforms.Normalize(mean, std)
// ((x - mean)/std), the step in front of the hot path (src/testing.py:99-104; src/app.py:39-42).
// One thread = one pixel.  Writes fp32 NCHW (what the reference modules take) and/or NHWC4 `dtype`
// (what the first-layer kernels take) — 3 bytes in per pixel instead of 12.
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ void normalize_u8_kernel(const unsigned char* __restrict__ img, float* __restrict__ out_nchw,
                                    typename TT::elem* __restrict__ out_nhwc4, size_t npix, size_t HW, float m0,
                                    float m1, float m2, float s0, float s1, float s2) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < npix; i += stride) {
    const float r = ((float)img[i * 3 + 0] / 255.0f - m0) / s0;
    const float g = ((float)img[i * 3 + 1] / 255.0f - m1) / s1;
    const float b = ((float)img[i * 3 + 2] / 255.0f - m2) / s2;
    if (out_nchw) {
      const size_t n = i / HW, sp = i - n * HW;
      float* o = out_nchw + n * 3 * HW + sp;
      o[0] = r; o[HW] = g; o[2 * HW] = b;
    }
    if (out_nhwc4) {
      typename TT::elem e[4] = {TT::from_f32(r), TT::from_f32(g), TT::from_f32(b), TT::from_f32(0.f)};
      u32x2_t w;
      __builtin_memcpy(&w, e, 8);
      *(u32x2_t*)(out_nhwc4 + i * 4) = w;
    }
  }
}

extern "C" int frmap_normalize_u8_hwc(const unsigned char* img, float* out_nchw_f32, void* out_nhwc4, int B, int H,
                                      int W, const float* mean3_host, const float* std3_host, int dtype, void* stream) {
  FRMAP_REQUIRE(img && (out_nchw_f32 || out_nhwc4) && mean3_host && std3_host, "normalize_u8_hwc: null pointer");
  FRMAP_REQUIRE(B > 0 && H > 0 && W > 0, "normalize_u8_hwc: empty input");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "normalize_u8_hwc: bad dtype");
  FRMAP_REQUIRE(std3_host[0] != 0.f && std3_host[1] != 0.f && std3_host[2] != 0.f, "normalize_u8_hwc: zero std");
  const size_t HW = (size_t)H * W, npix = (size_t)B * HW;
  const int blocks = (int)((npix + 255) / 256 < 16384 ? (npix + 255) / 256 : 16384);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(normalize_u8_kernel<BF16>, dim3(blocks), dim3(256), 0, st, img, out_nchw_f32, (__bf16*)out_nhwc4, npix, HW,
                       mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
  else
    hipLaunchKernelGGL(normalize_u8_kernel<F16>, dim3(blocks), dim3(256), 0, st, img, out_nchw_f32, (_Float16*)out_nhwc4, npix, HW,
                       mean3_host[0], mean3_host[1], mean3_host[2], std3_host[0], std3_host[1], std3_host[2]);
  FRMAP_LAUNCH_CHECK();
  return 0;
}
