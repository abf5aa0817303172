// Token-side kernels of the hybrid CNN-Transformer (gfx950): positional add + LayerNorm, 4-head
// self-attention over the 49 tokens of one face, token mean + final LayerNorm.
//
// Replaces, for eval mode, the non-GEMM parts of TransformerBlock.forward and HybridNet.get_embedding
// (/root/reference/src/face_models.py:636-648, 705-721): nn.LayerNorm (eps 1e-5), the
// scaled-dot-product attention inside nn.MultiheadAttention(512, 4) (softmax(QK^T/sqrt(128))V per
// head), `feats.mean(dim=0)` and the output LayerNorm.  The projections / MLP run on the MFMA conv
// kernel as 1x1 convolutions (conv_igemm.hip) with bias / GELU / residual fused in its epilogue.
//
// Layout: tokens are [B][L][D] (face-major) — exactly the NHWC trunk output B x 7 x 7 x 512 viewed as
// B x 49 x 512 — instead of the reference's L x B x D; every op here is per (face, token) or per face,
// so the permutation is free.
#include "frmap_common.h"

// ------------------------------------------------------------------------------------------------
// t = x (+ pos[l]) ; y = LayerNorm(t) * gamma + beta.   One wave per token row; D % 64 == 0, D <= 1024.
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ void add_pos_layernorm_kernel(const typename TT::elem* __restrict__ x, const float* __restrict__ pos,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         typename TT::elem* __restrict__ t_out, typename TT::elem* __restrict__ y_out,
                                         int rows, int L, int D, float eps) {
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int l = row % L;
  const int per = D >> 6;  // elements per lane (<= 16), strided by 64 so loads stay coalesced
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i >= per) break;
    const int d = i * 64 + lane;
    float f = TT::to_f32(x[(size_t)row * D + d]);
    if (pos) f += pos[(size_t)l * D + d];
    if (t_out) {
      const typename TT::elem r = TT::from_f32(f);
      t_out[(size_t)row * D + d] = r;
      f = TT::to_f32(r);  // the residual stream is stored in `dtype`; normalise what was stored
    }
    v[i] = f;
    s += f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i >= per) break;
    const float c = v[i] - mean;
    q += c * c;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i >= per) break;
    const int d = i * 64 + lane;
    y_out[(size_t)row * D + d] = TT::from_f32((v[i] - mean) * rstd * gamma[d] + beta[d]);
  }
}

extern "C" int frmap_add_pos_layernorm(const void* x, const float* pos, const float* gamma, const float* beta,
                                       void* t_out, void* y_out, int B, int L, int D, float eps, int dtype,
                                       void* stream) {
  FRMAP_REQUIRE(x && gamma && beta && y_out, "add_pos_layernorm: null pointer");
  FRMAP_REQUIRE(B > 0 && L > 0 && D > 0 && D % 64 == 0 && D <= 1024, "add_pos_layernorm: bad shape (D %% 64 == 0, D <= 1024)");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "add_pos_layernorm: bad dtype");
  const long long rows = (long long)B * L;
  FRMAP_REQUIRE(rows < (1ll << 31) / 64, "add_pos_layernorm: too many tokens");
  const int blocks = (int)((rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(add_pos_layernorm_kernel<BF16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)x, pos, gamma, beta,
                       (__bf16*)t_out, (__bf16*)y_out, (int)rows, L, D, eps);
  else
    hipLaunchKernelGGL(add_pos_layernorm_kernel<F16>, dim3(blocks), dim3(256), 0, st, (const _Float16*)x, pos, gamma, beta,
                       (_Float16*)t_out, (_Float16*)y_out, (int)rows, L, D, eps);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Multi-head self-attention over L <= 64 tokens, head dim 128.  qkv: [B][L][3D] (q | k | v, each D =
// H*128 wide, as nn.MultiheadAttention's in_proj produces); out: [B][L][D].  One workgroup per
// (face, head): K and V of the head sit in LDS as fp32 (pitch 129 -> conflict-free), each wave
// takes query rows round-robin; lane j scores key j, softmax is a wave reduction, P stays in LDS,
// and each lane then produces two of the 128 output dims.  fp32 math throughout.
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ __launch_bounds__(256) void mha_tokens_kernel(const typename TT::elem* __restrict__ qkv,
                                                         typename TT::elem* __restrict__ out, int L, int D, int H) {
  constexpr int DH = 128, PITCH = DH + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ks = (float*)smem;             // [L][PITCH]
  float* Vs = Ks + 64 * PITCH;          // [L][PITCH]
  float* Qs = Vs + 64 * PITCH;          // [4 waves][DH]
  float* Ps = Qs + 4 * DH;              // [4 waves][64]
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const typename TT::elem* base = qkv + (size_t)b * L * 3 * D;
  for (int i = tid; i < L * DH; i += 256) {
    const int j = i / DH, d = i - j * DH;
    Ks[j * PITCH + d] = TT::to_f32(base[(size_t)j * 3 * D + D + h * DH + d]);
    Vs[j * PITCH + d] = TT::to_f32(base[(size_t)j * 3 * D + 2 * D + h * DH + d]);
  }
  __syncthreads();
  const float scale = rsqrtf((float)DH);
  float* q = Qs + wave * DH;
  float* pr = Ps + wave * 64;
  for (int i = wave; i < L; i += 4) {
    q[lane] = TT::to_f32(base[(size_t)i * 3 * D + h * DH + lane]) * scale;
    q[lane + 64] = TT::to_f32(base[(size_t)i * 3 * D + h * DH + lane + 64]) * scale;
    // scores: lane j <-> key j   (same-wave LDS ops are ordered; no barrier needed)
    float sc = -INFINITY;
    if (lane < L) {
      float a = 0.f;
#pragma unroll 8
      for (int d = 0; d < DH; ++d) a += q[d] * Ks[lane * PITCH + d];
      sc = a;
    }
    float mx = sc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float e = lane < L ? __expf(sc - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    pr[lane] = e / sum;
    // out[i][d] = sum_j P[j] V[j][d], two dims per lane
    float o0 = 0.f, o1 = 0.f;
    for (int j = 0; j < L; ++j) {
      const float pj = pr[j];
      o0 += pj * Vs[j * PITCH + lane];
      o1 += pj * Vs[j * PITCH + lane + 64];
    }
    typename TT::elem* dst = out + ((size_t)b * L + i) * D + h * DH;
    dst[lane] = TT::from_f32(o0);
    dst[lane + 64] = TT::from_f32(o1);
  }
}

extern "C" int frmap_mha_tokens(const void* qkv, void* out, int B, int L, int D, int H, int dtype, void* stream) {
  FRMAP_REQUIRE(qkv && out, "mha_tokens: null pointer");
  FRMAP_REQUIRE(B > 0 && L > 0 && L <= 64 && H > 0 && D == H * 128, "mha_tokens: need L <= 64 and head dim 128 (D=%d H=%d L=%d)", D, H, L);
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "mha_tokens: bad dtype");
  FRMAP_REQUIRE((long long)B * H < (1ll << 31), "mha_tokens: too many heads");
  const int lds = (2 * 64 * 129 + 4 * 128 + 4 * 64) * 4;
  hipStream_t st = (hipStream_t)stream;
  static bool attr[2] = {false, false};
  const void* kern = dtype == FRMAP_BF16 ? (const void*)mha_tokens_kernel<BF16> : (const void*)mha_tokens_kernel<F16>;
  if (!attr[dtype]) {
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) {
      frmap_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return -2;
    }
    attr[dtype] = true;
  }
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(mha_tokens_kernel<BF16>, dim3(B * H), dim3(256), lds, st, (const __bf16*)qkv, (__bf16*)out, L, D, H);
  else
    hipLaunchKernelGGL(mha_tokens_kernel<F16>, dim3(B * H), dim3(256), lds, st, (const _Float16*)qkv, (_Float16*)out, L, D, H);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// out[b] = LayerNorm(mean_l t[b][l][:]) * gamma + beta   (fp32 out).  One workgroup per face.
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ __launch_bounds__(256) void mean_layernorm_kernel(const typename TT::elem* __restrict__ t,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ out,
                                                             int L, int D, float eps) {
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (D + 255) / 256;  // <= 4
  float m[4] = {0.f, 0.f, 0.f, 0.f};
  for (int l = 0; l < L; ++l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = i * 256 + tid;
      if (i < per && d < D) m[i] += TT::to_f32(t[((size_t)b * L + l) * D + d]);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    m[i] /= (float)L;
    if (i < per && i * 256 + tid < D) s += m[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < per && i * 256 + tid < D) q += (m[i] - mean) * (m[i] - mean);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  if (lane == 0) red[4 + wave] = q;
  __syncthreads();
  const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int d = i * 256 + tid;
    if (i < per && d < D) out[(size_t)b * D + d] = (m[i] - mean) * rstd * gamma[d] + beta[d];
  }
}

extern "C" int frmap_mean_layernorm(const void* t, const float* gamma, const float* beta, float* out, int B, int L,
                                    int D, float eps, int dtype, void* stream) {
  FRMAP_REQUIRE(t && gamma && beta && out, "mean_layernorm: null pointer");
  FRMAP_REQUIRE(B > 0 && L > 0 && D > 0 && D <= 1024, "mean_layernorm: bad shape (D <= 1024)");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "mean_layernorm: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == FRMAP_BF16)
    hipLaunchKernelGGL(mean_layernorm_kernel<BF16>, dim3(B), dim3(256), 0, st, (const __bf16*)t, gamma, beta, out, L, D, eps);
  else
    hipLaunchKernelGGL(mean_layernorm_kernel<F16>, dim3(B), dim3(256), 0, st, (const _Float16*)t, gamma, beta, out, L, D, eps);
  FRMAP_LAUNCH_CHECK();
  return 0;
}
