// Token-side kernels of the hybrid CNN-Transformer (gfx950): positional add + LayerNorm, 4-head
// self-attention over the 49 tokens of one face (QK^T and PV on MFMA), token mean + final LayerNorm.
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
// Multi-head self-attention over L <= 64 tokens, head dim 128, on the matrix cores.
// qkv: [B][L][3D] (q | k | v, each D = H*128 wide, as nn.MultiheadAttention's in_proj produces);
// out: [B][L][D].  One workgroup (4 waves) per (face, head); L is padded to 64 with zero rows.
//   S = Q K^T   : MFMA 16x16x32, wave w owns query rows 16w..16w+15 (4 key tiles x 4 k-steps);
//                 both operands are K-contiguous rows of the LDS images Q[64][128], K[64][128].
//   softmax     : a lane holds 4 query rows x 4 key tiles of S (fp32); the key axis lies across
//                 the 16 lanes of its lane group -> in-register + 4 xor-shuffles; keys >= L masked.
//   O = P V     : P (rounded to the storage dtype) goes through a wave-private LDS tile to become
//                 the A operand; V is staged TRANSPOSED (Vt[128][64]) so the B operand is K-contiguous;
//                 8 value tiles x 2 k-steps per wave.  fp32 accumulation throughout.
//   store       : O tile -> wave-private LDS -> 16-byte whole-line stores.
// ------------------------------------------------------------------------------------------------
template <typename TT>
__global__ __launch_bounds__(256) void mha_tokens_kernel(const typename TT::elem* __restrict__ qkv,
                                                         typename TT::elem* __restrict__ out, int L, int D, int H) {
  constexpr int DH = 128, LP = 64;
  constexpr int QP = (DH + 8) * 2;   // byte pitch of a Q / K row (272 B: 16-byte aligned, staggers banks)
  constexpr int VP = (LP + 8) * 2;   // byte pitch of a Vt row   (144 B)
  constexpr int OP = (DH + 8) * 2;   // byte pitch of an O row   (272 B)
  using vec8 = typename TT::vec8;
  using elem = typename TT::elem;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;                    // [64][QP]
  char* Ks = Qs + LP * QP;            // [64][QP]
  char* Vt = Ks + LP * QP;            // [128][VP]
  char* Ps = Vt + DH * VP;            // [4 waves][16][VP]   probabilities of the wave's 16 query rows
  char* Os = Ps + 4 * 16 * VP;        // [4 waves][16][OP]
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, g = lane >> 4;
  const elem* base = qkv + (size_t)b * L * 3 * D + h * DH;

  // ---- stage Q, K (row-major) and V (transposed); rows >= L are zero ---------------------------
  for (int i = tid; i < LP * (DH / 8); i += 256) {
    const int row = i >> 4, c8 = i & 15;  // 16 x 16-byte pieces per 128-wide row
    u32x4_t q = {0u, 0u, 0u, 0u}, k = q, v = q;
    if (row < L) {
      const elem* src = base + (size_t)row * 3 * D + c8 * 8;
      q = *(const u32x4_t*)(src);
      k = *(const u32x4_t*)(src + D);
      v = *(const u32x4_t*)(src + 2 * D);
    }
    *(u32x4_t*)(Qs + row * QP + c8 * 16) = q;
    *(u32x4_t*)(Ks + row * QP + c8 * 16) = k;
    elem ve[8];
    __builtin_memcpy(ve, &v, 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) *(elem*)(Vt + (c8 * 8 + e) * VP + row * 2) = ve[e];
  }
  __syncthreads();

  // ---- S = Q K^T for this wave's 16 query rows ------------------------------------------------
  f32x4_t sacc[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) sacc[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < DH / 32; ++ks) {
    const vec8 qa = *(const vec8*)(Qs + (wave * 16 + lr) * QP + ks * 64 + g * 16);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const vec8 kb = *(const vec8*)(Ks + (nt * 16 + lr) * QP + ks * 64 + g * 16);
      sacc[nt] = TT::mfma(qa, kb, sacc[nt]);  // D[row = query (g*4+r)][col = key (nt*16+lr)]
    }
  }
  // ---- softmax over keys: row r of this lane = query wave*16 + g*4 + r ---------------------------
  const float scale = rsqrtf((float)DH);
  char* pw = Ps + wave * 16 * VP;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v[4], mx = -INFINITY;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      v[nt] = (nt * 16 + lr) < L ? sacc[nt][r] * scale : -INFINITY;
      mx = fmaxf(mx, v[nt]);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      v[nt] = __expf(v[nt] - mx);
      sum += v[nt];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) *(elem*)(pw + (g * 4 + r) * VP + (nt * 16 + lr) * 2) = TT::from_f32(v[nt] * inv);
  }
  // ---- O = P V  (wave-private P tile; same-wave LDS ops are ordered) ---------------------------
  f32x4_t oacc[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) oacc[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < LP / 32; ++ks) {
    const vec8 pa = *(const vec8*)(pw + lr * VP + ks * 64 + g * 16);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const vec8 vb = *(const vec8*)(Vt + (nt * 16 + lr) * VP + ks * 64 + g * 16);
      oacc[nt] = TT::mfma(pa, vb, oacc[nt]);  // D[row = query (g*4+r)][col = d (nt*16+lr)]
    }
  }
  // ---- store through a wave-private tile: 16 rows x 256 B, 16 bytes per lane per pass ------------
  char* ow = Os + wave * 16 * OP;
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) *(elem*)(ow + (g * 4 + r) * OP + (nt * 16 + lr) * 2) = TT::from_f32(oacc[nt][r]);
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int idx = pass * 64 + lane, row = idx >> 4, c8 = idx & 15;
    const int i = wave * 16 + row;
    if (i < L) *(u32x4_t*)(out + ((size_t)b * L + i) * D + h * DH + c8 * 8) = *(const u32x4_t*)(ow + row * OP + c8 * 16);
  }
}

extern "C" int frmap_mha_tokens(const void* qkv, void* out, int B, int L, int D, int H, int dtype, void* stream) {
  FRMAP_REQUIRE(qkv && out, "mha_tokens: null pointer");
  FRMAP_REQUIRE(B > 0 && L > 0 && L <= 64 && H > 0 && D == H * 128, "mha_tokens: need L <= 64 and head dim 128 (D=%d H=%d L=%d)", D, H, L);
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "mha_tokens: bad dtype");
  FRMAP_REQUIRE((long long)B * H < (1ll << 31), "mha_tokens: too many heads");
  const int lds = 2 * 64 * 272 + 128 * 144 + 4 * 16 * 144 + 4 * 16 * 272;
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
