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
  if (D % 512 == 0) {   // 16-byte path: lane owns channels 8 * lane .. + 7 of every 512-channel block
    float v[2][8];
    float s = 0.f;
    const int nb = D >> 9;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i >= nb) break;
      const int d = i * 512 + lane * 8;
      float f[8];
      unpack8<TT>(*(const u32x4_t*)(x + (size_t)row * D + d), f);
      if (pos) {
        const f32x4_t p0 = *(const f32x4_t*)(pos + (size_t)l * D + d), p1 = *(const f32x4_t*)(pos + (size_t)l * D + d + 4);
        f[0] += p0[0]; f[1] += p0[1]; f[2] += p0[2]; f[3] += p0[3]; f[4] += p1[0]; f[5] += p1[1]; f[6] += p1[2]; f[7] += p1[3];
      }
      if (t_out) {
        const u32x4_t r = pack8<TT>(f);
        *(u32x4_t*)(t_out + (size_t)row * D + d) = r;
        unpack8<TT>(r, f);   // the residual stream is stored in `dtype`; normalise what was stored
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[i][j] = f[j]; s += f[j]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i >= nb) break;
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float c = v[i][j] - mean; q += c * c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i >= nb) break;
      const int d = i * 512 + lane * 8;
      const f32x4_t g0 = *(const f32x4_t*)(gamma + d), g1 = *(const f32x4_t*)(gamma + d + 4);
      const f32x4_t b0 = *(const f32x4_t*)(beta + d), b1 = *(const f32x4_t*)(beta + d + 4);
      float y[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y[j] = (v[i][j] - mean) * rstd * g0[j] + b0[j];
        y[4 + j] = (v[i][4 + j] - mean) * rstd * g1[j] + b1[j];
      }
      *(u32x4_t*)(y_out + (size_t)row * D + d) = pack8<TT>(y);
    }
    return;
  }
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
  const void* kern = dtype == FRMAP_BF16 ? (const void*)mha_tokens_kernel<BF16> : (const void*)mha_tokens_kernel<F16>;
  if (frmap_big_lds(kern, 96 * 1024)) return -2;
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
  __shared__ float part_sum[4][1024];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (D + 255) / 256;  // <= 4
  float m[4] = {0.f, 0.f, 0.f, 0.f};
  if (D % 8 == 0 && D <= 1024) {
    // 16-byte loads: 8-channel group c8 x token subset `sub` per thread (D / 8 groups x as many subsets as fit in 256 threads),
    // up to 7 independent loads in flight per thread; the subsets meet in LDS
    const int C8 = D >> 3, nsub = C8 <= 256 ? min(4, 256 / C8) : 1;
    if (tid < C8 * nsub) {
      const int c8 = tid % C8, sub = tid / C8;
      float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      const typename TT::elem* src = t + (size_t)b * L * D + c8 * 8;
      for (int l0 = sub; l0 < L; l0 += 7 * nsub) {
        u32x4_t r[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) r[u] = *(const u32x4_t*)(src + (size_t)min(l0 + u * nsub, L - 1) * D);
#pragma unroll
        for (int u = 0; u < 7; ++u) {
          float v[8];
          unpack8<TT>(r[u], v);
          const float w = l0 + u * nsub < L ? 1.0f : 0.0f;
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] = fmaf(w, v[j], a[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) part_sum[sub][c8 * 8 + j] = a[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = i * 256 + tid;
      if (i < per && d < D)
        for (int sub = 0; sub < nsub; ++sub) m[i] += part_sum[sub][d];
    }
  } else {
    for (int l = 0; l < L; ++l) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = i * 256 + tid;
        if (i < per && d < D) m[i] += TT::to_f32(t[((size_t)b * L + l) * D + d]);
      }
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

// ================================================================================================
// AttentionNet's attention block on the trunk's H x W x C map (`/root/reference/src/face_models.py:194-262`),
// one workgroup per image, everything after the fused 1x1 q/k/v projection in ONE kernel:
//   energy = q . k^T over the L = H*W positions -> softmax rows -> out[i][c] = sum_j attn[i][j] v[j][c]
//   y = gamma * out + x                                   (AttentionModule, `:233-252`)
//   gate = sigmoid(conv KSxKS([mean_c y, max_c y]) + b)    (SpatialAttention, `:194-211`)
//   map = y * gate ; pool = mean over positions            (+ AttentionNet's AdaptiveAvgPool2d(1), `:279`)
// qkv is the packed projection [B][L][2*Cq + C] (q | k | v) a single 1x1 conv emits; thread t owns
// channels CPT t .. CPT t + CPT - 1 of every position (adjacent: one LDS read / one store per position), so y never leaves registers, the channel
// mean / max are a wave reduction + 4-way LDS combine, and the final pooling is thread-local.
// All arithmetic is fp32 on the storage-dtype inputs.
// ================================================================================================
struct CnnAttnParams {
  const void* qkv;
  const void* x;
  const float* gamma;
  const float* sw;   // [2][KS][KS]
  const float* sb;   // [1]
  void* out_map;     // [B][L][C] storage dtype, or null
  float* out_pool;   // [B][C] fp32, or null
  int B, H, W, L, Cq, C, KS;
};

template <typename TT, int CPT, int LMAX>
__global__ __launch_bounds__(256) void cnn_attention_kernel(const CnnAttnParams p) {
  using elem = typename TT::elem;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int L = p.L, Cq = p.Cq, C = p.C, RS = 2 * Cq + C;  // row stride of qkv in elements
  const int QP = Cq + 1;                                     // padded pitch of the fp32 q / k images
  float* qs = (float*)smem;
  float* ks = qs + LMAX * QP;
  float* at = ks + LMAX * QP;            // attention, pitch LMAX + 1
  float* part = at + LMAX * (LMAX + 1);  // [4 waves][2][LMAX] channel sum / max partials
  float* gate = part + 4 * 2 * LMAX;     // [LMAX] mean, [LMAX] max, [LMAX] gate
  elem* vs = (elem*)(gate + 3 * LMAX);   // [L][C] storage dtype
  elem* xs = vs + (size_t)L * C;          // [L][C] the trunk map (residual input), staged with 16-byte loads
  const size_t b = blockIdx.x;
  const elem* qkv = (const elem*)p.qkv + b * (size_t)L * RS;
  const elem* xb = (const elem*)p.x + b * (size_t)L * C;

  for (int i = tid; i < L * Cq; i += 256) {
    const int r = i / Cq, c = i - r * Cq;
    qs[r * QP + c] = TT::to_f32(qkv[(size_t)r * RS + c]);
    ks[r * QP + c] = TT::to_f32(qkv[(size_t)r * RS + Cq + c]);
  }
  for (int i = tid; i < L * (C / 8); i += 256) {
    const int r = i / (C / 8), c8 = i - r * (C / 8);
    *(u32x4_t*)(vs + (size_t)r * C + c8 * 8) = *(const u32x4_t*)(qkv + (size_t)r * RS + 2 * Cq + c8 * 8);
    *(u32x4_t*)(xs + (size_t)r * C + c8 * 8) = *(const u32x4_t*)(xb + (size_t)r * C + c8 * 8);
  }
  __syncthreads();
  for (int e = tid; e < L * L; e += 256) {
    const int i = e / L, j = e - i * L;
    float s = 0.f;
    for (int c = 0; c < Cq; ++c) s = fmaf(qs[i * QP + c], ks[j * QP + c], s);
    at[i * (LMAX + 1) + j] = s;
  }
  __syncthreads();
  if (tid < L) {  // softmax over j, as F.softmax: exp(x - max) / sum
    float* row = at + tid * (LMAX + 1);
    float m = row[0];
    for (int j = 1; j < L; ++j) m = fmaxf(m, row[j]);
    float sum = 0.f;
    for (int j = 0; j < L; ++j) { const float e = expf(row[j] - m); row[j] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int j = 0; j < L; ++j) row[j] *= inv;
  }
  __syncthreads();

  float y[LMAX][CPT];
#pragma unroll
  for (int i = 0; i < LMAX; ++i)
#pragma unroll
    for (int cc = 0; cc < CPT; ++cc) y[i][cc] = 0.f;
  for (int j = 0; j < L; ++j) {
    float vj[CPT];
#pragma unroll
    for (int cc = 0; cc < CPT; ++cc) vj[cc] = TT::to_f32(vs[(size_t)j * C + tid * CPT + cc]);   // CPT adjacent channels: one LDS read
#pragma unroll
    for (int i = 0; i < LMAX; ++i) {
      const float a = i < L ? at[i * (LMAX + 1) + j] : 0.f;  // same address in every lane: an LDS broadcast
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) y[i][cc] = fmaf(a, vj[cc], y[i][cc]);
    }
  }
  const float gamma = p.gamma[0];
#pragma unroll
  for (int i = 0; i < LMAX; ++i) {
    float s = 0.f, m = -INFINITY;
    if (i < L) {
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        y[i][cc] = fmaf(gamma, y[i][cc], TT::to_f32(xs[(size_t)i * C + tid * CPT + cc]));
        s += y[i][cc];
        m = fmaxf(m, y[i][cc]);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s += __shfl_xor(s, o);
      m = fmaxf(m, __shfl_xor(m, o));
    }
    if (lane == 0 && i < L) {
      part[(wave * 2 + 0) * LMAX + i] = s;
      part[(wave * 2 + 1) * LMAX + i] = m;
    }
  }
  __syncthreads();
  if (tid < L) {
    float s = 0.f, m = -INFINITY;
    for (int w = 0; w < 4; ++w) {
      s += part[(w * 2 + 0) * LMAX + tid];
      m = fmaxf(m, part[(w * 2 + 1) * LMAX + tid]);
    }
    gate[tid] = s / (float)C;
    gate[LMAX + tid] = m;
  }
  __syncthreads();
  if (tid < L) {
    const int oy = tid / p.W, ox = tid - oy * p.W, KS = p.KS, pad = KS / 2;
    float g = p.sb[0];
    for (int ch = 0; ch < 2; ++ch)
      for (int dy = 0; dy < KS; ++dy) {
        const int iy = oy + dy - pad;
        if ((unsigned)iy >= (unsigned)p.H) continue;
        for (int dx = 0; dx < KS; ++dx) {
          const int ix = ox + dx - pad;
          if ((unsigned)ix < (unsigned)p.W) g = fmaf(p.sw[(ch * KS + dy) * KS + dx], gate[ch * LMAX + iy * p.W + ix], g);
        }
      }
    gate[2 * LMAX + tid] = 1.0f / (1.0f + expf(-g));
  }
  __syncthreads();
  float pool[CPT];
#pragma unroll
  for (int cc = 0; cc < CPT; ++cc) pool[cc] = 0.f;
  elem* om = p.out_map ? (elem*)p.out_map + b * (size_t)L * C : nullptr;
#pragma unroll
  for (int i = 0; i < LMAX; ++i)
    if (i < L) {
      const float gt = gate[2 * LMAX + i];
      float ov[CPT];
#pragma unroll
      for (int cc = 0; cc < CPT; ++cc) {
        const float o = y[i][cc] * gt;
        pool[cc] += o;
        ov[cc] = o;
      }
      if (om) {
        if (CPT == 2) *(unsigned*)(om + (size_t)i * C + tid * 2) = pack4<TT>(ov[0], ov[CPT - 1], 0.f, 0.f)[0];
        else
#pragma unroll
          for (int cc = 0; cc < CPT; ++cc) om[(size_t)i * C + tid * CPT + cc] = TT::from_f32(ov[cc]);
      }
    }
  if (p.out_pool)
#pragma unroll
    for (int cc = 0; cc < CPT; ++cc) p.out_pool[b * C + tid * CPT + cc] = pool[cc] / (float)L;
}

template <typename TT, int CPT, int LMAX>
static int cnn_attention_launch(const CnnAttnParams& p, hipStream_t st) {
  const size_t lds = (size_t)(2 * LMAX * (p.Cq + 1) + LMAX * (LMAX + 1) + 8 * LMAX + 3 * LMAX) * sizeof(float) +
                     (size_t)2 * p.L * p.C * sizeof(typename TT::elem);
  FRMAP_REQUIRE(lds <= 160 * 1024, "cnn_attention: %zu bytes of LDS needed (> 160 KB)", lds);
  auto kern = cnn_attention_kernel<TT, CPT, LMAX>;
  if (frmap_big_lds((const void*)kern, 160 * 1024)) return -2;
  hipLaunchKernelGGL(kern, dim3(p.B), dim3(256), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 0;
}

extern "C" int frmap_cnn_attention(const void* qkv, const void* x, const float* gamma, const float* spatial_w,
                                   const float* spatial_b, void* out_map, float* out_pool, int B, int H, int W, int Cq,
                                   int C, int KS, int dtype, void* stream) {
  FRMAP_REQUIRE(qkv && x && gamma && spatial_w && spatial_b, "cnn_attention: null pointer");
  FRMAP_REQUIRE(out_map || out_pool, "cnn_attention: no output requested");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "cnn_attention: bad dtype");
  FRMAP_REQUIRE(B > 0 && H > 0 && W > 0 && H * W <= 64, "cnn_attention: need 1 <= H*W <= 64 (got %dx%d)", H, W);
  FRMAP_REQUIRE(Cq > 0 && Cq <= 128 && Cq % 8 == 0, "cnn_attention: Cq=%d must be a multiple of 8, <= 128", Cq);
  FRMAP_REQUIRE(C == 256 || C == 512, "cnn_attention: C=%d must be 256 or 512", C);
  FRMAP_REQUIRE(KS > 0 && KS % 2 == 1 && KS <= 15, "cnn_attention: odd spatial kernel size <= 15 expected (got %d)", KS);
  CnnAttnParams p;
  p.qkv = qkv; p.x = x; p.gamma = gamma; p.sw = spatial_w; p.sb = spatial_b; p.out_map = out_map; p.out_pool = out_pool;
  p.B = B; p.H = H; p.W = W; p.L = H * W; p.Cq = Cq; p.C = C; p.KS = KS;
  hipStream_t st = (hipStream_t)stream;
  const bool big = p.L > 49;
#define FRMAP_CA(TT)                                                                                  \
  (C == 512 ? (big ? cnn_attention_launch<TT, 2, 64>(p, st) : cnn_attention_launch<TT, 2, 49>(p, st)) \
            : (big ? cnn_attention_launch<TT, 1, 64>(p, st) : cnn_attention_launch<TT, 1, 49>(p, st)))
  return dtype == FRMAP_BF16 ? FRMAP_CA(BF16) : FRMAP_CA(F16);
#undef FRMAP_CA
}
