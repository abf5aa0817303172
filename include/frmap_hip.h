/*
 * frmap_hip.h — C ABI of the MI355X (gfx950) embedding-extraction + gallery-matching kernels.
 *
 * The reference (henryhcooperr/FaceRecognition-MultiArchitecture-Pipeline) has no FFI / plugin
 * boundary of its own: its hot path is stock torch.nn ops called from Python
 * (src/face_models.py, src/app.py).  This header is the boundary a maintainer binds instead; each
 * entry point names the reference lines whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host"; the library never allocates,
 *     frees or synchronises: outputs and scratch are caller-owned, launches are asynchronous on
 *     `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - return value: 0 = launched, negative = rejected before any launch (frmap_last_error()
 *     gives the text); nothing is thrown across the ABI;
 *   - `dtype`: FRMAP_BF16 or FRMAP_F16 = storage + MFMA input type of activations / packed conv
 *     weights (accumulation is always fp32); heads and matching are fp32 throughout;
 *   - activations are NHWC ("channels last"), C a multiple of 32 for frmap_conv_igemm;
 *   - threading / devices (the reference calls model(x) from a daemon thread while the main thread
 *     matches, src/app.py:331-335,639): every entry point may be called from any host thread; like
 *     every HIP launch it targets the calling thread's CURRENT device, so the caller makes the
 *     device that owns the pointers and `stream` current first (hipSetDevice; the Python binding
 *     does it per call from the operands' device).  Per-kernel launch attributes are kept per
 *     (kernel, device), so one process may drive several GPUs.
 */
#ifndef FRMAP_HIP_H
#define FRMAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRMAP_BF16 0
#define FRMAP_F16 1

/* Batch-invariant planning.  By default the conv / Linear planners pick a layer's kernel and tile layout from the number of
 * tiles the launch would have (pixel split, in-workgroup split-K, first- or second-generation kernel), i.e. from the batch size:
 * the same face then gets different fp32 summation orders - results equal to rounding, not to the bit - in batches of
 * different sizes.  on = 1: layouts are chosen from the per-image geometry alone (never split-K by tile count), so a face's
 * embedding and match are bit-identical whatever batch, shard or rank it is computed in, at some cost in speed for small
 * batches.  on = 0: default planning; on = -1: back to the environment (FRMAP_BATCH_INVARIANT=1).  Process-wide. */
int frmap_set_batch_invariant(int on);

/* ABI version of this header (bumped on any signature change). */
int frmap_abi_version(void);
/* Text of the last rejected call on this thread ("" if none). Host pointer, do not free. */
const char* frmap_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Input layout: fp32 NCHW B×3×H×W  ->  NHWC4 (3 channels + one zero channel) in `dtype`.
 * Replaces the implicit layout the reference feeds nn.Conv2d with (src/testing.py:99-104,251).
 * ------------------------------------------------------------------------------------------- */
int frmap_pack_input_nchw_f32(const float* x_nchw, void* out_nhwc4, int B, int H, int W,
                              int dtype, void* stream);

/* uint8 HWC RGB B×H×W×3 -> ToTensor (u/255) + Normalize ((x-mean)/std), src/testing.py:99-104,
 * src/app.py:39-42.  Writes fp32 NCHW (out_nchw_f32) and/or NHWC4 `dtype` (out_nhwc4); either may be
 * NULL.  mean3_host / std3_host: 3 floats each in HOST memory. */
int frmap_normalize_u8_hwc(const unsigned char* img_u8, float* out_nchw_f32, void* out_nhwc4, int B, int H,
                           int W, const float* mean3_host, const float* std3_host, int dtype, void* stream);

/* `transforms.Resize((H, W))` for PIL-style 8-bit RGB images (src/testing.py:99-100, 561-566; src/training.py:305-310;
 * src/app.py:39), batched, bit-exact with Pillow's two-pass bilinear resampler (libImaging/Resample.c: integer FIR taps
 * with 22 fractional bits, the horizontal pass rounded to 8 bits before the vertical pass).  src_pool: the images' bytes
 * (HWC uint8 RGB, native sizes) back to back; items: B records
 *     struct { uint64 src_off; int32 H, W, bx_off, kx_off, ksx, by_off, ky_off, ksy; }      (40 bytes, device memory)
 * with offsets (int32 elements) into `tables` of the per-axis bounds [out][2] = (first input sample, taps) and coefficients
 * [out][ks] (ks = 0: that axis keeps its size) - built by the host in float64 with Pillow's operation order
 * (resize.py:bilinear_coeffs).  out: B x out_h x out_w x 3 uint8.  One workgroup resizes rows_per_block output rows;
 * lds_rows = the most input rows one workgroup touches (lds_rows * out_w * 4 bytes of LDS). */
int frmap_resize_bilinear_u8(const unsigned char* src_pool, const void* items, const int* tables, unsigned char* out,
                             int B, int out_h, int out_w, int rows_per_block, int lds_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Conv weight packing.  `w_oihw` = fp32 [Cout][Cin][KH][KW] with the BatchNorm scale already
 * folded in (w * gamma/sqrt(var+eps)); output is the kernel's LDS-image order in `dtype`.
 *   frmap_pack_conv_weight      : for frmap_conv_igemm      (Cin % 32 == 0, Cout % 64 == 0,
 *                                 KH == KW in {1,3});  out elems = Cout*Cin*KH*KW
 *   frmap_pack_conv_weight_c3   : for frmap_conv_small_cin  (Cin == 3);  out elems =
 *                                 Cout * frmap_small_cin_kpad(KH, KW)
 * ------------------------------------------------------------------------------------------- */
int frmap_pack_conv_weight(const float* w_oihw, void* w_packed, int Cout, int Cin, int KH, int KW,
                           int dtype, void* stream);
int frmap_small_cin_kpad(int KH, int KW);
int frmap_pack_conv_weight_c3(const float* w_oihw, void* w_packed, int Cout, int KH, int KW,
                              int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * First-layer convolution on NHWC4 input (Cin = 3) + per-channel shift + optional ReLU.
 *   ResNet/Siamese stem 7x7 s2 p3 -> 64   (torchvision resnet18.conv1/bn1/relu via
 *                                          src/face_models.py:67,463,658; face_models.py:115-117)
 *   BaselineNet conv1 3x3 s1 p1 -> 32     (src/face_models.py:21-22,38)
 * Implicit GEMM on v_mfma_f32_16x16x32 with the K axis laid along (kw, c) of each kernel row.
 * out: NHWC B×Ho×Wo×Cout in `dtype`.  Cout in {32, 64}.
 * ------------------------------------------------------------------------------------------- */
int frmap_conv_small_cin(const void* in_nhwc4, const void* w_packed, const float* shift, void* out,
                         int B, int Hi, int Wi, int Cout, int KH, int KW, int stride, int pad,
                         int relu, int dtype, void* stream);
/* The 3x3 s1 p1 3->32 layer with MaxPool2d(2, 2) fused into its epilogue: BaselineNet
 * `self.pool(F.relu(self.bn1(self.conv1(x))))` (src/face_models.py:38) in one launch.  The kernel walks
 * its output pixels in pool-major order (4 consecutive pixels = one 2x2 window), so the pooled value is
 * a max over 4 accumulator rows; the 32×H×W conv map never reaches HBM.  Even Hi, Wi.
 * out: B×(Hi/2)×(Wi/2)×32. */
int frmap_conv_small_cin_pool2(const void* in_nhwc4, const void* w_packed, const float* shift, void* out,
                               int B, int Hi, int Wi, int Cout, int relu, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused ResNet stem: fp32 NCHW B×3×Hi×Wi -> conv 7x7 s2 p3 (3->64) + shift + ReLU -> maxpool 3x3 s2 p1
 * -> NHWC B×Hq×Wq×64 in `dtype`, one kernel (the 64×112×112 conv map never reaches HBM).
 * Replaces conv1/bn1/relu/maxpool of torchvision resnet18 (src/face_models.py:67,463,658) plus the
 * input layout cast.  w_packed_c3: from frmap_pack_conv_weight_c3(…, 64, 7, 7).  Needs Wi <= ~224
 * (pooled width <= 56); wider inputs use frmap_pack_input + frmap_conv_small_cin + frmap_maxpool.
 * ------------------------------------------------------------------------------------------- */
int frmap_stem7x7_maxpool(const float* x_nchw, const void* w_packed_c3, const float* shift, void* out,
                          int B, int Hi, int Wi, int dtype, void* stream);
/* Same fusion with MaxPool2d(2, 2): SiameseNet conv.0-3 (conv 7x7 s2 p3 + bias + BN + ReLU + pool,
 * src/face_models.py:115-118); out = B×(Hc/2)×(Wc/2)×64. */
int frmap_stem7x7_maxpool2(const float* x_nchw, const void* w_packed_c3, const float* shift, void* out,
                           int B, int Hi, int Wi, int dtype, void* stream);
/* The same two fusions fed by the uint8 image itself: x_u8_hwc = B×Hi×Wi×3 RGB bytes (what
 * transforms.Resize leaves, src/testing.py:99-100); ToTensor (u/255) + Normalize((x-mean)/std)
 * (src/testing.py:101-104; mean/std: 3 floats each in HOST memory) are applied while the rows are
 * staged, through a per-channel 256-entry table rounded to `dtype` exactly as
 * frmap_normalize_u8_hwc + the fp32 entry points would — the 602 KB/face fp32 tensor never exists
 * (SURVEY.md §8f row 1).  pool3 != 0: MaxPool2d(3,2,1) (ResNet stem); 0: MaxPool2d(2,2) (Siamese).
 * Needs Wi % 4 == 0 and a 4-byte aligned tensor. */
int frmap_stem7x7_maxpool_u8(const unsigned char* x_u8_hwc, const float* mean3_host, const float* std3_host,
                             const void* w_packed_c3, const float* shift, void* out, int B, int Hi, int Wi,
                             int pool3, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution, NHWC, MFMA 16x16x32 (bf16 / f16), fused epilogue
 *     out = act( conv(in, w) + shift[cout] [+ residual] )      act (`relu` argument): 0 none, 1 ReLU,
 *                                                              2 exact GELU (nn.GELU, face_models.py:630)
 * Covers: every 3x3 (s1/s2, p1) and 1x1 (s1/s2, p0) convolution + folded BatchNorm (+ReLU)
 * (+ residual add) of the ResNet-18 BasicBlocks behind src/face_models.py:67,463,658, of
 * BaselineNet conv2/conv3 (src/face_models.py:23-26,39-40), of SiameseNet (src/face_models.py:
 * 121-141), and — as a 1x1 "conv" over H=W=1 — the wide Linear(+BatchNorm1d)(+ReLU) layers
 * (src/face_models.py:148-155, 629-632).
 *   in       : B×Hi×Wi×Cin   (dtype)
 *   w_packed : from frmap_pack_conv_weight
 *   shift    : fp32 [Cout]   (beta - mean*scale [+ bias*scale])
 *   residual : B×Ho×Wo×Cout (dtype) or NULL
 *   out      : B×Ho×Wo×Cout (dtype),  Ho = (Hi + 2*pad - K)/stride + 1
 * ------------------------------------------------------------------------------------------- */
int frmap_conv_igemm(const void* in, const void* w_packed, const float* shift, const void* residual,
                     void* out, int B, int Hi, int Wi, int Cin, int Cout, int K, int stride, int pad,
                     int relu, int dtype, void* stream);
/* conv 3x3 s1 p1 + shift (+ReLU) + MaxPool2d(2, 2) in one launch: BaselineNet conv2/conv3 blocks
 * (src/face_models.py:39-40) and SiameseNet's conv -> BN -> ReLU -> MaxPool2d(2) runs (:121-141).
 * relu in {0, 1} (the max is taken before shift + activation, which is exact for monotonic ones).
 * out: B×(Hi/2)×(Wi/2)×Cout.  frmap_conv_igemm_pool2_supported: 1 when the shape is taken (even Hi and Wi,
 * Cin % 32 == 0, Cout % 64 == 0, the tile's input rows fit LDS); otherwise run frmap_conv_igemm +
 * frmap_maxpool. */
int frmap_conv_igemm_pool2_supported(int B, int Hi, int Wi, int Cin, int Cout);
/* which kernel frmap_conv_igemm_pool2 runs the shape on: 3 = LDS-DMA ping-pong kernel (row pairs tile a 112-pixel
 * slice: Wi in {2,4,8,14,28,56}, Cin >= 128, Cout % 128 == 0), 2 = weights-resident wave kernel (Cin 32 / 64, Hi and Wi
 * multiples of 8), 1 = generic kernel, 0 = not taken.  frmap_conv3x3_pp_pool_layout: 1 when the ping-pong form fits. */
int frmap_conv_igemm_pool2_form(int B, int Hi, int Wi, int Cin, int Cout);
/* does a 1x1 conv / Linear layer (stride s, pad 0; Linear: Hi = Wi = 1, B = rows) take the LDS-DMA ping-pong kernel
 * (conv1x1_pp_kernel)?  1 = 224 px x 256 ch tiles, 2 = 448 px x 128 ch, 3 = 224 px x 128 ch with split-K, 0 = the
 * first-generation 1x1 kernel runs it (Cin % 32, Cout % 128, too few tiles). */
int frmap_conv1x1_pp_layout(int B, int Hi, int Wi, int Cin, int Cout, int stride);
int frmap_conv3x3_pp_pool_layout(int B, int Hi, int Wi, int Cin, int Cout);
int frmap_conv_igemm_pool2(const void* in, const void* w_packed, const float* shift, void* out, int B, int Hi,
                           int Wi, int Cin, int Cout, int relu, int dtype, void* stream);

/* Tuning / test hook for the second-generation 3x3 stride-1 kernel behind frmap_conv_igemm (conv_pp.hip: 8-wave
 * workgroups, LDS-DMA operands): enable (0 / 1, -1 = default), pixels per tile (<= 224, -1 = whole rows / images),
 * channel tile (128 / 256; 1282 = 128 channels with the wave groups splitting K; -1 = heuristic).  Process-wide; not
 * needed for normal use. */
int frmap_conv_pp_tuning(int enable, int tile_px, int bn);
/* A/B hook of the second-generation kernel's fragment-read placement: 1 = reads of k-step k + 1 interleaved with the
 * MFMAs of k-step k (conv3x3_pp_kernel<..., RI = true>) where the layout allows, 0 = one burst per phase, -1 = environment
 * (FRMAP_PP_RI). */
int frmap_conv_pp_ri(int v);
/* Further process-wide A/B hooks of the same kernel family (experiments recorded in DESIGN.md; defaults = the shipped path):
 * frmap_conv_pp_pitch: conflict-free LDS halo pitch on (1) / off (0); frmap_conv_pp_ds: the fused projection-shortcut form
 * on the second-generation kernel on (1) / off (0); frmap_conv_pp_im: LDS-DMA issued between the MFMAs (1) or in a burst (0). */
int frmap_conv_pp_pitch(int v);
int frmap_conv_pp_ds(int v);
int frmap_conv_pp_im(int v);
/* Which layout frmap_conv_igemm gives a 3x3 stride-1 pad-1 layer without a fused shortcut: 0 = a first-generation
 * kernel; conv3x3_pp_kernel with 1 = 224 px x 256 ch tiles, 2 = 448 px x 128 ch, 3 = 224 px x 128 ch split-K. */
int frmap_conv3x3_pp_layout(int B, int Hi, int Wi, int Cin, int Cout);
/* The same question for a 3x3 stride-2 pad-1 layer (conv3x3s2_pp_kernel: 1 = 224 px x 256 ch, 2 = 448 px x 128 ch). */
int frmap_conv3x3s2_pp_layout(int B, int Hi, int Wi, int Cin, int Cout);
/* ... and for frmap_conv_igemm_ds (3x3 stride-1 layer with a fused 1x1 stride-s projection shortcut): 0 / 1 / 2. */
int frmap_conv3x3_pp_ds_layout(int B, int Hi, int Wi, int Cin, int Cout, int ds_Hi, int ds_Wi, int ds_Cin, int ds_stride);

/* A 3x3 stride-1 pad-1 convolution with a ResNet projection shortcut folded in (BasicBlock.conv2 + bn2 + downsample
 * [conv1x1 stride s + bn] + add + ReLU of the first block of a stage, torchvision resnet.py via face_models.py:67):
 *   out = act( conv3x3(in, W) + conv1x1_stride_s(ds_in, W_ds) + shift ),   shift = shift_conv + shift_shortcut
 *   in        : B×Hi×Wi×Cin;  ds_in : B×ds_Hi×ds_Wi×ds_Cin with (ds_H - 1)/s + 1 == Hi (the block's input)
 *   w_packed  : frmap_pack_conv_weight(W [Cout][Cin][3][3]);  ds_w_packed : frmap_pack_conv_weight(W_ds [Cout][ds_Cin][1][1])
 * The shortcut runs as extra one-tap K stages of the same kernel: no 1x1 launch, no round trip of its output through
 * HBM, no residual read.  frmap_conv_igemm_ds_supported(...) != 0 tells whether a shape takes this kernel (3x3 s1 layers
 * of the register-prefetch kind with ds_Cin <= Cin); otherwise run the shortcut with frmap_conv_igemm and pass it as
 * `residual`. */
int frmap_conv_igemm_ds_supported(int B, int Hi, int Wi, int Cin, int Cout, int ds_Hi, int ds_Wi, int ds_Cin,
                                  int ds_stride);
int frmap_conv_igemm_ds(const void* in, const void* w_packed, const float* shift, const void* ds_in,
                        const void* ds_w_packed, void* out, int B, int Hi, int Wi, int Cin, int Cout,
                        int ds_Hi, int ds_Wi, int ds_Cin, int ds_stride, int relu, int dtype, void* stream);

/* Wide Linear (+folded BatchNorm1d) (+ReLU/GELU `act` as above) (+residual) on the same MFMA kernel:
 *   out[M][N] = act( x[M][K] · Wᵀ + shift [+ residual] ),  x / residual / out in `dtype`, w_packed from
 *   frmap_pack_conv_weight(W as [N][K][1][1]).  When the output has too few tiles to fill the GPU the K
 *   loop is split across workgroups (SiameseNet fc.1, src/face_models.py:148: 18432 -> 1024) and
 *   `workspace` (device, >= frmap_linear_mfma_workspace_bytes(M,K,N) bytes; may be NULL when that is 0)
 *   holds the fp32 partial slabs. */
size_t frmap_linear_mfma_workspace_bytes(int M, int K, int N);
int frmap_linear_mfma(const void* x, const void* w_packed, const float* shift, const void* residual,
                      void* out, void* workspace, int M, int K, int N, int act, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pools (NHWC, `dtype`).
 *   frmap_maxpool        : nn.MaxPool2d(k, s, p)  — (3,2,1) ResNet stem; (2,2,0) BaselineNet /
 *                          SiameseNet (src/face_models.py:27,118,127,136)
 *   frmap_avgpool_global : AdaptiveAvgPool2d(1) -> fp32 B×C (src/face_models.py:30,43; resnet
 *                          avgpool)
 *   frmap_avgpool_adaptive: AdaptiveAvgPool2d((OH,OW)) -> `dtype` B×OH×OW×C
 *                          (src/face_models.py:142), windows [floor(i*H/OH), ceil((i+1)*H/OH))
 * ------------------------------------------------------------------------------------------- */
int frmap_maxpool(const void* in, void* out, int B, int H, int W, int C, int k, int stride, int pad,
                  int dtype, void* stream);
int frmap_avgpool_global(const void* in, float* out_f32, int B, int HW, int C, int dtype, void* stream);
int frmap_avgpool_adaptive(const void* in, void* out, int B, int H, int W, int C, int OH, int OW,
                           int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * fp32 heads.
 *   frmap_linear_f32 : out[b][n] = [relu]( sum_k x[b][k]*w[n][k] * scale[n] + shift[n] )
 *                      (scale/shift may be NULL = 1/0).  nn.Linear (+ folded BatchNorm1d):
 *                      src/face_models.py:32-33,46-48; :75; :467-468,516-517; :488,580; :678
 *   frmap_l2_normalize_f32 : F.normalize(x, p=2, dim=1, eps) = x / max(||x||, eps)
 *                      (src/face_models.py:179,525,590)
 *   frmap_cast_to_f32 / frmap_cast_from_f32 : dtype <-> fp32 element casts for head glue
 * ------------------------------------------------------------------------------------------- */
int frmap_linear_f32(const float* x, const float* w, const float* scale, const float* shift,
                     float* out, int B, int K, int N, int relu, void* stream);
int frmap_l2_normalize_f32(const float* x, float* out, int B, int D, float eps, void* stream);
int frmap_cast_to_f32(const void* in, float* out, size_t n, int dtype, void* stream);
int frmap_cast_from_f32(const float* in, void* out, size_t n, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Token-side ops of the hybrid CNN-Transformer (src/face_models.py:618-721), tokens laid out
 * [B][L][D] in `dtype` (= the NHWC trunk output B×7×7×512 viewed as B×49×512).
 *   frmap_add_pos_layernorm : t = x (+ pos[l]) ; y = LayerNorm(t)*gamma+beta (eps)   (:639,644,690)
 *                             t_out (optional) receives t in `dtype`; pos fp32 [L][D] or NULL.
 *   frmap_mha_tokens        : softmax(QK^T/sqrt(128))V per head of nn.MultiheadAttention(512,4)
 *                             (:623,640); qkv = [B][L][3D] as in_proj emits; L <= 64, D = H*128.
 *   frmap_mean_layernorm    : LayerNorm(mean over L tokens) -> fp32 [B][D]               (:718-719)
 * The projections / MLP GEMMs are frmap_conv_igemm calls (K=1, H=W=1) with bias, GELU, residual fused.
 * ------------------------------------------------------------------------------------------- */
int frmap_add_pos_layernorm(const void* x, const float* pos, const float* gamma, const float* beta,
                            void* t_out, void* y_out, int B, int L, int D, float eps, int dtype,
                            void* stream);
int frmap_mha_tokens(const void* qkv, void* out, int B, int L, int D, int H, int dtype, void* stream);
int frmap_mean_layernorm(const void* t, const float* gamma, const float* beta, float* out_f32,
                         int B, int L, int D, float eps, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * AttentionNet's attention block on the trunk map   (/root/reference/src/face_models.py:194-262, GAP :279)
 *   qkv       : [B][H*W][2*Cq + C] (dtype): the 1x1 query | key | value projections (+bias) of x, packed
 *               (one frmap_conv_igemm with the three weight matrices stacked along Cout)
 *   x         : [B][H*W][C] (dtype), the trunk map (NHWC)
 *   gamma     : device float[1]  (AttentionModule.gamma, :220)
 *   spatial_w : device fp32 [2][KS][KS], spatial_b: device float[1]  (SpatialAttention.conv, :199; pad KS/2)
 *   out_map   : [B][H*W][C] (dtype) = (gamma * softmax(q k^T) v + x) * sigmoid(conv([mean_c, max_c]))  or NULL
 *   out_pool  : fp32 [B][C] = mean over the H*W positions of that map (AdaptiveAvgPool2d(1))            or NULL
 *   H*W <= 64, Cq <= 128 (multiple of 8), C in {256, 512}, KS odd.
 * ------------------------------------------------------------------------------------------- */
int frmap_cnn_attention(const void* qkv, const void* x, const float* gamma, const float* spatial_w,
                        const float* spatial_b, void* out_map, float* out_pool, int B, int H, int W, int Cq,
                        int C, int KS, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Gallery matching (fp32, exact-f32 MFMA).
 *   frmap_match_top1 : for each probe row e (B×D) the FIRST index minimising
 *                      || e - g_i + 1e-6 ||_2 over gallery rows g (G×D) and that distance —
 *                      the batched form of compare_faces' loop (src/app.py:58-63; F.pairwise_distance
 *                      eps semantics).  idx_out int32[B], dist_out fp32[B].  G == 0 -> idx -1,
 *                      dist +inf.  id_or_unknown_out (optional int32[B]) = idx if dist <= thresh else -1
 *                      (compare_faces' "Unknown", src/app.py:64).  packed_out (optional int32[B][2]) =
 *                      {id_or_unknown, bits of dist}: the 8-byte record the multi-GPU all-gather ships.
 *                      G <= 64 takes a one-launch exact scan.  G > 64: a GEMM scores every pair by the
 *                      expanded squared distance WITH a worst-case rounding bound, and every gallery row
 *                      that could be the minimiser within that bound is re-scored with the exact
 *                      ||(e - g) + 1e-6||_2 (float64 accumulation); the first row attaining the exact
 *                      minimum wins.  The index is therefore the reference loop's, not the expanded
 *                      form's (near-duplicate enrolments, un-normalised embeddings included).
 *   frmap_cosine_logits : logits[b][c] = s * <x_b/||x_b||, w_c/||w_c||>  and (optionally)
 *                      argmax_out[b] — class-centre match (src/hyperparameter_tuning.py:1038-1046,
 *                      src/face_models.py:889-893).  logits_out may be NULL.
 *   frmap_arcmargin_eval : ArcMarginProduct.forward in eval mode (src/face_models.py:351-429):
 *                      clamp(cos) -> acos -> target column cos(min(pi-1e-4, theta+m)) (or the
 *                      easy-margin rule) -> * min(s,24) -> NaN/Inf -> 0.  label: int64[B].
 *                      minmax_out (optional) fp32[2] receives max/min raw cosine (:358-360).
 * ------------------------------------------------------------------------------------------- */
/* Evaluation-loop glue (src/testing.py:175-177, 278-279):
 *   frmap_softmax_argmax    : probs = softmax(logits, dim=1) (optional), pred = first arg-max (optional)
 *   frmap_pairwise_distance : dist[b] = ||a_b - b_b + 1e-6||_2 ; same_out[b] = dist < thresh (optional) */
int frmap_softmax_argmax(const float* logits, float* probs_out, int32_t* pred_out, int B, int C, void* stream);
int frmap_pairwise_distance(const float* a, const float* b, float* dist_out, int32_t* same_out, float thresh,
                            int B, int D, void* stream);

/* Scratch frmap_cosine_logits / frmap_arcmargin_eval need (device, caller-owned, >= this many bytes, 16-byte
 * aligned). */
size_t frmap_head_workspace_bytes(int B, int C);
/* Scratch of frmap_match_top1 and frmap_match_top1_packed for B probes against G gallery rows (candidate records +
 * row statistics; device, caller-owned, 16-byte aligned). */
size_t frmap_match_workspace_bytes(int B, int G);
int frmap_match_top1(const float* emb, const float* gallery, int32_t* idx_out, float* dist_out,
                     int32_t* id_or_unknown_out, int32_t* packed_out, float thresh, void* workspace,
                     int B, int G, int D, void* stream);
/* The same match for LARGE galleries on the fp16 MFMA pipe (same outputs, same contract, same exact re-scoring of
 * every in-band candidate).  Every fp32 operand is split into two fp16
 * numbers (hi = fp16(S x), lo = fp16(S x - hi)); one K = 3 D GEMM accumulates a_hi.g_hi + a_hi.g_lo + a_lo.g_hi in
 * fp32 (the fp32 dot product to ~2^-22).  frmap_match_pack_gallery prepares a gallery ONCE: packed_out
 * (frmap_match_gallery_pack_bytes(G, D) bytes) and stat_w_out [G][4]; D % 32 == 0.  Each row is scaled by its own power
 * of two before the split, so any finite magnitude is safe.  frmap_match_top1_packed: workspace =
 * frmap_match_workspace_bytes(B, G) bytes, probe_split = B * 3 * D fp16 of scratch; `gallery` is still the fp32 matrix
 * (candidates are re-scored from it). */
size_t frmap_match_gallery_pack_bytes(int G, int D);
int frmap_match_pack_gallery(const float* gallery, void* packed_out, float* stat_w_out, int G, int D, void* stream);
/* Incremental enrolment (src/app.py:428-436 appends one identity and re-saves): rows [row_lo, row_hi) of a gallery that now
 * holds G rows were appended (row_hi == G) or edited in place; only their statistics and the 64-row tiles they touch are
 * re-packed.  packed_out / stat_w_out must be sized for the gallery's CAPACITY (>= G rows). */
int frmap_match_pack_gallery_rows(const float* gallery, void* packed_out, float* stat_w_out, int row_lo, int row_hi,
                                  int G, int D, void* stream);
int frmap_match_top1_packed(const float* emb, const float* gallery, const void* gallery_packed, const float* stat_w,
                            int32_t* idx_out, float* dist_out, int32_t* id_or_unknown_out, int32_t* packed_out,
                            float thresh, void* workspace, void* probe_split, int B, int G, int D, void* stream);

/* The tail of the ResNet-18 ('cnn') embed-and-match step for small galleries in ONE launch, one workgroup per face:
 * AdaptiveAvgPool2d(1) of the trunk map (face_models.py:100) -> optional F.normalize(eps) -> compare_faces' scan
 * (src/app.py:58-64) exactly as frmap_match_top1 does it for G <= 64.
 *   map : [B][HW][C] (dtype) trunk output (NHWC);  gallery : fp32 [G][C], 0 <= G <= 64
 *   emb_out : fp32 [B][C] pooled (and normalised, if asked) embedding, or NULL;  other outputs as frmap_match_top1. */
int frmap_gap_norm_match(const void* map, const float* gallery, float* emb_out, int32_t* idx_out, float* dist_out,
                         int32_t* id_or_unknown_out, int32_t* packed_out, float thresh, int normalize, float eps,
                         int B, int HW, int C, int G, int dtype, void* stream);
/* Pool + Linear + normalise heads in one launch: AdaptiveAvgPool2d(1) of the NHWC trunk map [B][HW][K] -> Linear(K, N)
 * (y * scale + shift: a folded BatchNorm1d, or scale = NULL and shift = the bias) -> optional ReLU -> F.normalize(eps).
 *   ArcFaceNet (src/face_models.py:573-590): embedding (no bias) + bn, relu = 0;
 *   BaselineNet (:41-46, 51-60): F.relu(self.fc1(pooled)), relu = 1 (pre_out = the reference's un-normalised embedding).
 * wt: the Linear weight transposed, fp32 [K][N], N in {256, 512}; pre_out / emb_out: fp32 [B][N] un-normalised /
 * unit-norm embeddings (either may be NULL). */
int frmap_gap_linear_norm(const void* map, const float* wt, const float* scale, const float* shift, float* pre_out,
                          float* emb_out, float eps, int B, int HW, int K, int N, int relu, int dtype, void* stream);
int frmap_cosine_logits(const float* x, const float* w, float* logits_out, int32_t* argmax_out,
                        void* workspace, int B, int C, int D, float s, void* stream);
int frmap_arcmargin_eval(const float* x, const float* w, const int64_t* label, float* logits_out,
                         float* minmax_out, void* workspace, int B, int C, int D, float s, float m,
                         int easy_margin, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Model handles: what `model(images)` / `model.get_embedding(images)` run in the reference
 * (src/testing.py:255-273, src/app.py:44) as ONE call, for the ResNet-18 families of get_model()
 * (src/face_models.py:785-813): 'baseline' = BaselineNet (:16-60), 'cnn' = ResNetTransfer (:62-102), 'siamese' = SiameseNet
 * (one tower of forward(x1, x2), :104-192; 256-d embeddings), 'arcface' = ArcFaceNet eval (:447-613), 'hybrid' = HybridNet
 * (:650-721), 'resnet18_trunk' = the bare trunk (features only; AttentionNet builds on it, :269).
 *
 *   frmap_model_create       model_type as get_model() spells it; num_classes sizes the classifier head; `dtype` = compute type.
 *                            An unknown type is rejected with the reference's message ("Invalid model type: ...", :813).
 *   frmap_model_load_tensor  one call per entry of the reference checkpoint's state_dict, under the SAME key
 *                            ("resnet.layer1.0.bn1.running_var", "backbone.conv1.weight" or its alias "features.0.weight",
 *                            "embedding.weight", "bn.weight", "val_classifier.bias", ...): fp32 data, `numel` elements, host
 *                            memory (on_device = 0) or device memory (1).  Returns 0 = taken, 1 = a key the inference path does
 *                            not use (ignored: training head, num_batches_tracked), -1 = wrong size.
 *   frmap_model_finalize     folds every eval-mode BatchNorm into its conv / linear (fp32), packs the weights into the kernels'
 *                            layout, fixes the layer plan; fails naming the first missing tensor.  Synchronises `stream` once.
 *                            After it the handle is immutable: any thread / stream may run forwards on it concurrently.
 *   frmap_model_forward      x: FRMAP_INPUT_F32_NCHW = fp32 [B][3][H][W] (the reference's tensor) or FRMAP_INPUT_U8_HWC = uint8
 *                            [B][H][W][3] (ToTensor + Normalize applied inside the stem, src/testing.py:99-104);
 *                            `what` selects the output written to the caller-owned `out`:
 *                              FRMAP_OUT_TRUNK_MAP  [B][h][w][512] in `dtype`  (children()[:-2], face_models.py:660)
 *                              FRMAP_OUT_POOLED     fp32 [B][512]              (children()[:-1] flattened, :100,464)
 *                              FRMAP_OUT_EMBEDDING  fp32 [B][frmap_model_embedding_dim]: get_embedding() - 'cnn': the pooled features
 *                                                   (:98-102); 'arcface': F.normalize(bn(embedding(pooled))) (:584-590); 'baseline':
 *                                                   relu(fc1(pooled)) (:51-60); 'siamese': the tower's unit-norm output (:161-179);
 *                                                   'hybrid': LayerNorm(token mean) (:705-721)
 *                              FRMAP_OUT_LOGITS     fp32 [B][num_classes]: forward() - 'cnn': resnet.fc (:93-96); 'arcface':
 *                                                   val_classifier over row-normalised weights (:576-580); 'baseline': fc2;
 *                                                   'hybrid': fc; 'siamese': none (rejected)
 *                            (TRUNK_MAP / POOLED are the ResNet trunk's outputs: 'cnn', 'arcface', 'hybrid', 'resnet18_trunk'.)
 *                            workspace: frmap_model_workspace_bytes(m, B, H, W) bytes, 256-byte aligned, caller-owned.
 *                            Nothing is allocated, freed or synchronised; all launches go to `stream`.
 *   frmap_model_embed_and_match  forward + compare_faces for every face (src/app.py:44,50-64): embedding (L2-normalised first if
 *                            `normalize`, for models whose embedding is not unit-norm) -> first arg-min of ||e - g + 1e-6||_2
 *                            over the fp32 gallery [G][512] -> outputs as frmap_match_top1.  gallery_packed / gallery_stat
 *                            (frmap_match_pack_gallery; may be NULL) put galleries of >= 512 rows on the MFMA pipe; 'cnn' with
 *                            G <= 64 pools, normalises and matches in one launch.  emb_out: optional fp32 [B][512].
 *                            workspace: frmap_model_match_workspace_bytes(m, B, H, W, G).
 *   frmap_model_trace / _trace_read  per-launch HIP-event timing of subsequent forwards (kernel label, algorithmic FLOPs and
 *                            bytes, microseconds) for roofline reports; read synchronises the recorded events and clears them.
 * ------------------------------------------------------------------------------------------- */
#define FRMAP_INPUT_F32_NCHW 0
#define FRMAP_INPUT_U8_HWC 1
#define FRMAP_OUT_TRUNK_MAP 0
#define FRMAP_OUT_POOLED 1
#define FRMAP_OUT_EMBEDDING 2
#define FRMAP_OUT_LOGITS 3
typedef struct frmap_model frmap_model;
typedef struct frmap_trace_record {
  char kernel[64];
  double flop;
  double bytes;
  float us;
} frmap_trace_record;
int frmap_model_create(frmap_model** out, const char* model_type, int num_classes, int dtype);
int frmap_model_load_tensor(frmap_model* m, const char* key, const void* data, size_t numel, int on_device);
int frmap_model_set_input_normalization(frmap_model* m, const float* mean3_host, const float* std3_host);
int frmap_model_finalize(frmap_model* m, void* stream);
int frmap_model_embedding_dim(const frmap_model* m);
size_t frmap_model_workspace_bytes(const frmap_model* m, int B, int H, int W);
size_t frmap_model_match_workspace_bytes(const frmap_model* m, int B, int H, int W, int G);
int frmap_model_forward(frmap_model* m, const void* x, int x_kind, int B, int H, int W, int what, void* out,
                        void* workspace, void* stream);
int frmap_model_embed_and_match(frmap_model* m, const void* x, int x_kind, int B, int H, int W, const float* gallery,
                                const void* gallery_packed, const float* gallery_stat, int G, float thresh, int normalize,
                                int32_t* idx_out, float* dist_out, int32_t* id_or_unknown_out, int32_t* packed_out,
                                float* emb_out, void* workspace, void* stream);
int frmap_model_trace(frmap_model* m, int enable);
int frmap_model_trace_read(frmap_model* m, frmap_trace_record* out, int max_records);
void frmap_model_destroy(frmap_model* m);

#ifdef __cplusplus
}
#endif
#endif /* FRMAP_HIP_H */
