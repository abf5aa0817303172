// Second-generation fused ResNet stem for gfx950 (round 3): fp32 NCHW (or uint8 HWC) input -> conv 7x7 s2 p3 (3 -> 64) +
// folded BN + ReLU -> maxpool 3x3 s2 p1 -> NHWC B x Hq x Wq x 64 in bf16 / f16, one kernel.  Replaces `conv1 -> bn1 -> relu ->
// maxpool` of the torchvision ResNet-18 behind /root/reference/src/face_models.py:67,463,658 (and ToTensor + Normalize of
// src/testing.py:101-104 for uint8 input), like stem_pool.hip, whose kernel issued 2.03x the useful MFMAs (K laid as
// 7 x 32 = 224 for 147 taps, and 7 conv rows computed per 6 new ones).  Two changes remove most of that:
//
//  * SPACE-TO-DEPTH K axis.  A 7x7 stride-2 convolution over pixels is a 4x4 stride-1 convolution over 2x2 "super-pixels"
//    of 12 values (dy, dx, c): out(y, x) = sum over super-rows y-2 .. y+1, super-columns x-2 .. x+1, with kh = 2 khh + dy - 1,
//    kw = 2 kww + dx - 1 (the taps kh = -1 / kw = -1 carry zero weights).  K = 4 x 4 x 12 = 192 = 6 MFMA k-steps instead of 7,
//    and the LDS image is [super-row][super-column][12] (24 B per super-pixel): a lane's 8 consecutive k are 16 contiguous,
//    8-byte aligned bytes (two ds_read_b64).
//  * ROW CARRY.  A workgroup walks DOWN one (image, column half): each step computes the 6 conv rows of 3 pooled rows; the
//    7th row a 3x3-s2 pool window needs (the row above) is the previous step's last row, kept in 16 accumulator registers.
//    Nothing is recomputed vertically, and consecutive steps share 3 of their 9 super-rows in an LDS ring of 16 rows, so a
//    step stages 6 new super-rows (12 input rows) where the old kernel staged 19 rows per tile.
//
// MFMAs per 3 pooled rows x 7 pooled columns x 64 channels: 6 k-steps x 6 x 4 = 144 (old: 196); staged bytes per step: 2/3.
// The next step's input rows are loaded into registers (one 4-pixel x 2-row x 3-plane item per thread: 24 registers) before
// the current step's MFMA phase and written to the ring after it: one barrier per step.
// Column strips as before: wave s owns conv columns 14 s - 1 .. 14 s + 14 of its half (16 columns -> 7 pooled columns).
//
// Execution form: two independent 4-wave workgroups per CU (one (image, half, row segment) unit each at a time).  An 8-wave
// ping-pong form (the two halves of an image in one workgroup, half a step apart behind workgroup barriers) was built and
// measured: same time (119.6 vs 117.2 us at 256 faces) at 20 % more cycles and a higher clock - the launch is bound by the energy
// of its MFMAs and its data movement, not by how its phases overlap - and its 143 KB of LDS would keep the two
// micro-batch streams of the graph mode from sharing a CU.
#include "frmap_common.h"
#include <stdlib.h>
#include <type_traits>

struct StemS2DParams {
  const float* x;           // fp32 NCHW input (U8 = false)
  const unsigned char* x8;  // uint8 HWC RGB input (U8 = true)
  float mean[3], std[3];
  const void* wpk;          // frmap_pack_conv_weight_c3(64, 7, 7): [64][240], k = kh * 32 + kw * 4 + c
  const float* shift;
  void* out;
  int N, Hi, Wi, Hc, Wc, Hq, Wq;
  int nhalves;   // column halves of 4 strips (28 pooled columns) each
  int nsteps;    // ceil(Hq / 3)
  int nseg, sps; // row segments per (image, half), steps per segment
  int nunits;
};

namespace {
constexpr int kRing = 16, kSWC = 64, kSPB = 24, kRowB = kSWC * kSPB;   // ring rows, super-columns per row, bytes
constexpr int kWP = 400;                                              // bytes per output channel of the weight image (192 + 8 elements)
constexpr int kPairs = 31;                                            // 4-pixel items per super-row (62 super-columns)
constexpr int kPitch = 4 * 64 + 16;                                   // epilogue transpose pitch (bytes per pixel row)

template <int N>
__device__ __forceinline__ float s2d_row_down(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x100 + N, 0xF, 0xF, false));
}
}  // namespace

template <typename TT, bool U8>
__global__ __launch_bounds__(256, 2) void stem_s2d_kernel(const StemS2DParams p) {
  constexpr int MI = 6, NI = 4, KS = 6;
  using vec8 = typename TT::vec8;
  using elem = typename TT::elem;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ring = smem;                                  // [16][64][24 B]
  char* wl = smem + kRing * kRowB;                    // [64][400 B]
  char* scratch_all = wl + 64 * kWP;                  // 4 waves x 16 x kPitch
  elem* lut = (elem*)(scratch_all + 4 * 16 * kPitch); // U8: [3][256]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, g = lane >> 4;
  const size_t HW = (size_t)p.Hi * p.Wi;
  const int plane_b = (int)(HW * 4);

  // ---- weights: scatter the [64][240] (k = kh * 32 + kw * 4 + c) image into the space-to-depth order, once per workgroup:
  // 16-byte loads along the source rows, shifts and masks only (kh = 2 khh + dy - 1 <=> khh = (kh + 1) >> 1, dy = (kh + 1) & 1)
  {
    for (int i = tid; i < 64 * kWP / 16; i += 256) ((u32x4_t*)wl)[i] = (u32x4_t){0u, 0u, 0u, 0u};   // zero taps (kh = -1, kw = -1) and padding
    if (tid < 64) ((float*)(scratch_all + 4 * 16 * kPitch + 3 * 256 * 2))[tid] = p.shift[tid];
    if (U8)
      for (int i = tid; i < 768; i += 256) {
        const int c = i >> 8, u = i & 255;
        lut[i] = TT::from_f32(((float)u / 255.0f - p.mean[c]) / p.std[c]);
      }
    __syncthreads();
    const u32x4_t* src = (const u32x4_t*)p.wpk;          // 30 pieces of 8 elements per output channel, the first 28 carry taps
    for (int i = tid; i < 64 * 28; i += 256) {
      const int co = i / 28, pc = i - co * 28;            // piece pc = elements 8 pc .. 8 pc + 7 of the row: kh = pc >> 2, kw = 2 (pc & 3) + {0, 1}
      const u32x4_t v = src[co * 30 + pc];
      elem e[8];
      __builtin_memcpy(e, &v, 16);
      const int kh = pc >> 2, khh = (kh + 1) >> 1, dy = (kh + 1) & 1;
      elem* dst = (elem*)(wl + co * kWP) + khh * 48 + dy * 6;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int kw = 2 * (pc & 3) + h;
        if (kw < 7) {
          elem* d = dst + ((kw + 1) >> 1) * 12 + ((kw + 1) & 1) * 3;
          d[0] = e[4 * h]; d[1] = e[4 * h + 1]; d[2] = e[4 * h + 2];
        }
      }
    }
  }

  // ---- per-lane constants of the MFMA phase ----
  const int lcb = (14 * wave + lr + 1) * kSPB;           // this lane's conv column -> first super-column of its taps, bytes
  const int woff = lr * kWP + g * 16;
  char* scratch = scratch_all + wave * (16 * kPitch);
  const float* shl = (const float*)(scratch_all + 4 * 16 * kPitch + 3 * 256 * 2);   // [64] folded BatchNorm shift (LDS copy)
  elem* outp = (elem*)p.out;
  const int j = lane >> 3, part = lane & 7;              // store item: pooled column j of the strip, 8-channel run `part`

  // ---- staging: one item = super-row r, 4-pixel group q (2 super-pixels): 2 input rows x 3 planes x 4 pixels ----
  struct Item {
    u32x4_t fv[U8 ? 1 : 2][3];   // fp32: [input row dy][plane] = 4 pixels
    u32x4_t fb[U8 ? 2 : 1];      // uint8: [input row dy] = 12 bytes (4 pixels x RGB)
    unsigned okm;                // uint8: bit dy = that input row lies inside the image (padding is 0 in NORMALISED space)
  };
  Item pend;                     // the next step's item of this thread, in flight across the MFMA phase

  for (int unit = blockIdx.x; unit < p.nunits; unit += gridDim.x) {
    const int seg = unit % p.nseg, u2 = unit / p.nseg;
    const int half = u2 % p.nhalves;
    const int n = __builtin_amdgcn_readfirstlane(u2 / p.nhalves);
    const int st_lo = seg * p.sps, st_hi = min(st_lo + p.sps, p.nsteps);
    if (st_lo >= st_hi) continue;
    const int strip = half * 4 + wave;
    const int xin0 = 112 * half - 8;                     // input column of super-column 0 of this half's ring rows
    auto rsrc = U8 ? __builtin_amdgcn_make_buffer_rsrc((void*)(p.x8 + (size_t)n * HW * 3), (short)0, (int)(HW * 3), 0x00020000)
                   : __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)n * 3 * HW), (short)0, 3 * plane_b, 0x00020000);

    auto issue = [&](Item& it, int sr0, int nrows, int round) {
      const int item = tid + 256 * round;
      const bool iv = item < nrows * kPairs;
      const int r = (iv ? item : 0) / kPairs, q = (iv ? item : 0) - r * kPairs;
      const int ix = xin0 + 4 * q, iy = 2 * (sr0 + r);
      const bool colok = iv && (unsigned)ix < (unsigned)p.Wi;
      if constexpr (U8) {
        it.okm = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const bool ok = colok && (unsigned)(iy + dy) < (unsigned)p.Hi;
          const auto v3 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, ok ? ((iy + dy) * p.Wi + ix) * 3 : 0x7FFFFFF0, 0, 0);
          it.fb[dy] = (u32x4_t){v3[0], v3[1], v3[2], 0u};
          it.okm |= ok ? (1u << dy) : 0u;
        }
      } else {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const bool ok = colok && (unsigned)(iy + dy) < (unsigned)p.Hi;
          const int o = ok ? ((iy + dy) * p.Wi + ix) * 4 : 0x7FFFFFF0;
#pragma unroll
          for (int c = 0; c < 3; ++c) it.fv[dy][c] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, c * plane_b, 0);
        }
      }
    };
    auto commit = [&](const Item& it, int sr0, int nrows, int round) {
      const int item = tid + 256 * round;
      if (item >= nrows * kPairs) return;
      const int r = item / kPairs, q = item - r * kPairs;
      char* dst = ring + ((sr0 + r) & (kRing - 1)) * kRowB + q * (2 * kSPB);
      elem e[24];   // super-pixel A (pixels 0, 1) then B (pixels 2, 3), each (dy, dx, c)
      if constexpr (U8) {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const bool ok = (it.okm >> dy) & 1u;
          const unsigned w0 = it.fb[dy][0], w1 = it.fb[dy][1], w2 = it.fb[dy][2];
          const unsigned bytes[12] = {w0 & 255u, (w0 >> 8) & 255u, (w0 >> 16) & 255u, w0 >> 24,
                                      w1 & 255u, (w1 >> 8) & 255u, (w1 >> 16) & 255u, w1 >> 24,
                                      w2 & 255u, (w2 >> 8) & 255u, (w2 >> 16) & 255u, w2 >> 24};
#pragma unroll
          for (int px = 0; px < 4; ++px)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const elem v = ok ? lut[c * 256 + bytes[3 * px + c]] : TT::from_f32(0.f);
              e[(px >> 1) * 12 + dy * 6 + (px & 1) * 3 + c] = v;
            }
        }
      } else {
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int px = 0; px < 4; ++px) {
              const unsigned bits = it.fv[dy][c][px];   // (copy the lane first: bit_cast of a vector-element lvalue reads element 0)
              e[(px >> 1) * 12 + dy * 6 + (px & 1) * 3 + c] = TT::from_f32(__uint_as_float(bits));
            }
      }
      u32x4_t w[3];
      __builtin_memcpy(w, e, 48);
#pragma unroll
      for (int k = 0; k < 3; ++k) *(u32x4_t*)(dst + 16 * k) = w[k];
    };

    // MFMA phase: conv rows y0 .. y0 + NR - 1 of this wave's 16-column strip
    auto mfma_rows = [&](auto& acc, auto nr_tag, int y0) {
      constexpr int NR = decltype(nr_tag)::value;
#pragma unroll
      for (int mi = 0; mi < NR; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = *(const f32x4_t*)(shl + ni * 16 + g * 4);   // the folded BatchNorm shift rides in the accumulator (max and + commute)
      // pixel operand: k-step s, lane group g -> flat k = 32 s + 8 g of the 192: super-row khh = k / 48, byte offset 2 (k % 48);
      // recomputed per use (3 VALU ops under the MFMAs) instead of being held in 12 registers across the whole step
      int gg = g;
      asm volatile("" : "+v"(gg));
      // (a register-double-buffered fragment pipeline pinned with sched_barriers, and reads interleaved with the MFMAs through
      //  sched_group_barriers, were both built and measured: no change in launch time)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int f = 32 * s + 8 * gg, khh = (f * 1366) >> 16;   // f / 48 for f < 192
        const int poff = 2 * (f - khh * 48);
        vec8 wf[NI], pf[NR];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) wf[ni] = *(const vec8*)(wl + ni * 16 * kWP + woff + s * 64);
#pragma unroll
        for (int mi = 0; mi < NR; ++mi) {
          const char* a = ring + ((y0 + mi - 2 + khh) & (kRing - 1)) * kRowB + lcb + poff;
          const u32x2_t lo = *(const u32x2_t*)a, hi = *(const u32x2_t*)(a + 8);
          const u32x4_t w = {lo[0], lo[1], hi[0], hi[1]};
          __builtin_memcpy(&pf[mi], &w, 16);
        }
#pragma unroll
        for (int mi = 0; mi < NR; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = TT::mfma(wf[ni], pf[mi], acc[mi][ni]);
      }
    };

    const int cc = 14 * strip + lr - 1;                  // this lane's conv column
    const int cc_lo = 14 * strip - 1;                    // (wave-uniform) first conv column of the strip
    const bool colv = (unsigned)cc < (unsigned)p.Wc;
    // conv column -1 (lane 0 of the first strip) is skipped on the read side of the transpose; lane 15 is never read; a strip
    // whose lanes 0 .. 14 reach past the last conv column (odd sizes) takes the masking path
    const bool mask_cols = cc_lo + 14 >= p.Wc;
    int rd_off[3];                                          // transpose rows (conv columns) of this lane's pooling window
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      int row = 2 * (j < 7 ? j : 6) + k;
      if (cc_lo < 0 && row == 0) row = 1;
      rd_off[k] = row * kPitch + part * 32;
    }
    const bool active = strip * 7 < p.Wq;                // this strip has pooled columns at all

    // ---- prologue: the first step's super-rows (plus one more above when the carry row has to be computed) ----
    const int y00 = 6 * st_lo;                           // first conv row of the segment
    const bool pre = st_lo > 0;
    {
      const int sr_lo = y00 - 2 - (pre ? 1 : 0), nrows = 9 + (pre ? 1 : 0);
      __syncthreads();                                   // the previous unit's readers are done with the ring (and, first time, the weights are complete)
      Item second;                                       // (its registers are free here: no accumulators are live yet)
      issue(pend, sr_lo, nrows, 0);
      issue(second, sr_lo, nrows, 1);                    // both rounds' loads in flight together
      commit(pend, sr_lo, nrows, 0);
      commit(second, sr_lo, nrows, 1);
      __syncthreads();
    }
    if (st_lo + 1 < st_hi) issue(pend, y00 + 7, 6, 0);         // the second step's 6 new super-rows land under the first step's MFMAs

    f32x4_t carry[NI];                                       // the conv row above this step's first one (16 registers)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) carry[ni] = (f32x4_t){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (pre && active) {
      f32x4_t a1[1][NI];
      mfma_rows(a1, std::integral_constant<int, 1>{}, y00 - 1);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) carry[ni][e] = colv ? a1[0][ni][e] : -INFINITY;
    }

    // ---- the two groups half a step apart: group 0 runs MFMA(i) in phase 2 i and its epilogue in phase 2 i + 1, group 1 one
    //      phase later; a workgroup barrier closes every phase (it also publishes the epilogue's ring writes to the group's next
    //      MFMA phase and retires the MFMA phase's ring reads before the rows are overwritten two phases later)
    for (int st = st_lo; st < st_hi; ++st) {
      const int y0 = 6 * st, py0 = 3 * st;
      f32x4_t acc[MI][NI];
      if (active) mfma_rows(acc, std::integral_constant<int, MI>{}, y0);
      if (active) {
        if (mask_cols || y0 + MI > p.Hc) {   // (rare: odd conv sizes, last rows) conv positions outside the image act as -inf under the max
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const bool v = colv && (y0 + mi) < p.Hc;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc[mi][ni][e] = v ? acc[mi][ni][e] : -INFINITY;
          }
        }
        // pooling: the VERTICAL max of a pooled row's three conv rows is register-local (48 v_max3 per step); its result goes
        // through the wave's LDS transpose, and the HORIZONTAL max is taken on the far side, where a lane owns 8 channels of one
        // pooled column and reads the three conv columns 2 j, 2 j + 1, 2 j + 2 of its window (8 v_max3 per pooled row) - a third
        // of the vector instructions of pooling with lane shuffles before the transpose.
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            f32x4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float top = pr == 0 ? carry[ni][e] : acc[2 * pr - 1][ni][e];
              v[e] = fmaxf(fmaxf(top, acc[2 * pr][ni][e]), acc[2 * pr + 1][ni][e]);           // rows 2 py - 1, 2 py, 2 py + 1
            }
            *(f32x4_t*)(scratch + lr * kPitch + ni * 64 + g * 16) = v;
          }
          const int py = py0 + pr, px = 7 * strip + j;
          f32x4_t c0[3], c1[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            c0[k] = *(const f32x4_t*)(scratch + rd_off[k]);
            c1[k] = *(const f32x4_t*)(scratch + rd_off[k] + 16);
          }
          if (j < 7 && px < p.Wq && py < p.Hq) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o[e] = fmaxf(fmaxf(fmaxf(c0[0][e], c0[1][e]), c0[2][e]), 0.f);
              o[4 + e] = fmaxf(fmaxf(fmaxf(c1[0][e], c1[1][e]), c1[2][e]), 0.f);
            }
            *(u32x4_t*)(outp + (((size_t)n * p.Hq + py) * p.Wq + px) * 64 + part * 8) = pack8<TT>(o);
          }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) carry[ni] = acc[MI - 1][ni];
      }
      if (st + 1 < st_hi) commit(pend, y0 + 7, 6, 0);          // next step's rows into ring slots this step did not read
      __syncthreads();                                       // publishes the ring writes to the next step's MFMA phase
      // (AFTER the barrier: __syncthreads() drains vmcnt too - loads issued before it would be waited for on the spot)
      if (st + 2 < st_hi) issue(pend, y0 + 13, 6, 0);          // the step after's loads, in flight across the next MFMA phase
    }
  }
}

// 1 = launched, 0 = shape not taken (caller uses stem_pool_kernel), < 0 = error
int frmap_stem_s2d(const float* x_nchw, const unsigned char* x_u8, const float* mean3, const float* std3, const void* w_packed_c3,
                   const float* shift, void* out, int B, int Hi, int Wi, int dtype, hipStream_t st) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("FRMAP_STEM_S2D"); on = e ? atoi(e) : 1; }
  if (!on) return 0;
  if (Wi % 4 || Hi < 7 || Wi < 7) return 0;
  if (!x_u8 && ((uintptr_t)x_nchw & 15)) return 0;
  if (x_u8 && ((uintptr_t)x_u8 & 3)) return 0;
  StemS2DParams p;
  p.x = x_nchw; p.x8 = x_u8; p.wpk = w_packed_c3; p.shift = shift; p.out = out;
  for (int c = 0; c < 3; ++c) { p.mean[c] = mean3 ? mean3[c] : 0.f; p.std[c] = std3 ? std3[c] : 1.f; }
  p.N = B; p.Hi = Hi; p.Wi = Wi;
  p.Hc = (Hi + 6 - 7) / 2 + 1; p.Wc = (Wi + 6 - 7) / 2 + 1;
  p.Hq = (p.Hc + 2 - 3) / 2 + 1; p.Wq = (p.Wc + 2 - 3) / 2 + 1;
  const int nstrips = (p.Wq + 6) / 7;
  if (nstrips > 8) return 0;
  p.nhalves = (nstrips + 3) / 4;
  p.nsteps = (p.Hq + 2) / 3;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, v = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  // one workgroup per (image, half, row segment); segments only while whole columns would leave workgroup slots (2 per CU) empty
  const long long cols = (long long)B * p.nhalves;
  int nseg = (int)((2ll * ncu + cols - 1) / cols);
  if (nseg < 1) nseg = 1;
  if (nseg > p.nsteps) nseg = p.nsteps;
  p.sps = (p.nsteps + nseg - 1) / nseg;
  p.nseg = (p.nsteps + p.sps - 1) / p.sps;
  const long long nunits = cols * p.nseg;
  if (nunits >= (1ll << 31)) return 0;
  p.nunits = (int)nunits;
  const unsigned grid = (unsigned)(nunits < 2ll * ncu ? nunits : 2ll * ncu);
  const int lds = kRing * kRowB + 64 * kWP + 4 * 16 * kPitch + 3 * 256 * 2 + 64 * 4;   // 69.4 KB: two workgroups per CU
  typedef void (*kern_t)(const StemS2DParams);
  const kern_t kern = x_u8 ? (dtype == FRMAP_BF16 ? (kern_t)stem_s2d_kernel<BF16, true> : (kern_t)stem_s2d_kernel<F16, true>)
                           : (dtype == FRMAP_BF16 ? (kern_t)stem_s2d_kernel<BF16, false> : (kern_t)stem_s2d_kernel<F16, false>);
  if (frmap_big_lds((const void*)kern, 160 * 1024)) return -2;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, p);
  FRMAP_LAUNCH_CHECK();
  return 1;
}
