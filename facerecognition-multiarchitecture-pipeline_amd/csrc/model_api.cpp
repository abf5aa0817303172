// Model-level entry points of the C ABI (include/frmap_hip.h, "Model handles"): what `model(images)` runs in the reference
// (/root/reference/src/testing.py:255-273, src/app.py:44) for the ResNet-18 families of get_model()
// (/root/reference/src/face_models.py:62-102 'cnn', :447-613 'arcface') as ONE call a non-Python host can make.
//
// A handle owns the inference form of a checkpoint: BatchNorm folded into the neighbouring conv (fp32, on the host), weights
// packed into the kernels' LDS-image order, the per-layer kernel plan (which fused entry point takes which layer).  It is
// immutable after frmap_model_finalize, so any number of host threads / streams may run forwards on it at once; everything a
// forward writes lives in the CALLER's workspace and output buffers (nothing is allocated, freed or synchronised in a forward).
// The layer sequence is torchvision's ResNet-18 v1 (BasicBlock, stride on the first 3x3 of a stage, 1x1-s2 + BN projection
// shortcut; un-vendored third-party topology) exactly as face_models.py's Python planner issues it.
#include <math.h>
#include <string.h>

#include "model_api.h"

namespace {

const int kStagePlanes[4] = {64, 128, 256, 512};

// ---- state_dict keys ------------------------------------------------------------------------------------------------
// 'cnn': resnet.<child>...; 'arcface': backbone.<child>... and the aliased features.<i>... (nn.Sequential over the same
// tensors, face_models.py:464); canonical form here: trunk.<child>...
bool canonical_key(int kind, const std::string& key, std::string* out) {
  static const char* feat_child[8] = {"conv1", "bn1", nullptr, nullptr, "layer1", "layer2", "layer3", "layer4"};
  auto starts = [&](const char* p) { return key.compare(0, strlen(p), p) == 0; };
  if (key.size() > 19 && key.compare(key.size() - 19, 19, "num_batches_tracked") == 0) return false;
  if (starts("features.")) {
    const size_t dot = key.find('.', 9);
    if (dot == std::string::npos) return false;
    const int idx = atoi(key.substr(9, dot - 9).c_str());
    if (idx < 0 || idx > 7 || !feat_child[idx]) return false;
    *out = std::string("trunk.") + feat_child[idx] + key.substr(dot);
    return true;
  }
  if (kind >= KIND_BASELINE) return frmap_family_key(kind, key, out);
  if (kind == KIND_CNN) {
    if (starts("resnet.fc.1.")) { *out = "fc." + key.substr(12); return true; }
    if (starts("resnet.fc.")) return false;
    if (starts("resnet.")) { *out = "trunk." + key.substr(7); return true; }
    return false;
  }
  if (kind == KIND_ARCFACE) {
    if (starts("backbone.fc.")) return false;
    if (starts("backbone.")) { *out = "trunk." + key.substr(9); return true; }
    if (starts("embedding.") || starts("bn.") || starts("val_classifier.")) { *out = key; return true; }
    return false;  // arcface.weight (training head), ...
  }
  // bare trunk: accept resnet. / backbone. prefixes or the plain torchvision names
  if (starts("resnet.")) { *out = "trunk." + key.substr(7); }
  else if (starts("backbone.")) { *out = "trunk." + key.substr(9); }
  else if (starts("trunk.")) { *out = key; }
  else { *out = "trunk." + key; }
  return out->compare(0, 9, "trunk.fc.") != 0;
}

void add_bn(std::map<std::string, size_t>& want, const std::string& p, int c) {
  want[p + ".weight"] = c; want[p + ".bias"] = c; want[p + ".running_mean"] = c; want[p + ".running_var"] = c;
}

std::map<std::string, size_t> expected_tensors(const frmap_model* m) {
  std::map<std::string, size_t> want;
  if (m->kind == KIND_BASELINE || m->kind == KIND_SIAMESE) {
    frmap_family_expected(m, &want);
    return want;
  }
  want["trunk.conv1.weight"] = 64 * 3 * 7 * 7;
  add_bn(want, "trunk.bn1", 64);
  int inpl = 64;
  for (int li = 0; li < 4; ++li) {
    const int planes = kStagePlanes[li];
    for (int bi = 0; bi < 2; ++bi) {
      const std::string p = "trunk.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
      const int cin = bi == 0 ? inpl : planes;
      want[p + ".conv1.weight"] = (size_t)planes * cin * 9;
      add_bn(want, p + ".bn1", planes);
      want[p + ".conv2.weight"] = (size_t)planes * planes * 9;
      add_bn(want, p + ".bn2", planes);
      if (bi == 0 && li > 0) {
        want[p + ".downsample.0.weight"] = (size_t)planes * cin;
        add_bn(want, p + ".downsample.1", planes);
      }
    }
    inpl = planes;
  }
  if (m->kind == KIND_CNN) {
    want["fc.weight"] = (size_t)m->num_classes * 512;
    want["fc.bias"] = m->num_classes;
  } else if (m->kind == KIND_ARCFACE) {
    want["embedding.weight"] = 512 * 512;
    add_bn(want, "bn", 512);
    want["val_classifier.weight"] = (size_t)m->num_classes * 512;
    want["val_classifier.bias"] = m->num_classes;
  } else if (m->kind == KIND_HYBRID) {
    frmap_family_expected(m, &want);
  }
  return want;
}

}  // namespace

// ---- device memory owned by the handle --------------------------------------------------------------------------------
void* frmap_model_dev_alloc(frmap_model* m, size_t bytes) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
  m->allocs.push_back(p);
  return p;
}
float* frmap_model_dev_upload(frmap_model* m, const std::vector<float>& v) {
  float* p = (float*)frmap_model_dev_alloc(m, v.size() * sizeof(float));
  if (p && hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return p;
}

// eval BatchNorm as y = x * scale + shift (fp32, the Python planner's `_bn_scale_shift`)
void frmap_model_bn_fold(const frmap_model* m, const std::string& p, int c, std::vector<float>* scale, std::vector<float>* shift) {
  const auto& g = m->raw.at(p + ".weight"); const auto& b = m->raw.at(p + ".bias");
  const auto& mu = m->raw.at(p + ".running_mean"); const auto& var = m->raw.at(p + ".running_var");
  scale->resize(c); shift->resize(c);
  for (int i = 0; i < c; ++i) {
    const float s = g[i] / sqrtf(var[i] + 1e-5f);
    volatile float ms = mu[i] * s;   // (two roundings, as two torch kernels give; no contraction into an fma)
    (*scale)[i] = s; (*shift)[i] = b[i] - ms;
  }
}

int frmap_model_pack(frmap_model* m, const std::string& wkey, const std::string& bkey, const std::string& bnkey, int cout, int cin,
                     int k, int stride, int pad, PackedConv* pc, hipStream_t st, const std::vector<float>* w_override) {
  std::vector<float> scale(cout, 1.f), shift(cout, 0.f);
  if (!bnkey.empty()) frmap_model_bn_fold(m, bnkey, cout, &scale, &shift);
  if (!bkey.empty()) {   // a conv / linear bias in front of the BatchNorm lands in the shift: (x + b) * s + t = x * s + (t + b * s)
    const auto& b = m->raw.at(bkey);
    for (int o = 0; o < cout; ++o) { volatile float bs = b[o] * scale[o]; shift[o] = shift[o] + bs; }
  }
  std::vector<float> w = w_override ? *w_override : m->raw.at(wkey);
  const size_t per = (size_t)cin * k * k;
  if (!bnkey.empty())
    for (int o = 0; o < cout; ++o)
      for (size_t i = 0; i < per; ++i) w[o * per + i] *= scale[o];
  float* wdev = nullptr;
  if (hipMalloc((void**)&wdev, w.size() * sizeof(float)) != hipSuccess) { frmap_set_error("model_finalize: out of device memory"); return -2; }
  hipMemcpy(wdev, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice);
  const size_t elems = cin == 3 ? (size_t)cout * frmap_small_cin_kpad(k, k) : w.size();
  pc->wpk = frmap_model_dev_alloc(m, elems * 2);
  pc->shift = frmap_model_dev_upload(m, shift);
  pc->cout = cout; pc->cin = cin; pc->k = k; pc->stride = stride; pc->pad = pad;
  int rc = -2;
  if (pc->wpk && pc->shift)
    rc = cin == 3 ? frmap_pack_conv_weight_c3(wdev, pc->wpk, cout, k, k, m->dtype, st)
                  : frmap_pack_conv_weight(wdev, pc->wpk, cout, cin, k, k, m->dtype, st);
  hipStreamSynchronize(st);
  hipFree(wdev);
  pc->shift_host = shift;
  return rc;
}

frmap_traced::frmap_traced(frmap_run& r_, const char* fmt, double flop, double bytes) : r(r_), on(r_.m->trace && !r_.dry) {
  if (!on) return;
  snprintf(rec.kernel, sizeof(rec.kernel), fmt, r.dt);
  rec.flop = flop; rec.bytes = bytes;
  hipEventCreate(&rec.e0); hipEventCreate(&rec.e1);
  hipEventRecord(rec.e0, r.st);
}
frmap_traced::~frmap_traced() {
  if (!on) return;
  hipEventRecord(rec.e1, r.st);
  std::lock_guard<std::mutex> lock(r.m->trace_mu);
  r.m->recs.push_back(rec);
}

namespace {

auto& dev_alloc = frmap_model_dev_alloc;
auto& dev_upload = frmap_model_dev_upload;
auto& bn_fold = frmap_model_bn_fold;
int pack_conv(frmap_model* m, const std::string& wkey, const std::string& bnkey, int cout, int cin, int k, int stride, int pad,
              PackedConv* pc, std::vector<float>* shift_host, hipStream_t st) {
  const int rc = frmap_model_pack(m, wkey, "", bnkey, cout, cin, k, stride, pad, pc, st);
  if (shift_host) *shift_host = pc->shift_host;
  return rc;
}

// ---- one forward ------------------------------------------------------------------------------------------------------
using Run = frmap_run;
using Traced = frmap_traced;

const char* dt_name(int dtype) { return dtype == FRMAP_BF16 ? "BF16" : "F16"; }

void conv(Run& r, const PackedConv& c, const void* in, int Hi, int Wi, const void* residual, void* out, int relu) {
  if (r.rc) return;
  const int Ho = (Hi + 2 * c.pad - c.k) / c.stride + 1, Wo = (Wi + 2 * c.pad - c.k) / c.stride + 1;
  const char* name = "conv_igemm_kernel<%s>";
  if (c.k == 3 && c.stride == 1) {
    if (frmap_conv3x3_pp_layout(r.B, Hi, Wi, c.cin, c.cout)) name = "conv3x3_pp_kernel<%s>";
    else if (c.cin == 64 && Hi % 8 == 0 && Wi % 8 == 0) name = "conv3x3_c64_wave_kernel<%s>";
    else name = "conv3x3_fast_kernel<%s, false>";
  } else if (c.k == 3 && c.stride == 2) {
    name = frmap_conv3x3s2_pp_layout(r.B, Hi, Wi, c.cin, c.cout) ? "conv3x3s2_pp_kernel<%s>" : "conv3x3s2_fast_kernel<%s>";
  } else if (c.k == 1) {
    name = (c.cin >= 128 && frmap_conv1x1_pp_layout(r.B, Hi, Wi, c.cin, c.cout, c.stride)) ? "conv1x1_pp_kernel<%s>" : "conv1x1_kernel<%s>";
  }
  const double M = (double)r.B * Ho * Wo;
  Traced t(r, name, 2.0 * M * c.cout * c.cin * c.k * c.k,
           2.0 * ((double)r.B * Hi * Wi * c.cin + M * c.cout * (residual ? 2 : 1) + (double)c.cout * c.cin * c.k * c.k));
  r.rc = frmap_conv_igemm(in, c.wpk, c.shift, residual, out, r.B, Hi, Wi, c.cin, c.cout, c.k, c.stride, c.pad, relu, r.m->dtype, r.st);
}

// ResNet-18 trunk: x (fp32 NCHW or uint8 HWC) -> NHWC [B][h][w][512] map in `dtype`.  `slots`: three activation buffers of
// slot_bytes(B, H, W) each.  The last block writes into `final_out` when given.  Returns the map's location and size.
struct MapOut { void* p; int h, w; };

size_t act_slot_bytes(int B, int H, int W) {
  const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;
  const size_t conv_out = (size_t)B * Hc * Wc * 64 * 2, in4 = (size_t)B * H * W * 4 * 2;
  const size_t s = conv_out > in4 ? conv_out : in4;
  return (s + 255) / 256 * 256;
}

MapOut trunk_features(Run& r, const void* x, int x_kind, int H, int W, char* slots, size_t slot, void* final_out) {
  frmap_model* m = r.m;
  void* buf[3] = {slots, slots + slot, slots + 2 * slot};
  const int Hc = (H + 6 - 7) / 2 + 1, Wc = (W + 6 - 7) / 2 + 1;
  const int Hq = (Hc + 2 - 3) / 2 + 1, Wq = (Wc + 2 - 3) / 2 + 1;
  int cur = 0;
  const double stem_flop = 2.0 * r.B * Hc * Wc * 64 * 147;
  const double stem_out = (double)r.B * Hq * Wq * 64 * 2;
  if (x_kind == 1) {
    if (Wq <= 56 && W % 4 == 0) {
      Traced t(r, ((uintptr_t)x & 3) == 0 ? "stem_s2d_kernel<%s, U8>" : "stem_pool_u8_kernel<%s>", stem_flop, (double)r.B * H * W * 3 + stem_out);
      r.rc = frmap_stem7x7_maxpool_u8((const unsigned char*)x, m->mean, m->stdv, m->stem.wpk, m->stem.shift, buf[0], r.B, H, W, 1, m->dtype, r.st);
    } else {
      r.rc = frmap_normalize_u8_hwc((const unsigned char*)x, nullptr, buf[1], r.B, H, W, m->mean, m->stdv, m->dtype, r.st);
      if (!r.rc) r.rc = frmap_conv_small_cin(buf[1], m->stem.wpk, m->stem.shift, buf[2], r.B, H, W, 64, 7, 7, 2, 3, 1, m->dtype, r.st);
      if (!r.rc) r.rc = frmap_maxpool(buf[2], buf[0], r.B, Hc, Wc, 64, 3, 2, 1, m->dtype, r.st);
    }
  } else if (Wq <= 56) {
    // (stem_s2d.hip takes 16-byte aligned tensors with W % 4 == 0; stem_pool.hip the rest)
    Traced t(r, (W % 4 == 0 && ((uintptr_t)x & 15) == 0) ? "stem_s2d_kernel<%s>" : "stem_pool_kernel<%s>", stem_flop, (double)r.B * H * W * 3 * 4 + stem_out);
    r.rc = frmap_stem7x7_maxpool((const float*)x, m->stem.wpk, m->stem.shift, buf[0], r.B, H, W, m->dtype, r.st);
  } else {  // wider than the fused kernel's column strips
    r.rc = frmap_pack_input_nchw_f32((const float*)x, buf[1], r.B, H, W, m->dtype, r.st);
    if (!r.rc) r.rc = frmap_conv_small_cin(buf[1], m->stem.wpk, m->stem.shift, buf[2], r.B, H, W, 64, 7, 7, 2, 3, 1, m->dtype, r.st);
    if (!r.rc) r.rc = frmap_maxpool(buf[2], buf[0], r.B, Hc, Wc, 64, 3, 2, 1, m->dtype, r.st);
  }
  int h = Hq, w = Wq;
  for (int bi = 0; bi < 8 && !r.rc; ++bi) {
    const Block& b = m->blocks[bi];
    const bool last = bi == 7 && final_out;
    void* xin = buf[cur];
    void* hb = buf[(cur + 1) % 3];
    void* third = buf[(cur + 2) % 3];
    const int ho = (h + 2 - 3) / b.c1.stride + 1, wo = (w + 2 - 3) / b.c1.stride + 1;
    conv(r, b.c1, xin, h, w, nullptr, hb, 1);
    if (!b.has_ds) {
      void* dst = last ? final_out : third;
      conv(r, b.c2, hb, ho, wo, xin, dst, 1);
      if (!last) cur = (cur + 2) % 3;
      else buf[cur] = dst;
    } else if (b.ds.k == 1 && frmap_conv_igemm_ds_supported(r.B, ho, wo, b.c2.cin, b.c2.cout, h, w, b.ds.cin, b.ds.stride)) {
      void* dst = last ? final_out : third;
      if (!r.rc) {
        const double M = (double)r.B * ho * wo;
        const char* name = frmap_conv3x3_pp_ds_layout(r.B, ho, wo, b.c2.cin, b.c2.cout, h, w, b.ds.cin, b.ds.stride)
                               ? "conv3x3_pp_kernel<%s, DS>" : "conv3x3_fast_kernel<%s, true>";
        Traced t(r, name, 2.0 * M * b.c2.cout * (b.c2.cin * 9 + b.ds.cin),
                 2.0 * (M * b.c2.cin + M * b.c2.cout + (double)b.c2.cout * b.c2.cin * 9 + (double)b.c2.cout * b.ds.cin + M * b.ds.cin));
        r.rc = frmap_conv_igemm_ds(hb, b.c2.wpk, b.fshift, xin, b.ds.wpk, dst, r.B, ho, wo, b.c2.cin, b.c2.cout, h, w, b.ds.cin,
                                   b.ds.stride, 1, m->dtype, r.st);
      }
      if (!last) cur = (cur + 2) % 3;
      else buf[cur] = dst;
    } else {
      conv(r, b.ds, xin, h, w, nullptr, third, 0);           // shortcut -> third; the block's input is dead after this
      void* dst = last ? final_out : xin;
      conv(r, b.c2, hb, ho, wo, third, dst, 1);
      if (last) buf[cur] = dst;
    }
    h = ho; w = wo;
  }
  return MapOut{buf[cur], h, w};
}

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

int check_ready(const frmap_model* m, const char* what) {
  if (!m) { frmap_set_error("%s: null model", what); return -1; }
  if (!m->finalized) { frmap_set_error("%s: frmap_model_finalize has not run", what); return -1; }
  int dev = -1;
  hipGetDevice(&dev);
  if (dev != m->device) { frmap_set_error("%s: the model lives on device %d but the calling thread's current device is %d", what, m->device, dev); return -1; }
  return 0;
}

}  // namespace

extern "C" int frmap_model_create(frmap_model** out, const char* model_type, int num_classes, int dtype) {
  FRMAP_REQUIRE(out && model_type, "model_create: null pointer");
  FRMAP_REQUIRE(dtype == FRMAP_BF16 || dtype == FRMAP_F16, "model_create: bad dtype %d", dtype);
  int kind;
  if (!strcmp(model_type, "cnn")) kind = KIND_CNN;
  else if (!strcmp(model_type, "arcface")) kind = KIND_ARCFACE;
  else if (!strcmp(model_type, "resnet18_trunk")) kind = KIND_TRUNK;
  else if (!strcmp(model_type, "baseline")) kind = KIND_BASELINE;
  else if (!strcmp(model_type, "siamese")) kind = KIND_SIAMESE;
  else if (!strcmp(model_type, "hybrid")) kind = KIND_HYBRID;
  else { frmap_set_error("Invalid model type: %s (model handles exist for 'baseline', 'cnn', 'siamese', 'arcface', 'hybrid', 'resnet18_trunk')", model_type); return -1; }
  FRMAP_REQUIRE(kind == KIND_TRUNK || kind == KIND_SIAMESE || num_classes > 0, "model_create: num_classes=%d", num_classes);
  frmap_model* m = new (std::nothrow) frmap_model();
  FRMAP_REQUIRE(m, "model_create: out of host memory");
  m->kind = kind; m->dtype = dtype; m->num_classes = num_classes;
  if (hipGetDevice(&m->device) != hipSuccess) { delete m; frmap_set_error("model_create: no HIP device"); return -2; }
  *out = m;
  return 0;
}

extern "C" int frmap_model_load_tensor(frmap_model* m, const char* key, const void* data, size_t numel, int on_device) {
  FRMAP_REQUIRE(m && key && data, "model_load_tensor: null pointer");
  FRMAP_REQUIRE(!m->finalized, "model_load_tensor: the model is finalized (create a new handle for new weights)");
  std::string ck;
  if (!canonical_key(m->kind, key, &ck)) return 1;   // not part of the inference path (training head, counters, fc of a bare trunk)
  const auto want = expected_tensors(m);
  const auto it = want.find(ck);
  if (it == want.end()) return 1;
  FRMAP_REQUIRE(it->second == numel, "model_load_tensor: %s has %zu elements, expected %zu", key, numel, it->second);
  std::vector<float> v(numel);
  if (on_device) {
    const hipError_t e = hipMemcpy(v.data(), data, numel * sizeof(float), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { frmap_set_error("model_load_tensor: %s", hipGetErrorString(e)); return -2; }
  } else {
    memcpy(v.data(), data, numel * sizeof(float));
  }
  m->raw[ck] = std::move(v);
  return 0;
}

extern "C" int frmap_model_set_input_normalization(frmap_model* m, const float* mean3_host, const float* std3_host) {
  FRMAP_REQUIRE(m && mean3_host && std3_host, "model_set_input_normalization: null pointer");
  for (int i = 0; i < 3; ++i) { m->mean[i] = mean3_host[i]; m->stdv[i] = std3_host[i]; }
  return 0;
}

extern "C" int frmap_model_finalize(frmap_model* m, void* stream) {
  FRMAP_REQUIRE(m, "model_finalize: null model");
  FRMAP_REQUIRE(!m->finalized, "model_finalize: already finalized");
  hipStream_t st = (hipStream_t)stream;
  for (const auto& kv : expected_tensors(m))
    FRMAP_REQUIRE(m->raw.count(kv.first), "model_finalize: tensor %s was never loaded", kv.first.c_str());
  const bool has_trunk = m->kind != KIND_BASELINE && m->kind != KIND_SIAMESE;
  int rc = has_trunk ? pack_conv(m, "trunk.conv1.weight", "trunk.bn1", 64, 3, 7, 2, 3, &m->stem, nullptr, st) : 0;
  int inpl = 64;
  for (int li = 0; li < 4 && !rc && has_trunk; ++li) {
    const int planes = kStagePlanes[li];
    for (int bi = 0; bi < 2 && !rc; ++bi) {
      Block& b = m->blocks[li * 2 + bi];
      const std::string p = "trunk.layer" + std::to_string(li + 1) + "." + std::to_string(bi);
      const int cin = bi == 0 ? inpl : planes, stride = (bi == 0 && li > 0) ? 2 : 1;
      std::vector<float> s2, sd;
      rc = pack_conv(m, p + ".conv1.weight", p + ".bn1", planes, cin, 3, stride, 1, &b.c1, nullptr, st);
      if (!rc) rc = pack_conv(m, p + ".conv2.weight", p + ".bn2", planes, planes, 3, 1, 1, &b.c2, &s2, st);
      if (!rc && bi == 0 && li > 0) {
        b.has_ds = true;
        rc = pack_conv(m, p + ".downsample.0.weight", p + ".downsample.1", planes, cin, 1, stride, 0, &b.ds, &sd, st);
        for (int i = 0; i < planes; ++i) s2[i] += sd[i];
        b.fshift = dev_upload(m, s2);
        if (!rc && !b.fshift) rc = -2;
      }
    }
    inpl = planes;
  }
  if (!rc && m->kind == KIND_CNN) {
    m->fc_w = dev_upload(m, m->raw.at("fc.weight"));
    m->fc_b = dev_upload(m, m->raw.at("fc.bias"));
    if (!m->fc_w || !m->fc_b) rc = -2;
  }
  if (!rc && m->kind == KIND_ARCFACE) {
    std::vector<float> scale, shift;
    bn_fold(m, "bn", 512, &scale, &shift);
    const auto& w = m->raw.at("embedding.weight");     // [N][K] -> [K][N] for the fused head kernel
    std::vector<float> wt(w.size());
    for (int n = 0; n < 512; ++n)
      for (int k = 0; k < 512; ++k) wt[(size_t)k * 512 + n] = w[(size_t)n * 512 + k];
    m->emb_wt = dev_upload(m, wt);
    m->bn_scale = dev_upload(m, scale);
    m->bn_shift = dev_upload(m, shift);
    float* cls = dev_upload(m, m->raw.at("val_classifier.weight"));
    m->cls_wn = (float*)dev_alloc(m, (size_t)m->num_classes * 512 * sizeof(float));
    m->cls_b = dev_upload(m, m->raw.at("val_classifier.bias"));
    if (!m->emb_wt || !m->bn_scale || !m->bn_shift || !cls || !m->cls_wn || !m->cls_b) rc = -2;
    if (!rc) rc = frmap_l2_normalize_f32(cls, m->cls_wn, m->num_classes, 512, 1e-12f, st);   // face_models.py:576
  }
  if (!rc && m->kind >= KIND_BASELINE) rc = frmap_family_finalize(m, st);
  if (rc == -2 && !*frmap_last_error()) frmap_set_error("model_finalize: out of device memory");
  if (rc) return rc;
  if (hipStreamSynchronize(st) != hipSuccess) { frmap_set_error("model_finalize: stream error"); return -2; }
  m->raw.clear();
  m->finalized = true;
  return 0;
}

extern "C" int frmap_model_embedding_dim(const frmap_model* m) { return !m ? 0 : m->kind == KIND_SIAMESE ? 256 : 512; }

extern "C" size_t frmap_model_workspace_bytes(const frmap_model* m, int B, int H, int W) {
  if (!m || B <= 0 || H <= 0 || W <= 0) return 0;
  const size_t heads = 2 * align256((size_t)B * 512 * sizeof(float)) + 256;
  if (m->kind >= KIND_BASELINE) {   // the families' bump arena, sized by running their forward without launching
    Run r{const_cast<frmap_model*>(m), nullptr, B, "", 0, true};
    const size_t fam = frmap_family_forward(r, nullptr, FRMAP_INPUT_F32_NCHW, H, W, FRMAP_OUT_LOGITS, nullptr, nullptr, nullptr, nullptr);
    if (m->kind == KIND_HYBRID) return 3 * act_slot_bytes(B, H, W) + align256((size_t)B * 49 * 512 * 2) + fam + heads;
    return fam + heads;
  }
  return 3 * act_slot_bytes(B, H, W) + heads;
}

extern "C" size_t frmap_model_match_workspace_bytes(const frmap_model* m, int B, int H, int W, int G) {
  if (!m || B <= 0) return 0;
  return frmap_model_workspace_bytes(m, B, H, W) + align256(frmap_match_workspace_bytes(B, G)) + align256((size_t)B * 3 * 512 * 2);   // (+ the probes' fp16 split)
}

extern "C" int frmap_model_forward(frmap_model* m, const void* x, int x_kind, int B, int H, int W, int what, void* out,
                                   void* workspace, void* stream) {
  if (int rc = check_ready(m, "model_forward")) return rc;
  FRMAP_REQUIRE(x && out && workspace, "model_forward: null pointer");
  FRMAP_REQUIRE(B > 0 && H >= 7 && W >= 7, "model_forward: bad shape B=%d H=%d W=%d", B, H, W);
  FRMAP_REQUIRE(x_kind == FRMAP_INPUT_F32_NCHW || x_kind == FRMAP_INPUT_U8_HWC, "model_forward: bad input kind %d", x_kind);
  FRMAP_REQUIRE(what >= FRMAP_OUT_TRUNK_MAP && what <= FRMAP_OUT_LOGITS, "model_forward: bad output selector %d", what);
  FRMAP_REQUIRE(m->kind != KIND_TRUNK || what <= FRMAP_OUT_POOLED, "model_forward: a bare trunk has no embedding / logits head");
  Run r{m, (hipStream_t)stream, B, dt_name(m->dtype)};
  if (m->kind == KIND_BASELINE || m->kind == KIND_SIAMESE) {
    FRMAP_REQUIRE(what != FRMAP_OUT_POOLED, "model_forward: '%s' has no pooled-trunk output", m->kind == KIND_BASELINE ? "baseline" : "siamese");
    FRMAP_REQUIRE(m->kind != KIND_SIAMESE || what != FRMAP_OUT_LOGITS, "model_forward: 'siamese' has no classifier head (forward(x1, x2) = two embeddings)");
    frmap_family_forward(r, x, x_kind, H, W, what, out, nullptr, (char*)workspace, nullptr);
    return r.rc;
  }
  const size_t slot = act_slot_bytes(B, H, W);
  char* ws = (char*)workspace;
  float* scratch0 = (float*)(ws + 3 * slot);
  if (m->kind == KIND_HYBRID && what >= FRMAP_OUT_EMBEDDING) {
    {   // (checked BEFORE anything is launched: a rejected call launches nothing)
      int th = ((H + 6 - 7) / 2 + 1 + 2 - 3) / 2 + 1, tw = ((W + 6 - 7) / 2 + 1 + 2 - 3) / 2 + 1;
      for (int i = 0; i < 3; ++i) { th = (th - 1) / 2 + 1; tw = (tw - 1) / 2 + 1; }
      FRMAP_REQUIRE(th * tw == 49, "model_forward: HybridNet expects a 49-token feature map (224x224 input), got %d", th * tw);
    }
    void* tmap = ws + 3 * slot;
    const MapOut hm = trunk_features(r, x, x_kind, H, W, ws, slot, tmap);
    if (r.rc) return r.rc;
    (void)hm;
    frmap_family_forward(r, nullptr, x_kind, H, W, what, out, nullptr, ws + 3 * slot + align256((size_t)B * 49 * 512 * 2), tmap);
    return r.rc;
  }
  const MapOut map = trunk_features(r, x, x_kind, H, W, ws, slot, what == FRMAP_OUT_TRUNK_MAP ? out : nullptr);
  if (r.rc) return r.rc;
  const int HW = map.h * map.w;
  if (what == FRMAP_OUT_TRUNK_MAP) return 0;
  FRMAP_REQUIRE(m->kind != KIND_HYBRID || what == FRMAP_OUT_POOLED, "model_forward: bad output selector");
  const bool arc = m->kind == KIND_ARCFACE;
  if (what == FRMAP_OUT_POOLED || (!arc && what == FRMAP_OUT_EMBEDDING))
    return frmap_avgpool_global(map.p, (float*)out, B, HW, 512, m->dtype, r.st);
  if (!arc) {  // 'cnn' logits: avgpool -> dropout (eval: identity) -> fc   (face_models.py:93-96)
    int rc = frmap_avgpool_global(map.p, scratch0, B, HW, 512, m->dtype, r.st);
    if (!rc) rc = frmap_linear_f32(scratch0, m->fc_w, nullptr, m->fc_b, (float*)out, B, 512, m->num_classes, 0, r.st);
    return rc;
  }
  // 'arcface': avgpool + embedding + bn + F.normalize in one launch (face_models.py:573-590)
  float* emb = what == FRMAP_OUT_EMBEDDING ? (float*)out : scratch0;
  {
    Traced t(r, "gap_linear_norm_kernel<%s>", 2.0 * B * 512 * 512, 2.0 * B * HW * 512 + 4.0 * 512 * 512 + 4.0 * B * 512);
    r.rc = frmap_gap_linear_norm(map.p, m->emb_wt, m->bn_scale, m->bn_shift, nullptr, emb, 1e-12f, B, HW, 512, 512, 0, m->dtype, r.st);
  }
  if (r.rc || what == FRMAP_OUT_EMBEDDING) return r.rc;
  return frmap_linear_f32(emb, m->cls_wn, nullptr, m->cls_b, (float*)out, B, 512, m->num_classes, 0, r.st);   // :576-580
}

extern "C" int frmap_model_embed_and_match(frmap_model* m, const void* x, int x_kind, int B, int H, int W, const float* gallery,
                                           const void* gallery_packed, const float* gallery_stat, int G, float thresh,
                                           int normalize, int32_t* idx_out, float* dist_out, int32_t* id_or_unknown_out,
                                           int32_t* packed_out, float* emb_out, void* workspace, void* stream) {
  if (int rc = check_ready(m, "model_embed_and_match")) return rc;
  FRMAP_REQUIRE(x && idx_out && dist_out && workspace, "model_embed_and_match: null pointer");
  FRMAP_REQUIRE(m->kind != KIND_TRUNK, "model_embed_and_match: a bare trunk has no embedding");
  FRMAP_REQUIRE(B > 0 && H >= 7 && W >= 7 && G >= 0 && (G == 0 || gallery), "model_embed_and_match: bad shape B=%d H=%d W=%d G=%d", B, H, W, G);
  FRMAP_REQUIRE(x_kind == FRMAP_INPUT_F32_NCHW || x_kind == FRMAP_INPUT_U8_HWC, "model_embed_and_match: bad input kind %d", x_kind);
  Run r{m, (hipStream_t)stream, B, dt_name(m->dtype)};
  const size_t slot = act_slot_bytes(B, H, W);
  char* ws = (char*)workspace;
  char* match_ws = ws + frmap_model_workspace_bytes(m, B, H, W);
  if (m->kind >= KIND_BASELINE) {
    // the family forward fills the embedding the matcher needs: its unit-norm copy when `normalize`, else what get_embedding returns
    const int D = frmap_model_embedding_dim(m);
    const size_t model_ws = frmap_model_workspace_bytes(m, B, H, W);
    float* e0 = (float*)(ws + model_ws - 2 * align256((size_t)B * 512 * sizeof(float)) - 256);
    float* e1 = e0 + align256((size_t)B * 512 * sizeof(float)) / sizeof(float);
    float* emb = emb_out ? emb_out : e0;
    if (m->kind == KIND_HYBRID) {
      int th = ((H + 6 - 7) / 2 + 1 + 2 - 3) / 2 + 1, tw = ((W + 6 - 7) / 2 + 1 + 2 - 3) / 2 + 1;
      for (int i = 0; i < 3; ++i) { th = (th - 1) / 2 + 1; tw = (tw - 1) / 2 + 1; }
      FRMAP_REQUIRE(th * tw == 49, "model_embed_and_match: HybridNet expects a 49-token feature map (224x224 input), got %d", th * tw);
      void* tmap = ws + 3 * slot;
      trunk_features(r, x, x_kind, H, W, ws, slot, tmap);
      if (r.rc) return r.rc;
      frmap_family_forward(r, nullptr, x_kind, H, W, FRMAP_OUT_EMBEDDING, normalize ? e1 : emb, normalize ? emb : nullptr,
                           ws + 3 * slot + align256((size_t)B * 49 * 512 * 2), tmap);
    } else if (m->kind == KIND_BASELINE) {
      frmap_family_forward(r, x, x_kind, H, W, FRMAP_OUT_EMBEDDING, normalize ? e1 : emb, normalize ? emb : nullptr, ws, nullptr);
    } else {
      frmap_family_forward(r, x, x_kind, H, W, FRMAP_OUT_EMBEDDING, emb, nullptr, ws, nullptr);   // (already unit-norm)
    }
    if (r.rc) return r.rc;
    if (gallery_packed && gallery_stat && G >= 512 && D % 32 == 0) {
      void* split = match_ws + align256(frmap_match_workspace_bytes(B, G));
      return frmap_match_top1_packed(emb, gallery, gallery_packed, gallery_stat, idx_out, dist_out, id_or_unknown_out, packed_out, thresh,
                                     match_ws, split, B, G, D, r.st);
    }
    return frmap_match_top1(emb, gallery, idx_out, dist_out, id_or_unknown_out, packed_out, thresh, match_ws, B, G, D, r.st);
  }
  float* scratch0 = (float*)(ws + 3 * slot);
  float* scratch1 = (float*)(ws + 3 * slot + align256((size_t)B * 512 * sizeof(float)));
  const MapOut map = trunk_features(r, x, x_kind, H, W, ws, slot, nullptr);
  if (r.rc) return r.rc;
  const int HW = map.h * map.w;
  if (m->kind == KIND_CNN && G <= 64) {
    // pool + (normalise) + compare_faces' scan in one launch, one workgroup per face
    Traced t(r, "gap_norm_match_kernel<%s>", 2.0 * B * 512 * G, 2.0 * B * HW * 512 + 4.0 * G * 512 + 16.0 * B);
    return frmap_gap_norm_match(map.p, gallery, emb_out, idx_out, dist_out, id_or_unknown_out, packed_out, thresh, normalize ? 1 : 0,
                                1e-12f, B, HW, 512, G, m->dtype, r.st);
  }
  float* emb = emb_out ? emb_out : scratch0;
  if (m->kind == KIND_CNN) {
    r.rc = frmap_avgpool_global(map.p, normalize ? scratch1 : emb, B, HW, 512, m->dtype, r.st);
    if (!r.rc && normalize) r.rc = frmap_l2_normalize_f32(scratch1, emb, B, 512, 1e-12f, r.st);
  } else {
    Traced t(r, "gap_linear_norm_kernel<%s>", 2.0 * B * 512 * 512, 2.0 * B * HW * 512 + 4.0 * 512 * 512 + 4.0 * B * 512);
    r.rc = frmap_gap_linear_norm(map.p, m->emb_wt, m->bn_scale, m->bn_shift, nullptr, emb, 1e-12f, B, HW, 512, 512, 0, m->dtype, r.st);
  }
  if (r.rc) return r.rc;
  if (gallery_packed && gallery_stat && G >= 512) {
    void* split = match_ws + align256(frmap_match_workspace_bytes(B, G));
    Traced t(r, "match_top1 (conv1x1_pp_kernel<F16, MATCH> + finalize)", 6.0 * B * 512 * G, 4.0 * B * 512 + 6.0 * G * 512 + 16.0 * B);
    return frmap_match_top1_packed(emb, gallery, gallery_packed, gallery_stat, idx_out, dist_out, id_or_unknown_out, packed_out, thresh,
                                   match_ws, split, B, G, 512, r.st);
  }
  Traced t(r, "match_top1 (gemm_nt_f32_kernel + finalize)", 2.0 * B * 512 * G, 4.0 * B * 512 + 4.0 * G * 512 + 16.0 * B);
  return frmap_match_top1(emb, gallery, idx_out, dist_out, id_or_unknown_out, packed_out, thresh, match_ws, B, G, 512, r.st);
}

extern "C" int frmap_model_trace(frmap_model* m, int enable) {
  FRMAP_REQUIRE(m, "model_trace: null model");
  std::lock_guard<std::mutex> lock(m->trace_mu);
  m->trace = enable != 0;
  return 0;
}

extern "C" int frmap_model_trace_read(frmap_model* m, frmap_trace_record* out, int max_records) {
  FRMAP_REQUIRE(m && (out || max_records == 0), "model_trace_read: null pointer");
  std::lock_guard<std::mutex> lock(m->trace_mu);
  int n = 0;
  for (auto& rec : m->recs) {
    hipEventSynchronize(rec.e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, rec.e0, rec.e1);
    if (n < max_records) {
      memset(&out[n], 0, sizeof(out[n]));
      strncpy(out[n].kernel, rec.kernel, sizeof(out[n].kernel) - 1);
      out[n].flop = rec.flop; out[n].bytes = rec.bytes; out[n].us = ms * 1e3f;
      ++n;
    }
    hipEventDestroy(rec.e0); hipEventDestroy(rec.e1);
  }
  m->recs.clear();
  return n;
}

extern "C" void frmap_model_destroy(frmap_model* m) {
  if (!m) return;
  for (auto& rec : m->recs) { hipEventDestroy(rec.e0); hipEventDestroy(rec.e1); }
  for (void* p : m->allocs) hipFree(p);
  delete m;
}
