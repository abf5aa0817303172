// Model handles of the non-ResNet-transfer families (include/frmap_hip.h, "Model handles"; state in model_api.h):
//   'baseline'  BaselineNet   /root/reference/src/face_models.py:16-60    conv-bn-relu-pool x3 -> GAP -> fc1 + ReLU (-> fc2)
//   'siamese'   SiameseNet    :104-192   7x7 stem + pool, five 3x3 convs (three pooled), AdaptiveAvgPool2d((6, 6)), three Linear
//                                        (+BatchNorm1d + ReLU), F.normalize;  forward(x1, x2) = two forwards of one tower
//   'hybrid'    HybridNet     :650-721   ResNet-18 trunk (model_api.cpp) -> + pos -> pre-LN transformer block (:618-648) ->
//                                        token mean -> LayerNorm (-> fc)
// Same contract as the ResNet handles: tensors arrive under the reference's state_dict keys, BatchNorm / bias folding in fp32 on
// the host, a forward launches the fused per-op entry points on the caller's stream and writes only into the caller's workspace.
// The workspace is a bump arena; frmap_model_workspace_bytes runs the same code without launching (`dry`) to size it.
#include <math.h>
#include <string.h>

#include "model_api.h"

namespace {

void add_bn(std::map<std::string, size_t>* want, const std::string& p, int c) {
  (*want)[p + ".weight"] = c; (*want)[p + ".bias"] = c; (*want)[p + ".running_mean"] = c; (*want)[p + ".running_var"] = c;
}

// siamese tower: (conv index in nn.Sequential, bn index, Cin, Cout, MaxPool2d(2) after it)
struct SiaConv { int ci, bi, cin, cout; bool pool; };
const SiaConv kSia[6] = {{0, 1, 3, 64, true}, {4, 5, 64, 128, false}, {7, 8, 128, 128, true},
                         {11, 12, 128, 256, false}, {14, 15, 256, 256, true}, {18, 19, 256, 512, false}};

struct Arena {
  char* base;
  size_t off = 0;
  void* take(size_t bytes) {
    void* p = base ? base + off : nullptr;
    off += (bytes + 255) / 256 * 256;
    return p;
  }
};

#define GO(expr)                           \
  do {                                     \
    if (!r.dry && !r.rc) r.rc = (expr);    \
  } while (0)

// conv 3x3 s1 p1 + shift + ReLU, then MaxPool2d(2, 2) when `pool` - one launch where the fused kernels take the shape
void* conv3(frmap_run& r, Arena& a, const PackedConv& c, const void* in, int& H, int& W, bool pool) {
  const size_t esz = 2;
  if (pool && H % 2 == 0 && W % 2 == 0 && frmap_conv_igemm_pool2_supported(r.B, H, W, c.cin, c.cout)) {
    void* out = a.take((size_t)r.B * (H / 2) * (W / 2) * c.cout * esz);
    {
      frmap_traced t(r, "conv3x3 + 2x2 max-pool<%s>", 2.0 * r.B * H * W * c.cout * c.cin * 9,
                     2.0 * ((double)r.B * H * W * c.cin + (double)r.B * (H / 2) * (W / 2) * c.cout + (double)c.cout * c.cin * 9));
      GO(frmap_conv_igemm_pool2(in, c.wpk, c.shift, out, r.B, H, W, c.cin, c.cout, 1, r.m->dtype, r.st));
    }
    H /= 2; W /= 2;
    return out;
  }
  void* full = a.take((size_t)r.B * H * W * c.cout * esz);
  {
    frmap_traced t(r, "conv_igemm 3x3<%s>", 2.0 * r.B * H * W * c.cout * c.cin * 9,
                   2.0 * ((double)r.B * H * W * (c.cin + c.cout) + (double)c.cout * c.cin * 9));
    GO(frmap_conv_igemm(in, c.wpk, c.shift, nullptr, full, r.B, H, W, c.cin, c.cout, 3, 1, 1, 1, r.m->dtype, r.st));
  }
  if (!pool) return full;
  void* out = a.take((size_t)r.B * (H / 2) * (W / 2) * c.cout * esz);
  GO(frmap_maxpool(full, out, r.B, H, W, c.cout, 2, 2, 0, r.m->dtype, r.st));
  H /= 2; W /= 2;
  return out;
}

void* nhwc4_input(frmap_run& r, Arena& a, const void* x, int x_kind, int H, int W) {
  void* x4 = a.take((size_t)r.B * H * W * 4 * 2);
  if (x_kind == FRMAP_INPUT_U8_HWC)
    GO(frmap_normalize_u8_hwc((const unsigned char*)x, nullptr, x4, r.B, H, W, r.m->mean, r.m->stdv, r.m->dtype, r.st));
  else
    GO(frmap_pack_input_nchw_f32((const float*)x, x4, r.B, H, W, r.m->dtype, r.st));
  return x4;
}

void* linear(frmap_run& r, Arena& a, const PackedConv& l, const void* x2d, int M, int act, const void* residual) {
  void* out = a.take((size_t)M * l.cout * 2);
  const size_t wsb = frmap_linear_mfma_workspace_bytes(M, l.cin, l.cout);
  void* ws = wsb ? a.take(wsb) : nullptr;
  frmap_traced t(r, "linear_mfma<%s> (1x1 MFMA kernels)", 2.0 * M * l.cin * l.cout,
                 2.0 * ((double)M * (l.cin + l.cout * (residual ? 2 : 1)) + (double)l.cin * l.cout));
  GO(frmap_linear_mfma(x2d, l.wpk, l.shift, residual, out, ws, M, l.cin, l.cout, act, r.m->dtype, r.st));
  return out;
}

}  // namespace

bool frmap_family_key(int kind, const std::string& key, std::string* out) {
  if (key.size() > 19 && key.compare(key.size() - 19, 19, "num_batches_tracked") == 0) return false;
  if (kind == KIND_HYBRID) {
    if (key.compare(0, 4, "cnn.") == 0) {
      if (key.compare(0, 7, "cnn.fc.") == 0) return false;
      *out = "trunk." + key.substr(4);
      return true;
    }
    *out = key;   // pos_encoding, transformer.*, norm.*, fc.*  (features.* aliases are mapped by the caller)
    return true;
  }
  *out = key;     // baseline / siamese: the module's own names
  return true;
}

void frmap_family_expected(const frmap_model* m, std::map<std::string, size_t>* want) {
  if (m->kind == KIND_BASELINE) {
    const int ch[4] = {3, 32, 64, 128};
    for (int i = 1; i <= 3; ++i) {
      const std::string n = std::to_string(i);
      (*want)["conv" + n + ".weight"] = (size_t)ch[i] * ch[i - 1] * 9;
      (*want)["conv" + n + ".bias"] = ch[i];
      add_bn(want, "bn" + n, ch[i]);
    }
    (*want)["fc1.weight"] = 512 * 128; (*want)["fc1.bias"] = 512;
    (*want)["fc2.weight"] = (size_t)m->num_classes * 512; (*want)["fc2.bias"] = m->num_classes;
  } else if (m->kind == KIND_SIAMESE) {
    for (const SiaConv& c : kSia) {
      const int k = c.cin == 3 ? 7 : 3;
      (*want)["conv." + std::to_string(c.ci) + ".weight"] = (size_t)c.cout * c.cin * k * k;
      (*want)["conv." + std::to_string(c.ci) + ".bias"] = c.cout;
      add_bn(want, "conv." + std::to_string(c.bi), c.cout);
    }
    (*want)["fc.1.weight"] = (size_t)1024 * 18432; (*want)["fc.1.bias"] = 1024; add_bn(want, "fc.2", 1024);
    (*want)["fc.5.weight"] = 512 * 1024; (*want)["fc.5.bias"] = 512; add_bn(want, "fc.6", 512);
    (*want)["fc.8.weight"] = 256 * 512; (*want)["fc.8.bias"] = 256;
  } else if (m->kind == KIND_HYBRID) {
    (*want)["pos_encoding"] = 49 * 512;
    (*want)["transformer.attention.in_proj_weight"] = 1536 * 512; (*want)["transformer.attention.in_proj_bias"] = 1536;
    (*want)["transformer.attention.out_proj.weight"] = 512 * 512; (*want)["transformer.attention.out_proj.bias"] = 512;
    for (const char* n : {"transformer.norm1", "transformer.norm2", "norm"}) {
      (*want)[std::string(n) + ".weight"] = 512; (*want)[std::string(n) + ".bias"] = 512;
    }
    (*want)["transformer.ff.0.weight"] = 2048 * 512; (*want)["transformer.ff.0.bias"] = 2048;
    (*want)["transformer.ff.3.weight"] = 512 * 2048; (*want)["transformer.ff.3.bias"] = 512;
    (*want)["fc.weight"] = (size_t)m->num_classes * 512; (*want)["fc.bias"] = m->num_classes;
  }
}

int frmap_family_finalize(frmap_model* m, hipStream_t st) {
  int rc = 0;
  if (m->kind == KIND_BASELINE) {
    const int ch[4] = {3, 32, 64, 128};
    for (int i = 1; i <= 3 && !rc; ++i) {
      const std::string n = std::to_string(i);
      rc = frmap_model_pack(m, "conv" + n + ".weight", "conv" + n + ".bias", "bn" + n, ch[i], ch[i - 1], 3, 1, 1, &m->convs[i - 1], st);
    }
    const auto& w = m->raw.at("fc1.weight");   // [512][128] -> [128][512] for the pool + Linear + normalise head
    std::vector<float> wt(w.size());
    for (int n = 0; n < 512; ++n)
      for (int k = 0; k < 128; ++k) wt[(size_t)k * 512 + n] = w[(size_t)n * 128 + k];
    m->head_wt = frmap_model_dev_upload(m, wt);
    m->head_b = frmap_model_dev_upload(m, m->raw.at("fc1.bias"));
    m->fc_w = frmap_model_dev_upload(m, m->raw.at("fc2.weight"));
    m->fc_b = frmap_model_dev_upload(m, m->raw.at("fc2.bias"));
    if (!rc && !(m->head_wt && m->head_b && m->fc_w && m->fc_b)) rc = -2;
  } else if (m->kind == KIND_SIAMESE) {
    for (int i = 0; i < 6 && !rc; ++i) {
      const SiaConv& c = kSia[i];
      const int k = c.cin == 3 ? 7 : 3;
      rc = frmap_model_pack(m, "conv." + std::to_string(c.ci) + ".weight", "conv." + std::to_string(c.ci) + ".bias",
                            "conv." + std::to_string(c.bi), c.cout, c.cin, k, c.cin == 3 ? 2 : 1, c.cin == 3 ? 3 : 1, &m->convs[i], st);
    }
    // fc.1 consumes the NCHW flatten (index c * 36 + s, face_models.py:171); the pooled tensor here is NHWC (index s * 512 + c):
    // permute the weight's input axis once
    if (!rc) {
      const auto& w = m->raw.at("fc.1.weight");
      std::vector<float> wp(w.size());
      for (int n = 0; n < 1024; ++n)
        for (int c = 0; c < 512; ++c)
          for (int s = 0; s < 36; ++s) wp[(size_t)n * 18432 + s * 512 + c] = w[(size_t)n * 18432 + c * 36 + s];
      rc = frmap_model_pack(m, "", "fc.1.bias", "fc.2", 1024, 18432, 1, 1, 0, &m->lin[0], st, &wp);
    }
    if (!rc) rc = frmap_model_pack(m, "fc.5.weight", "fc.5.bias", "fc.6", 512, 1024, 1, 1, 0, &m->lin[1], st);
    if (!rc) rc = frmap_model_pack(m, "fc.8.weight", "fc.8.bias", "", 256, 512, 1, 1, 0, &m->lin[2], st);
  } else if (m->kind == KIND_HYBRID) {
    rc = frmap_model_pack(m, "transformer.attention.in_proj_weight", "transformer.attention.in_proj_bias", "", 1536, 512, 1, 1, 0, &m->lin[0], st);
    if (!rc) rc = frmap_model_pack(m, "transformer.attention.out_proj.weight", "transformer.attention.out_proj.bias", "", 512, 512, 1, 1, 0, &m->lin[1], st);
    if (!rc) rc = frmap_model_pack(m, "transformer.ff.0.weight", "transformer.ff.0.bias", "", 2048, 512, 1, 1, 0, &m->lin[2], st);
    if (!rc) rc = frmap_model_pack(m, "transformer.ff.3.weight", "transformer.ff.3.bias", "", 512, 2048, 1, 1, 0, &m->lin[3], st);
    m->pos = frmap_model_dev_upload(m, m->raw.at("pos_encoding"));
    const char* names[3] = {"transformer.norm1", "transformer.norm2", "norm"};
    for (int i = 0; i < 3; ++i) {
      m->ln[2 * i] = frmap_model_dev_upload(m, m->raw.at(std::string(names[i]) + ".weight"));
      m->ln[2 * i + 1] = frmap_model_dev_upload(m, m->raw.at(std::string(names[i]) + ".bias"));
      if (!m->ln[2 * i] || !m->ln[2 * i + 1]) rc = rc ? rc : -2;
    }
    m->fc_w = frmap_model_dev_upload(m, m->raw.at("fc.weight"));
    m->fc_b = frmap_model_dev_upload(m, m->raw.at("fc.bias"));
    if (!rc && !(m->pos && m->fc_w && m->fc_b)) rc = -2;
  }
  return rc;
}

// `what`: FRMAP_OUT_TRUNK_MAP (last conv map), FRMAP_OUT_EMBEDDING, FRMAP_OUT_LOGITS.  `unit_out` (optional, fp32 [B][dim]):
// F.normalize(embedding) for the matcher.  hybrid: `trunk_map` = the ResNet trunk's [B][49][512] map (already computed).
size_t frmap_family_forward(frmap_run& r, const void* x, int x_kind, int H, int W, int what, void* out, float* unit_out, char* ws,
                            const void* trunk_map) {
  frmap_model* m = r.m;
  Arena a{r.dry ? nullptr : ws};
  const int B = r.B;
  if (m->kind == KIND_BASELINE) {
    int h = H, w = W;
    void* x4 = nhwc4_input(r, a, x, x_kind, H, W);
    void* c1;
    if (H % 2 == 0 && W % 2 == 0) {   // self.pool(F.relu(self.bn1(self.conv1(x)))) in one launch (face_models.py:38)
      c1 = a.take((size_t)B * (H / 2) * (W / 2) * 32 * 2);
      frmap_traced t(r, "conv_small_cin_kernel<%s, POOL>", 2.0 * B * H * W * 32 * 27, 2.0 * ((double)B * H * W * 4 + (double)B * (H / 2) * (W / 2) * 32));
      GO(frmap_conv_small_cin_pool2(x4, m->convs[0].wpk, m->convs[0].shift, c1, B, H, W, 32, 1, m->dtype, r.st));
      h /= 2; w /= 2;
    } else {
      void* full = a.take((size_t)B * H * W * 32 * 2);
      GO(frmap_conv_small_cin(x4, m->convs[0].wpk, m->convs[0].shift, full, B, H, W, 32, 3, 3, 1, 1, 1, m->dtype, r.st));
      c1 = a.take((size_t)B * (H / 2) * (W / 2) * 32 * 2);
      GO(frmap_maxpool(full, c1, B, H, W, 32, 2, 2, 0, m->dtype, r.st));
      h /= 2; w /= 2;
    }
    void* c2 = conv3(r, a, m->convs[1], c1, h, w, true);
    void* map = conv3(r, a, m->convs[2], c2, h, w, true);
    if (what == FRMAP_OUT_TRUNK_MAP) {
      if (!r.dry && !r.rc) r.rc = hipMemcpyAsync(out, map, (size_t)B * h * w * 128 * 2, hipMemcpyDeviceToDevice, r.st) == hipSuccess ? 0 : -2;
      return a.off;
    }
    // adaptive_pool + fc1 + ReLU (face_models.py:41-46) in one launch; its unit-norm copy for the matcher comes with it
    float* pre = what == FRMAP_OUT_EMBEDDING ? (float*)out : (float*)a.take((size_t)B * 512 * 4);
    {
      frmap_traced t(r, "gap_linear_norm_kernel<%s>", 2.0 * B * 128 * 512, 2.0 * B * h * w * 128 + 4.0 * 128 * 512 + 4.0 * B * 512);
      GO(frmap_gap_linear_norm(map, m->head_wt, nullptr, m->head_b, pre, unit_out, 1e-12f, B, h * w, 128, 512, 1, m->dtype, r.st));
    }
    if (what == FRMAP_OUT_LOGITS) GO(frmap_linear_f32(pre, m->fc_w, nullptr, m->fc_b, (float*)out, B, 512, m->num_classes, 0, r.st));
    return a.off;
  }
  if (m->kind == KIND_SIAMESE) {
    int h, w;
    void* cur;
    const int Wc = (W + 6 - 7) / 2 + 1, Hc = (H + 6 - 7) / 2 + 1;
    int first = 1;
    if (Wc / 2 <= 64 && (x_kind != FRMAP_INPUT_U8_HWC || W % 4 == 0)) {   // conv.0-3 fused: 7x7 conv + bias + BN + ReLU + MaxPool2d(2, 2)
      cur = a.take((size_t)B * (Hc / 2) * (Wc / 2) * 64 * 2);
      frmap_traced t(r, "stem_pool_kernel<%s> (2x2 pool)", 2.0 * B * Hc * Wc * 64 * 147, (double)B * H * W * 3 * 4 + (double)B * (Hc / 2) * (Wc / 2) * 64 * 2);
      if (x_kind == FRMAP_INPUT_U8_HWC)
        GO(frmap_stem7x7_maxpool_u8((const unsigned char*)x, m->mean, m->stdv, m->convs[0].wpk, m->convs[0].shift, cur, B, H, W, 0, m->dtype, r.st));
      else
        GO(frmap_stem7x7_maxpool2((const float*)x, m->convs[0].wpk, m->convs[0].shift, cur, B, H, W, m->dtype, r.st));
    } else {
      void* x4 = nhwc4_input(r, a, x, x_kind, H, W);
      void* full = a.take((size_t)B * Hc * Wc * 64 * 2);
      GO(frmap_conv_small_cin(x4, m->convs[0].wpk, m->convs[0].shift, full, B, H, W, 64, 7, 7, 2, 3, 1, m->dtype, r.st));
      cur = a.take((size_t)B * (Hc / 2) * (Wc / 2) * 64 * 2);
      GO(frmap_maxpool(full, cur, B, Hc, Wc, 64, 2, 2, 0, m->dtype, r.st));
    }
    h = Hc / 2; w = Wc / 2;
    for (int i = first; i < 6; ++i) cur = conv3(r, a, m->convs[i], cur, h, w, kSia[i].pool);   // face_models.py:121-146
    if (what == FRMAP_OUT_TRUNK_MAP) {
      if (!r.dry && !r.rc) r.rc = hipMemcpyAsync(out, cur, (size_t)B * h * w * 512 * 2, hipMemcpyDeviceToDevice, r.st) == hipSuccess ? 0 : -2;
      return a.off;
    }
    void* pooled = a.take((size_t)B * 36 * 512 * 2);
    GO(frmap_avgpool_adaptive(cur, pooled, B, h, w, 512, 6, 6, m->dtype, r.st));     // NHWC [B][6][6][512] == [B][18432]
    void* f1 = linear(r, a, m->lin[0], pooled, B, 1, nullptr);
    void* f2 = linear(r, a, m->lin[1], f1, B, 1, nullptr);
    void* f3 = linear(r, a, m->lin[2], f2, B, 0, nullptr);
    float* f32 = (float*)a.take((size_t)B * 256 * 4);
    GO(frmap_cast_to_f32(f3, f32, (size_t)B * 256, m->dtype, r.st));
    GO(frmap_l2_normalize_f32(f32, (float*)out, B, 256, 1e-12f, r.st));             // face_models.py:179
    if (unit_out && unit_out != out)
      if (!r.dry && !r.rc) r.rc = hipMemcpyAsync(unit_out, out, (size_t)B * 256 * 4, hipMemcpyDeviceToDevice, r.st) == hipSuccess ? 0 : -2;
    return a.off;
  }
  // ---- hybrid: tokens [B][49][512] = the trunk map (face_models.py:705-721) ----
  const int L = 49, D = 512, M = B * L;
  void* t = a.take((size_t)M * D * 2);
  void* n1 = a.take((size_t)M * D * 2);
  GO(frmap_add_pos_layernorm(trunk_map, m->pos, m->ln[0], m->ln[1], t, n1, B, L, D, 1e-5f, m->dtype, r.st));
  void* qkv = linear(r, a, m->lin[0], n1, M, 0, nullptr);
  void* att = a.take((size_t)M * D * 2);
  {
    frmap_traced tr(r, "mha_tokens_kernel<%s>", 4.0 * B * L * L * D, 2.0 * M * D * 4);
    GO(frmap_mha_tokens(qkv, att, B, L, D, 4, m->dtype, r.st));
  }
  void* t2 = linear(r, a, m->lin[1], att, M, 0, t);                                  // x + attn_out
  void* n2 = a.take((size_t)M * D * 2);
  GO(frmap_add_pos_layernorm(t2, nullptr, m->ln[2], m->ln[3], nullptr, n2, B, L, D, 1e-5f, m->dtype, r.st));
  void* hdn = linear(r, a, m->lin[2], n2, M, 2, nullptr);                            // Linear -> GELU
  void* t3 = linear(r, a, m->lin[3], hdn, M, 0, t2);                                 // x + ff_out
  float* emb = what == FRMAP_OUT_EMBEDDING ? (float*)out : (float*)a.take((size_t)B * D * 4);
  GO(frmap_mean_layernorm(t3, m->ln[4], m->ln[5], emb, B, L, D, 1e-5f, m->dtype, r.st));
  if (what == FRMAP_OUT_LOGITS) GO(frmap_linear_f32(emb, m->fc_w, nullptr, m->fc_b, (float*)out, B, D, m->num_classes, 0, r.st));
  if (unit_out) GO(frmap_l2_normalize_f32(emb, unit_out, B, D, 1e-12f, r.st));
  return a.off;
}
