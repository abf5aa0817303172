// Private to model_api.cpp / model_families.cpp: the state behind a `frmap_model` handle (include/frmap_hip.h, "Model handles").
#pragma once
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "frmap_common.h"

enum { KIND_CNN = 0, KIND_ARCFACE = 1, KIND_TRUNK = 2, KIND_BASELINE = 3, KIND_SIAMESE = 4, KIND_HYBRID = 5 };

struct PackedConv {
  void* wpk = nullptr;
  float* shift = nullptr;
  int cout = 0, cin = 0, k = 0, stride = 1, pad = 0;
  std::vector<float> shift_host;
};

struct Block {
  PackedConv c1, c2, ds;
  bool has_ds = false;
  float* fshift = nullptr;  // c2.shift + ds.shift: shift of the fused conv2 + projection-shortcut launch
};

struct TraceRec {
  char kernel[64];
  double flop, bytes;
  hipEvent_t e0, e1;
};

struct frmap_model {
  int kind = KIND_CNN, dtype = FRMAP_BF16, num_classes = 0, device = 0;
  bool finalized = false;
  std::map<std::string, std::vector<float>> raw;  // canonical key -> fp32 host copy (dropped by finalize)
  PackedConv stem;
  Block blocks[8];
  float* fc_w = nullptr;      // cnn: resnet.fc.1 [num_classes][512] / bias
  float* fc_b = nullptr;
  float* emb_wt = nullptr;    // arcface: embedding.weight transposed [512][512], folded BatchNorm1d
  float* bn_scale = nullptr;
  float* bn_shift = nullptr;
  float* cls_wn = nullptr;    // arcface: val_classifier.weight with unit rows (face_models.py:576), bias
  float* cls_b = nullptr;
  float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};  // src/testing.py:102-103
  // 'baseline' (face_models.py:16-60), 'siamese' (:104-192), 'hybrid' token block (:618-721): model_families.cpp
  PackedConv convs[6];        // baseline: conv1..3; siamese: conv.0, .4, .7, .11, .14, .18
  PackedConv lin[4];          // siamese: fc.1 (+bn), fc.5 (+bn), fc.8; hybrid: in_proj, out_proj, ff.0, ff.3  (1x1-packed, shift = folded bias)
  float* head_wt = nullptr;   // baseline: fc1.weight transposed [128][512]
  float* head_b = nullptr;    // baseline: fc1.bias
  float* pos = nullptr;       // hybrid: pos_encoding [49][512]
  float* ln[6] = {};          // hybrid: norm1 (w, b), norm2 (w, b), norm (w, b)
  std::vector<void*> allocs;
  std::mutex trace_mu;
  bool trace = false;
  std::vector<TraceRec> recs;
};


// helpers shared by the two files (model_api.cpp)
void* frmap_model_dev_alloc(frmap_model* m, size_t bytes);
float* frmap_model_dev_upload(frmap_model* m, const std::vector<float>& v);
void frmap_model_bn_fold(const frmap_model* m, const std::string& p, int c, std::vector<float>* scale, std::vector<float>* shift);
// fold (optional) BatchNorm `bnkey` and (optional) bias `bkey` into an OIHW conv / [N][K] linear weight and pack it
int frmap_model_pack(frmap_model* m, const std::string& wkey, const std::string& bkey, const std::string& bnkey, int cout, int cin,
                     int k, int stride, int pad, PackedConv* pc, hipStream_t st, const std::vector<float>* w_override = nullptr);

// per-launch trace (frmap_model_trace): brackets one launch with events
struct frmap_run {
  frmap_model* m;
  hipStream_t st;
  int B;
  const char* dt;
  int rc = 0;
  bool dry = false;   // size the workspace only: no launches
};
struct frmap_traced {
  frmap_run& r;
  TraceRec rec;
  bool on;
  frmap_traced(frmap_run& r_, const char* fmt, double flop, double bytes);
  ~frmap_traced();
};

// model_families.cpp
void frmap_family_expected(const frmap_model* m, std::map<std::string, size_t>* want);
bool frmap_family_key(int kind, const std::string& key, std::string* out);
int frmap_family_finalize(frmap_model* m, hipStream_t st);
// forward of 'baseline' / 'siamese' / the token block of 'hybrid'; returns bytes of workspace used (dry run: needed)
size_t frmap_family_forward(frmap_run& r, const void* x, int x_kind, int H, int W, int what, void* out, float* unit_out, char* ws,
                            const void* trunk_map);
