/* A plain C host of the model-level C ABI (include/frmap_hip.h, "Model handles"): no Python, no PyTorch - the HIP runtime for
 * device memory, libfrmap_hip.so for everything else.  It does what `model(images)` + `compare_faces` do in the reference
 * (/root/reference/src/testing.py:255-273, src/app.py:44,50-64): load a checkpoint's state_dict entry by entry under the
 * reference's key names, embed a batch of fp32 NCHW faces, match every face against a gallery.
 *
 *   cabi_host <model_type> <num_classes> <dtype: bf16|f16> <weights.bin> <input.bin> <gallery.bin|-> <out.bin>
 *
 * File formats (little endian; written by tests/test_model_cabi_gpu.py):
 *   weights.bin : int32 n; n x { int32 key_len; key bytes; int64 numel; numel x float32 }
 *   input.bin   : int32 B, H, W; B*3*H*W x float32 (NCHW)
 *   gallery.bin : int32 G, D; float32 thresh; int32 normalize; G*D x float32        ("-": no match step)
 *   out.bin     : int32 B, D; B*D x float32 embeddings; [B x int32 id-or-unknown; B x float32 distance]
 *
 * Build:  gcc examples/cabi_host.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o examples/cabi_host \
 *             -Lfacerecognition-multiarchitecture-pipeline_amd -l:libfrmap_hip.so -L/opt/rocm/lib -lamdhip64 \
 *             -Wl,-rpath,'$ORIGIN/../facerecognition-multiarchitecture-pipeline_amd' -Wl,-rpath,/opt/rocm/lib
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "frmap_hip.h"

#define DIE(...)                                   \
  do {                                             \
    fprintf(stderr, "cabi_host: " __VA_ARGS__);    \
    fprintf(stderr, "\n");                         \
    exit(1);                                       \
  } while (0)
#define HIPOK(e)                                                              \
  do {                                                                        \
    hipError_t e__ = (e);                                                     \
    if (e__ != hipSuccess) DIE("%s: %s", #e, hipGetErrorString(e__));         \
  } while (0)
#define FRMAPOK(e)                                                            \
  do {                                                                        \
    int rc__ = (e);                                                           \
    if (rc__ < 0) DIE("%s: rc %d: %s", #e, rc__, frmap_last_error());         \
  } while (0)

static void rd(FILE* f, void* p, size_t n) {
  if (fread(p, 1, n, f) != n) DIE("short read");
}

int main(int argc, char** argv) {
  if (argc != 8) DIE("usage: cabi_host <model_type> <num_classes> <bf16|f16> <weights.bin> <input.bin> <gallery.bin|-> <out.bin>");
  const char* type = argv[1];
  const int classes = atoi(argv[2]);
  const int dtype = strcmp(argv[3], "bf16") == 0 ? FRMAP_BF16 : FRMAP_F16;
  HIPOK(hipSetDevice(0));

  /* ---- the checkpoint, entry by entry, under the reference's state_dict keys ---- */
  frmap_model* m = NULL;
  FRMAPOK(frmap_model_create(&m, type, classes, dtype));
  FILE* f = fopen(argv[4], "rb");
  if (!f) DIE("cannot open %s", argv[4]);
  int32_t n = 0, used = 0;
  rd(f, &n, 4);
  for (int i = 0; i < n; ++i) {
    int32_t klen;
    int64_t numel;
    char key[256];
    rd(f, &klen, 4);
    if (klen <= 0 || klen >= (int)sizeof(key)) DIE("bad key length");
    rd(f, key, (size_t)klen);
    key[klen] = 0;
    rd(f, &numel, 8);
    float* buf = (float*)malloc((size_t)numel * 4);
    rd(f, buf, (size_t)numel * 4);
    const int rc = frmap_model_load_tensor(m, key, buf, (size_t)numel, 0 /* host memory */);
    if (rc < 0) DIE("load_tensor(%s): %s", key, frmap_last_error());
    used += rc == 0;
    free(buf);
  }
  fclose(f);
  FRMAPOK(frmap_model_finalize(m, NULL));

  /* ---- a batch of faces: fp32 NCHW, as the reference feeds its model ---- */
  f = fopen(argv[5], "rb");
  if (!f) DIE("cannot open %s", argv[5]);
  int32_t dims[3];
  rd(f, dims, 12);
  const int B = dims[0], H = dims[1], W = dims[2];
  const size_t xbytes = (size_t)B * 3 * H * W * 4;
  float* xh = (float*)malloc(xbytes);
  rd(f, xh, xbytes);
  fclose(f);
  float* xd;
  HIPOK(hipMalloc((void**)&xd, xbytes));
  HIPOK(hipMemcpy(xd, xh, xbytes, hipMemcpyHostToDevice));
  free(xh);

  const int D = frmap_model_embedding_dim(m);
  float *embd, *embh = (float*)malloc((size_t)B * D * 4);
  void* ws;
  HIPOK(hipMalloc((void**)&embd, (size_t)B * D * 4));
  HIPOK(hipMalloc(&ws, frmap_model_workspace_bytes(m, B, H, W)));
  FRMAPOK(frmap_model_forward(m, xd, FRMAP_INPUT_F32_NCHW, B, H, W, FRMAP_OUT_EMBEDDING, embd, ws, NULL));   /* model.get_embedding(x) */
  HIPOK(hipDeviceSynchronize());
  HIPOK(hipMemcpy(embh, embd, (size_t)B * D * 4, hipMemcpyDeviceToHost));

  /* ---- compare_faces for every face, in one call ---- */
  int32_t* idsh = NULL;
  float* disth = NULL;
  if (strcmp(argv[6], "-") != 0) {
    f = fopen(argv[6], "rb");
    if (!f) DIE("cannot open %s", argv[6]);
    int32_t gd[2], normalize;
    float thresh;
    rd(f, gd, 8); rd(f, &thresh, 4); rd(f, &normalize, 4);
    const int G = gd[0];
    if (gd[1] != D) DIE("gallery rows have %d values, the model embeds into %d", gd[1], D);
    float* gh = (float*)malloc((size_t)G * D * 4);
    rd(f, gh, (size_t)G * D * 4);
    fclose(f);
    float *gdv, *stat = NULL, *distd;
    void *packed = NULL, *ws2;
    int32_t *idxd, *idsd;
    HIPOK(hipMalloc((void**)&gdv, (size_t)G * D * 4));
    HIPOK(hipMemcpy(gdv, gh, (size_t)G * D * 4, hipMemcpyHostToDevice));
    free(gh);
    if (G >= 512 && D % 32 == 0) {   /* large galleries: prepared once for the MFMA match path */
      HIPOK(hipMalloc(&packed, frmap_match_gallery_pack_bytes(G, D)));
      HIPOK(hipMalloc((void**)&stat, (size_t)G * 16));
      FRMAPOK(frmap_match_pack_gallery(gdv, packed, stat, G, D, NULL));
    }
    HIPOK(hipMalloc(&ws2, frmap_model_match_workspace_bytes(m, B, H, W, G)));
    HIPOK(hipMalloc((void**)&idxd, (size_t)B * 4));
    HIPOK(hipMalloc((void**)&idsd, (size_t)B * 4));
    HIPOK(hipMalloc((void**)&distd, (size_t)B * 4));
    FRMAPOK(frmap_model_embed_and_match(m, xd, FRMAP_INPUT_F32_NCHW, B, H, W, gdv, packed, stat, G, thresh, normalize, idxd, distd, idsd,
                                        NULL, NULL, ws2, NULL));
    HIPOK(hipDeviceSynchronize());
    idsh = (int32_t*)malloc((size_t)B * 4);
    disth = (float*)malloc((size_t)B * 4);
    HIPOK(hipMemcpy(idsh, idsd, (size_t)B * 4, hipMemcpyDeviceToHost));
    HIPOK(hipMemcpy(disth, distd, (size_t)B * 4, hipMemcpyDeviceToHost));
  }

  f = fopen(argv[7], "wb");
  if (!f) DIE("cannot write %s", argv[7]);
  int32_t od[2] = {B, D};
  fwrite(od, 4, 2, f);
  fwrite(embh, 4, (size_t)B * D, f);
  if (idsh) { fwrite(idsh, 4, (size_t)B, f); fwrite(disth, 4, (size_t)B, f); }
  fclose(f);
  frmap_model_destroy(m);
  printf("cabi_host: %s %s B=%d D=%d tensors used %d%s\n", type, argv[3], B, D, used, idsh ? " + match" : "");
  return 0;
}
