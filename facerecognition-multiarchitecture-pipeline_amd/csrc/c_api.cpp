// Error plumbing, version of the C ABI (include/frmap_hip.h) and per-device kernel attributes.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <set>
#include <utility>

#include "../../include/frmap_hip.h"

static thread_local char g_err[512] = "";

void frmap_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: a process that launches on several
// GPUs (the app-style use of SURVEY.md §8b: a model on any device, called from any thread) must raise it once per
// (kernel, device), not once per process.  Returns 0, or -2 with the error text set.
int frmap_big_lds(const void* kern, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) {
    frmap_set_error("hipGetDevice: %s", hipGetErrorString(e));
    return -2;
  }
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(kern, dev);
  if (done.count(key)) return 0;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    frmap_set_error("hipFuncSetAttribute(device %d): %s", dev, hipGetErrorString(e));
    return -2;
  }
  done.insert(key);
  return 0;
}

// Batch-invariant planning (frmap_set_batch_invariant): the conv planners then choose a layer's kernel and tile layout from the
// per-image geometry alone - never from the tile count, never split-K - so a face's result does not depend on the batch it
// arrives in (a 1-face call and a 1024-face call give that face the same bits).  -1 = unset: environment FRMAP_BATCH_INVARIANT.
static int g_invariant = -1;
int frmap_batch_invariant() {
  if (g_invariant >= 0) return g_invariant;
  static int env = -1;
  if (env < 0) { const char* e = getenv("FRMAP_BATCH_INVARIANT"); env = (e && atoi(e) != 0) ? 1 : 0; }
  return env;
}
extern "C" int frmap_set_batch_invariant(int on) { g_invariant = on < 0 ? -1 : (on != 0); return 0; }

extern "C" int frmap_abi_version(void) { return 9; }
extern "C" const char* frmap_last_error(void) { return g_err; }
