// Error reporting, hipGraph helpers and per-kernel-class event timing for libseva_hip.so.
#include "seva_common.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

namespace {

thread_local char g_err[512] = "";

struct ProfRecord {
  int cls;
  double work, bytes;
  hipEvent_t e0, e1;
};
std::mutex g_prof_mu;
std::vector<ProfRecord> g_prof;
bool g_prof_on = false;

}  // namespace

namespace {
struct KnobEntry {
  const char* name;  // seva_set_knob() name; environment variable = "SEVA_" + upper-case name
  const char* env;
  int SevaKnobs::*field;
};
const KnobEntry kKnobs[] = {
    {"gemm_chunks", "SEVA_GEMM_CHUNKS", &SevaKnobs::gemm_chunks}, {"gemm_dbg", "SEVA_GEMM_DBG", &SevaKnobs::gemm_dbg},
    {"gemm_stagger", "SEVA_GEMM_STAGGER", &SevaKnobs::gemm_stagger},
    {"gemm_bm", "SEVA_GEMM_BM", &SevaKnobs::gemm_bm}, {"gemm_bn", "SEVA_GEMM_BN", &SevaKnobs::gemm_bn},
    {"gemm_astat", "SEVA_GEMM_ASTAT", &SevaKnobs::gemm_astat}, {"attn_dbg", "SEVA_ATTN_DBG", &SevaKnobs::attn_dbg},
    {"attn_no_tr", "SEVA_ATTN_NO_TR", &SevaKnobs::attn_no_tr}, {"attn_two", "SEVA_ATTN_TWO", &SevaKnobs::attn_two},
    {"attn_split", "SEVA_ATTN_SPLIT", &SevaKnobs::attn_split},
    {"gn_min_iter", "SEVA_GN_MIN_ITER", &SevaKnobs::gn_min_iter},
    {"conv_win", "SEVA_CONV_WIN", &SevaKnobs::conv_win},
};
SevaKnobs knobs_from_env() {
  SevaKnobs k;
  for (const auto& e : kKnobs) {
    const char* v = getenv(e.env);
    k.*(e.field) = (v && v[0]) ? atoi(v) : -1;
  }
  return k;
}
}  // namespace

SevaKnobs g_seva_knobs = knobs_from_env();  // once, at library load

void seva_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int seva_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    seva_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return SEVA_ERR_LAUNCH;
  }
  return SEVA_OK;
}

SevaProfScope::SevaProfScope(int cls_, double work_, hipStream_t stream_, double bytes_)
    : cls(cls_), work(work_), bytes(bytes_ < 0 ? work_ : bytes_), stream(stream_), e0(nullptr), e1(nullptr), on(g_prof_on) {
  if (!on) return;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
    on = false;
    return;
  }
  (void)hipEventRecord(e0, stream);
}

SevaProfScope::~SevaProfScope() {
  if (!on) return;
  (void)hipEventRecord(e1, stream);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back({cls, work, bytes, e0, e1});
}

extern "C" {

const char* seva_last_error(void) { return g_err; }
int seva_abi_version(void) { return 8; }
const char* seva_target_arch(void) { return "gfx950"; }

int seva_set_knob(const char* name, int value) {
  SEVA_REQUIRE(name != nullptr, "set_knob: null name");
  for (const auto& e : kKnobs)
    if (strcmp(e.name, name) == 0) {
      g_seva_knobs.*(e.field) = value;
      return SEVA_OK;
    }
  seva_set_error("set_knob: unknown knob '%s'", name);
  return SEVA_ERR_ARG;
}

int seva_get_knob(const char* name, int* value) {
  SEVA_REQUIRE(name != nullptr && value != nullptr, "get_knob: null argument");
  for (const auto& e : kKnobs)
    if (strcmp(e.name, name) == 0) {
      *value = g_seva_knobs.*(e.field);
      return SEVA_OK;
    }
  seva_set_error("get_knob: unknown knob '%s'", name);
  return SEVA_ERR_ARG;
}

int seva_graph_begin(seva_stream_t stream) {
  hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {
    seva_set_error("graph_begin: %s", hipGetErrorString(e));
    return SEVA_ERR_LAUNCH;
  }
  return SEVA_OK;
}

int seva_graph_end(seva_stream_t stream, void** graph_exec_out) {
  SEVA_REQUIRE(graph_exec_out != nullptr, "graph_end: null out");
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
  if (e != hipSuccess || graph == nullptr) {
    seva_set_error("graph_end: %s", hipGetErrorString(e));
    return SEVA_ERR_LAUNCH;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    seva_set_error("graph_instantiate: %s", hipGetErrorString(e));
    return SEVA_ERR_LAUNCH;
  }
  *graph_exec_out = (void*)exec;
  return SEVA_OK;
}

int seva_graph_launch(void* graph_exec, seva_stream_t stream) {
  SEVA_REQUIRE(graph_exec != nullptr, "graph_launch: null graph");
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
  if (e != hipSuccess) {
    seva_set_error("graph_launch: %s", hipGetErrorString(e));
    return SEVA_ERR_LAUNCH;
  }
  return SEVA_OK;
}

int seva_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return SEVA_OK;
}

int seva_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return SEVA_OK;
}

int seva_prof_collect(double* ms, int64_t* launches, double* work, double* bytes) {
  SEVA_REQUIRE(ms && launches && work && bytes, "prof_collect: null out");
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    seva_set_error("prof_collect: %s", hipGetErrorString(e));
    return SEVA_ERR_LAUNCH;
  }
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < SEVA_PROF_CLASSES; ++i) {
    ms[i] = 0.0;
    launches[i] = 0;
    work[i] = 0.0;
    bytes[i] = 0.0;
  }
  for (auto& r : g_prof) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess && r.cls >= 0 && r.cls < SEVA_PROF_CLASSES) {
      ms[r.cls] += t;
      launches[r.cls] += 1;
      work[r.cls] += r.work;
      bytes[r.cls] += r.bytes;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_prof.clear();
  return SEVA_OK;
}

}  // extern "C"
