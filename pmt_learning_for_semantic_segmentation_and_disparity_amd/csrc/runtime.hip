// Host-side runtime glue of libsdhip.so: ABI version and the thread-local
// error channel.  No global mutable state besides the per-thread message, so
// entry points are re-entrant from autograd's backward threads.
#include "sdhip_common.h"
#include <vector>

#define SDHIP_ABI_VERSION 2

static thread_local char g_err[512] = "";

void sdhip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int sdhip_abi_version(void) { return SDHIP_ABI_VERSION; }
extern "C" const char* sdhip_last_error(void) { return g_err; }

// ---- diagnostic / tuning switches: read ONCE (library load), never per launch --------------------------------------
// A stray environment variable must not silently change numerics mid-run: the table is filled at load, a set DIAG switch
// is announced on stderr, and only an explicit sdhip_diag_reload() (tests / tools doing A/B runs in one process) re-reads it.
static SdhipDiag g_diag;
static bool g_diag_init = false;

static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static bool env_set(const char* name) {
  const char* v = getenv(name);
  if (v && *v) { fprintf(stderr, "[libsdhip] diagnostic switch %s=%s is set: kernels / launch heuristics differ from production\n", name, v); return true; }
  return false;
}
static void diag_fill() {
  g_diag.conv_generic = env_set("SDHIP_CONV_GENERIC");
  g_diag.conv_big = env_set("SDHIP_CONV_BIG");
  g_diag.conv_no_thin = env_set("SDHIP_CONV_NO_THIN");
  g_diag.conv_no_gemm = env_set("SDHIP_CONV_NO_GEMM");
  g_diag.conv_no_band = env_set("SDHIP_CONV_NO_BAND");
  g_diag.conv_no_band3 = env_set("SDHIP_CONV_NO_BAND3");
  g_diag.wgrad_generic = env_set("SDHIP_WGRAD_GENERIC");
  g_diag.wgrad_no_pack = env_set("SDHIP_WGRAD_NO_PACK");
  g_diag.wgrad_force_pack = env_set("SDHIP_WGRAD_FORCE_PACK");
  g_diag.wgrad_no_half = env_set("SDHIP_WGRAD_NO_HALF");
  g_diag.thin_wgrad_reg = env_set("SDHIP_THIN_WGRAD_REG");
  g_diag.corr_no_tiled = env_set("SDHIP_CORR_NO_TILED");
  g_diag.wgrad_no_half32 = env_set("SDHIP_WGRAD_NO_HALF32");
  g_diag.wgrad_split2d = env_set("SDHIP_WGRAD_SPLIT2D");
  g_diag.tune_s2_small = env_int("SDHIP_TUNE_S2_SMALL", 1);
  g_diag.tune_big = env_int("SDHIP_TUNE_BIG", 512);
  g_diag.tune_split = env_int("SDHIP_TUNE_SPLIT", 1024);
  g_diag.tune_thin_blocks = env_int("SDHIP_TUNE_THIN_BLOCKS", 1024);
  g_diag.tune_fused_blocks = env_int("SDHIP_TUNE_FUSED_BLOCKS", 768);
  g_diag.tune_gemm_dbg = env_int("SDHIP_TUNE_GEMM_DBG", 0);
  g_diag.tune_band_dbg = env_int("SDHIP_TUNE_BAND_DBG", 0);
  g_diag.tune_atomic_tbs = getenv("SDHIP_TUNE_ATOMIC_TBS") ? atof(getenv("SDHIP_TUNE_ATOMIC_TBS")) : 1.3;
  g_diag_init = true;
}
const SdhipDiag& sdhip_diag() {
  if (!g_diag_init) diag_fill();
  return g_diag;
}
__attribute__((constructor)) static void diag_at_load() { diag_fill(); }
extern "C" void sdhip_diag_reload(void) { diag_fill(); }

// ---- recovery from a stream capture that failed half way (train.TrainStep) -------------------------------------------
// An illegal call during hipStreamBeginCapture..EndCapture invalidates the capture, but the stream stays in capture mode
// until hipStreamEndCapture has been called on it; a caller that raised out of its capture block (PyTorch's graph context
// does, and then skips its own clean-up) leaves the process unable to launch anything.  This ends the capture, drops the
// partial graph and clears the sticky error.  Returns 1 if a capture was still open, 0 if not, < 0 on failure.
extern "C" int sdhip_abort_capture(void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  hipError_t e = hipStreamIsCapturing(s, &st);
  (void)hipGetLastError();
  int was_open = 0;
  if (e != hipSuccess || st != hipStreamCaptureStatusNone) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(s, &g);        // returns hipErrorStreamCaptureInvalidated for a broken capture: expected
    (void)hipGetLastError();
    if (g) (void)hipGraphDestroy(g);
    was_open = 1;
  }
  st = hipStreamCaptureStatusNone;
  e = hipStreamIsCapturing(s, &st);
  (void)hipGetLastError();
  // ROCm 7.2 leaves an invalidated capture stream in the Invalidated state even after EndCapture; that stream is dropped by
  // the caller and every other stream works again.  Only a stream that is still ACTIVELY capturing is an error.
  if (e == hipSuccess && st == hipStreamCaptureStatusActive) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "abort_capture: the stream is still capturing");
  return was_open;
}

// ---- node inventory of a captured step (tests: "kernel nodes only", node-count budgets) --------------------------------
// graph: a hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()).  counts[0..3] = kernel, memset, memcpy,
// every other node type.  Returns the total number of nodes, < 0 on failure.
extern "C" int sdhip_graph_node_counts(void* graph, int* counts) {
  SDHIP_CHECK_ARG(graph && counts, "graph_node_counts: null pointer");
  size_t n = 0;
  if (hipGraphGetNodes((hipGraph_t)graph, nullptr, &n) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "graph_node_counts: hipGraphGetNodes failed");
  std::vector<hipGraphNode_t> nodes(n);
  if (n && hipGraphGetNodes((hipGraph_t)graph, nodes.data(), &n) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "graph_node_counts: hipGraphGetNodes failed");
  counts[0] = counts[1] = counts[2] = counts[3] = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) SDHIP_FAIL(SDHIP_ERR_LAUNCH, "graph_node_counts: hipGraphNodeGetType failed");
    if (t == hipGraphNodeTypeKernel) ++counts[0];
    else if (t == hipGraphNodeTypeMemset) ++counts[1];
    else if (t == hipGraphNodeTypeMemcpy) ++counts[2];
    else ++counts[3];
  }
  return (int)n;
}
