// hipGraph helpers: capture one sampling step (all kernels enqueued through this C ABI on a
// stream) and replay it once per step.  Launch-bound inner loops (~150 kernels per UNet
// forward) collapse to one graph launch; per-step scalars come from device memory
// (az_step_begin), so the same executable graph serves every step.
#include "common.h"

struct AzGraph {
  hipGraph_t graph;
  hipGraphExec_t exec;
};

extern "C" {

int az_graph_begin(az_stream_t stream) {
  hipError_t e = hipStreamBeginCapture(az_s(stream), hipStreamCaptureModeThreadLocal);
  return e == hipSuccess ? AZ_OK : (int)e;
}

int az_graph_end(az_stream_t stream, AzGraph** out) {
  AZ_REQUIRE(out, AZ_E_NULL);
  *out = nullptr;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(az_s(stream), &g);
  if (e != hipSuccess) return (int)e;
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    return (int)e;
  }
  AzGraph* r = new AzGraph{g, x};
  *out = r;
  return AZ_OK;
}

int az_graph_launch(AzGraph* g, az_stream_t stream) {
  AZ_REQUIRE(g, AZ_E_NULL);
  hipError_t e = hipGraphLaunch(g->exec, az_s(stream));
  return e == hipSuccess ? AZ_OK : (int)e;
}

int az_graph_destroy(AzGraph* g) {
  if (!g) return AZ_OK;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
  return AZ_OK;
}

int az_graph_num_nodes(AzGraph* g, int64_t* n) {
  AZ_REQUIRE(g && n, AZ_E_NULL);
  size_t k = 0;
  hipError_t e = hipGraphGetNodes(g->graph, nullptr, &k);
  *n = (int64_t)k;
  return e == hipSuccess ? AZ_OK : (int)e;
}

}  // extern "C"
