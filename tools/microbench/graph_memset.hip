// Do memset nodes of a captured HIP graph keep their place in the stream order on replay?
// Records N x { clear a counter (hipMemsetAsync or a kernel), add into it from many workgroups, copy it out } on one
// stream, replays the graph many times and counts the iterations whose copied value is not the expected sum.
// Also writes the graph as DOT (node kinds and dependency edges) for both variants.
//   hipcc --offload-arch=gfx950 -O2 -o build/graph_memset tools/microbench/graph_memset.hip && build/graph_memset [dir]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
  } while (0)

__global__ void add_kernel(float* acc, int n) {
  // a little work first so that the launch is not instantaneous
  float v = 1.0f;
  for (int i = 0; i < 64; ++i) v = v * 1.0000001f + 0.0f;
  if (threadIdx.x == 0) atomicAdd(&acc[blockIdx.x % n], v > 0.f ? 1.0f : 0.0f);
}
__global__ void zero_kernel(float* acc, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) acc[i] = 0.0f;
}
__global__ void copy_kernel(const float* acc, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = acc[i];
}

static long run(bool use_memset, int pairs, int replays, const std::string& dot) {
  const int n = 128, blocks = 4096;
  float *acc, *out;
  CHECK(hipMalloc(&acc, sizeof(float) * n));
  CHECK(hipMalloc(&out, sizeof(float) * n * pairs));
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  hipGraph_t graph;
  hipGraphExec_t exec;
  CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int p = 0; p < pairs; ++p) {
    if (use_memset) CHECK(hipMemsetAsync(acc, 0, sizeof(float) * n, s));
    else hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(128), 0, s, acc, n);
    hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(64), 0, s, acc, n);
    hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(128), 0, s, acc, out + p * n, n);
  }
  CHECK(hipStreamEndCapture(s, &graph));
  if (!dot.empty()) CHECK(hipGraphDebugDotPrint(graph, dot.c_str(), 0));
  size_t nodes = 0;
  CHECK(hipGraphGetNodes(graph, nullptr, &nodes));
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  std::vector<float> host(n * pairs);
  long bad = 0;
  for (int r = 0; r < replays; ++r) {
    CHECK(hipGraphLaunch(exec, s));
    if (r % 8 == 7 || r == replays - 1) {   // mostly back to back, a host read now and then
      CHECK(hipStreamSynchronize(s));
      CHECK(hipMemcpy(host.data(), out, sizeof(float) * n * pairs, hipMemcpyDeviceToHost));
      for (int i = 0; i < n * pairs; ++i) bad += host[i] != (float)(blocks / n);
    }
  }
  printf("%s: %zu nodes, %d replays, wrong values read back: %ld\n", use_memset ? "memset nodes" : "kernel nodes", nodes,
         replays, bad);
  CHECK(hipGraphExecDestroy(exec));
  CHECK(hipGraphDestroy(graph));
  CHECK(hipStreamDestroy(s));
  CHECK(hipFree(acc));
  CHECK(hipFree(out));
  return bad;
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "";
  run(true, 50, 400, dir.empty() ? "" : dir + "/graph_memset_nodes.dot");
  run(false, 50, 400, dir.empty() ? "" : dir + "/graph_kernel_nodes.dot");
  return 0;
}
