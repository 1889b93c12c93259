// Probe: may a captured kernel node's arguments be replaced per launch (hipGraphExecKernelNodeSetParams) while earlier launches of
// the same executable graph are still in flight?  And what does the eager-launch -> graph-launch transition cost against having
// the same kernel as the graph's first node?
//
//   1. graph = spin(100 us) -> mark(out, idx, val) -> spin(20 us); N launches back to back, the mark node's (idx, val) replaced
//      before each; the host runs far ahead of the GPU.  Every out[i] must be i: an update that reached an in-flight launch shows
//      as a hole.
//   2. the same with the node's FUNCTION and grid replaced (mark -> mark_other).
//   3. timing: [eager short kernel + graph of 13 dependent short kernels] against [graph of 14, first node's arguments replaced
//      per launch], 2000 iterations each; the host cost of the update call.
//   4. what one graph LAUNCH costs on top of its nodes: K-node graphs back to back, K = 7 .. 56, linear fit.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/graph_setparams tools/probe/graph_setparams.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                          \
  do {                                                                                                 \
    hipError_t e_ = (x);                                                                               \
    if (e_ != hipSuccess) {                                                                            \
      printf("%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));                       \
      exit(2);                                                                                         \
    }                                                                                                  \
  } while (0)

struct MarkArgs {          // (by-value struct of the hand-over launch's size class)
  int* out;
  int idx, val;
  long pad[60];
};

__global__ void spin_kernel(long ticks) {
  long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}
__global__ void mark_kernel(MarkArgs a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) a.out[a.idx] = a.val;
}
__global__ void mark_other_kernel(MarkArgs a) {
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) a.out[a.idx] = a.val + 1000000 * (int)gridDim.x;
}
__global__ void step_kernel(float* p, int n) {          // a short dependent kernel: 256 workgroups touching 1 MB
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
__global__ void first_kernel(MarkArgs a, float* p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += (float)a.val;
  if (i == 0) a.out[a.idx] = a.val;
}

static hipGraphNode_t find_node(hipGraph_t g, void* func) {
  size_t n = 0;
  CK(hipGraphGetNodes(g, nullptr, &n));
  std::vector<hipGraphNode_t> nodes(n);
  CK(hipGraphGetNodes(g, nodes.data(), &n));
  for (auto nd : nodes) {
    hipGraphNodeType t;
    CK(hipGraphNodeGetType(nd, &t));
    if (t != hipGraphNodeTypeKernel) continue;
    hipKernelNodeParams p;
    CK(hipGraphKernelNodeGetParams(nd, &p));
    if (p.func == func) return nd;
  }
  printf("node not found among %zu\n", n);
  exit(2);
}

int main(int argc, char** argv) {
  int N = argc > 1 ? atoi(argv[1]) : 200;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int* out;
  CK(hipMalloc(&out, sizeof(int) * (N + 8)));
  CK(hipMemset(out, 0xff, sizeof(int) * (N + 8)));
  std::vector<int> host(N + 8);

  // ---- 1 / 2: in-flight safety ----
  {
    hipGraph_t g;
    hipGraphExec_t ex;
    MarkArgs a{out, N, -5};
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    spin_kernel<<<1, 64, 0, st>>>(10000);          // 100 us at 100 MHz
    mark_kernel<<<4, 64, 0, st>>>(a);
    spin_kernel<<<1, 64, 0, st>>>(2000);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipGraphNode_t nd = find_node(g, (void*)mark_kernel);
    auto t0 = std::chrono::steady_clock::now();
    double upd = 0;
    for (int i = 0; i < N; ++i) {
      MarkArgs ai{out, i, i};
      void* kp[1] = {&ai};
      hipKernelNodeParams p{};
      p.func = (void*)mark_kernel;
      p.gridDim = dim3(4);
      p.blockDim = dim3(64);
      p.kernelParams = kp;
      auto u0 = std::chrono::steady_clock::now();
      hipError_t e = hipGraphExecKernelNodeSetParams(ex, nd, &p);
      upd += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - u0).count();
      if (e != hipSuccess) { printf("SetParams: %s\n", hipGetErrorString(e)); return 2; }
      CK(hipGraphLaunch(ex, st));
    }
    double host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipStreamSynchronize(st));
    double all_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipMemcpy(host.data(), out, sizeof(int) * (N + 8), hipMemcpyDeviceToHost));
    int holes = 0;
    for (int i = 0; i < N; ++i) holes += host[i] != i;
    printf("1. %d launches, arguments replaced before each: host loop %.0f us (GPU done after %.0f us), update call %.2f us mean, holes %d, example slot %d\n",
           N, host_us, all_us, upd / N, holes, host[N]);

    // 2. function + grid replaced
    CK(hipMemset(out, 0xff, sizeof(int) * (N + 8)));
    int bad_rc = 0;
    for (int i = 0; i < N; ++i) {
      MarkArgs ai{out, i, i};
      void* kp[1] = {&ai};
      hipKernelNodeParams p{};
      bool other = i & 1;
      p.func = other ? (void*)mark_other_kernel : (void*)mark_kernel;
      p.gridDim = dim3(other ? 7 : 4);
      p.blockDim = dim3(other ? 128 : 64);
      p.kernelParams = kp;
      hipError_t e = hipGraphExecKernelNodeSetParams(ex, nd, &p);
      if (e != hipSuccess) { if (!bad_rc) printf("2. SetParams with another function: %s\n", hipGetErrorString(e)); bad_rc++; (void)hipGetLastError(); }
      CK(hipGraphLaunch(ex, st));
    }
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(host.data(), out, sizeof(int) * (N + 8), hipMemcpyDeviceToHost));
    holes = 0;
    for (int i = 0; i < N; ++i) holes += host[i] != ((i & 1) ? i + 7000000 : i);
    printf("2. function and grid replaced on odd launches: refused %d times, wrong slots %d (slot 1 = %d, slot 2 = %d)\n", bad_rc, holes, host[1], host[2]);
    CK(hipGraphExecDestroy(ex));
    CK(hipGraphDestroy(g));
  }

  // ---- 3: transition cost ----
  {
    const int n = 256 * 1024, K = 13, IT = 2000;
    float* p;
    CK(hipMalloc(&p, sizeof(float) * n));
    CK(hipMemset(p, 0, sizeof(float) * n));
    hipGraph_t g13, g14;
    hipGraphExec_t e13, e14;
    MarkArgs a{out, N, 0};
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int k = 0; k < K; ++k) step_kernel<<<n / 256, 256, 0, st>>>(p, n);
    CK(hipStreamEndCapture(st, &g13));
    CK(hipGraphInstantiate(&e13, g13, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    first_kernel<<<n / 256, 256, 0, st>>>(a, p, n);
    for (int k = 0; k < K; ++k) step_kernel<<<n / 256, 256, 0, st>>>(p, n);
    CK(hipStreamEndCapture(st, &g14));
    CK(hipGraphInstantiate(&e14, g14, nullptr, nullptr, 0));
    hipGraphNode_t nd = find_node(g14, (void*)first_kernel);
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    for (int rep = 0; rep < 2; ++rep) {
      float ms_a, ms_b, ms_c;
      CK(hipEventRecord(ev0, st));
      for (int i = 0; i < IT; ++i) {
        MarkArgs ai{out, N + 1, i};
        first_kernel<<<n / 256, 256, 0, st>>>(ai, p, n);
        CK(hipGraphLaunch(e13, st));
      }
      CK(hipEventRecord(ev1, st));
      CK(hipEventSynchronize(ev1));
      CK(hipEventElapsedTime(&ms_a, ev0, ev1));
      CK(hipEventRecord(ev0, st));
      for (int i = 0; i < IT; ++i) {
        MarkArgs ai{out, N + 2, i};
        void* kp[3];
        float* pp = p;
        int nn = n;
        kp[0] = &ai; kp[1] = &pp; kp[2] = &nn;
        hipKernelNodeParams q{};
        q.func = (void*)first_kernel;
        q.gridDim = dim3(n / 256);
        q.blockDim = dim3(256);
        q.kernelParams = kp;
        CK(hipGraphExecKernelNodeSetParams(e14, nd, &q));
        CK(hipGraphLaunch(e14, st));
      }
      CK(hipEventRecord(ev1, st));
      CK(hipEventSynchronize(ev1));
      CK(hipEventElapsedTime(&ms_b, ev0, ev1));
      CK(hipEventRecord(ev0, st));
      for (int i = 0; i < IT; ++i) CK(hipGraphLaunch(e14, st));
      CK(hipEventRecord(ev1, st));
      CK(hipEventSynchronize(ev1));
      CK(hipEventElapsedTime(&ms_c, ev0, ev1));
      printf("3. per iteration: eager first + graph(13) %.2f us | graph(14), first node's arguments replaced %.2f us | graph(14) untouched %.2f us\n",
             ms_a * 1000 / IT, ms_b * 1000 / IT, ms_c * 1000 / IT);
    }
    CK(hipMemcpy(host.data(), out, sizeof(int) * (N + 8), hipMemcpyDeviceToHost));
    printf("   last values written: eager %d, replaced %d (both must be %d)\n", host[N + 1], host[N + 2], IT - 1);

    // ---- 4: what ONE graph launch costs on top of its nodes: graphs of K dependent kernels, K = 7 .. 56, back to back ----
    printf("4. per iteration of a K-node graph launched back to back (us), and per node:\n");
    double t7 = 0, t56 = 0;
    for (int KK : {7, 14, 28, 56}) {
      hipGraph_t g;
      hipGraphExec_t e;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int k = 0; k < KK; ++k) step_kernel<<<n / 256, 256, 0, st>>>(p, n);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
      const int it = 2000 * 14 / KK;
      for (int i = 0; i < 50; ++i) CK(hipGraphLaunch(e, st));
      float ms;
      CK(hipEventRecord(ev0, st));
      for (int i = 0; i < it; ++i) CK(hipGraphLaunch(e, st));
      CK(hipEventRecord(ev1, st));
      CK(hipEventSynchronize(ev1));
      CK(hipEventElapsedTime(&ms, ev0, ev1));
      const double us = ms * 1000 / it;
      printf("   K = %2d: %7.2f us per launch, %.3f us per node\n", KK, us, us / KK);
      if (KK == 7) t7 = us;
      if (KK == 56) t56 = us;
      CK(hipGraphExecDestroy(e));
      CK(hipGraphDestroy(g));
    }
    const double per_node = (t56 - t7) / 49.0;
    printf("   linear fit over K = 7 .. 56: %.3f us per node + %.2f us per graph launch\n", per_node, t7 - 7 * per_node);
    // the same K kernels launched eagerly (the host must keep up: 2000 x 14 launches)
    {
      float ms;
      for (int i = 0; i < 200; ++i) step_kernel<<<n / 256, 256, 0, st>>>(p, n);
      CK(hipEventRecord(ev0, st));
      for (int i = 0; i < 2000 * 14; ++i) step_kernel<<<n / 256, 256, 0, st>>>(p, n);
      CK(hipEventRecord(ev1, st));
      CK(hipEventSynchronize(ev1));
      CK(hipEventElapsedTime(&ms, ev0, ev1));
      printf("   eager, back to back: %.3f us per kernel\n", ms * 1000 / (2000 * 14));
    }
  }
  return 0;
}
