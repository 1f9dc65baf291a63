// launch_probe.hip -- what the fixed part of one search call costs on this box:
// small H2D copy, kernel launches, and the ways of waiting for the result.
// Diagnostic only (hipcc --offload-arch=gfx950 -O2 tools/launch_probe.hip -o tools/launch_probe).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Big { float q[768]; };

__global__ void k_empty(float *p) { if (p && threadIdx.x == 9999) p[0] = 1.0f; }
__global__ void k_fetch(const float *host, float *dev, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dev[i] = host[i];
}
__global__ void k_use(const float *dev, float *out, int n) {
  __shared__ float s[768];
  for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = dev[i];
  __syncthreads();
  if (threadIdx.x == 0 && s[5] == 12345.0f) out[blockIdx.x] = s[0];
}
__global__ void k_arg(Big b, float *out) {
  __shared__ float s[768];
  for (int i = threadIdx.x; i < 768; i += blockDim.x) s[i] = b.q[i];
  __syncthreads();
  if (threadIdx.x == 0 && s[5] == 12345.0f) out[blockIdx.x] = s[0];
}
__global__ void k_flag(volatile unsigned *flag, unsigned v) {
  if (threadIdx.x == 0) {
    __threadfence_system();
    *flag = v;
  }
}

template <class F>
double bench(F f, int n = 2000) {
  for (int i = 0; i < 50; ++i) f(i);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f(i + 50);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}

int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "spin")) { CK(hipSetDeviceFlags(hipDeviceScheduleSpin)); printf("(hipDeviceScheduleSpin)\n"); }
  if (argc > 1 && !strcmp(argv[1], "yield")) { CK(hipSetDeviceFlags(hipDeviceScheduleYield)); printf("(hipDeviceScheduleYield)\n"); }
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float *hq, *dq, *dout, *hq_dev;
  unsigned *hflag, *hflag_dev;
  CK(hipHostMalloc((void **)&hq, 4096, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void **)&hq_dev, hq, 0));
  CK(hipHostMalloc((void **)&hflag, 64, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void **)&hflag_dev, hflag, 0));
  CK(hipMalloc((void **)&dq, 4096));
  CK(hipMalloc((void **)&dout, 4096 * 4));
  memset(hq, 0, 4096);
  *hflag = 0;
  Big big;
  memset(&big, 0, sizeof big);
  const int G = 512;

  printf("sync only                         %7.2f us\n", bench([&](int) { hipStreamSynchronize(s); }));
  printf("memcpy 3K H2D + sync              %7.2f us\n", bench([&](int) {
    hipMemcpyAsync(dq, hq, 3072, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }));
  printf("1 kernel + sync                   %7.2f us\n", bench([&](int) {
    k_use<<<G, 256, 0, s>>>(dq, dout, 768); hipStreamSynchronize(s); }));
  printf("2 kernels + sync                  %7.2f us\n", bench([&](int) {
    k_use<<<G, 256, 0, s>>>(dq, dout, 768); k_empty<<<1, 1024, 0, s>>>(dout); hipStreamSynchronize(s); }));
  printf("memcpy + 2 kernels + sync         %7.2f us\n", bench([&](int) {
    hipMemcpyAsync(dq, hq, 3072, hipMemcpyHostToDevice, s);
    k_use<<<G, 256, 0, s>>>(dq, dout, 768); k_empty<<<1, 1024, 0, s>>>(dout); hipStreamSynchronize(s); }));
  printf("fetch kernel + 2 kernels + sync   %7.2f us\n", bench([&](int) {
    k_fetch<<<1, 256, 0, s>>>(hq_dev, dq, 768);
    k_use<<<G, 256, 0, s>>>(dq, dout, 768); k_empty<<<1, 1024, 0, s>>>(dout); hipStreamSynchronize(s); }));
  printf("host-mapped query, 2 kernels+sync %7.2f us\n", bench([&](int) {
    k_use<<<G, 256, 0, s>>>(hq_dev, dout, 768); k_empty<<<1, 1024, 0, s>>>(dout); hipStreamSynchronize(s); }));
  printf("kernarg query, 2 kernels + sync   %7.2f us\n", bench([&](int) {
    k_arg<<<G, 256, 0, s>>>(big, dout); k_empty<<<1, 1024, 0, s>>>(dout); hipStreamSynchronize(s); }));
  printf("2 kernels + flag spin             %7.2f us\n", bench([&](int i) {
    k_use<<<G, 256, 0, s>>>(dq, dout, 768); k_flag<<<1, 64, 0, s>>>(hflag_dev, (unsigned)i + 1);
    while (*(volatile unsigned *)hflag != (unsigned)i + 1) {} }));
  printf("kernarg + flag spin               %7.2f us\n", bench([&](int i) {
    k_arg<<<G, 256, 0, s>>>(big, dout); k_flag<<<1, 64, 0, s>>>(hflag_dev, (unsigned)i + 1);
    while (*(volatile unsigned *)hflag != (unsigned)i + 1) {} }));
  printf("fetch + 2 kernels + flag spin     %7.2f us\n", bench([&](int i) {
    k_fetch<<<1, 256, 0, s>>>(hq_dev, dq, 768);
    k_use<<<G, 256, 0, s>>>(dq, dout, 768); k_flag<<<1, 64, 0, s>>>(hflag_dev, (unsigned)i + 1);
    while (*(volatile unsigned *)hflag != (unsigned)i + 1) {} }));
  hipStreamSynchronize(s);
  // waiting by polling the runtime instead of sleeping on the completion interrupt
  printf("memcpy + 2 kernels + query spin   %7.2f us\n", bench([&](int) {
    hipMemcpyAsync(dq, hq, 3072, hipMemcpyHostToDevice, s);
    k_use<<<G, 256, 0, s>>>(dq, dout, 768); k_empty<<<1, 1024, 0, s>>>(dout);
    while (hipStreamQuery(s) == hipErrorNotReady) {} }));
  {
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    printf("memcpy + 2 kernels + event spin   %7.2f us\n", bench([&](int) {
      hipMemcpyAsync(dq, hq, 3072, hipMemcpyHostToDevice, s);
      k_use<<<G, 256, 0, s>>>(dq, dout, 768); k_empty<<<1, 1024, 0, s>>>(dout);
      hipEventRecord(ev, s);
      while (hipEventQuery(ev) == hipErrorNotReady) {} }));
    printf("memcpy + 2 kernels + event sync   %7.2f us\n", bench([&](int) {
      hipMemcpyAsync(dq, hq, 3072, hipMemcpyHostToDevice, s);
      k_use<<<G, 256, 0, s>>>(dq, dout, 768); k_empty<<<1, 1024, 0, s>>>(dout);
      hipEventRecord(ev, s);
      hipEventSynchronize(ev); }));
  }
  // the same chain as an instantiated graph (r05: the one form the earlier rounds had not priced): captured once,
  // launched per call; then with one kernel node's parameters rewritten before every launch (what a search would do:
  // limit, thresholds and pointers change from call to call)
  {
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipMemcpyAsync(dq, hq, 3072, hipMemcpyHostToDevice, s);
    k_use<<<G, 256, 0, s>>>(dq, dout, 768);
    k_empty<<<1, 1024, 0, s>>>(dout);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    printf("graph(memcpy + 2 kernels) + sync  %7.2f us\n", bench([&](int) { hipGraphLaunch(exec, s); hipStreamSynchronize(s); }));
    size_t nn = 0;
    CK(hipGraphGetNodes(graph, nullptr, &nn));
    hipGraphNode_t nodes[8];
    CK(hipGraphGetNodes(graph, nodes, &nn));
    hipGraphNode_t knode = nullptr;
    for (size_t i = 0; i < nn; ++i) {
      hipGraphNodeType t;
      CK(hipGraphNodeGetType(nodes[i], &t));
      if (t == hipGraphNodeTypeKernel && !knode) knode = nodes[i];
    }
    if (knode) {
      hipKernelNodeParams kp;
      CK(hipGraphKernelNodeGetParams(knode, &kp));
      int n768 = 768;
      void *args[3] = {&dq, &dout, &n768};
      kp.kernelParams = args;
      printf("graph + node params set + sync    %7.2f us\n", bench([&](int i) {
        n768 = 768 - (i & 1);
        hipGraphExecKernelNodeSetParams(exec, knode, &kp);
        hipGraphLaunch(exec, s); hipStreamSynchronize(s); }));
    }
    // two kernels only (the query already on the device)
    hipGraph_t g2;
    hipGraphExec_t e2;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    k_use<<<G, 256, 0, s>>>(dq, dout, 768);
    k_empty<<<1, 1024, 0, s>>>(dout);
    CK(hipStreamEndCapture(s, &g2));
    CK(hipGraphInstantiate(&e2, g2, nullptr, nullptr, 0));
    printf("graph(2 kernels) + sync           %7.2f us\n", bench([&](int) { hipGraphLaunch(e2, s); hipStreamSynchronize(s); }));
    // five kernels (config 5's chain) plain against graph
    printf("memcpy + 5 kernels + sync         %7.2f us\n", bench([&](int) {
      hipMemcpyAsync(dq, hq, 3072, hipMemcpyHostToDevice, s);
      k_use<<<G, 256, 0, s>>>(dq, dout, 768); k_empty<<<256, 256, 0, s>>>(dout); k_empty<<<1, 1024, 0, s>>>(dout);
      k_use<<<32, 256, 0, s>>>(dq, dout, 768); k_empty<<<1, 1024, 0, s>>>(dout); hipStreamSynchronize(s); }));
    hipGraph_t g5;
    hipGraphExec_t e5;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipMemcpyAsync(dq, hq, 3072, hipMemcpyHostToDevice, s);
    k_use<<<G, 256, 0, s>>>(dq, dout, 768); k_empty<<<256, 256, 0, s>>>(dout); k_empty<<<1, 1024, 0, s>>>(dout);
    k_use<<<32, 256, 0, s>>>(dq, dout, 768); k_empty<<<1, 1024, 0, s>>>(dout);
    CK(hipStreamEndCapture(s, &g5));
    CK(hipGraphInstantiate(&e5, g5, nullptr, nullptr, 0));
    printf("graph(memcpy + 5 kernels) + sync  %7.2f us\n", bench([&](int) { hipGraphLaunch(e5, s); hipStreamSynchronize(s); }));
  }
  // events around the pair (what profiling mode adds)
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("memcpy + ev + 2 kernels + ev + sync %6.2f us\n", bench([&](int) {
    hipMemcpyAsync(dq, hq, 3072, hipMemcpyHostToDevice, s);
    hipEventRecord(e0, s);
    k_use<<<G, 256, 0, s>>>(dq, dout, 768);
    hipEventRecord(e1, s);
    k_empty<<<1, 1024, 0, s>>>(dout); hipStreamSynchronize(s); }));
  return 0;
}
