// vmm_probe.hip -- can the row slab live in one reserved virtual range that grows by mapping
// physical chunks (hipMemAddressReserve / hipMemCreate / hipMemMap) instead of hipMalloc +
// copy?  Measures, for a buffer of <GB>: the read-only streaming rate over hipMalloc memory and
// over a mapped range built from chunks of several sizes, and what reserve / create / map /
// set-access cost.  Build: hipcc --offload-arch=gfx950 -O3 tools/vmm_probe.hip -o tools/vmm_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void read_kernel(const f32x4 *__restrict__ p, size_t n16, float *out) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  for (; i + 7 * stride < n16; i += 8 * stride) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; i < n16; i += stride) acc += __builtin_nontemporal_load(p + i);
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123456.789f) out[0] = s;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int read_rate(const void *p, size_t bytes, float *out, hipStream_t s, double *gbps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double best = 0;
  for (int rep = 0; rep < 5; ++rep) {
    float ms;
    CK(hipEventRecord(e0, s));
    read_kernel<<<256 * 8, 256, 0, s>>>((const f32x4 *)p, bytes / 16, out);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double r = bytes / (ms * 1e-3) / 1e9;
    if (rep && r > best) best = r;
  }
  *gbps = best;
  return 0;
}

int main(int argc, char **argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 30.72;
  float *out;
  CK(hipMalloc((void **)&out, 64));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gmin = 0, grec = 0;
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  const size_t bytes = (size_t)(gb * 1e9) / (1 << 21) * (1 << 21);
  printf("{\"granularity_min\": %zu, \"granularity_recommended\": %zu, \"buffer_GB\": %.2f}\n", gmin, grec, bytes / 1e9);
  {
    void *a;
    double t0 = now();
    CK(hipMalloc(&a, bytes));
    const double t_alloc = now() - t0;
    CK(hipMemset(a, 0, bytes));
    CK(hipDeviceSynchronize());
    double r;
    if (read_rate(a, bytes, out, s, &r)) return 1;
    t0 = now();
    CK(hipFree(a));
    printf("{\"kind\": \"hipMalloc\", \"read_GBps\": %.0f, \"alloc_ms\": %.2f, \"free_ms\": %.2f}\n", r, t_alloc * 1e3, (now() - t0) * 1e3);
  }
  const size_t reserve = (size_t)320 << 30;  // more than the card holds: the range a shard would reserve once
  const size_t align = argc > 2 ? (size_t)atoll(argv[2]) << 20 : 0;  // alignment of the reserved range, MB
  const size_t chunk_sizes[] = {bytes, (size_t)1 << 30, (size_t)256 << 20, (size_t)32 << 20, (size_t)2 << 20};
  for (size_t chunk : chunk_sizes) {
    if (chunk == ((size_t)2 << 20) && bytes > ((size_t)8 << 30)) continue;  // thousands of 2-MB maps: small buffers only
    void *base = nullptr;
    double t0 = now();
    CK(hipMemAddressReserve(&base, reserve, align, nullptr, 0));
    const double t_reserve = now() - t0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    double t_create = 0, t_map = 0, t_access = 0;
    size_t mapped = 0;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = 0;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const bool uniform = argc > 3 && atoi(argv[3]);  // every chunk the same size (the range ends past `bytes`)
    while (mapped < bytes) {
      const size_t sz = uniform ? chunk : std::min(chunk, bytes - mapped);
      hipMemGenericAllocationHandle_t h;
      t0 = now();
      CK(hipMemCreate(&h, sz, &prop, 0));
      t_create += now() - t0;
      t0 = now();
      CK(hipMemMap((char *)base + mapped, sz, 0, h, 0));
      t_map += now() - t0;
      t0 = now();
      {
        hipError_t e = hipMemSetAccess((char *)base + mapped, sz, &acc, 1);
        if (e != hipSuccess) {
          fprintf(stderr, "hipMemSetAccess(base + %zu, %zu) chunk %zu: %s\n", mapped, sz, handles.size(), hipGetErrorString(e));
          return 1;
        }
      }
      t_access += now() - t0;
      handles.push_back(h);
      mapped += sz;
    }
    CK(hipMemsetAsync(base, 0, bytes, s));
    CK(hipStreamSynchronize(s));
    double r;
    if (read_rate(base, bytes, out, s, &r)) return 1;
    // a copy into the range from ordinary memory, as an insert would do
    void *src;
    CK(hipMalloc(&src, 64 << 20));
    CK(hipMemcpyAsync(base, src, 64 << 20, hipMemcpyDeviceToDevice, s));
    CK(hipStreamSynchronize(s));
    CK(hipFree(src));
    t0 = now();
    CK(hipMemUnmap(base, mapped));
    for (auto h : handles) CK(hipMemRelease(h));
    CK(hipMemAddressFree(base, reserve));
    const double t_free = now() - t0;
    printf("{\"kind\": \"mapped\", \"base\": \"%p\", \"chunk_MB\": %.0f, \"chunks\": %zu, \"read_GBps\": %.0f, \"reserve_ms\": %.3f, \"create_ms\": %.2f, "
           "\"map_ms\": %.2f, \"set_access_ms\": %.2f, \"release_ms\": %.2f}\n",
           base, chunk / 1048576.0, handles.size(), r, t_reserve * 1e3, t_create * 1e3, t_map * 1e3, t_access * 1e3, t_free * 1e3);
  }
  return 0;
}
