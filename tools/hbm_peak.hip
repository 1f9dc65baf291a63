// hbm_peak.hip -- what this box's HBM delivers to the simplest possible kernels, as the
// yardstick beside the 8 TB/s spec figure (SURVEY §8d "confirm on the box"):
//   read   : every lane sums 16-byte non-temporal loads, 8 in flight, nothing written
//   copy   : 16-byte loads + stores (bytes counted = read + written)
//   memcpy : hipMemcpyDtoDAsync (bytes counted = read + written)
// Build: hipcc --offload-arch=gfx950 -O3 tools/hbm_peak.hip -o tools/hbm_peak ; prints one JSON line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

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
  if (s == 123456.789f) out[0] = s;  // never true: keeps the loads alive
}

__global__ __launch_bounds__(256) void copy_kernel(const f32x4 *__restrict__ p, f32x4 *__restrict__ q, size_t n16) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], q + i + u * stride);
  }
  for (; i < n16; i += stride) q[i] = p[i];
}

// prefix read: the first `pre16` 16-byte words of every row of `row16` words (the funnel
// stage's access pattern: 512 B of each 3 KiB row); one wave-instruction covers 1 KiB.
__global__ __launch_bounds__(256) void prefix_kernel(const f32x4 *__restrict__ p, size_t rows, uint32_t row16,
                                                     uint32_t pre16, float *out) {
  const size_t total = rows * pre16;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  for (; i + 7 * stride < total; i += 8 * stride) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t j = i + u * stride;
      const size_t r = j / pre16, c = j - r * pre16;
      v[u] = __builtin_nontemporal_load(p + r * row16 + c);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123456.789f) out[0] = s;
}

int main(int argc, char **argv) {
  const double gb = argc > 1 ? atof(argv[1]) : 30.72;  // bytes read per launch, GB (default: the 10M x 768 corpus)
  const size_t bytes = (size_t)(gb * 1e9) / 4096 * 4096;
  float *a, *b, *out;
  CK(hipMalloc((void **)&a, bytes));
  CK(hipMalloc((void **)&b, bytes));
  CK(hipMalloc((void **)&out, 64));
  CK(hipMemset(a, 0, bytes));
  CK(hipMemset(b, 0, bytes));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t n16 = bytes / 16;
  double best_read = 0, best_copy = 0, best_memcpy = 0;
  int best_read_blocks = 0, best_copy_blocks = 0;
  const int per_cu[] = {2, 4, 8, 16};
  float ms;
  for (int pc : per_cu) {
    const int blocks = cus * pc;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, s));
      read_kernel<<<blocks, 256, 0, s>>>((const f32x4 *)a, n16, out);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double r = bytes / (ms * 1e-3) / 1e9;
      if (rep && r > best_read) { best_read = r; best_read_blocks = pc; }
    }
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(e0, s));
      copy_kernel<<<blocks, 256, 0, s>>>((const f32x4 *)a, (f32x4 *)b, n16);
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double r = 2.0 * bytes / (ms * 1e-3) / 1e9;
      if (rep && r > best_copy) { best_copy = r; best_copy_blocks = pc; }
    }
  }
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0, s));
    CK(hipMemcpyDtoDAsync((hipDeviceptr_t)b, (hipDeviceptr_t)a, bytes, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double r = 2.0 * bytes / (ms * 1e-3) / 1e9;
    if (rep && r > best_memcpy) best_memcpy = r;
  }
  if (argc > 2) {  // hbm_peak <GB> <prefix floats>: strided prefix read of 768-float rows
    const uint32_t pre16 = (uint32_t)atoi(argv[2]) / 4, row16 = 768 / 4;
    const size_t rows = bytes / 3072;
    double best = 0;
    int best_pc = 0;
    for (int pc : per_cu) {
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, s));
        prefix_kernel<<<cus * pc, 256, 0, s>>>((const f32x4 *)a, rows, row16, pre16, out);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double r = (double)rows * pre16 * 16 / (ms * 1e-3) / 1e9;
        if (rep && r > best) { best = r; best_pc = pc; }
      }
    }
    printf("{\"rows\": %zu, \"prefix_floats\": %u, \"prefix_read_useful_GBps\": %.0f, \"blocks_per_cu\": %d}\n", rows,
           pre16 * 4, best, best_pc);
  }
  printf("{\"device\": \"%s\", \"cus\": %d, \"buffer_GB\": %.2f, \"read_GBps\": %.0f, \"read_blocks_per_cu\": %d, "
         "\"copy_GBps\": %.0f, \"copy_blocks_per_cu\": %d, \"memcpy_d2d_GBps\": %.0f}\n",
         prop.name, cus, bytes / 1e9, best_read, best_read_blocks, best_copy, best_copy_blocks, best_memcpy);
  return 0;
}
