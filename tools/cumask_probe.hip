// cumask_probe.hip -- can the tail of batch group g (exact rescoring, select) and the head of group g + 1 (sample pass,
// thresholds) run INSIDE group g + 1's pass over the rows?  K2s keeps one block per CU resident for the whole pass (all
// of a CU's LDS), so nothing else is dispatched beside it: the only room is CUs the pass does not use.  This probe
// answers the three questions that design rests on:
//   1. hipExtStreamCreateWithCUMask on this part: which physical CUs (XCC, SE, CU) does bit i of the mask stand for?
//   2. what does the K2s pass lose on 248 / 240 / 224 CUs (it is power-limited, DESIGN 4.5: perhaps nothing)?
//   3. does a kernel on the complementary mask really run beside the pass, and what does the pass lose then?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ivettore_amd/csrc tools/cumask_probe.hip -o tools/cumask_probe
// Run:   tools/cumask_probe [rows]            one JSON line per measurement
#define VT_ENV_IMPLEMENTATION  // (this program's own copy of the library's settings table: csrc/vt_env.h)
#include "../vettore_amd/csrc/vt_batch_shadow.hip"

#include <cmath>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_kernel(float *p, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t h = (uint32_t)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xFFFF) / 32768.0f - 1.0f;
  }
}

// where a block runs: HW_ID (id 4) and XCC_ID (id 20) of its first wave; a short spin so that one launch spreads
// over every CU its queue may use
__global__ void whereami_kernel(uint32_t *out, int spin) {
  if (threadIdx.x == 0) {
    const uint32_t hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    const uint32_t xcc = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20);
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  const long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}

// a stand-in for the exact rescoring: gathers `per_q` rows of 3 KB per "query" (random rows), sums them
__global__ __launch_bounds__(256) void gather_kernel(const float *X, uint32_t ld, uint32_t rows, uint32_t per_block, float *out, uint32_t seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (uint32_t i = wave; i < per_block; i += 4) {
    uint32_t h = (blockIdx.x * 7919u + i) * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float *row = X + (size_t)(h % rows) * ld;
    for (uint32_t c = lane * 4; c < ld; c += 256) {
      const float4 v = *reinterpret_cast<const float4 *>(row + c);
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
}

static std::set<uint32_t> cus_of(const std::vector<uint32_t> &h, int blocks) {
  std::set<uint32_t> s;
  for (int b = 0; b < blocks; ++b) {
    const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
    const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    s.insert(xcc << 12 | se << 8 | sh << 4 | cu);
  }
  return s;
}

int main(int argc, char **argv) {
  const uint32_t rows = argc > 1 ? (uint32_t)atoll(argv[1]) : 10000000u;
  const uint32_t d = 768, ld = 768, nq_pad = 256;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d}\n", prop.gcnArchName, ncu);

  // ---- 1. the mask's bit order
  const int wblocks = 8192;
  uint32_t *dwhere;
  CK(hipMalloc(&dwhere, wblocks * 8));
  std::vector<uint32_t> hwhere(2 * wblocks);
  const int words = (ncu + 31) / 32;
  auto where_under = [&](const std::vector<uint32_t> &mask, std::set<uint32_t> *out) -> int {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    hipLaunchKernelGGL(whereami_kernel, dim3(wblocks), dim3(64), 0, s, dwhere, 20000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(hwhere.data(), dwhere, wblocks * 8, hipMemcpyDeviceToHost));
    *out = cus_of(hwhere, wblocks);
    CK(hipStreamDestroy(s));
    return 0;
  };
  std::vector<uint32_t> all(words, 0xFFFFFFFFu);
  std::set<uint32_t> full;
  if (where_under(all, &full)) return 1;
  printf("{\"mask\": \"all\", \"distinct_cus\": %zu}\n", full.size());
  for (int cleared : {1, 8, 16, 32}) {
    std::vector<uint32_t> m = all;
    for (int b = 0; b < cleared; ++b) m[b / 32] &= ~(1u << (b % 32));
    std::set<uint32_t> got;
    if (where_under(m, &got)) return 1;
    std::map<uint32_t, int> missing_per_xcc;
    std::string miss;
    for (uint32_t c : full)
      if (!got.count(c)) {
        missing_per_xcc[c >> 12] += 1;
        char buf[32];
        snprintf(buf, sizeof buf, "%s%x", miss.empty() ? "" : " ", c);
        miss += buf;
      }
    std::string per;
    for (auto &kv : missing_per_xcc) per += (per.empty() ? "" : ", ") + std::to_string(kv.first) + ": " + std::to_string(kv.second);
    printf("{\"mask\": \"bits 0..%d cleared\", \"distinct_cus\": %zu, \"missing_per_xcc\": {%s}, \"missing_xcc_se_sh_cu\": \"%s\"}\n", cleared - 1,
           got.size(), per.c_str(), miss.c_str());
    // and the complement: only those bits set
    std::vector<uint32_t> cm(words, 0u);
    for (int b = 0; b < cleared; ++b) cm[b / 32] |= 1u << (b % 32);
    std::set<uint32_t> cgot;
    if (where_under(cm, &cgot)) return 1;
    size_t overlap = 0;
    for (uint32_t c : cgot) overlap += got.count(c);
    printf("{\"mask\": \"only bits 0..%d\", \"distinct_cus\": %zu, \"shared_with_the_complement\": %zu}\n", cleared - 1, cgot.size(), overlap);
    fflush(stdout);
  }

  // ---- 2. the K2s pass on fewer CUs
  const uint32_t rows_img = (rows + 255) / 256 * 256;
  float *X, *Q, *dtau, *gout;
  void *imgS, *shadow;
  vt::BatchCand *cand;
  uint32_t *cnt;
  CK(hipMalloc(&X, (size_t)rows_img * ld * 4));
  CK(hipMalloc(&shadow, (size_t)rows_img * ld * 2));
  CK(hipMalloc(&Q, (size_t)256 * ld * 4));
  CK(hipMalloc(&imgS, vt::batch_shadow_image_bytes(ld)));
  CK(hipMalloc(&dtau, 256 * 4));
  CK(hipMalloc(&cand, (size_t)256 * 8192 * sizeof(vt::BatchCand)));
  CK(hipMalloc(&cnt, 256 * 4));
  CK(hipMalloc(&gout, 65536 * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, X, (size_t)rows_img * ld, 1u);
  hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, Q, (size_t)256 * ld, 2u);
  std::vector<float> ht(256, 1e30f);
  CK(hipMemcpy(dtau, ht.data(), 256 * 4, hipMemcpyHostToDevice));
  CK(vt::launch_batch_q_image16(Q, ld, nq_pad, imgS, 0));
  CK(vt::launch_shadow_build(X, ld, rows, rows_img, ld, shadow, 0));
  CK(hipDeviceSynchronize());
  vt::BatchScoreArgs a{};
  a.X = X; a.stride = ld; a.Q = Q; a.ld = ld; a.nq_pad = nq_pad; a.n = rows; a.n_total = rows;
  a.tau = dtau; a.cand = cand; a.cand_count = cnt; a.cand_cap = 8192; a.Xshadow = shadow; a.Qimage = imgS;
  hipEvent_t e0, e1, g0, g1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&g0));
  CK(hipEventCreate(&g1));
  for (int rep = 0; rep < 60; ++rep) CK(vt::launch_batch_scores_shadow(a, false, (uint32_t)ncu, 0));  // clocks
  CK(hipDeviceSynchronize());
  auto time_pass = [&](hipStream_t s, uint32_t grid, float *best_out) -> int {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, s));
      CK(vt::launch_batch_scores_shadow(a, false, grid, s));
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    *best_out = best;
    return 0;
  };
  for (int round = 0; round < 2; ++round)
    for (uint32_t grid : {(uint32_t)ncu, (uint32_t)ncu - 8, (uint32_t)ncu - 16, (uint32_t)ncu - 32, (uint32_t)ncu - 64}) {
      float ms;
      if (time_pass(0, grid, &ms)) return 1;
      printf("{\"k2s_pass\": \"unmasked\", \"grid\": %u, \"ms\": %.4f, \"TBps_bf16\": %.2f}\n", grid, ms, (double)rows * d * 2 / (ms * 1e-3) / 1e12);
      fflush(stdout);
    }

  // ---- 3. the pass on a masked stream, a gather kernel on the complement
  for (int reserved : {8, 16, 32}) {
    std::vector<uint32_t> mpass = all, mside(words, 0u);
    for (int b = 0; b < reserved; ++b) {
      mpass[b / 32] &= ~(1u << (b % 32));
      mside[b / 32] |= 1u << (b % 32);
    }
    hipStream_t sp, ss;
    CK(hipExtStreamCreateWithCUMask(&sp, (uint32_t)words, mpass.data()));
    CK(hipExtStreamCreateWithCUMask(&ss, (uint32_t)words, mside.data()));
    float alone;
    if (time_pass(sp, (uint32_t)(ncu - reserved), &alone)) return 1;
    // the side work alone on its CUs: 256 "queries" x 8 blocks x 114 rows of 3 KB = 715 MB
    float side_alone = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipEventRecord(g0, ss));
      hipLaunchKernelGGL(gather_kernel, dim3(2048), dim3(256), 0, ss, X, ld, rows, 114u, gout, 77u + rep);
      CK(hipEventRecord(g1, ss));
      CK(hipEventSynchronize(g1));
      float ms;
      CK(hipEventElapsedTime(&ms, g0, g1));
      if (rep > 0) side_alone = std::min(side_alone, ms);
    }
    // together: the side kernel starts first (it is what is pending when the pass begins)
    float together = 1e30f, side_together = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(g0, ss));
      hipLaunchKernelGGL(gather_kernel, dim3(2048), dim3(256), 0, ss, X, ld, rows, 114u, gout, 99u + rep);
      CK(hipEventRecord(g1, ss));
      CK(hipEventRecord(e0, sp));
      CK(vt::launch_batch_scores_shadow(a, false, (uint32_t)(ncu - reserved), sp));
      CK(hipEventRecord(e1, sp));
      CK(hipEventSynchronize(e1));
      CK(hipEventSynchronize(g1));
      float ms, gms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipEventElapsedTime(&gms, g0, g1));
      if (rep > 0) {
        together = std::min(together, ms);
        side_together = std::min(side_together, gms);
      }
    }
    // and the same pair with NO masks (what two plain streams do today)
    float plain_pass = 1e30f, plain_side = 1e30f;
    {
      hipStream_t p0, p1;
      CK(hipStreamCreateWithFlags(&p0, hipStreamNonBlocking));
      CK(hipStreamCreateWithFlags(&p1, hipStreamNonBlocking));
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(g0, p1));
        hipLaunchKernelGGL(gather_kernel, dim3(2048), dim3(256), 0, p1, X, ld, rows, 114u, gout, 199u + rep);
        CK(hipEventRecord(g1, p1));
        CK(hipEventRecord(e0, p0));
        CK(vt::launch_batch_scores_shadow(a, false, (uint32_t)ncu, p0));
        CK(hipEventRecord(e1, p0));
        CK(hipEventSynchronize(e1));
        CK(hipEventSynchronize(g1));
        float ms, gms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipEventElapsedTime(&gms, g0, g1));
        if (rep > 0) {
          plain_pass = std::min(plain_pass, ms);
          plain_side = std::min(plain_side, gms);
        }
      }
      CK(hipStreamDestroy(p0));
      CK(hipStreamDestroy(p1));
    }
    printf("{\"reserved_cus\": %d, \"pass_alone_ms\": %.4f, \"side_alone_ms\": %.4f, \"pass_beside_side_ms\": %.4f, \"side_beside_pass_ms\": %.4f, "
           "\"plain_streams_pass_ms\": %.4f, \"plain_streams_side_ms\": %.4f}\n",
           reserved, alone, side_alone, together, side_together, plain_pass, plain_side);
    fflush(stdout);
    CK(hipStreamDestroy(sp));
    CK(hipStreamDestroy(ss));
  }
  return 0;
}
