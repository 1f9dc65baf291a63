// callers_native.cpp -- T NATIVE threads calling one handle through the C ABI, for bench.py's
// `side.callers*` legs (Python threads queue for the interpreter lock before they can call again,
// which hides how the library lets concurrent callers share a pass -- DESIGN 6.3).
// Built by `make` into vettore_amd/lib/libvt_callers.so; bench.py falls back to Python threads
// when it is missing.  Not part of the product library.
//   kind 0: vt_flat_search(limit)
//   kind 1: vt_flat_quantized_search(candidates, limit)
//   kind 2: vt_flat_funnel_search(stages = [param], candidates, limit)
// Every query's answer "alone" is taken first; while the threads run, every `check_every`-th answer
// of a thread is compared with it (ids and raw bits).
#include "vettore_flat.h"

#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
struct Answer {
  std::vector<std::string> ids;
  std::vector<float> raw;
};
int call(vt_flat *h, int kind, const float *q, size_t d, size_t limit, size_t param, size_t candidates, vt_hits **out) {
  if (kind == 1) return vt_flat_quantized_search(h, q, d, candidates, limit, out);
  if (kind == 2) return vt_flat_funnel_search(h, q, d, &param, 1, candidates, limit, out);
  return vt_flat_search(h, q, d, limit, out);
}
Answer take(vt_hits *hits) {
  Answer a;
  const size_t n = vt_hits_len(hits);
  for (size_t i = 0; i < n; ++i) {
    size_t len = 0;
    const char *id = vt_hits_id(hits, i, &len);
    a.ids.emplace_back(id, len);
    a.raw.push_back(vt_hits_raw(hits, i));
  }
  return a;
}
bool same(const Answer &a, const Answer &b) {
  return a.ids == b.ids && a.raw.size() == b.raw.size() &&
         (a.raw.empty() || std::memcmp(a.raw.data(), b.raw.data(), a.raw.size() * sizeof(float)) == 0);
}
}  // namespace

extern "C" int vt_callers_run(vt_flat *h, const float *queries, size_t nq, size_t d, size_t limit, int kind, size_t param,
                              size_t candidates, int threads, double seconds, int check_every, unsigned long long *searches,
                              unsigned long long *mismatches, unsigned long long *failures) {
  if (!h || !queries || nq == 0 || threads <= 0 || !searches || !mismatches || !failures) return 1;
  std::vector<Answer> alone(nq);
  for (size_t i = 0; i < nq; ++i) {
    vt_hits *hits = nullptr;
    if (call(h, kind, queries + i * d, d, limit, param, candidates, &hits) != 0) return 2;
    alone[i] = take(hits);
    vt_hits_free(hits);
  }
  std::atomic<bool> stop{false};
  std::atomic<unsigned long long> total{0}, wrong{0}, failed{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      unsigned long long mine = 0;
      for (size_t i = (size_t)t; !stop.load(std::memory_order_relaxed); i += (size_t)threads) {
        const size_t j = i % nq;
        vt_hits *hits = nullptr;
        if (call(h, kind, queries + j * d, d, limit, param, candidates, &hits) != 0) {
          failed += 1;
          break;
        }
        if (check_every > 0 && mine % (unsigned)check_every == 0 && !same(take(hits), alone[j])) wrong += 1;
        vt_hits_free(hits);
        mine += 1;
      }
      total += mine;
    });
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  stop = true;
  for (auto &th : pool) th.join();
  *searches = total.load();
  *mismatches = wrong.load();
  *failures = failed.load();
  return 0;
}
