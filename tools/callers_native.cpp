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
#include <condition_variable>
#include <cstring>
#include <mutex>
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

// ---- callers that MEET (tests/test_gpu_coalesce.py, tests/test_gpu_shadow.py) ------------------------------------
// `threads` native threads make `rounds` calls each, all of them leaving a barrier together at the start of every
// round.  With libvettore_hip_hooks.so and `hold` != 0 the handle's first caller of a round keeps its slot until the
// other threads have queued behind it (test_coalesce_hold_until, host/vt_concurrency.h), so that who travels with whom
// is a fact the test can assert instead of a matter of timing.  Thread t calls entry point kinds[t] with params[t] /
// candidates[t]; in round r its query is number (7 t + r) mod nq.  Every answer is compared (ids, raw bits) with the
// same call made alone before the threads start.
// Returns 0, 1 (arguments), 2 (a call alone failed), 3 (the library has no hold hook: not the hooks build).
namespace {
struct Barrier {
  std::mutex mu;
  std::condition_variable cv;
  int waiting = 0, generation = 0, parties;
  explicit Barrier(int n) : parties(n) {}
  void arrive() {
    std::unique_lock<std::mutex> g(mu);
    const int gen = generation;
    if (++waiting == parties) {
      waiting = 0;
      generation += 1;
      cv.notify_all();
    } else {
      cv.wait(g, [&] { return generation != gen; });
    }
  }
};
}  // namespace

extern "C" int vt_callers_meet(vt_flat *h, const float *queries, size_t nq, size_t d, size_t limit, const int *kinds,
                               const size_t *params, const size_t *candidates, int threads, int rounds, int hold,
                               unsigned long long *mismatches, unsigned long long *failures) {
  if (!h || !queries || nq == 0 || threads <= 0 || rounds <= 0 || !kinds || !params || !candidates || !mismatches || !failures)
    return 1;
  auto query_of = [&](int t, int r) { return (size_t)(7 * t + r) % nq; };
  std::vector<std::vector<Answer>> alone((size_t)threads, std::vector<Answer>((size_t)rounds));
  for (int t = 0; t < threads; ++t)
    for (int r = 0; r < rounds; ++r) {
      vt_hits *hits = nullptr;
      if (call(h, kinds[t], queries + query_of(t, r) * d, d, limit, params[t], candidates[t], &hits) != 0) return 2;
      alone[t][r] = take(hits);
      vt_hits_free(hits);
    }
  if (hold && vt_debug_set("test_coalesce_hold_until", threads) != 0) return 3;
  std::atomic<unsigned long long> wrong{0}, failed{0};
  Barrier barrier(threads);
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      for (int r = 0; r < rounds; ++r) {
        barrier.arrive();
        vt_hits *hits = nullptr;
        if (call(h, kinds[t], queries + query_of(t, r) * d, d, limit, params[t], candidates[t], &hits) != 0) {
          failed += 1;  // (and stays for the barriers: the others must not wait for ever)
          continue;
        }
        if (!same(take(hits), alone[t][r])) wrong += 1;
        vt_hits_free(hits);
      }
    });
  for (auto &th : pool) th.join();
  if (hold) vt_debug_set("test_coalesce_hold_until", 0);
  *mismatches = wrong.load();
  *failures = failed.load();
  return 0;
}
