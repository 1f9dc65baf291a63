// vt_coalesce.h -- searches that meet on one handle travel as one batch.
// Part of vt_index.cpp's translation unit (included there, in this order, exactly once): the host
// side is one TU on purpose -- everything below the C ABI lives in an anonymous namespace.
#pragma once

namespace {

// ---- searches that meet on one handle go together -----------------------------------------
// The reference's readers share an RwLock and scale with the host's cores.  Here every search
// is a pass over the corpus in HBM, and callers that run side by side on their own streams
// each read all of it.  So a search that finds another one running waits for it; whoever waits
// first then leads everything that has queued up with its limit as ONE batch -- the batch path
// gives every query the hits its own search would get, bit for bit -- and the others wake up
// with their lists.  An idle handle adds nothing: the first caller runs at once, alone.  Small
// corpora (latency-bound, not bandwidth-bound) keep more than one operation in flight
// (coalesce_slots).  What would force work a lone search avoids (a strict re-rank after unsorted inserts) is not batched: those
// callers are released to search side by side as before.  `VT_COALESCE=0` switches it off.
constexpr size_t kCoalesceMax = 256;
// Operations in flight before callers start to queue: a pass over a large corpus owns the
// memory system, two over a medium one still overlap their fixed costs, and searches of a
// corpus of a few MB are all fixed cost -- there every reader context runs side by side and
// only the callers beyond them travel together (tools/reader_probe.cpp).
unsigned coalesce_slots(uint64_t corpus_bytes) {
  if (const char *e = std::getenv("VT_COALESCE_SLOTS")) return (unsigned)std::max(1, std::atoi(e));
  return corpus_bytes < (64ull << 20) ? (unsigned)kMaxContexts : corpus_bytes < (1ull << 30) ? 2u : 1u;
}

bool coalescing_enabled() {
  const char *e = std::getenv("VT_COALESCE");
  return !(e && e[0] == '0');
}

// Runs the members of one batch (all with the leader's limit and query length).
void run_coalesced(vt_flat *h, std::vector<vt_flat::Waiting *> &members) {
  const size_t limit = members[0]->limit, n = members[0]->n;
  auto alone = [&](vt_flat::Waiting *w) {
    w->status = search_direct(h, w->query, w->n, w->limit, w->out);
    if (w->status != VT_OK) w->error = g_last_error;
  };
  if (members.size() == 1) {
    alone(members[0]);
    return;
  }
  // every query is judged on its own (flat.rs:97-101), as if it had come alone
  std::vector<vt_flat::Waiting *> good;
  {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    const long dim = handle_dim(h);
    for (vt_flat::Waiting *w : members) {
      const int st = h->poisoned ? poisoned_status() : validate_vector(w->query, w->n, dim);
      if (st != VT_OK) {
        w->status = st;
        w->error = st == VT_ERR_POISONED ? g_last_error : std::string();
      } else {
        good.push_back(w);
      }
    }
  }
  if (good.size() < 2) {
    for (vt_flat::Waiting *w : good) alone(w);
    return;
  }
  std::vector<float> qs(good.size() * n);
  for (size_t i = 0; i < good.size(); ++i) std::memcpy(&qs[i * n], good[i]->query, n * sizeof(float));
  std::vector<vt_hits *> outs(good.size(), nullptr);
  const int st = batch_direct(h, qs.data(), good.size(), n, limit, outs.data());
  if (st == VT_OK) {
    for (size_t i = 0; i < good.size(); ++i) *good[i]->out = outs[i];
    return;
  }
  // one query's failure ("metric overflow", a dimension that changed under us) is that query's own
  for (vt_flat::Waiting *w : good) alone(w);
}

int coalesced_search(vt_flat *h, const float *query, size_t n, size_t limit, vt_hits **out) {
  if (limit == 0 || limit > (size_t)vt::kMaxFusedK || n == 0 || !coalescing_enabled()) return search_direct(h, query, n, limit, out);
  vt_flat::Coalescer &co = h->co;
  const unsigned max_active = coalesce_slots(h->approx_bytes.load(std::memory_order_relaxed));
  vt_flat::Waiting me(query, n, limit, out);
  std::vector<vt_flat::Waiting *> members;
  members.reserve(kCoalesceMax);  // (no allocation once others depend on this caller)
  {
    std::unique_lock<std::mutex> lk(co.mu);
    if (co.active < max_active && co.waiting.empty()) {
      co.active += 1;  // nobody to wait for, nobody to take along
    } else {
      co.waiting.push_back(&me);
      co.gather.notify_one();
      me.wake.wait(lk, [&] { return me.state != vt_flat::Waiting::QUEUED; });
      if (me.state == vt_flat::Waiting::DONE) {
        if (me.status != VT_OK) g_last_error = me.error;
        return me.status;
      }
      if (me.state == vt_flat::Waiting::ALONE) {
        lk.unlock();
        return search_direct(h, query, n, limit, out);
      }
      // LEADS (the operation that just finished passed its slot on: `active` already counts this one).
      // Callers that have just been answered are about to come back -- give them a moment (a few
      // % of a pass) before the next pass over the corpus starts without them
      if (co.last_batch > 1 && co.waiting.size() + 1 < co.last_batch) {
        const double window = std::min(300e-6, 0.03 * co.last_seconds);
        const size_t want = co.last_batch - 1;
        co.gather.wait_for(lk, std::chrono::duration<double>(window), [&] { return co.waiting.size() >= want; });
      }
      for (auto it = co.waiting.begin(); it != co.waiting.end() && members.size() + 1 < kCoalesceMax;) {
        if ((*it)->limit == limit && (*it)->n == n) {
          members.push_back(*it);
          it = co.waiting.erase(it);
        } else {
          ++it;
        }
      }
    }
    members.insert(members.begin(), &me);
  }
  // a batch needs strictly current id ranks; a lone search after unsorted inserts does not
  // (lazy ranks): then nobody is made to wait for a re-rank -- everyone searches alone
  bool disband = false;
  if (members.size() > 1 && !h->multi()) {
    std::shared_lock<std::shared_mutex> rl(h->rw);
    disband = shard_stale(h->shards[0].get(), NEED_STRICT_RANKS, limit);
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (disband) {
    {
      std::lock_guard<std::mutex> g(co.mu);
      for (size_t i = 1; i < members.size(); ++i) {
        members[i]->state = vt_flat::Waiting::ALONE;
        members[i]->wake.notify_one();
      }
    }
    members.resize(1);
  }
  try {
    run_coalesced(h, members);
  } catch (...) {  // (host memory, most likely) -- nobody may be left waiting
    for (vt_flat::Waiting *w : members) {
      if (w->out && *w->out) {
        delete *w->out;
        *w->out = nullptr;
      }
      w->status = VT_ERR_NOMEM;
      w->error = "out of host memory";
    }
  }
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  {
    std::lock_guard<std::mutex> g(co.mu);
    // (notified under the lock: a member may return -- and its Waiting leave the stack -- the
    // moment it can take the lock and see DONE)
    for (size_t i = 1; i < members.size(); ++i) {
      members[i]->state = vt_flat::Waiting::DONE;
      members[i]->wake.notify_one();
    }
    co.last_batch = members.size();
    co.last_seconds = seconds;
    if (members.size() > 1) {
      co.batches += 1;
      co.batched_queries += members.size();
    }
    // the longest-waiting caller leads next, in this operation's slot (it takes the others along)
    if (!co.waiting.empty()) {
      co.waiting.front()->state = vt_flat::Waiting::LEADS;
      co.waiting.front()->wake.notify_one();
      co.waiting.pop_front();
    } else {
      co.active -= 1;
    }
  }
  if (me.status != VT_OK) g_last_error = me.error;
  return me.status;
}

}  // namespace
