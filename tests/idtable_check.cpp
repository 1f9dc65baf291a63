// Randomised check of vettore_amd/csrc/host/vt_idtable.h against std::unordered_map, driven the
// way the shard drives it: find-or-insert of appended rows, swap-delete (erase + move_row of the
// last row), growth across many rebuilds, ids of every length incl. empty and shared prefixes.
#include "../vettore_amd/csrc/host/vt_idtable.h"

#include <cstdio>
#include <random>
#include <unordered_map>

using vt_host::IdTable;

static std::string make_id(std::mt19937_64 &rng, int style) {
  const uint64_t v = rng() % (style == 0 ? 300 : style == 1 ? 20000 : 5000000);
  switch (style) {
    case 0: return std::string((size_t)(v % 40), 'a' + (char)(v % 3)) + std::to_string(v);  // long runs, few values
    case 1: return "doc-" + std::to_string(v);
    default: {
      std::string s((size_t)(rng() % 24), '\0');
      for (auto &c : s) c = (char)(rng() & 0xFF);
      return s;
    }
  }
}

int main() {
  std::mt19937_64 rng(12345);
  for (int round = 0; round < 6; ++round) {
    std::vector<std::string> ids;
    IdTable t(&ids);
    std::unordered_map<std::string, uint32_t> ref;
    if (round % 2) t.reserve(1000);
    const int steps = round < 3 ? 200000 : 60000;
    for (int step = 0; step < steps; ++step) {
      const std::string id = make_id(rng, round % 3);
      const uint64_t h = vt_host::hash_id(id.data(), id.size());
      const uint32_t got = t.find(id.data(), id.size(), h);
      auto it = ref.find(id);
      if ((it == ref.end()) != (got == IdTable::kNone) || (it != ref.end() && it->second != got)) {
        std::printf("find mismatch round %d step %d\n", round, step);
        return 1;
      }
      const uint64_t op = rng() % 10;
      if (got == IdTable::kNone) {
        if (op < 7) {  // append
          const uint32_t row = (uint32_t)ids.size();
          t.insert(h, row);
          ids.push_back(id);
          ref.emplace(id, row);
        }
      } else if (op < 4) {  // swap-delete, exactly as shard_delete does it
        const uint32_t r = got, last = (uint32_t)ids.size() - 1;
        if (!t.erase(id.data(), id.size(), h)) { std::printf("erase failed\n"); return 1; }
        ref.erase(id);
        if (r != last) {
          const std::string &moved = ids[last];
          if (!t.move_row(moved.data(), moved.size(), vt_host::hash_id(moved.data(), moved.size()), r)) { std::printf("move failed\n"); return 1; }
          ref[moved] = r;
          ids[r] = std::move(ids[last]);
        }
        ids.pop_back();
      }
      if (t.size() != ref.size() || ids.size() != ref.size()) { std::printf("size mismatch\n"); return 1; }
    }
    for (auto &kv : ref) {  // every survivor still found, at its row
      if (t.find(kv.first.data(), kv.first.size(), vt_host::hash_id(kv.first.data(), kv.first.size())) != kv.second) {
        std::printf("final mismatch round %d\n", round);
        return 1;
      }
    }
    if (t.slots() * 7 < t.size() * 10) { std::printf("overfull\n"); return 1; }
  }
  std::printf("ok\n");
  return 0;
}
