// vt_idtable.h -- id bytes -> row: the host's stand-in for the keys of the reference's
// HashMap<String, Vec<f32>> (flat.rs:13-17).  Stand-alone (no HIP, no other header of this
// directory): tests/test_idtable.py builds it with g++ and checks it against std::unordered_map.
//
// Open addressing with linear probing over 16-byte slots {hash, row + 1}; the id bytes themselves
// stay where they already are -- the shard's `ids[row]` -- and are only touched when a slot's
// 64-bit hash matches.  A std::unordered_map<std::string, uint32_t> cost a node allocation, a
// second copy of every id and two cache misses per new id: 300 ns per row of a bulk load (the
// largest part of it) and ~90 bytes per row; this is one miss, no allocation, <= 32 bytes.
// Deletion shifts the following entries back (no tombstones), using the stored hashes only.
#pragma once

#include <cstdint>
#include <cstring>
#include <random>
#include <string>
#include <vector>

namespace vt_host {

// (seeded once per process, like the random keys of Rust's HashMap: ids come from callers, and a
// fixed function would let them be chosen to pile up in one probe run)
inline uint64_t hash_seed() {
  static const uint64_t seed = [] {
    std::random_device rd;
    return ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ 0x9E3779B97F4A7C15ull;
  }();
  return seed;
}

inline uint64_t hash_id(const char *p, size_t n) {
  uint64_t h = hash_seed() ^ (n * 0xff51afd7ed558ccdull);
  while (n >= 8) {
    uint64_t w;
    std::memcpy(&w, p, 8);
    h = (h ^ w) * 0x9FB21C651E98DF25ull;
    h ^= h >> 32;
    p += 8;
    n -= 8;
  }
  uint64_t w = 0;
  if (n) std::memcpy(&w, p, n);
  h = (h ^ w) * 0x9FB21C651E98DF25ull;
  h ^= h >> 29;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 32;
  return h;
}

class IdTable {
 public:
  static constexpr uint32_t kNone = 0xFFFFFFFFu;

  explicit IdTable(const std::vector<std::string> *ids) : ids_(ids) {}

  size_t size() const { return size_; }
  size_t slots() const { return slots_.size(); }
  void clear() {
    slots_.clear();
    size_ = 0;
    mask_ = 0;
  }
  // Room for `n` ids without growing on the way.
  void reserve(size_t n) {
    size_t want = 16;
    while (want * 7 < n * 10) want *= 2;  // load <= 0.7
    if (want > slots_.size()) rebuild(want);
  }
  void prefetch(uint64_t hash) const {
    if (!slots_.empty()) __builtin_prefetch(&slots_[hash & mask_]);
  }
  // Row of the id, or kNone.
  uint32_t find(const char *id, size_t len, uint64_t hash) const {
    if (slots_.empty()) return kNone;
    for (size_t i = hash & mask_;; i = (i + 1) & mask_) {
      const Slot &s = slots_[i];
      if (s.row1 == 0) return kNone;
      if (s.hash == hash && same(s.row1 - 1, id, len)) return s.row1 - 1;
    }
  }
  // The id must not be present (find first).  `row` is where its bytes will be: ids[row].
  void insert(uint64_t hash, uint32_t row) {
    if ((size_ + 1) * 10 > slots_.size() * 7) rebuild(slots_.empty() ? 16 : slots_.size() * 2);
    place(hash, row + 1);
    size_ += 1;
  }
  // The id now lives in another row (a swap-delete moved it).  Call while ids[old row] still holds the bytes.
  bool move_row(const char *id, size_t len, uint64_t hash, uint32_t new_row) {
    if (slots_.empty()) return false;
    for (size_t i = hash & mask_;; i = (i + 1) & mask_) {
      Slot &s = slots_[i];
      if (s.row1 == 0) return false;
      if (s.hash == hash && same(s.row1 - 1, id, len)) {
        s.row1 = new_row + 1;
        return true;
      }
    }
  }
  // Forgets the id.  Call while ids[row] still holds the bytes.
  bool erase(const char *id, size_t len, uint64_t hash) {
    if (slots_.empty()) return false;
    size_t i = hash & mask_;
    for (;; i = (i + 1) & mask_) {
      const Slot &s = slots_[i];
      if (s.row1 == 0) return false;
      if (s.hash == hash && same(s.row1 - 1, id, len)) break;
    }
    // backward shift: close the hole with the entries of the run that may move up
    size_t hole = i;
    for (size_t j = (i + 1) & mask_;; j = (j + 1) & mask_) {
      const Slot &s = slots_[j];
      if (s.row1 == 0) break;
      const size_t home = s.hash & mask_;
      // s may fill the hole iff its home is not inside (hole, j] (cyclically)
      const bool stays = hole <= j ? (home > hole && home <= j) : (home > hole || home <= j);
      if (!stays) {
        slots_[hole] = s;
        hole = j;
      }
    }
    slots_[hole] = Slot{0, 0, 0};
    size_ -= 1;
    return true;
  }

 private:
  struct Slot {
    uint64_t hash;
    uint32_t row1;  // row + 1; 0 = empty
    uint32_t pad;
  };
  bool same(uint32_t row, const char *id, size_t len) const {
    const std::string &s = (*ids_)[row];
    return s.size() == len && (len == 0 || std::memcmp(s.data(), id, len) == 0);
  }
  static void place_in(std::vector<Slot> &slots, size_t mask, uint64_t hash, uint32_t row1) {
    size_t i = hash & mask;
    while (slots[i].row1 != 0) i = (i + 1) & mask;
    slots[i] = Slot{hash, row1, 0};
  }
  void place(uint64_t hash, uint32_t row1) { place_in(slots_, mask_, hash, row1); }
  // (the new slots exist before the old ones go: an allocation that fails leaves the table as it was)
  void rebuild(size_t want) {
    std::vector<Slot> fresh(want, Slot{0, 0, 0});
    for (const Slot &s : slots_)
      if (s.row1) place_in(fresh, want - 1, s.hash, s.row1);
    slots_.swap(fresh);
    mask_ = want - 1;
  }

  const std::vector<std::string> *ids_;
  std::vector<Slot> slots_;
  size_t size_ = 0, mask_ = 0;
};

}  // namespace vt_host
