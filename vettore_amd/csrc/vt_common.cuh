// vt_common.cuh -- device helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include "vt_device.h"
#include "vt_env.h"

#include <float.h>

// rustc never contracts a*b+c; neither may these files, whatever the command line says.
#pragma clang fp contract(off)

namespace vt {
namespace dev {

constexpr int kWave = 64;
constexpr int kU = 8;  // 1-KiB loads in flight per wave

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

enum { OP_DOT = 0, OP_L2 = 1, OP_L1 = 2, OP_LINF = 3, OP_HAM = 4, OP_JAC = 5 };
enum { M_L2 = 0, M_L2SQ = 1, M_COS = 2, M_IP = 3, M_NIP = 4, M_L1 = 5, M_LINF = 6, M_HAM = 7, M_JAC = 8 };
constexpr int kErrOverflow = 4;  // VT_ERR_OVERFLOW

__host__ __device__ inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

__host__ __device__ inline int metric_op(int metric) {
  switch (metric) {
    case M_L2: case M_L2SQ: return OP_L2;
    case M_L1: return OP_L1;
    case M_LINF: return OP_LINF;
    case M_HAM: return OP_HAM;
    case M_JAC: return OP_JAC;
    default: return OP_DOT;
  }
}

// Neighbour lane (lane ^ 1) through DPP quad_perm [1,0,3,2]: folds into the
// consuming v_add_f32.
__device__ __forceinline__ float dpp_xor1(float v) {
  int i = __builtin_bit_cast(int, v);
  i = __builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true);
  return __builtin_bit_cast(float, i);
}

// f32::total_cmp as an order-preserving u32.
__device__ __forceinline__ uint32_t orderable(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ bool finite_f32(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }

// The wave-local LDS exchange fence: orders this wave's LDS writes before its
// later LDS reads by other lanes (one wave = one instruction stream; the fence
// only stops the compiler from moving accesses across it).
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-private candidate buffer in LDS holding (at least) the k smallest keys
// seen so far: keys[CAP] + payload[CAP], k + 64 <= CAP.
//  * `thr` (wave-uniform) is an upper bound: only keys < thr can still be among
//    the k smallest.  It is the exact k-th smallest as of the last compaction.
//  * offer(): every lane whose candidate passes appends it at n + (its rank among
//    the passing lanes) -- one ballot, one popcount, two ds_write_b64, whatever
//    the number of hits; nothing is serialised.
//  * when a full wave of hits might not fit, compact(): the lanes pull the
//    buffer into registers, a 64-step bitwise search (ballot + popcount per
//    step) finds the k-th smallest key, the survivors are written back densely
//    and thr tightens.  One compaction buys CAP - k - 63 insertions.
template <int CAP>
struct WaveTopK {
  static constexpr int kRegs = (CAP + kWave - 1) / kWave;
  uint64_t *bk;  // LDS keys[CAP]
  uint64_t *bp;  // LDS payload[CAP]: row | raw bits << 32
  uint32_t n;    // wave-uniform: entries in the buffer
  uint32_t k;
  uint64_t thr;  // wave-uniform

  static constexpr size_t lds_bytes() { return (size_t)CAP * 16; }

  __device__ __forceinline__ void init(void *lds, uint32_t k_) {
    bk = reinterpret_cast<uint64_t *>(lds);
    bp = bk + CAP;
    n = 0;
    k = k_;
    thr = kEmptyKey;
  }

  // Keeps the k smallest keys at bk/bp[0..n), tightens thr.
  __device__ __forceinline__ void compact(int lane) {
    wave_lds_fence();
    uint64_t key[kRegs], pay[kRegs];
#pragma unroll
    for (int j = 0; j < kRegs; ++j) {
      const uint32_t i = lane + kWave * j;
      const bool in = i < n && i < (uint32_t)CAP;
      key[j] = in ? bk[i] : kEmptyKey;
      pay[j] = in ? bp[i] : 0ull;
    }
    uint64_t T = kEmptyKey;
    if (n > k) {
      T = 0;  // k-th smallest: the largest T with count(key < T) < k
      for (int b = 63; b >= 0; --b) {
        const uint64_t c = T | (1ull << b);
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < kRegs; ++j) cnt += __popcll(__ballot(key[j] < c));
        if (cnt < k) T = c;
      }
      thr = T;
    }
    wave_lds_fence();
    const uint64_t lt = (1ull << lane) - 1;
    uint32_t base = 0;
    // Keys are NOT always distinct: rows inserted out of id order share one sentinel rank until
    // the next re-rank (lazy mode), so two of them with the same f32 rank carry the same key, and
    // callers may pass duplicate ids.  "<= T" can then hold more than k entries; the ones left
    // out must be among the equals of T, never a smaller key: the smaller ones are filed first.
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int j = 0; j < kRegs; ++j) {
        const bool has = key[j] != kEmptyKey && (pass == 0 ? key[j] < T : key[j] == T);
        const uint64_t m = __ballot(has);
        const uint32_t pos = base + __popcll(m & lt);
        if (has && pos < k) {
          bk[pos] = key[j];
          bp[pos] = pay[j];
        }
        base += __popcll(m);
      }
    }
    n = base < k ? base : k;
    wave_lds_fence();
  }

  // Offers one candidate per lane (valid lanes only).
  __device__ __forceinline__ void offer(bool valid, uint64_t ck, uint32_t crow, float craw, int lane) {
    bool hit = valid && ck < thr;
    uint64_t m = __ballot(hit);
    if (!m) return;
    if (n + (uint32_t)__popcll(m) > (uint32_t)CAP) {
      compact(lane);
      hit = hit && ck < thr;
      m = __ballot(hit);
      if (!m) return;
    }
    if (hit) {
      const uint32_t pos = n + __popcll(m & ((1ull << lane) - 1));
      bk[pos] = ck;
      bp[pos] = (uint64_t)crow | ((uint64_t)__float_as_uint(craw) << 32);
    }
    n += __popcll(m);
  }

  // Offers the `cnt` entries of another wave's (compacted) buffer: keys at `ok`, payload at ok + CAP.
  __device__ __forceinline__ void absorb(const uint64_t *ok, uint32_t cnt, int lane) {
    const uint64_t *op = ok + CAP;
    for (uint32_t base = 0; base < cnt; base += kWave) {
      const uint32_t i = base + lane;
      const bool valid = i < cnt;
      const uint64_t ck = valid ? ok[i] : kEmptyKey;
      const uint64_t cp = valid ? op[i] : 0ull;
      offer(valid, ck, (uint32_t)cp, __uint_as_float((uint32_t)(cp >> 32)), lane);
    }
  }

  // Block-level merge at the end of a kernel: every wave compacts, wave 0 then
  // absorbs the other waves' buffers (`wave_bytes` apart in LDS) and is the only
  // one left holding a list.  Must be called by all waves of the block.
  __device__ __forceinline__ void merge_block(int wib, int nwaves, uint32_t *s_counts, int lane) {
    compact(lane);
    if (lane == 0) s_counts[wib] = n;
    __syncthreads();
    if (wib == 0)
      for (int w = 1; w < nwaves; ++w) absorb(bk + (size_t)w * 2 * CAP, s_counts[w], lane);  // wave w's keys (its payload follows at + CAP)
  }

  // Writes the k best (unsorted) to keys/pay[0..k), padding with kEmptyKey.
  __device__ __forceinline__ void store(uint64_t *keys, Payload *pay, int lane) {
    compact(lane);
    for (uint32_t i = lane; i < k; i += kWave) {
      if (i < n) {
        keys[i] = bk[i];
        const uint64_t p = bp[i];
        Payload q;
        q.row = (uint32_t)p;
        q.raw = __uint_as_float((uint32_t)(p >> 32));
        pay[i] = q;
      } else {
        keys[i] = kEmptyKey;
      }
    }
  }
};

constexpr int kCapSmall = 128;  // k <= 64
constexpr int kCapLarge = 320;  // k <= 256

// distances.rs:92-98 f64_to_f32
__device__ __forceinline__ bool f64_to_f32(double v, float *out) {
  if (isfinite(v) && v >= -(double)FLT_MAX && v <= (double)FLT_MAX) {
    *out = (float)v;
    return true;
  }
  return false;
}

}  // namespace dev
}  // namespace vt
