// vt_common.cuh -- device helpers shared by the gfx950 kernels (wave64 only).
#pragma once
#include "vt_device.h"

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

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    uint64_t t = __shfl_xor(v, o, kWave);
    v = t > v ? t : v;
  }
  return v;
}

__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int src) {
  uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, src);
  uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}

// Wave-wide buffer holding (at least) the k smallest keys seen so far, spread
// over the lanes' registers: slot (lane, j), 64*R slots, k < 64*R.
//  * `thr` (wave-uniform) is an upper bound: only keys < thr can still be among
//    the k smallest.  It is the exact k-th smallest as of the last compaction.
//  * A candidate that passes is written into any free slot (a wave-uniform
//    free-slot bitmask per register row): ~20 instructions, no reduction.
//  * When no slot is free, compact(): a 64-step bitwise search finds the k-th
//    smallest key of the buffer, everything above it is dropped, thr tightens.
//    One compaction buys 64*R - k cheap insertions.
template <int R>
struct WaveTopK {
  uint64_t key[R];
  uint32_t row[R];
  float raw[R];
  uint64_t freem[R];  // wave-uniform
  uint64_t thr;       // wave-uniform
  uint32_t k;

  __device__ __forceinline__ void init(uint32_t k_, int lane) {
    (void)lane;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      key[j] = kEmptyKey;
      row[j] = 0;
      raw[j] = 0.f;
      freem[j] = ~0ull;
    }
    thr = kEmptyKey;
    k = k_;
  }

  // Keeps the k smallest keys, frees every other slot, tightens thr.
  __device__ __forceinline__ void compact() {
    uint32_t live = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) live += __popcll(__ballot(key[j] != kEmptyKey));
    if (live > k) {
      uint64_t T = 0;  // k-th smallest: the largest T with count(key < T) < k
      for (int b = 63; b >= 0; --b) {
        const uint64_t c = T | (1ull << b);
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < R; ++j) cnt += __popcll(__ballot(key[j] < c));
        if (cnt < k) T = c;
      }
#pragma unroll
      for (int j = 0; j < R; ++j)
        if (key[j] > T) key[j] = kEmptyKey;
      thr = T;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) freem[j] = __ballot(key[j] == kEmptyKey);
  }

  // Wave-uniform arguments; precondition ck < thr.
  __device__ __forceinline__ void push(uint64_t ck, uint32_t crow, float craw, int lane) {
    bool any = false;
#pragma unroll
    for (int j = 0; j < R; ++j) any |= freem[j] != 0;
    if (!any) {
      compact();
      if (!(ck < thr)) return;
    }
    bool done = false;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (!done && freem[j] != 0) {
        const int l = __ffsll((unsigned long long)freem[j]) - 1;
        freem[j] &= freem[j] - 1;
        if (lane == l) {
          key[j] = ck;
          row[j] = crow;
          raw[j] = craw;
        }
        done = true;
      }
    }
  }

  // Offers one candidate per lane (valid lanes only).
  __device__ __forceinline__ void offer(bool valid, uint64_t ck, uint32_t crow, float craw, int lane) {
    uint64_t m = __ballot(valid && ck < thr);
    while (m) {
      const int src = __ffsll((unsigned long long)m) - 1;
      m &= m - 1;
      const uint64_t k2 = readlane_u64(ck, src);
      if (k2 < thr) {
        const uint32_t r2 = __builtin_amdgcn_readlane(crow, src);
        const float f2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(craw), src));
        push(k2, r2, f2, lane);
      }
    }
  }

  // Writes the k best (unsorted) to keys/pay[0..k), padding with kEmptyKey.
  __device__ __forceinline__ void store(uint64_t *keys, Payload *pay, int lane) {
    compact();
    uint32_t base = 0;
    const uint64_t lt = (1ull << lane) - 1;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const bool has = key[j] != kEmptyKey;
      const uint64_t m = __ballot(has);
      const uint32_t pos = base + __popcll(m & lt);
      if (has && pos < k) {
        keys[pos] = key[j];
        Payload p;
        p.row = row[j];
        p.raw = raw[j];
        pay[pos] = p;
      }
      base += __popcll(m);
    }
    for (uint32_t i = base + lane; i < k; i += kWave) keys[i] = kEmptyKey;
  }
};

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
