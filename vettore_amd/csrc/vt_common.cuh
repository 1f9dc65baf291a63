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

// Wave-wide list of the k smallest keys seen so far, spread over the lanes'
// registers: slot = lane + 64*j.  `thr` (wave-uniform) is the largest key in
// the list, i.e. the k-th best; a candidate enters only if key < thr.
template <int R>
struct WaveTopK {
  uint64_t key[R];
  uint32_t row[R];
  float raw[R];
  uint64_t thr;

  __device__ __forceinline__ void init(uint32_t k, int lane) {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      key[j] = (uint32_t)(lane + kWave * j) < k ? kEmptyKey : 0ull;
      row[j] = 0;
      raw[j] = 0.f;
    }
    thr = kEmptyKey;
  }

  // Wave-uniform arguments; precondition ck < thr.
  __device__ __forceinline__ void push(uint64_t ck, uint32_t crow, float craw, int lane) {
    bool has = false;
#pragma unroll
    for (int j = 0; j < R; ++j) has |= (key[j] == thr);
    const uint64_t b = __ballot(has);
    const int owner = __ffsll((unsigned long long)b) - 1;
    if (lane == owner) {
      bool done = false;
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if (!done && key[j] == thr) {
          key[j] = ck;
          row[j] = crow;
          raw[j] = craw;
          done = true;
        }
      }
    }
    uint64_t lm = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) lm = key[j] > lm ? key[j] : lm;
    thr = uniform_u64(wave_max_u64(lm));
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

  __device__ __forceinline__ void store(uint64_t *keys, Payload *pay, uint32_t k, int lane) const {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint32_t slot = lane + kWave * j;
      if (slot < k) {
        keys[slot] = key[j];
        Payload p;
        p.row = row[j];
        p.raw = raw[j];
        pay[slot] = p;
      }
    }
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
