// vt_device.hip -- gfx950 (MI355X, CDNA4, wave64) kernels for the Vettore flat
// hot path.  Written for this target only; compile with
//   hipcc --offload-arch=gfx950 -ffp-contract=off
//
// Arithmetic contract: every raw metric value is computed with exactly the
// reference's f32 operation order (native/vettore/src/distances.rs:197-308):
// per 8-float chunk eight separately rounded products (no FMA), one horizontal
// add in the lane order of wide::f32x8::reduce_add, then `acc += chunk_sum`
// sequentially over the chunks, then the scalar tail.  The result is therefore
// bit-identical to the CPU oracle (oracle/vt_oracle.c) for the selected order,
// not merely within tolerance.
//
// How that is made HBM-bound (K1, scan_topk_kernel):
//   * the corpus is one row-major slab; a wave reads a tile of 32 rows as a run
//     of fully coalesced 1-KiB wave loads (16 B per lane), kept kU deep in
//     flight in a register ring (nontemporal: the slab never fits L2/MALL);
//   * the query sits in LDS; each lane pair holds one 8-float chunk, so the
//     chunk sum costs 4 v_mul + 3..7 v_add, one of them a DPP quad_perm add;
//   * chunk sums go to a per-wave LDS panel S[32 rows][chunks] (row stride
//     4*odd dwords: conflict-free ds_read_b128); then lane r walks row r's
//     chunk sums in order -- the reference's sequential `acc +=` chain -- so 32
//     rows are chained in parallel, once per tile (~1% of the tile's time);
//   * each lane then owns one finished row: rank key, compare against the
//     wave's current k-th best (wave-uniform threshold); rows that pass are
//     rare and are inserted one by one into a register-resident wave-wide list.
#include "vt_device.h"

#include <float.h>

// rustc never contracts a*b+c; neither may this file, whatever the command line says.
#pragma clang fp contract(off)

namespace vt {

namespace {

constexpr int kWave = 64;
constexpr int kU = 8;  // 1-KiB loads in flight per wave

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

enum { OP_DOT = 0, OP_L2 = 1, OP_L1 = 2, OP_LINF = 3, OP_HAM = 4, OP_JAC = 5 };
enum { M_L2 = 0, M_L2SQ = 1, M_COS = 2, M_IP = 3, M_NIP = 4, M_L1 = 5, M_LINF = 6, M_HAM = 7, M_JAC = 8 };
constexpr int kErrOverflow = 4;  // VT_ERR_OVERFLOW

__host__ __device__ inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// Row layout of the per-wave LDS panel.
struct PanelShape {
  uint32_t ld;         // padded row length in floats (multiple of 8)
  uint32_t cfull;      // full 8-float chunks per row (d / 8)
  uint32_t tail;       // d % 8
  uint32_t tail_base;  // dword offset of the 8 tail-product slots
  uint32_t ss;         // dwords per panel row, 4 * odd
};
__host__ __device__ inline PanelShape panel_shape(uint32_t d) {
  PanelShape p;
  p.ld = round_up(d, 8);
  p.cfull = d / 8;
  p.tail = d % 8;
  p.tail_base = round_up(p.cfull, 4);
  uint32_t need = p.tail ? p.tail_base + 8 : (p.cfull ? p.cfull : 1);
  uint32_t ss = round_up(need, 4);
  if (((ss >> 2) & 1u) == 0) ss += 4;
  p.ss = ss;
  return p;
}

__device__ __forceinline__ int metric_op(int metric) {
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

  __device__ __forceinline__ void store(Entry *dst, uint32_t k, int lane) const {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint32_t slot = lane + kWave * j;
      if (slot < k) {
        Entry e;
        e.key = key[j];
        e.row = row[j];
        e.raw = raw[j];
        dst[slot] = e;
      }
    }
  }
};

// ---- per-element operation of each metric family ---------------------------
template <int OP>
__device__ __forceinline__ float elem(int op_rt, float q, float x) {
  const int op = OP >= 0 ? OP : op_rt;
  switch (op) {
    case OP_DOT: return __fmul_rn(q, x);
    case OP_L2: {
      const float t = __fsub_rn(q, x);
      return __fmul_rn(t, t);
    }
    case OP_L1:
    case OP_LINF: return fabsf(__fsub_rn(q, x));
    case OP_HAM: return ((q != 0.0f) != (x != 0.0f)) ? 1.0f : 0.0f;
    default:  // OP_JAC: hamming count + 4096 * (x != 0); exact in f32 for d < 4096
      return (((q != 0.0f) != (x != 0.0f)) ? 1.0f : 0.0f) + ((x != 0.0f) ? 4096.0f : 0.0f);
  }
}

template <int OP>
__device__ __forceinline__ float comb(int op_rt, float a, float b) {
  const int op = OP >= 0 ? OP : op_rt;
  return op == OP_LINF ? fmaxf(a, b) : __fadd_rn(a, b);
}

// wide::f32x8::reduce_add of the chunk held by a lane pair (even lane: l0..l3,
// odd lane: l4..l7); both lanes return the chunk sum.
template <int OP, int ORDER>
__device__ __forceinline__ float chunk_sum(int op_rt, int order_rt, float p0, float p1, float p2, float p3, int odd) {
  const int order = ORDER >= 0 ? ORDER : order_rt;
  if (order == 1) {  // AVX: ((l0+l4)+(l2+l6)) + ((l1+l5)+(l3+l7))
    const float u0 = comb<OP>(op_rt, p0, dpp_xor1(p0));
    const float u1 = comb<OP>(op_rt, p1, dpp_xor1(p1));
    const float u2 = comb<OP>(op_rt, p2, dpp_xor1(p2));
    const float u3 = comb<OP>(op_rt, p3, dpp_xor1(p3));
    const float t = comb<OP>(op_rt, odd ? u1 : u0, odd ? u3 : u2);
    return comb<OP>(op_rt, t, dpp_xor1(t));
  }
  float e;
  if (order == 2)  // SEQ: (((l0+l1)+l2)+l3) + (((l4+l5)+l6)+l7)
    e = comb<OP>(op_rt, comb<OP>(op_rt, comb<OP>(op_rt, p0, p1), p2), p3);
  else  // PAIR: ((l0+l1)+(l2+l3)) + ((l4+l5)+(l6+l7))
    e = comb<OP>(op_rt, comb<OP>(op_rt, p0, p1), comb<OP>(op_rt, p2, p3));
  return comb<OP>(op_rt, e, dpp_xor1(e));
}

// distances.rs:92-98 f64_to_f32
__device__ __forceinline__ bool f64_to_f32(double v, float *out) {
  if (isfinite(v) && v >= -(double)FLT_MAX && v <= (double)FLT_MAX) {
    *out = (float)v;
    return true;
  }
  return false;
}

// distances.rs:70-90 recover_metric_overflow (+ the f64 branch of l2(),
// distances.rs:140-147), run by the one lane whose f32 result was non-finite.
__device__ __noinline__ bool recover_overflow(int metric, const float *q, const float *x, uint32_t d, float *out) {
  double acc = 0.0;
  switch (metric) {
    case M_L2:
    case M_L2SQ:
      for (uint32_t i = 0; i < d; ++i) {
        const double t = (double)q[i] - (double)x[i];
        acc += t * t;
      }
      if (metric == M_L2) {
        // l2(): (f64 sqrt) as f32, accepted when finite; the later
        // recover path reaches the same value or fails identically.
        const float v = (float)sqrt(acc);
        if (finite_f32(v)) {
          *out = v;
          return true;
        }
        return false;
      }
      return f64_to_f32(acc, out);
    case M_COS:
    case M_IP:
    case M_NIP:
      for (uint32_t i = 0; i < d; ++i) acc += (double)q[i] * (double)x[i];
      return f64_to_f32(metric == M_NIP ? -acc : acc, out);
    case M_L1:
      for (uint32_t i = 0; i < d; ++i) acc += fabs((double)q[i] - (double)x[i]);
      return f64_to_f32(acc, out);
    case M_LINF:
      for (uint32_t i = 0; i < d; ++i) acc = fmax(acc, fabs((double)q[i] - (double)x[i]));
      return f64_to_f32(acc, out);
    default: return false;
  }
}

struct ScanDev {
  ScanArgs a;
  PanelShape p;
  uint32_t ntiles;
  uint32_t nseg;  // 1-KiB segments per tile = kTileRows * ld / 256
};

// ---------------------------------------------------------------------------
// K1: flat scan + fused wave64 top-k.  Replaces the hot loop of
// FlatIndex::search (flat.rs:104-118) and of vector_top_k (search.rs:49-70).
// OP / ORDER < 0: taken from the arguments at run time (generic build).
// GENERAL: rows addressed through `gather` and/or stride != ld (prefix scan).
// ALIGNED: segments per tile is a multiple of kU, so the load ring needs no
// guards and every load is unconditional (clamped at the end of the stream).
// TAIL: d % 8 may be non-zero (per-segment tail-chunk test compiled in).
// ---------------------------------------------------------------------------
template <int OP, int ORDER, int R, bool GENERAL, bool ALIGNED, bool TAIL>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void scan_topk_kernel(const ScanDev sd) {
  extern __shared__ __align__(16) float lds[];
  const ScanArgs &a = sd.a;
  const PanelShape &p = sd.p;
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = threadIdx.x >> 6;
  const int odd = lane & 1;
  float *qs = lds;
  float *S = lds + p.ld + wib * (kTileRows * p.ss);

  for (uint32_t i = threadIdx.x; i < p.ld; i += blockDim.x) qs[i] = a.q[i];
  __syncthreads();

  const int op_rt = metric_op(a.metric);
  const uint32_t total_waves = gridDim.x * kWavesPerBlock;
  const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wib;
  const uint32_t nseg = sd.nseg;
  const uint32_t ntiles = sd.ntiles;

  WaveTopK<R> tk;
  tk.init(a.k, lane);

  if (wave_global < ntiles) {
    // last tile this wave owns: loads past it are clamped onto it
    const uint32_t last_tile = wave_global + ((ntiles - 1 - wave_global) / total_waves) * total_waves;
    const uint32_t tile_f4 = kTileRows * p.ld / 4;  // float4 per tile (contiguous mode)
    const f32x4 *X4 = reinterpret_cast<const f32x4 *>(a.X);

    // address of segment `s` of tile `t` for this lane
    auto seg_ptr = [&](uint32_t t, uint32_t s) -> const f32x4 * {
      if (!GENERAL) return X4 + (size_t)t * tile_f4 + (size_t)s * kWave + lane;
      const uint32_t f = s * 256u + (uint32_t)lane * 4u;
      const uint32_t ri = f / p.ld, col = f - ri * p.ld;
      const uint32_t gi = t * kTileRows + ri;
      uint32_t src = 0;
      if (gi < a.n) src = a.gather ? a.gather[(size_t)gi * a.gather_stride] : gi;
      return reinterpret_cast<const f32x4 *>(a.X + (size_t)src * a.stride + col);
    };

    f32x4 buf[kU];
    constexpr bool aligned = ALIGNED;
    // ring slot u holds the next segment congruent to u (mod kU) of the stream
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      if (aligned || (uint32_t)u < nseg) buf[u] = __builtin_nontemporal_load(seg_ptr(wave_global, u));
    }

    for (uint32_t t = wave_global; t < ntiles; t += total_waves) {
      const uint32_t grow = t * kTileRows + lane;  // lanes 0..31 own a row
      const bool row_valid = lane < kTileRows && grow < a.n;
      uint32_t src_row = grow;
      if (GENERAL && row_valid && a.gather) src_row = a.gather[(size_t)grow * a.gather_stride];
      uint32_t my_rank = src_row;
      if (row_valid && a.id_rank) my_rank = a.id_rank[src_row];

      // this lane's (row in tile, column) for segment 0
      uint32_t rowi = ((uint32_t)lane * 4u) / p.ld;
      uint32_t col = (uint32_t)lane * 4u - rowi * p.ld;
      const uint32_t step_r = 256u / p.ld, step_c = 256u - step_r * p.ld;

      for (uint32_t s = 0; s < nseg; s += kU) {
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (aligned || s + u < nseg) {
            const f32x4 x = buf[u];
            // refill slot u with the next segment it will serve
            uint32_t ns = s + u + kU, nt = t;
            if (ns >= nseg) {
              ns = u;
              nt = t + total_waves;
            }
            if (aligned) {
              nt = nt < last_tile ? nt : last_tile;
              buf[u] = __builtin_nontemporal_load(seg_ptr(nt, ns));
            } else if (ns < nseg && nt < ntiles) {
              buf[u] = __builtin_nontemporal_load(seg_ptr(nt, ns));
            }

            const f32x4 qv = *reinterpret_cast<const f32x4 *>(qs + col);
            const float p0 = elem<OP>(op_rt, qv.x, x.x);
            const float p1 = elem<OP>(op_rt, qv.y, x.y);
            const float p2 = elem<OP>(op_rt, qv.z, x.z);
            const float p3 = elem<OP>(op_rt, qv.w, x.w);
            const uint32_t c = col >> 3;
            float *Srow = S + rowi * p.ss;
            if (!TAIL || c < p.cfull) {
              const float sum = chunk_sum<OP, ORDER>(op_rt, a.order, p0, p1, p2, p3, odd);
              if (!odd) Srow[c] = sum;
            } else {
              // tail chunk: the reference adds these products one by one
              *reinterpret_cast<f32x4 *>(Srow + p.tail_base + odd * 4) = f32x4{p0, p1, p2, p3};
            }
            rowi += step_r;
            col += step_c;
            if (col >= p.ld) {
              col -= p.ld;
              rowi += 1;
            }
          }
        }
      }

      // panel complete: hand rows to lanes (wave-local LDS exchange)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      float acc = 0.0f;
      if (lane < kTileRows) {
        const float *Sr = S + lane * p.ss;
        uint32_t c = 0;
        for (; c + 4 <= p.cfull; c += 4) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(Sr + c);
          acc = comb<OP>(op_rt, acc, v.x);
          acc = comb<OP>(op_rt, acc, v.y);
          acc = comb<OP>(op_rt, acc, v.z);
          acc = comb<OP>(op_rt, acc, v.w);
        }
        for (; c < p.cfull; ++c) acc = comb<OP>(op_rt, acc, Sr[c]);
        for (uint32_t j = 0; j < p.tail; ++j) acc = comb<OP>(op_rt, acc, Sr[p.tail_base + j]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // distances.rs:42-68 compute(): value, finiteness, f64 recovery
      float raw = acc;
      const int metric = a.metric;
      if (metric == M_NIP) raw = -acc;
      else if (metric == M_L2) raw = finite_f32(acc) ? __builtin_sqrtf(acc) : acc;
      else if (metric == M_HAM) raw = acc;
      else if (metric == M_JAC) {
        const uint32_t tot = (uint32_t)acc;
        const uint32_t xnz = tot >> 12, ham = tot & 4095u;
        const uint32_t uni = (a.q_nonzero + xnz + ham) >> 1;
        const uint32_t inter = (a.q_nonzero + xnz - ham) >> 1;
        raw = uni == 0 ? 0.0f : __fsub_rn(1.0f, __fdiv_rn((float)inter, (float)uni));
      }
      bool valid = row_valid;
      if (valid && !finite_f32(raw)) {
        float rec;
        if (recover_overflow(metric, qs, a.X + (size_t)src_row * a.stride, a.d, &rec)) {
          raw = rec;
        } else {
          atomicMax(a.status, kErrOverflow);
          valid = false;
        }
      }
      // distances.rs:113-119 rank_value, flat.rs:34-40 ordering
      float rank = raw;
      if (metric == M_COS) rank = __fsub_rn(1.0f, raw);
      else if (metric == M_IP) rank = -raw;
      const uint64_t key = ((uint64_t)orderable(rank) << 32) | my_rank;
      if (a.has_lo) valid = valid && key > a.lo_key;
      tk.offer(valid, key, src_row, raw, lane);
    }
  }
  tk.store(a.partial + (size_t)wave_global * a.k, a.k, lane);
}

// ---------------------------------------------------------------------------
// K3: merge of partial lists -> final k, sorted ascending.  Replaces
// `hits.sort()` (flat.rs:120-121, search.rs:107-110).  One 1024-thread block.
// ---------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(1024) void merge_topk_kernel(const Entry *__restrict__ in, uint32_t m, uint32_t k,
                                                          Entry *__restrict__ out, uint32_t *__restrict__ out_count) {
  extern __shared__ __align__(16) unsigned char smem[];
  Entry *stage = reinterpret_cast<Entry *>(smem);  // [16][k]
  Entry *fin = stage + 16 * k;                     // [k]
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;

  WaveTopK<R> tk;
  tk.init(k, lane);
  for (uint32_t base = wave * kWave; base < m; base += 1024) {
    const uint32_t i = base + lane;
    Entry e;
    e.key = kEmptyKey;
    e.row = 0;
    e.raw = 0.f;
    if (i < m) e = in[i];
    tk.offer(i < m, e.key, e.row, e.raw, lane);
  }
  tk.store(stage + wave * k, k, lane);
  __syncthreads();
  if (wave == 0) {
    WaveTopK<R> f;
    f.init(k, lane);
    for (uint32_t base = 0; base < 16 * k; base += kWave) {
      const uint32_t i = base + lane;
      Entry e;
      e.key = kEmptyKey;
      e.row = 0;
      e.raw = 0.f;
      if (i < 16 * k) e = stage[i];
      f.offer(i < 16 * k, e.key, e.row, e.raw, lane);
    }
    f.store(fin, k, lane);
  }
  __syncthreads();
  // rank sort: keys are distinct (id_rank is unique per row)
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
    const Entry e = fin[i];
    if (e.key == kEmptyKey) continue;
    uint32_t pos = 0;
    for (uint32_t j = 0; j < k; ++j) pos += (fin[j].key < e.key || (fin[j].key == e.key && j < i)) ? 1u : 0u;
    out[pos] = e;
  }
  if (threadIdx.x == 0) {
    uint32_t cnt = 0;
    for (uint32_t j = 0; j < k; ++j) cnt += fin[j].key != kEmptyKey ? 1u : 0u;
    *out_count = cnt;
  }
}

// ---------------------------------------------------------------------------
// K4: packed sign-bit Hamming scan + fused top-k.  Replaces binary_top_k
// (search.rs:76-92) + packed_hamming (distances.rs:426-437, word_mask :472-481).
// A wave reads a tile of 64 rows (64*words u64) as coalesced 16-B-per-lane
// loads, counts bits per word, and regroups the per-word counts by row in LDS.
// ---------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void hamming_topk_kernel(const HammingArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = threadIdx.x >> 6;
  const uint32_t W = a.words;
  uint64_t *qs = reinterpret_cast<uint64_t *>(smem);  // [W] query words, [W] masks
  uint64_t *ms = qs + W;
  uint32_t *cnt = reinterpret_cast<uint32_t *>(ms + W) + wib * (kWave * W);  // [64*W] per wave
  const uint32_t rem = a.d % 64;
  for (uint32_t i = threadIdx.x; i < W; i += blockDim.x) {
    qs[i] = a.qbits[i];
    ms[i] = (i + 1 == W && rem != 0) ? ((1ull << rem) - 1) : ~0ull;
  }
  __syncthreads();

  const uint32_t total_waves = gridDim.x * kWavesPerBlock;
  const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wib;
  const uint32_t ntiles = (a.n + kWave - 1) / kWave;
  const uint32_t tile_words = kWave * W;
  const uint32_t nload = (tile_words + 127) / 128;  // 2 words per lane per load
  const uint64_t total_words = (uint64_t)a.n * W;

  WaveTopK<R> tk;
  tk.init(a.k, lane);
  for (uint32_t t = wave_global; t < ntiles; t += total_waves) {
    const uint64_t tile_base = (uint64_t)t * tile_words;
    uint32_t w0 = (2u * lane) % W;  // word index within its row
    const uint32_t stepw = 128u % W;
    for (uint32_t j = 0; j < nload; ++j) {
      const uint32_t f = j * 128u + 2u * lane;  // word offset in tile
      if (f < tile_words) {
        const uint64_t g = tile_base + f;
        uint64_t x0 = 0, x1 = 0;
        if (g + 1 < total_words) {
          const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(a.bits + g));
          x0 = v.x;
          x1 = v.y;
        } else if (g < total_words) {
          x0 = a.bits[g];
        }
        const uint32_t w1 = (w0 + 1 == W) ? 0 : w0 + 1;
        const uint32_t c0 = __popcll((x0 ^ qs[w0]) & ms[w0]);
        const uint32_t c1 = __popcll((x1 ^ qs[w1]) & ms[w1]);
        cnt[f] = c0;
        if (f + 1 < tile_words) cnt[f + 1] = c1;
      }
      w0 += stepw;
      if (w0 >= W) w0 -= W;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t grow = t * kWave + lane;
    uint32_t ham = 0;
    for (uint32_t i = 0; i < W; ++i) ham += cnt[lane * W + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bool valid = grow < a.n;
    const uint32_t my_rank = (valid && a.id_rank) ? a.id_rank[grow] : grow;
    const float raw = (float)ham;  // distance as f32 (distances.rs:436)
    const uint64_t key = ((uint64_t)orderable(raw) << 32) | my_rank;
    if (a.has_lo) valid = valid && key > a.lo_key;
    tk.offer(valid, key, grow, raw, lane);
  }
  tk.store(a.partial + (size_t)wave_global * a.k, a.k, lane);
}

// ---------------------------------------------------------------------------
// K5: sign packing (compress_sign_bits, distances.rs:413-423).  One wave per
// 64 coordinates: lane j tests v[j] >= 0.0, the wave ballot IS the word.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sign_pack_kernel(const float *__restrict__ rows, size_t stride, uint32_t n,
                                                        uint32_t d, uint64_t *__restrict__ bits) {
  const int lane = threadIdx.x & (kWave - 1);
  const uint32_t W = (d + 63) / 64;
  const uint64_t total = (uint64_t)n * W;
  const uint64_t nwaves = (uint64_t)gridDim.x * (blockDim.x / kWave);
  for (uint64_t w = (uint64_t)blockIdx.x * (blockDim.x / kWave) + (threadIdx.x >> 6); w < total; w += nwaves) {
    const uint32_t r = (uint32_t)(w / W), wi = (uint32_t)(w - (uint64_t)r * W);
    const uint32_t j = wi * 64 + lane;
    bool bit = false;
    if (j < d) bit = rows[(size_t)r * stride + j] >= 0.0f;
    const uint64_t word = __ballot(bit);
    if (lane == 0) bits[w] = word;
  }
}

__global__ __launch_bounds__(256) void check_finite_kernel(const float *__restrict__ rows, size_t stride, uint32_t n,
                                                           uint32_t d, int *flag) {
  const uint64_t total = (uint64_t)n * d;
  bool bad = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / d, c = i - r * d;
    bad |= !finite_f32(rows[r * stride + c]);
  }
  if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ __launch_bounds__(256) void pad_rows_kernel(const float *__restrict__ src, uint32_t n, uint32_t d,
                                                       float *__restrict__ dst, size_t dst_stride) {
  const uint64_t total = (uint64_t)n * dst_stride;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = i / dst_stride, c = i - r * dst_stride;
    dst[i] = c < d ? src[r * d + c] : 0.0f;
  }
}

// K6 (cosine part): one candidate per lane, sequential f64 sums in index order
// (distances.rs:179-185 f64_dot), then distances.rs:160-177.
__global__ __launch_bounds__(64) void cosine_rerank_kernel(const CosineRerankArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const uint32_t src = a.gather[(size_t)i * a.gather_stride];
  const float *x = a.X + (size_t)src * a.stride;
  double qq = 0.0, xx = 0.0, qx = 0.0;
  for (uint32_t j = 0; j < a.d; ++j) {
    const double qv = (double)a.q[j], xv = (double)x[j];
    qq += qv * qv;
    xx += xv * xv;
    qx += qv * xv;
  }
  const double ln = sqrt(qq), rn = sqrt(xx);
  float raw = 0.0f;
  bool ok = true;
  if (!(ln == 0.0 || rn == 0.0)) {
    double sim = qx / (ln * rn);
    if (!isfinite(sim)) {
      ok = false;
      atomicMax(a.status, kErrOverflow);
    } else {
      sim = sim < -1.0 ? -1.0 : (sim > 1.0 ? 1.0 : sim);
      raw = (float)sim;
    }
  }
  Entry e;
  e.row = src;
  e.raw = raw;
  const uint32_t rk = a.id_rank ? a.id_rank[src] : src;
  e.key = ok ? (((uint64_t)orderable(__fsub_rn(1.0f, raw)) << 32) | rk) : kEmptyKey;
  a.out[i] = e;
}

// K7: normalize_l2 (distances.rs:350-361), one row per lane.
__global__ __launch_bounds__(64) void normalize_l2_kernel(const float *__restrict__ in, uint32_t n, uint32_t d,
                                                          float *__restrict__ out) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const float *x = in + (size_t)r * d;
  float *y = out + (size_t)r * d;
  double acc = 0.0;
  for (uint32_t j = 0; j < d; ++j) {
    const double v = (double)x[j];
    acc += v * v;
  }
  const double norm = sqrt(acc);
  if (norm == 0.0) {
    for (uint32_t j = 0; j < d; ++j) y[j] = 0.0f;
  } else {
    for (uint32_t j = 0; j < d; ++j) y[j] = (float)((double)x[j] / norm);
  }
}

constexpr size_t kMaxLds = 160 * 1024;

template <typename K>
hipError_t allow_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes);
}

template <int OP, int ORDER, int R, bool GENERAL, bool ALIGNED, bool TAIL>
hipError_t launch_scan_t(const ScanDev &sd, uint32_t blocks, size_t lds, hipStream_t s) {
  auto kern = scan_topk_kernel<OP, ORDER, R, GENERAL, ALIGNED, TAIL>;
  hipError_t e = allow_lds(kern, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, sd);
  return hipGetLastError();
}

}  // namespace

size_t scan_lds_bytes(uint32_t d) {
  if (d == 0) return 0;
  const PanelShape p = panel_shape(d);
  const size_t bytes = ((size_t)p.ld + (size_t)kWavesPerBlock * kTileRows * p.ss) * 4;
  return bytes <= kMaxLds ? bytes : 0;
}

hipError_t launch_scan(const ScanArgs &a, uint32_t blocks, hipStream_t s) {
  ScanDev sd;
  sd.a = a;
  sd.p = panel_shape(a.d);
  sd.ntiles = (a.n + kTileRows - 1) / kTileRows;
  sd.nseg = kTileRows * sd.p.ld / 256;
  const size_t lds = scan_lds_bytes(a.d);
  if (lds == 0 || a.k == 0 || a.k > (uint32_t)kMaxFusedK) return hipErrorInvalidValue;
  const bool general = a.gather != nullptr || a.stride != sd.p.ld;
  const int op = (a.metric == 0 || a.metric == 1) ? OP_L2 : ((a.metric >= 2 && a.metric <= 4) ? OP_DOT : -1);
  if (!general && a.k <= 64 && op >= 0 && sd.nseg % kU == 0 && sd.p.tail == 0) {
    // hot instantiations: operation and lane order resolved at compile time
#define VT_SCAN_CASE(OPV, ORD) \
  if (op == OPV && a.order == ORD) return launch_scan_t<OPV, ORD, 1, false, true, false>(sd, blocks, lds, s);
    VT_SCAN_CASE(OP_DOT, 0)
    VT_SCAN_CASE(OP_DOT, 1)
    VT_SCAN_CASE(OP_DOT, 2)
    VT_SCAN_CASE(OP_L2, 0)
    VT_SCAN_CASE(OP_L2, 1)
    VT_SCAN_CASE(OP_L2, 2)
#undef VT_SCAN_CASE
  }
  if (a.k <= 64) {
    return general ? launch_scan_t<-1, -1, 1, true, false, true>(sd, blocks, lds, s)
                   : launch_scan_t<-1, -1, 1, false, false, true>(sd, blocks, lds, s);
  }
  return general ? launch_scan_t<-1, -1, 4, true, false, true>(sd, blocks, lds, s)
                 : launch_scan_t<-1, -1, 4, false, false, true>(sd, blocks, lds, s);
}

hipError_t launch_merge(const Entry *in, uint32_t m, uint32_t k, Entry *out, uint32_t *out_count, hipStream_t s) {
  if (k == 0 || k > (uint32_t)kMaxFusedK) return hipErrorInvalidValue;
  const size_t lds = (size_t)17 * k * sizeof(Entry);
  if (k <= 64) {
    auto kern = merge_topk_kernel<1>;
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, s, in, m, k, out, out_count);
  } else {
    auto kern = merge_topk_kernel<4>;
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, s, in, m, k, out, out_count);
  }
  return hipGetLastError();
}

size_t hamming_lds_bytes(uint32_t words) {
  const size_t bytes = (size_t)words * 16 + (size_t)kWavesPerBlock * kWave * words * 4;
  return bytes <= kMaxLds ? bytes : 0;
}

hipError_t launch_hamming(const HammingArgs &a, uint32_t blocks, hipStream_t s) {
  const size_t lds = hamming_lds_bytes(a.words);
  if (lds == 0 || a.k == 0 || a.k > (uint32_t)kMaxFusedK) return hipErrorInvalidValue;
  if (a.k <= 64) {
    auto kern = hamming_topk_kernel<1>;
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a);
  } else {
    auto kern = hamming_topk_kernel<4>;
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a);
  }
  return hipGetLastError();
}

hipError_t launch_sign_pack(const float *rows, size_t stride, uint32_t n, uint32_t d, uint64_t *bits, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(sign_pack_kernel, dim3(2048), dim3(256), 0, s, rows, stride, n, d, bits);
  return hipGetLastError();
}

hipError_t launch_check_finite(const float *rows, size_t stride, uint32_t n, uint32_t d, int *flag, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(check_finite_kernel, dim3(2048), dim3(256), 0, s, rows, stride, n, d, flag);
  return hipGetLastError();
}

hipError_t launch_pad_rows(const float *src, uint32_t n, uint32_t d, float *dst, size_t dst_stride, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(pad_rows_kernel, dim3(2048), dim3(256), 0, s, src, n, d, dst, dst_stride);
  return hipGetLastError();
}

hipError_t launch_cosine_rerank(const CosineRerankArgs &a, hipStream_t s) {
  if (a.n == 0) return hipSuccess;
  hipLaunchKernelGGL(cosine_rerank_kernel, dim3((a.n + 63) / 64), dim3(64), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_normalize_l2(const float *in, uint32_t n, uint32_t d, float *out, hipStream_t s) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(normalize_l2_kernel, dim3((n + 63) / 64), dim3(64), 0, s, in, n, d, out);
  return hipGetLastError();
}

}  // namespace vt
