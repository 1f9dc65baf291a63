// vt_prefix_multi.hip -- K1p: K1's arithmetic over the first d coordinates of every row for up to eight
// queries per sweep (gfx950).  PrefixMultiArgs in vt_device.h says what it is for: stage 1 of funnel_search on an
// L2 / dot / L1 / Linf collection (collection.ex:245-260 -> search.rs:56-60 -> distances.rs:42-68) for several
// callers at once.
//
// A prefix is a narrow thing: 64..256 floats of a 768-float row, i.e. 256-B..1-KiB runs at the rows' 3-KiB stride.
// The walk is K6b's (vt_kernels.hip, cosine_scan_multi_kernel): a wave parks a 64-row x 64-float panel in its own
// LDS slice -- sixteen 1-KiB loads (4 rows x 256 B each), the next panel's already on their way -- and then lane r
// walks row r, chunk by chunk.  What differs is the arithmetic, which is K1's to the bit: per 8-float chunk eight
// separately rounded elements (q*x, (q-x)^2, |q-x|; never an FMA), their horizontal combination in the lane order
// of wide::f32x8::reduce_add (`order`: the four orders of vt_scan.cuh's chunk_sum, here inside one lane), then
// acc = acc + chunk sum down the row and the scalar tail one element at a time.  The queries are wave-uniform:
// eight floats per query and chunk through the scalar cache, used as SGPR operands.
// Cost: 16 (dot) to 24 (L2) VALU instructions per query and chunk, 8 queries: 1 024..1 536 per 16-KiB panel and wave
// -- the kernel sits between the VALU and the prefix reads (DESIGN.md 5.3 has the measured rates).
#include "vt_scan.cuh"

namespace vt {

using namespace dev;

namespace {

// K6b's panel is 64 rows x 64 floats, rows 272 B apart in LDS (PANEL = 64, K1p's r04 form).  PANEL = 32 (r05, the
// default): half the panel -- 8 wave loads of 8 rows x 128 B, 9 KB of LDS per wave, half the prefetch registers (109-113
// VGPRs against 173-192).  It was built to put three or four waves on a SIMD where two were; that is SLOWER -- but at the
// same two blocks per CU the narrow panel is 1-7 % ahead everywhere (A.15), so it stays.
constexpr int kPmRows = 64;
template <int PANEL> struct PmShape {
  static constexpr int kPanel = PANEL, kStride = PANEL + 4;
  static constexpr int kLanesPerRow = PANEL / 4, kRowsPerLoad = 64 / kLanesPerRow, kLoads = kPmRows / kRowsPerLoad;
};

template <int OP>
__device__ __forceinline__ float pm_elem(float q, float x) {
  return elem<OP>(0, q, x);
}

// wide::f32x8::reduce_add of one chunk's eight elements e[0..7], all in this lane (vt_scan.cuh chunk_sum: the even
// lane of a pair holds l0..l3, the odd one l4..l7)
template <int OP, int ORDER>
__device__ __forceinline__ float pm_reduce(const float (&e)[8]) {
  if (ORDER == 1)  // AVX: ((l0+l4)+(l2+l6)) + ((l1+l5)+(l3+l7))
    return comb<OP>(0, comb<OP>(0, comb<OP>(0, e[0], e[4]), comb<OP>(0, e[2], e[6])),
                    comb<OP>(0, comb<OP>(0, e[1], e[5]), comb<OP>(0, e[3], e[7])));
  if (ORDER == 2)  // SEQ: (((l0+l1)+l2)+l3) + (((l4+l5)+l6)+l7)
    return comb<OP>(0, comb<OP>(0, comb<OP>(0, comb<OP>(0, e[0], e[1]), e[2]), e[3]),
                    comb<OP>(0, comb<OP>(0, comb<OP>(0, e[4], e[5]), e[6]), e[7]));
  if (ORDER == 3)  // SSE2: ((l0+l2)+(l1+l3)) + ((l4+l6)+(l5+l7))
    return comb<OP>(0, comb<OP>(0, comb<OP>(0, e[0], e[2]), comb<OP>(0, e[1], e[3])),
                    comb<OP>(0, comb<OP>(0, e[4], e[6]), comb<OP>(0, e[5], e[7])));
  // PAIR: ((l0+l1)+(l2+l3)) + ((l4+l5)+(l6+l7))
  return comb<OP>(0, comb<OP>(0, comb<OP>(0, e[0], e[1]), comb<OP>(0, e[2], e[3])),
                  comb<OP>(0, comb<OP>(0, e[4], e[5]), comb<OP>(0, e[6], e[7])));
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// One chunk of one query: acc' = acc (+|max) reduce_add(elements).  x / q as four pairs (l0,l1) (l2,l3) (l4,l5) (l6,l7) --
// the pairs a ds_read_b128 / s_load_dwordx8 deliver.  The SSE2 and AVX orders add pairs to pairs: the elements and the
// first levels of the tree are packed instructions over two ELEMENTS of one query (v_pk_mul_f32 / v_pk_add_f32; left to
// itself the compiler packs over two QUERIES instead and pays for it with a register move per operand pair).
template <int OP, int ORDER>
__device__ __forceinline__ float pm_chunk(float acc, const f32x2 (&x)[4], const f32x2 (&q)[4]) {
  if (OP == OP_L1 && (ORDER == 3 || ORDER == 1)) {
    // |q - x|: packed differences; the absolute values are operand modifiers of the (plain) adds that follow
    f32x2 t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) t[i] = q[i] - x[i];
    auto ab = [](float v) { return __builtin_fabsf(v); };
    if (ORDER == 3)
      return acc + (((ab(t[0].x) + ab(t[1].x)) + (ab(t[0].y) + ab(t[1].y))) + ((ab(t[2].x) + ab(t[3].x)) + (ab(t[2].y) + ab(t[3].y))));
    return acc + (((ab(t[0].x) + ab(t[2].x)) + (ab(t[1].x) + ab(t[3].x))) + ((ab(t[0].y) + ab(t[2].y)) + (ab(t[1].y) + ab(t[3].y))));
  }
  if ((OP == OP_DOT || OP == OP_L2) && (ORDER == 3 || ORDER == 1)) {
    f32x2 e[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (OP == OP_DOT) {
        e[i] = q[i] * x[i];
      } else {
        const f32x2 t = q[i] - x[i];
        e[i] = t * t;
      }
    }
    if (ORDER == 3) {  // ((l0+l2)+(l1+l3)) + ((l4+l6)+(l5+l7))
      const f32x2 a = e[0] + e[1], b = e[2] + e[3];
      return acc + ((a.x + a.y) + (b.x + b.y));
    }
    // AVX: ((l0+l4)+(l2+l6)) + ((l1+l5)+(l3+l7))
    const f32x2 t = (e[0] + e[2]) + (e[1] + e[3]);
    return acc + (t.x + t.y);
  }
  float e[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    e[2 * i] = pm_elem<OP>(q[i].x, x[i].x);
    e[2 * i + 1] = pm_elem<OP>(q[i].y, x[i].y);
  }
  return comb<OP>(0, acc, pm_reduce<OP, ORDER>(e));
}

template <int OP, int ORDER, int PANEL>
__global__ __launch_bounds__(kWavesPerBlock *kWave) void prefix_multi_kernel(const PrefixMultiArgs a) {
  using Sh = PmShape<PANEL>;
  constexpr int kPmPanel = Sh::kPanel, kPmStride = Sh::kStride;
  extern __shared__ __align__(16) float pm_lds[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  typedef const __attribute__((address_space(4))) float *cf_p;
  cf_p qs = (cf_p)(uintptr_t)a.Q;
  const uint32_t qst = a.q_stride;
  float *S = pm_lds + wib * (kPmRows * kPmStride);
  const bool dense = a.sample != nullptr;
  const uint32_t total_waves = gridDim.x * kWavesPerBlock;
  const uint32_t wave_global = blockIdx.x * kWavesPerBlock + wib;
  const uint32_t ntiles_all = (a.n + kPmRows - 1) / kPmRows;
  const uint32_t step = dense ? a.sample_stride : 1u;      // dense: every step-th tile
  const uint32_t ntiles = (ntiles_all + step - 1) / step;   // tiles this launch walks
  const uint32_t npanel = (a.d + kPmPanel - 1) / kPmPanel;
  const uint32_t cfull = a.d / 8, tail = a.d % 8;
  // the thresholds, once, through the scalar cache: read per tile as vector loads they each brought an s_waitcnt vmcnt(0)
  // into the epilogue -- eight times per tile the wave waited for its NEXT tile's sixteen panel loads
  float tauv[kPrefixMultiMax];
#pragma unroll
  for (uint32_t q = 0; q < kPrefixMultiMax; ++q) tauv[q] = INFINITY;
  if (!dense) {
    cf_p tp = (cf_p)(uintptr_t)a.tau;
#pragma unroll
    for (uint32_t q = 0; q < kPrefixMultiMax; ++q)
      if (q < a.nq) tauv[q] = tp[q];
  }
  f32x4 v[Sh::kLoads];
  const int lrow = lane / Sh::kLanesPerRow, lcol = (lane % Sh::kLanesPerRow) * 4;  // this lane's row within a load, its column
  auto issue = [&](uint32_t ti, uint32_t p) {
    const uint32_t t = ti * step;
#pragma unroll
    for (int s = 0; s < Sh::kLoads; ++s) {
      uint32_t r = t * kPmRows + Sh::kRowsPerLoad * s + lrow;
      r = r < a.n ? r : a.n - 1;
      v[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(a.X + (size_t)r * a.stride + p * kPmPanel + lcol));
    }
  };
  if (wave_global < ntiles) issue(wave_global, 0);
  for (uint32_t ti = wave_global; ti < ntiles; ti += total_waves) {
    const uint32_t grow = ti * step * kPmRows + lane;
    const bool valid_row = grow < a.n;
    float acc[kPrefixMultiMax];
#pragma unroll
    for (uint32_t q = 0; q < kPrefixMultiMax; ++q) acc[q] = 0.0f;
    for (uint32_t p = 0; p < npanel; ++p) {
#pragma unroll
      for (int s = 0; s < Sh::kLoads; ++s) *reinterpret_cast<f32x4 *>(S + (Sh::kRowsPerLoad * s + lrow) * kPmStride + lcol) = v[s];
      wave_lds_fence();
      if (p + 1 < npanel) issue(ti, p + 1);
      else if (ti + total_waves < ntiles) issue(ti + total_waves, 0);
      const float *Sr = S + lane * kPmStride;
      const uint32_t c0 = p * (kPmPanel / 8);
      const uint32_t nfull = cfull > c0 ? (cfull - c0 < (uint32_t)(kPmPanel / 8) ? cfull - c0 : (uint32_t)(kPmPanel / 8)) : 0u;
      for (uint32_t c = 0; c < nfull; ++c) {
        const f32x4 xa = *reinterpret_cast<const f32x4 *>(Sr + c * 8);
        const f32x4 xb = *reinterpret_cast<const f32x4 *>(Sr + c * 8 + 4);
        const f32x2 x[4] = {f32x2{xa.x, xa.y}, f32x2{xa.z, xa.w}, f32x2{xb.x, xb.y}, f32x2{xb.z, xb.w}};
        cf_p qp = qs + (size_t)(c0 + c) * 8;
        // (all eight slots, straight-line, four queries' floats at a time: the scalar loads of a half go out together;
        // an unused slot reads a readable row)
#pragma unroll
        for (uint32_t h = 0; h < kPrefixMultiMax; h += 4) {
          f32x2 w[4][4];
#pragma unroll
          for (uint32_t q = 0; q < 4; ++q)
#pragma unroll
            for (int l = 0; l < 4; ++l) w[q][l] = f32x2{qp[(size_t)(h + q) * qst + 2 * l], qp[(size_t)(h + q) * qst + 2 * l + 1]};
#pragma unroll
          for (uint32_t q = 0; q < 4; ++q) acc[h + q] = pm_chunk<OP, ORDER>(acc[h + q], x, w[q]);
        }
      }
      // the scalar tail: the chunk after the last full one, element by element (distances.rs:212-216 and its siblings)
      if (tail && cfull >= c0 && cfull < c0 + (uint32_t)(kPmPanel / 8)) {
        const uint32_t cl = cfull - c0;
        cf_p qp = qs + (size_t)cfull * 8;
        for (uint32_t j = 0; j < tail; ++j) {
          const float xv = Sr[cl * 8 + j];
#pragma unroll
          for (uint32_t q = 0; q < kPrefixMultiMax; ++q) acc[q] = comb<OP>(0, acc[q], pm_elem<OP>(qp[(size_t)q * qst + j], xv));
        }
      }
      wave_lds_fence();
    }
    // distances.rs:42-68 compute(): value, finiteness, f64 recovery; :113-119 rank_value
    const int metric = a.metric;
#pragma unroll
    for (uint32_t q = 0; q < kPrefixMultiMax; ++q) {
      if (q >= a.nq) break;
      float raw = acc[q];
      if (metric == M_NIP) raw = -raw;
      else if (metric == M_L2) raw = finite_f32(raw) ? __builtin_sqrtf(raw) : raw;
      bool valid = valid_row;
      if (valid && !finite_f32(raw)) {
        raw = recover_overflow(metric, a.Q + (size_t)q * qst, a.X + (size_t)grow * a.stride, a.d);
        if (raw != raw) {
          if (!dense) atomicMax(a.status, kErrOverflow);
          valid = false;
        }
      }
      const float rank = metric == M_IP ? -raw : raw;
      const float good = -rank;  // larger = better: the order the sample's threshold is taken in
      if (dense) {
        if (a.sample_maxima) {  // the tile's best score, one value per query and tile
          float m = valid ? good : -INFINITY;
#pragma unroll
          for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
          if (lane == 0 && ti < a.sample_rows) a.sample[(size_t)q * a.sample_rows + ti] = m;
          continue;
        }
        const uint32_t i = ti * kPmRows + lane;  // position in the sample
        if (i < a.sample_rows) a.sample[(size_t)q * a.sample_rows + i] = valid ? good : -INFINITY;
        continue;
      }
      const bool hit = valid && good >= tauv[q];
      const uint64_t m = __ballot(hit);
      if (m) {
        uint32_t base = 0;
        if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&a.cand_count[q], (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, (int)__builtin_ctzll(m), kWave);
        const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1));
        if (hit && pos < a.cand_cap) {
          // (the id rank only here, in the rare lanes that list a row: a load per tile would be one more wait on vmcnt)
          const uint32_t my_rank = a.id_rank ? a.id_rank[grow] : grow;
          a.cand_keys[(size_t)q * a.cand_cap + pos] = ((uint64_t)orderable(rank) << 32) | my_rank;
          Payload pv;
          pv.row = grow;
          pv.raw = raw;
          a.cand_pay[(size_t)q * a.cand_cap + pos] = pv;
        }
      }
    }
  }
}

// (r05: 32-float panels -- alternating in one process they were 1-7 % ahead of the 64-float ones of r04 at every prefix
// length and metric, 7 % at full width under dot: DESIGN_APPENDIX A.15; r06: the 64-float build has left the library)
constexpr int kPmPanel = 32;

template <int OP, int ORDER>
hipError_t launch_pm(const PrefixMultiArgs &a, uint32_t blocks, size_t lds, hipStream_t s) {
  hipError_t e = allow_lds(prefix_multi_kernel<OP, ORDER, kPmPanel>, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((prefix_multi_kernel<OP, ORDER, kPmPanel>), dim3(blocks), dim3(kWavesPerBlock * kWave), lds, s, a);
  return hipGetLastError();
}

template <int OP>
hipError_t launch_pm_op(const PrefixMultiArgs &a, uint32_t blocks, size_t lds, hipStream_t s) {
  switch (a.order) {
    case 1: return launch_pm<OP, 1>(a, blocks, lds, s);
    case 2: return launch_pm<OP, 2>(a, blocks, lds, s);
    case 3: return launch_pm<OP, 3>(a, blocks, lds, s);
    default: return launch_pm<OP, 0>(a, blocks, lds, s);
  }
}

}  // namespace

bool prefix_multi_supports(int metric) {
  return metric == M_L2 || metric == M_L2SQ || metric == M_IP || metric == M_NIP || metric == M_L1 || metric == M_LINF;
}

size_t prefix_multi_lds_bytes() { return (size_t)kWavesPerBlock * kPmRows * (kPmPanel + 4) * sizeof(float); }
int prefix_multi_blocks_per_cu() { return 2; }  // (three or four blocks of the narrow panel fit a CU, and are slower: A.15)

hipError_t launch_prefix_multi(const PrefixMultiArgs &a, uint32_t blocks, hipStream_t s) {
  if (!prefix_multi_supports(a.metric) || !a.Q || ((uintptr_t)a.Q & 31) || a.q_stride % 8 != 0 || a.q_stride < a.d || a.nq == 0 ||
      a.nq > kPrefixMultiMax || a.n == 0 || a.d == 0 || a.stride < padded_dim(a.d) || a.order < 0 || a.order > 3)
    return hipErrorInvalidValue;
  if (a.sample ? (a.sample_stride == 0 || a.sample_rows == 0) : (!a.tau || !a.cand_keys || !a.cand_pay || !a.cand_count || !a.status))
    return hipErrorInvalidValue;
  const size_t lds = prefix_multi_lds_bytes();
  switch (metric_op(a.metric)) {
    case OP_DOT: return launch_pm_op<OP_DOT>(a, blocks, lds, s);
    case OP_L2: return launch_pm_op<OP_L2>(a, blocks, lds, s);
    case OP_L1: return launch_pm_op<OP_L1>(a, blocks, lds, s);
    default: return launch_pm_op<OP_LINF>(a, blocks, lds, s);
  }
}

}  // namespace vt
