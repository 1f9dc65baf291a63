// vt_kernels.hip -- gfx950 kernels around the scan: K3 radix-select merge, K4
// packed-Hamming scan, K5 sign packing, K6 cosine rerank, K7 normalisation and
// the ingest helpers; plus the scan dispatch.  Written for gfx950 only.
#include "vt_scan.cuh"

namespace vt {

using namespace dev;

namespace {

// ---------------------------------------------------------------------------
// K3: top-k of the partial lists, sorted ascending.  Replaces `hits.sort()`
// (flat.rs:120-121, search.rs:107-110) and the cross-wave merge the reference's
// single heap never needed.  One 1024-thread block: MSD radix select on the u64
// keys (8-bit digits, starting at the highest bit in which the keys differ),
// then compaction of the k winners and a rank sort in LDS.  Keys are unique
// (id_rank is unique per row), so "<= threshold" selects exactly k.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void select_topk_kernel(const uint64_t *__restrict__ keys,
                                                           const Payload *__restrict__ pay, uint32_t m, uint32_t k,
                                                           uint64_t lo_key, int has_lo, int *dev_status,
                                                           ResultBlock *out) {
  extern __shared__ __align__(16) unsigned char smem[];
  uint64_t *sel_key = reinterpret_cast<uint64_t *>(smem);         // [k]
  uint32_t *sel_idx = reinterpret_cast<uint32_t *>(sel_key + k);  // [k]
  __shared__ uint32_t hist[256];
  __shared__ uint64_t red_min[16], red_max[16];
  __shared__ uint32_t red_cnt[16];
  __shared__ uint32_t s_bin, s_below, s_bincount, s_sel;
  const uint32_t tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;

  auto live = [&](uint64_t key) { return key != kEmptyKey && (!has_lo || key > lo_key); };

  // pass A: range and count of the live keys
  uint64_t mn = ~0ull, mx = 0;
  uint32_t cnt = 0;
  for (uint32_t i = tid; i < m; i += 1024) {
    const uint64_t key = keys[i];
    if (live(key)) {
      mn = key < mn ? key : mn;
      mx = key > mx ? key : mx;
      cnt += 1;
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const uint64_t a = __shfl_xor(mn, o, kWave), b = __shfl_xor(mx, o, kWave);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
    cnt += __shfl_xor(cnt, o, kWave);
  }
  if (lane == 0) {
    red_min[wave] = mn;
    red_max[wave] = mx;
    red_cnt[wave] = cnt;
  }
  if (tid == 0) s_sel = 0;
  __syncthreads();
  mn = ~0ull;
  mx = 0;
  uint32_t nvalid = 0;
  for (int w = 0; w < 16; ++w) {
    mn = red_min[w] < mn ? red_min[w] : mn;
    mx = red_max[w] > mx ? red_max[w] : mx;
    nvalid += red_cnt[w];
  }

  uint64_t T = ~0ull - 1;  // select every live key
  if (nvalid > k && mn != mx) {
    uint32_t krem = k;
    int hb = 63 - __clzll((long long)(mn ^ mx));
    uint64_t mask = hb == 63 ? 0ull : (~0ull << (hb + 1));
    uint64_t prefix = mx & mask;
    for (;;) {
      const int width = hb + 1 < 8 ? hb + 1 : 8;
      const int shift = hb + 1 - width;
      const uint32_t dmask = (1u << width) - 1;
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (uint32_t i = tid; i < m; i += 1024) {
        const uint64_t key = keys[i];
        if (live(key) && (key & mask) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & dmask], 1u);
      }
      __syncthreads();
      if (wave == 0) {
        // bins 4*lane .. 4*lane+3; inclusive scan over lanes
        const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
        const uint32_t mine = h0 + h1 + h2 + h3;
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
          const uint32_t t = __shfl_up(incl, o, kWave);
          if (lane >= o) incl += t;
        }
        const uint32_t excl = incl - mine;
        if (excl < krem && krem <= incl) {
          uint32_t below = excl, b = 4 * lane, c = h0;
          if (below + c < krem) {
            below += c; b += 1; c = h1;
            if (below + c < krem) {
              below += c; b += 1; c = h2;
              if (below + c < krem) { below += c; b += 1; c = h3; }
            }
          }
          s_bin = b;
          s_below = below;
          s_bincount = c;
        }
      }
      __syncthreads();
      const uint32_t b = s_bin, bincount = s_bincount;
      krem -= s_below;
      prefix |= (uint64_t)b << shift;
      mask |= (uint64_t)dmask << shift;
      if (bincount == krem || shift == 0) {  // the whole bin is selected / exact key reached
        T = prefix | (shift ? ((1ull << shift) - 1) : 0ull);
        break;
      }
      hb = shift - 1;
    }
  }

  // compaction of the winners
  for (uint32_t i = tid; i < m; i += 1024) {
    const uint64_t key = keys[i];
    if (live(key) && key <= T) {
      const uint32_t pos = atomicAdd(&s_sel, 1u);
      if (pos < k) {
        sel_key[pos] = key;
        sel_idx[pos] = i;
      }
    }
  }
  __syncthreads();
  const uint32_t nsel = s_sel < k ? s_sel : k;
  // rank sort (keys distinct; ties only for caller-supplied duplicate ids)
  for (uint32_t j = tid; j < nsel; j += 1024) {
    const uint64_t kj = sel_key[j];
    uint32_t pos = 0;
    for (uint32_t x = 0; x < nsel; ++x) {
      const uint64_t kx = sel_key[x];
      pos += (kx < kj || (kx == kj && x < j)) ? 1u : 0u;
    }
    const Payload p = pay[sel_idx[j]];
    Entry e;
    e.key = kj;
    e.row = p.row;
    e.raw = p.raw;
    out->e[pos] = e;
  }
  if (tid == 0) {
    out->count = nsel;
    out->status = *dev_status;
    *dev_status = 0;
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
  const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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
    wave_lds_fence();
    const uint32_t grow = t * kWave + lane;
    uint32_t ham = 0;
    for (uint32_t i = 0; i < W; ++i) ham += cnt[lane * W + i];
    wave_lds_fence();
    bool valid = grow < a.n;
    const uint32_t my_rank = (valid && a.id_rank) ? a.id_rank[grow] : grow;
    const float raw = (float)ham;  // distance as f32 (distances.rs:436)
    const uint64_t key = ((uint64_t)orderable(raw) << 32) | my_rank;
    if (a.has_lo) valid = valid && key > a.lo_key;
    tk.offer(valid, key, grow, raw, lane);
  }
  tk.store(a.part_keys + (size_t)wave_global * a.k, a.part_pay + (size_t)wave_global * a.k, a.k, lane);
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
  const uint32_t src = a.gather ? a.gather[(size_t)i * a.gather_stride] : i;
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
  const uint32_t rk = a.id_rank ? a.id_rank[src] : src;
  a.out_keys[i] = ok ? (((uint64_t)orderable(1.0f - raw) << 32) | rk) : kEmptyKey;
  Payload p;
  p.row = src;
  p.raw = raw;
  a.out_pay[i] = p;
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

size_t scan_lds_for(const ScanShape &p) {
  return ((size_t)p.ld + (size_t)kWavesPerBlock * kTileRows * p.ss) * sizeof(float);
}

}  // namespace

size_t scan_lds_bytes(uint32_t d) {
  ScanShape p;
  if (!make_scan_shape(d, 1, &p)) return 0;
  const size_t bytes = scan_lds_for(p);
  return bytes <= kMaxLds ? bytes : 0;
}

hipError_t launch_scan(const ScanArgs &a, uint32_t blocks, hipStream_t s) {
  ScanDev sd;
  sd.a = a;
  if (!make_scan_shape(a.d, a.n, &sd.p)) return hipErrorInvalidValue;
  const size_t lds = scan_lds_for(sd.p);
  if (lds > kMaxLds || a.k == 0 || a.k > (uint32_t)kMaxFusedK || a.stride < sd.p.ld || a.stride % 4 != 0)
    return hipErrorInvalidValue;
  if (a.gather != nullptr) return launch_scan_general(sd, blocks, lds, s);
  const bool padded = (a.d % kRowAlign) != 0;
  switch (metric_op(a.metric)) {
    case OP_DOT: return launch_scan_dot(sd, blocks, lds, padded, s);
    case OP_L2: return launch_scan_l2(sd, blocks, lds, padded, s);
    default: return launch_scan_misc(sd, blocks, lds, padded, s);
  }
}

hipError_t launch_select(const uint64_t *keys, const Payload *pay, uint32_t m, uint32_t k, uint64_t lo_key, int has_lo,
                         int *dev_status, ResultBlock *out, hipStream_t s) {
  if (k == 0 || k > (uint32_t)kMaxFusedK) return hipErrorInvalidValue;
  const size_t lds = (size_t)k * 12;
  hipLaunchKernelGGL(select_topk_kernel, dim3(1), dim3(1024), lds, s, keys, pay, m, k, lo_key, has_lo, dev_status, out);
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
