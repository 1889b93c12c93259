// Embedding path on gfx950: fused multi-table lookup, duplicate-row plan (LSD radix sort +
// segment heads), atomic-free segmented gradient reduction, Adam (dense and row-sparse), and the
// device-side batch gather.  All kernels are HBM-bound byte movers: 64-wide waves, 16-byte lanes,
// grids capped at 8 workgroups per CU with grid-stride loops.
#include "tt_common.h"
#include "tt_gemm.h"
#include "tt_riders.h"

#include <stdlib.h>

namespace {

constexpr int kThreads = 256;

// ------------------------------------------------------------------------------------------------
// slot decoding shared by lookup (forward) and gradient (backward)
// ------------------------------------------------------------------------------------------------
struct SideDev {
  const int64_t* ids;
  const int64_t* off;
  const int64_t* vocab;
  char* out;          // lookup output / gradient source
  int64_t ld;
  uint32_t slot_base; // first slot of this side
  int32_t K;
  int32_t dtype;
  uint32_t magic;     // floor(2^32 / K): slot -> (sample, key) without an integer division (gradient kernels)
};

struct SideSet {
  SideDev s[TT_MAX_SIDES];
  int32_t n;
  int32_t E;
  uint32_t C;          // VEC-wide chunks per row
  uint32_t total_slots;
  int32_t table_rows;  // lookup kernels: rows of the table the decoded row indexes (0: unchecked)
  uint32_t* dev_err;   // ... and the context's sticky error word (TT_DEVERR_ROW_RANGE)
};

// A decoded row that does not lie in the table: key offsets / vocabularies (device arrays the host cannot check without a
// synchronisation) that belong to another table, or precomputed rows from elsewhere.  Reading it would be a GPU memory fault;
// the launch reads the last row instead and raises the sticky error word (tt_ctx_check_device_errors -> TT_ERR_DEVICE).
__device__ __forceinline__ int64_t row_in_table(int64_t row, int32_t table_rows, uint32_t* dev_err) {
#ifdef TT_NO_ROW_CHECK                                      // measurement builds only (tools/r04_b13.sh: what the check costs)
  return row;
#endif
  if (table_rows > 0 && (uint64_t)row >= (uint64_t)table_rows) {
    if (dev_err) atomicOr(dev_err, TT_DEVERR_ROW_RANGE);
    row = table_rows - 1;
  }
  return row;
}

__device__ __forceinline__ int side_of(const SideSet& a, uint32_t slot) {
  int si = 0;
#pragma unroll
  for (int i = 1; i < TT_MAX_SIDES; ++i)
    if (i < a.n && slot >= a.s[i].slot_base) si = i;
  return si;
}

// ------------------------------------------------------------------------------------------------
// a4 + a5: lookup.  Task = (slot, chunk); consecutive lanes take consecutive chunks of one row, so
// an E=32 f32 row is one 128-B line read by 8 lanes and a wave-instruction gathers 8 rows.
// U independent tasks per thread are issued before any store to keep >= U*16 B per lane in flight.
// ------------------------------------------------------------------------------------------------
template <int VEC, int U>
__global__ __launch_bounds__(kThreads) void lookup_kernel(SideSet a, const float* __restrict__ table,
                                                          int32_t* __restrict__ rows_out) {
  const uint32_t total = a.total_slots * a.C;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t base = blockIdx.x * blockDim.x + threadIdx.x; base < total; base += stride * U) {
    float v[U][VEC];
    char* dst[U];
    int dt[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t task = base + u * stride;
      ok[u] = task < total;
      if (ok[u]) {
        const uint32_t slot = task / a.C;
        const uint32_t chunk = task - slot * a.C;
        const int si = side_of(a, slot);
        const SideDev& s = a.s[si];
        const uint32_t local = slot - s.slot_base;
        const uint32_t b = local / (uint32_t)s.K;
        const uint32_t k = local - b * (uint32_t)s.K;
        int64_t id = s.ids[local];
        const int64_t hi = s.vocab[k] - 1;
        id = id < 0 ? 0 : (id > hi ? hi : id);                 // clamp: cat_embed.py:117
        const int64_t row = row_in_table(s.off[k] + id, a.table_rows, a.dev_err);
        if (chunk == 0 && rows_out) rows_out[slot] = (int32_t)row;
        if (table == nullptr) { ok[u] = false; continue; }      // rows-only mode (wave-uniform)
        const float* src = table + row * a.E + chunk * VEC;
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(src);
          v[u][0] = t.x; v[u][1 % VEC] = t.y; v[u][2 % VEC] = t.z; v[u][3 % VEC] = t.w;
        } else {
          v[u][0] = src[0];
        }
        dt[u] = s.dtype;
        const int64_t col = (int64_t)b * s.ld + (int64_t)k * a.E + chunk * VEC;
        dst[u] = s.out + col * (s.dtype == TT_BF16 ? 2 : 4);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      if (dt[u] == TT_F32) {
        if (VEC == 4) {
          *reinterpret_cast<float4*>(dst[u]) = make_float4(v[u][0], v[u][1 % VEC], v[u][2 % VEC], v[u][3 % VEC]);
        } else {
          *reinterpret_cast<float*>(dst[u]) = v[u][0];
        }
      } else {
        if (VEC == 4) {
          ushort4 o;
          o.x = tt_f2bf(v[u][0]); o.y = tt_f2bf(v[u][1 % VEC]); o.z = tt_f2bf(v[u][2 % VEC]); o.w = tt_f2bf(v[u][3 % VEC]);
          *reinterpret_cast<ushort4*>(dst[u]) = o;
        } else {
          *reinterpret_cast<uint16_t*>(dst[u]) = tt_f2bf(v[u][0]);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// a4 + a5, wave-chunk form (16-byte lanes, E % 4 == 0, E/4 a power of two <= 64).  A wave owns 64
// consecutive slots: lane l decodes slot l ONCE (coalesced 512-B id read, clamp, row, destination) and
// parks {source, destination} in a wave-private LDS table; the wave then walks the chunk 64/C rows at a
// time (C = E/4 lanes per row), issuing ALL row loads of the chunk before the first store, so every lane
// keeps C x 16 B in flight and a wave-instruction still reads whole 128-B lines.
// ------------------------------------------------------------------------------------------------
constexpr int kProfileMaxWg = 8192;
struct SlotRec {
  const float* src;
  char* dst;
};
using f32x4n = __attribute__((ext_vector_type(4))) float;

// W = 16-byte pieces per lane (round 4: with W = 2 a lane moves 32 B of a row -- two loads, ONE 16-byte bf16 store: half the
// store instructions, 3 % faster on tables in and out of the caches);  ROWS = the fused rows come precomputed (int32, slot
// order: the hand-over launch of a captured step has decoded and clamped the ids already -- tt_embed_lookup_rows_fwd) instead
// of being decoded from int64 ids, key offsets and vocabularies: no gathered offset / vocabulary loads, no 64-bit clamp, half
// the index bytes (lab, tools/probe/lookup_lab.hip: 9.2 -> 8.5 us back to back, 14.3 -> 11.9 us behind a cache-evicting copy).
template <int C, int SPW, int W, bool ROWS>   // C 16-byte chunks per row, SPW slots per wave pass
__global__ __launch_bounds__(kThreads) void lookup_wave_kernel(SideSet a, const float* __restrict__ table,
                                                              int32_t* __restrict__ rows_out, const int32_t* __restrict__ rows_in,
                                                              unsigned long long* __restrict__ ring, int ring_slots) {
  // measurement only: per-workgroup start/end stamps.  Workgroup b keeps its OWN launch counter (ring[b]) and writes the pair
  // of launch n into slot n % ring_slots of its column: no cross-workgroup traffic, no extra launch; the host reduces afterwards
  const bool stamp = ring && blockIdx.x < kProfileMaxWg && threadIdx.x == 0;
  unsigned long long t_start = 0, n_launch = 0;
  if (stamp) {
    t_start = __builtin_amdgcn_s_memrealtime();
    n_launch = ring[blockIdx.x];
  }
  __shared__ SlotRec recs[kThreads / 64][SPW];
  __shared__ int dts[kThreads / 64][SPW];
  static_assert(C % W == 0, "W pieces per lane must divide the row");
  constexpr int LPR = C / W;                        // lanes per row
  constexpr int RPI = 64 / LPR;                     // rows per wave-instruction
  constexpr int NIT = SPW / RPI;                    // wave-instructions per pass
  static_assert(SPW % RPI == 0 && NIT >= 1, "SPW must be a multiple of the rows per wave-instruction");
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t nchunks = (a.total_slots + SPW - 1) / SPW;
  const uint32_t wstride = gridDim.x * (kThreads / 64);
  for (uint32_t chunk = blockIdx.x * (kThreads / 64) + wave; chunk < nchunks; chunk += wstride) {
    const uint32_t slot = chunk * SPW + lane;
    SlotRec rec{nullptr, nullptr};
    int dt = TT_F32;
    if (lane < SPW && slot < a.total_slots) {
      const int si = side_of(a, slot);
      const SideDev& s = a.s[si];
      const uint32_t local = slot - s.slot_base;
      const uint32_t b = local / (uint32_t)s.K;
      const uint32_t k = local - b * (uint32_t)s.K;
      int64_t row;
      if (ROWS) {
        row = rows_in[slot];                               // checked where they were formed (tt_batch_ingest*: table_rows), not here
      } else {
        int64_t id = s.ids[local];
        const int64_t hi = s.vocab[k] - 1;
        id = id < 0 ? 0 : (id > hi ? hi : id);             // clamp: cat_embed.py:117
        row = row_in_table(s.off[k] + id, a.table_rows, a.dev_err);
        if (rows_out) rows_out[slot] = (int32_t)row;
      }
      rec.src = table + row * a.E;
      dt = s.dtype;
      rec.dst = s.out + ((int64_t)b * s.ld + (int64_t)k * a.E) * (dt == TT_BF16 ? 2 : 4);
    }
    if (lane < SPW) {
      recs[wave][lane] = rec;
      dts[wave][lane] = dt;
    }
    __builtin_amdgcn_wave_barrier();
    float4 v[NIT][W];
    const uint32_t sub = lane / LPR, part = lane % LPR;
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const SlotRec r = recs[wave][j * RPI + sub];
#pragma unroll
      for (int w = 0; w < W; ++w) {
        v[j][w] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r.src != nullptr) v[j][w] = *reinterpret_cast<const float4*>(r.src + (part * W + w) * 4);
      }
    }
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const SlotRec r = recs[wave][j * RPI + sub];
      if (r.dst == nullptr) continue;
      if (dts[wave][j * RPI + sub] == TT_F32) {            // non-temporal: the rows are read next by another kernel, not this one
#pragma unroll
        for (int w = 0; w < W; ++w) {
          f32x4n t;
          t[0] = v[j][w].x; t[1] = v[j][w].y; t[2] = v[j][w].z; t[3] = v[j][w].w;
          __builtin_nontemporal_store(t, reinterpret_cast<f32x4n*>(r.dst + (part * W + w) * 16));
        }
      } else if (W == 2) {
        uint4 o;
        o.x = (uint32_t)tt_f2bf(v[j][0].x) | ((uint32_t)tt_f2bf(v[j][0].y) << 16);
        o.y = (uint32_t)tt_f2bf(v[j][0].z) | ((uint32_t)tt_f2bf(v[j][0].w) << 16);
        o.z = (uint32_t)tt_f2bf(v[j][W - 1].x) | ((uint32_t)tt_f2bf(v[j][W - 1].y) << 16);
        o.w = (uint32_t)tt_f2bf(v[j][W - 1].z) | ((uint32_t)tt_f2bf(v[j][W - 1].w) << 16);
        *reinterpret_cast<uint4*>(r.dst + part * 16) = o;
      } else {
        ushort4 o;
        o.x = tt_f2bf(v[j][0].x); o.y = tt_f2bf(v[j][0].y); o.z = tt_f2bf(v[j][0].z); o.w = tt_f2bf(v[j][0].w);
        *reinterpret_cast<ushort4*>(r.dst + part * 8) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (ring) {                                            // measurement only: wait for this workgroup's stores
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (stamp) {
      unsigned long long* pair = ring + kProfileMaxWg + ((n_launch % (unsigned long long)ring_slots) * kProfileMaxWg + blockIdx.x) * 2;
      pair[0] = t_start;
      pair[1] = __builtin_amdgcn_s_memrealtime();
      ring[blockIdx.x] = n_launch + 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Dedup plan: LSD radix sort (BITS per pass) of (row, slot) pairs.
//   hist    : per-tile digit histogram        -> hist[digit * nblk + tile]
//   scan    : exclusive scan of that array (one workgroup)
//   scatter : stable ranking by wave ballots (no sorting network, no atomics on the output side)
// ------------------------------------------------------------------------------------------------
constexpr int kSortTile = 4096;                 // elements per workgroup
constexpr int kWaveSpan = kSortTile / 4;        // contiguous elements per wave

template <int BITS>
__global__ __launch_bounds__(kThreads) void sort_hist_kernel(const uint32_t* __restrict__ keys, uint32_t M, int shift,
                                                            uint32_t* __restrict__ hist, uint32_t nblk) {
  constexpr uint32_t R = 1u << BITS;
  __shared__ uint32_t h[R];
  for (uint32_t d = threadIdx.x; d < R; d += kThreads) h[d] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
  for (uint32_t i = threadIdx.x; i < kSortTile; i += kThreads) {
    const uint32_t idx = base + i;
    if (idx < M) atomicAdd(&h[(keys[idx] >> shift) & (R - 1)], 1u);
  }
  __syncthreads();
  for (uint32_t d = threadIdx.x; d < R; d += kThreads) hist[blockIdx.x * R + d] = h[d];
}

// per digit: exclusive scan over the tiles (hist[tile][digit], in place) and the digit total.
// grid = R/64 workgroups; thread = (digit, one of kScanSeg tile segments).  Round 4: 16 segments (1024 threads) instead of 4 -- at
// configs[4] (608 tiles) a thread walked 152 tiles twice and the launch took 42 us on 32 CUs; now 38 tiles: the same sums in the same
// order per segment, segment totals added in segment order (integers: any order gives the same bits)
constexpr int kScanSeg = 16;
template <int BITS>
__global__ __launch_bounds__(64 * kScanSeg) void sort_colscan_kernel(uint32_t* __restrict__ hist, uint32_t nblk, uint32_t* __restrict__ total) {
  constexpr uint32_t R = 1u << BITS;
  __shared__ uint32_t sh[kScanSeg][64];
  const uint32_t d = blockIdx.x * 64 + (threadIdx.x & 63), seg = threadIdx.x >> 6;
  const uint32_t per = (nblk + kScanSeg - 1) / kScanSeg;
  const uint32_t lo = seg * per < nblk ? seg * per : nblk;
  const uint32_t hi = lo + per < nblk ? lo + per : nblk;
  uint32_t s = 0;
  for (uint32_t b = lo; b < hi; ++b) s += hist[b * R + d];
  sh[seg][threadIdx.x & 63] = s;
  __syncthreads();
  uint32_t run = 0;
  for (uint32_t g = 0; g < seg; ++g) run += sh[g][threadIdx.x & 63];
  for (uint32_t b = lo; b < hi; ++b) {
    const uint32_t t = hist[b * R + d];
    hist[b * R + d] = run;
    run += t;
  }
  if (seg == kScanSeg - 1) total[d] = run;
}

template <int BITS>
__global__ __launch_bounds__(kThreads) void sort_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                               const uint32_t* __restrict__ vals_in,
                                                               uint32_t* __restrict__ keys_out,
                                                               uint32_t* __restrict__ vals_out, uint32_t M, int shift,
                                                               const uint32_t* __restrict__ hist_scanned,
                                                               const uint32_t* __restrict__ total) {
  constexpr uint32_t R = 1u << BITS;
  constexpr uint32_t PER = R / kThreads;              // digits per thread in the digit-base scan
  __shared__ uint32_t woff_s[4][R];
  __shared__ uint32_t dbase[R];
  __shared__ uint32_t wsum[4];
  volatile uint32_t(*woff)[R] = woff_s;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (uint32_t d = tid; d < 4 * R; d += kThreads) (&woff_s[0][0])[d] = 0;
  __syncthreads();
  const uint32_t wbase = blockIdx.x * kSortTile + wave * kWaveSpan;
  // phase 1: per-wave digit counts
  for (uint32_t it = 0; it < kWaveSpan / 64; ++it) {
    const uint32_t idx = wbase + it * 64 + lane;
    if (idx < M) atomicAdd(&woff_s[wave][(keys_in[idx] >> shift) & (R - 1)], 1u);
  }
  __syncthreads();
  // digit bases: exclusive scan of the digit totals (R <= 2048 values) inside the workgroup
  {
    uint32_t loc[PER], c = 0;
#pragma unroll
    for (uint32_t j = 0; j < PER; ++j) { loc[j] = total[tid * PER + j]; c += loc[j]; }
    uint32_t x = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(x, o);
      if (lane >= (uint32_t)o) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    uint32_t run = x - c;
    for (uint32_t w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (uint32_t j = 0; j < PER; ++j) { dbase[tid * PER + j] = run; run += loc[j]; }
  }
  __syncthreads();
  // phase 2: counts -> starting output offset of (wave, digit)
  for (uint32_t d = tid; d < R; d += kThreads) {
    uint32_t base = dbase[d] + hist_scanned[blockIdx.x * R + d];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t c = woff_s[w][d];
      woff_s[w][d] = base;
      base += c;
    }
  }
  __syncthreads();
  // phase 3: stable rank inside each 64-element batch from ballots, running offsets in LDS
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  for (uint32_t it = 0; it < kWaveSpan / 64; ++it) {
    const uint32_t idx = wbase + it * 64 + lane;
    const bool valid = idx < M;
    const uint32_t key = valid ? keys_in[idx] : 0u;
    const uint32_t val = valid ? (vals_in ? vals_in[idx] : idx) : 0u;
    const uint32_t d = (key >> shift) & (R - 1);
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < BITS; ++bit) {
      const bool one = (d >> bit) & 1u;
      const uint64_t bm = __ballot(one);
      peers &= one ? bm : ~bm;
    }
    uint32_t pos = 0;
    const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
    if (valid) pos = woff[wave][d] + rank;
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) woff[wave][d] = pos + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      keys_out[pos] = key;
      vals_out[pos] = val;
    }
  }
}

// stable merge of G ascending runs of length C (the owner's received buckets: every source sends its distinct rows in
// ascending order, pads = the largest value at the end): the position of an element is its own index plus, per other
// run, the number of elements that sort before it (<= for lower runs, < for higher ones) -- binary searches in L2,
// no radix passes (80 -> 30 us for 98 k ids; at G = 1 the run is already the answer)
__global__ __launch_bounds__(kThreads) void merge_runs_kernel(const int32_t* __restrict__ rows, uint32_t G, uint32_t C,
                                                             uint32_t* __restrict__ keys_out, int32_t* __restrict__ sorted_src) {
  const uint32_t total = G * C;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const uint32_t r = e / C, i = e - r * C;
    const uint32_t x = (uint32_t)rows[e];
    uint32_t pos = i;
    for (uint32_t q = 0; q < G; ++q) {
      if (q == r) continue;
      const int32_t* run = rows + (size_t)q * C;
      uint32_t lo = 0, hi = C;                          // first index whose value is > x (q < r) or >= x (q > r)
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint32_t v = (uint32_t)run[mid];
        const bool before = q < r ? v <= x : v < x;
        if (before) lo = mid + 1; else hi = mid;
      }
      pos += lo;
    }
    keys_out[pos] = x;
    sorted_src[pos] = (int32_t)e;
  }
}

// segment heads -------------------------------------------------------------------------------
__device__ __forceinline__ bool is_head(const uint32_t* keys, uint32_t i) { return i == 0 || keys[i] != keys[i - 1]; }

__global__ __launch_bounds__(kThreads) void head_count_kernel(const uint32_t* __restrict__ keys, uint32_t M,
                                                             uint32_t* __restrict__ blockcount) {
  __shared__ uint32_t cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kSortTile;
  uint32_t c = 0;
  for (uint32_t i = threadIdx.x; i < kSortTile; i += kThreads) {
    const uint32_t idx = base + i;
    if (idx < M && is_head(keys, idx)) ++c;
  }
  atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) blockcount[blockIdx.x] = cnt;
}

__global__ __launch_bounds__(kThreads) void head_write_kernel(const uint32_t* __restrict__ keys, uint32_t M,
                                                             const uint32_t* __restrict__ blockcount,
                                                             int32_t* __restrict__ n_unique,
                                                             int32_t* __restrict__ unique_rows,
                                                             int32_t* __restrict__ seg_offsets, uint32_t drop_from = 0xFFFFFFFFu) {
  __shared__ uint32_t wsum[4], wbefore[4], wall[4];
  constexpr uint32_t PER = kSortTile / kThreads;   // 16 contiguous elements per thread
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t lo = blockIdx.x * kSortTile + tid * PER;
  uint32_t flags = 0, c = 0;
#pragma unroll
  for (uint32_t j = 0; j < PER; ++j) {
    const uint32_t idx = lo + j;
    if (idx < M && is_head(keys, idx)) { flags |= 1u << j; ++c; }
  }
  uint32_t x = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o);
    if (lane >= (uint32_t)o) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  // offset of this tile = sum of the head counts of the tiles before it (block 0 also forms the total)
  uint32_t before = 0, all = 0;
  const uint32_t nblk = gridDim.x;
  for (uint32_t b = tid; b < nblk; b += kThreads) {
    const uint32_t v = blockcount[b];
    all += v;
    if (b < blockIdx.x) before += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); all += __shfl_xor(all, o); }
  if (lane == 0) { wbefore[wave] = before; wall[wave] = all; }
  __syncthreads();
  before = wbefore[0] + wbefore[1] + wbefore[2] + wbefore[3];
  all = wall[0] + wall[1] + wall[2] + wall[3];
  uint32_t u = before + x - c;
  for (uint32_t w = 0; w < wave; ++w) u += wsum[w];
#pragma unroll
  for (uint32_t j = 0; j < PER; ++j) {
    if (flags & (1u << j)) {
      unique_rows[u] = (int32_t)keys[lo + j];
      seg_offsets[u] = (int32_t)(lo + j);
      ++u;
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    // drop_from: the largest key is a pad value (>= drop_from) that the plan's consumers must not see as a row: it stays
    // in unique_rows / seg_offsets (seg_offsets[n_unique] is still the end of the last real row) but is not counted
    n_unique[0] = (int32_t)(all - ((M > 0 && keys[M - 1] >= drop_from) ? 1u : 0u));
    seg_offsets[all] = (int32_t)M;
  }
}

// chained aggregates (tt_common.h: tt_ctx::chain): a workgroup's value with the ready bit; a reader polls until the bit shows.
// All workgroups of these grids are resident at once (<= 2048 of 256 threads) and dispatched in index order, so a predecessor is
// always running or done; the poll is bounded all the same -- a buffer left dirty by an aborted launch must not hang the device
// (`stuck` then leaves an EMPTY plan, n_unique = 0: wrong results, but no consumer indexes anything with it).
// (relaxed agent-scope atomics: the word itself is all that travels, and nothing else has to become visible with it -- an acquire
//  load per poll invalidates the XCD's caches every time: 72 us for 608 tiles against 24 us for the two launches it replaced)
__device__ __forceinline__ void chain_publish(uint32_t* slot, uint32_t v) {
  __hip_atomic_store(slot, v | kChainReady, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t chain_wait(const uint32_t* slot, bool& stuck, int max_spin) {
  for (int spin = 0; spin < max_spin; ++spin) {
    const uint32_t v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v & kChainReady) return v & ~kChainReady;
    __builtin_amdgcn_s_sleep(1);
  }
  stuck = true;
  return 0u;
}

// head_count_kernel + head_write_kernel as ONE launch: a tile publishes its head count in chain[1 + tile] and adds up the tiles before
// it; the last tile also writes the totals.  chain[0] counts the tiles that are through with their reads: the last one clears.
__global__ __launch_bounds__(kThreads) void head_chained_kernel(const uint32_t* __restrict__ keys, uint32_t M, uint32_t* __restrict__ chain,
                                                               int32_t* __restrict__ n_unique, int32_t* __restrict__ unique_rows,
                                                               int32_t* __restrict__ seg_offsets, uint32_t drop_from, uint32_t* __restrict__ dev_err,
                                                               int max_spin) {
  __shared__ uint32_t wsum[4], wbefore[4];
  __shared__ uint32_t s_stuck, s_last;
  constexpr uint32_t PER = kSortTile / kThreads;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nblk = gridDim.x;
  const uint32_t lo = blockIdx.x * kSortTile + tid * PER;
  if (tid == 0) s_stuck = 0;
  uint32_t flags = 0, c = 0;
#pragma unroll
  for (uint32_t j = 0; j < PER; ++j) {
    const uint32_t idx = lo + j;
    if (idx < M && is_head(keys, idx)) { flags |= 1u << j; ++c; }
  }
  uint32_t x = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o);
    if (lane >= (uint32_t)o) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (tid == 0) chain_publish(chain + 1 + blockIdx.x, total);
  uint32_t before = 0;
  bool stuck = false;
  for (uint32_t b = tid; b < blockIdx.x; b += kThreads) before += chain_wait(chain + 1 + b, stuck, max_spin);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
  if (lane == 0) wbefore[wave] = before;
  if (stuck) {
    s_stuck = 1;
    // ANY tile whose wait expired raises the context's sticky device error word: the last tile only knows about
    // itself, and a plan with a middle tile's prefix missing is corrupt, not empty.  tt_ctx_check_device_errors reports it, clears
    // it and zeroes the chain buffers (a half-run launch leaves ready bits behind)
    atomicOr(dev_err, TT_DEVERR_CHAIN_TIMEOUT);
  }
  __syncthreads();
  before = wbefore[0] + wbefore[1] + wbefore[2] + wbefore[3];
  uint32_t u = before + x - c;
  for (uint32_t w = 0; w < wave; ++w) u += wsum[w];
#pragma unroll
  for (uint32_t j = 0; j < PER; ++j) {
    if (flags & (1u << j)) {
      unique_rows[u] = (int32_t)keys[lo + j];
      seg_offsets[u] = (int32_t)(lo + j);
      ++u;
    }
  }
  if (blockIdx.x == nblk - 1 && tid == 0) {              // (drop_from: see head_write_kernel)
    const uint32_t all = before + total;
    n_unique[0] = s_stuck ? 0 : (int32_t)(all - ((M > 0 && keys[M - 1] >= drop_from) ? 1u : 0u));   // (stuck: an empty plan reads nothing)
    seg_offsets[all] = (int32_t)M;
  }
  // everybody's reads of the chain are done once every tile has been here: the last one leaves the buffer all-zero
  if (tid == 0) s_last = atomicAdd(chain, 1u) == nblk - 1 ? 1u : 0u;
  __syncthreads();
  if (s_last) {
    for (uint32_t b = tid; b < nblk; b += kThreads) chain[1 + b] = 0u;
    if (tid == 0) chain[0] = 0u;
  }
}

// ------------------------------------------------------------------------------------------------
// Keyed dedup plan: one 1024-thread workgroup per (side, key) sorts that key's B slot rows in LDS.
// ------------------------------------------------------------------------------------------------
constexpr int kKeyedB = 8192;          // max ids per key (LDS: 2 x 32 KB keys + 2 x 16 KB values + 16 KB histograms + 32 KB lane sets)
constexpr int kKeyedThreads = 1024;

struct KeyedArgs {
  int32_t side_base[TT_MAX_SIDES + 1];   // first slot of side i
  int32_t key_base[TT_MAX_SIDES + 1];    // first key instance of side i
  int32_t K[TT_MAX_SIDES];
  int32_t n_sides;
  int32_t B;
  int32_t parts;                         // workgroups per key (range partition of the key's rows)
};
constexpr int kKeyedMaxParts = 8;
// (PlanLong, the long-row list the compaction builds for the gradient reduction: tt_riders.h)

__global__ __launch_bounds__(kKeyedThreads) void keyed_sort_kernel(KeyedArgs a, const int32_t* __restrict__ rows,
                                                                  int32_t* __restrict__ sorted_src, int32_t* __restrict__ uniq_stage,
                                                                  int32_t* __restrict__ seg_stage, int32_t* __restrict__ ucount,
                                                                  int32_t* __restrict__ ubase, int32_t* __restrict__ uend, bool key_major,
                                                                  int32_t* __restrict__ long_counters) {
  __shared__ uint32_t keys[2][kKeyedB];
  __shared__ uint16_t vals[2][kKeyedB];
  __shared__ uint32_t whist[16][256];
  __shared__ unsigned long long peers_mask[16][256];   // lane sets per (wave, digit); all zero between batches
  __shared__ uint32_t red[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // `parts` workgroups per key: every one loads the key's B rows and builds the same bucket histogram, then scatters, ranks and
  // writes only the buckets of ITS share of the key's row range (whole rows: a row never straddles two shares) -- the sorted
  // positions follow from the common prefix, so there is nothing to merge.  Every workgroup of a key reaches the same decision
  // about the LSD fallback (same histogram); share 0 then sorts the whole key alone.
  const int P = a.parts, ki = (int)blockIdx.x / P, part = (int)blockIdx.x % P, B = a.B;
  if (long_counters && blockIdx.x == 0 && tid < 2) long_counters[tid] = 0;      // the compaction (next launch) counts from zero
  for (int d = tid; d < 16 * 256; d += kKeyedThreads) (&peers_mask[0][0])[d] = 0ull;
  int side = 0;
#pragma unroll
  for (int i = 1; i < TT_MAX_SIDES; ++i)
    if (i < a.n_sides && ki >= a.key_base[i]) side = i;
  const int K = a.K[side], k = ki - a.key_base[side], sbase = a.side_base[side];
  // load this key's ids (stride K in the slot-major array) and find the row range
  uint32_t lo = 0xFFFFFFFFu, hi = 0;
  {
    // stride-K gather (one id per 4*K-byte step): all loads of a thread are issued before the first LDS store
    // (every key of a side reads the same 4*K-byte-strided lines, one dword of each: the workgroups start at different
    // samples so that they do not all ask for the same line at the same moment: 10.5 -> 7.9 us for the 32 notice keys)
    constexpr int PERL = kKeyedB / kKeyedThreads;
    uint32_t r[PERL];
    // key_major: the rows arrive as [key][sample] (tt_batch_ingest) -- this key's B rows are one contiguous run: 6.5 -> 2 us
    const int rot = key_major ? 0 : (int)(((unsigned)k * 264u) % (unsigned)B);
#pragma unroll
    for (int j = 0; j < PERL; ++j) {
      int b = tid + j * kKeyedThreads + rot;
      b = b >= B ? b - B : b;
      r[j] = tid + j * kKeyedThreads < B ? (uint32_t)rows[key_major ? sbase + k * B + b : sbase + b * K + k] : 0u;
    }
#pragma unroll
    for (int j = 0; j < PERL; ++j) {
      int b = tid + j * kKeyedThreads + rot;
      b = b >= B ? b - B : b;
      if (tid + j * kKeyedThreads < B) {
        keys[0][b] = r[j];
        vals[0][b] = (uint16_t)b;
        lo = r[j] < lo ? r[j] : lo;
        hi = r[j] > hi ? r[j] : hi;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if (lane == 0) { red[wave] = lo; red[16 + wave] = hi; }
  __syncthreads();
  lo = red[0]; hi = red[16];
  for (int w = 1; w < 16; ++w) { lo = red[w] < lo ? red[w] : lo; hi = red[16 + w] > hi ? red[16 + w] : hi; }
  int bits = 0;
  while (bits < 32 && ((hi - lo) >> bits) != 0) ++bits;
  const int passes = (bits + 7) / 8;
  // digits of equal width (a 9-bit range as 5 + 4 bits, not 8 + 1: a 1-bit digit sends all 64 lanes of a batch
  // to two LDS words and that pass ran 2x longer than an 8-bit one)
  const int dbits = passes > 0 ? (bits + passes - 1) / passes : 8;
  const uint32_t dmask = (1u << dbits) - 1u;
  // every wave owns a contiguous span of the array: stable LSD passes with per-wave digit histograms
  const int span = (B + 15) / 16;
  const int wlo = wave * span < B ? wave * span : B, whi = wlo + span < B ? wlo + span : B;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  int cur = 0;
  volatile uint32_t(*wh)[256] = whist;
  volatile unsigned long long(*pm)[256] = peers_mask;
  constexpr int NB = kKeyedB / 16 / 64;                // batches of 64 lanes per wave span
  // ---- bucket + rank instead of the LSD passes (~12 us each) -----------------------------------------------------------
  // The (row, slot) pairs are all different, so the sorted order is unique and the algorithm need not be stable: distribute
  // the pairs over buckets that are monotone in (row, slot) with plain LDS atomics (whatever order they are served in), scan,
  // scatter, then every element ranks itself inside its bucket by comparing (row, slot) with the bucket's other members.
  // (buckets of ~2 members: with 1024 buckets of 8 the ranking step's LDS reads -- members^2 per bucket -- took longer than
  // the passes they replace.)
  // A bucket with more than kBucketCap members (hot rows: Zipf ids) sends the whole key to the LSD passes below, which
  // produce the same unique order.
  bool sorted_by_buckets = false;
  const uint32_t range = hi - lo + 1u;
  int e_lo = 0, e_hi = B;                              // this workgroup's share of the sorted positions
  if (B > 64) {
    constexpr uint32_t kBucketCap = 32;
    // bucket = leading part of the pair (row - lo, slot), <= 4096 buckets, monotone in (row, slot):
    //   range <= 4096 : (row, slot >> sh) with 4096 / range (rounded down to a power of two) sub-buckets per row
    //   wider         : row scaled to [0, 4096) by one 32 x 32 -> 64 multiply
    // Up to 19 bits of row range the pair travels as ONE word (row - lo) * 8192 + slot (half the LDS reads of the ranking
    // step); wider keys keep two arrays.
    const bool fine = range <= 4096u, one_word = range <= (1u << 19);
    int sub_log = 0;
    while (sub_log < 13 && (range << (sub_log + 1)) <= 4096u) ++sub_log;
    const int sh = 13 - sub_log;
    const int nbk = fine ? (int)(range << sub_log) : 4096;
    const uint32_t M = fine ? 0u : (uint32_t)((4096ull << 32) / (uint64_t)range);
    auto bucket_of = [&](uint32_t rl, uint32_t slot) -> uint32_t {
      return fine ? (rl << sub_log) | (slot >> sh) : (uint32_t)(((uint64_t)rl * M) >> 32);
    };
    uint32_t* cnt = &whist[0][0];                                             // [4096]
    uint32_t* start = reinterpret_cast<uint32_t*>(&peers_mask[0][0]);         // [4096] (the lane sets are zeroed again below)
    uint32_t* w1 = &keys[1][0];                                               // scattered: composite, or row - lo ...
    uint16_t* s1 = &vals[1][0];                                               // ... and slot (two-array form)
#pragma unroll
    for (int q = 0; q < 4; ++q) cnt[tid * 4 + q] = 0;
    __syncthreads();
    constexpr int PERL = kKeyedB / kKeyedThreads;
    uint32_t kk[PERL], pp[PERL];
#pragma unroll
    for (int j = 0; j < PERL; ++j) {
      const int b = tid + j * kKeyedThreads;
      kk[j] = b < B ? keys[0][b] - lo : 0u;
      pp[j] = 0;
      if (b < B) pp[j] = atomicAdd(&cnt[bucket_of(kk[j], (uint32_t)b)], 1u);
    }
    __syncthreads();
    uint32_t c4[4], tot = 0, mx = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      c4[q] = tid * 4 + q < nbk ? cnt[tid * 4 + q] : 0u;
      tot += c4[q];
      mx = c4[q] > mx ? c4[q] : mx;
    }
    if (!__syncthreads_or(mx > kBucketCap)) {
      uint32_t x = tot;                                 // exclusive scan: thread = 4 consecutive buckets
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
      }
      if (lane == 63) red[wave] = x;
      __syncthreads();
      uint32_t run = x - tot;
      for (int w = 0; w < wave; ++w) run += red[w];
#pragma unroll
      for (int q = 0; q < 4; ++q) { start[tid * 4 + q] = run; run += c4[q]; }
      __syncthreads();
      // this share's buckets: rows [r_lo, r_hi) of the key's range (fine: a row owns 2^sub_log buckets; wider: a row owns one)
      uint32_t bk_lo, bk_hi;
      if (fine) {
        const uint32_t per = (range + (uint32_t)P - 1u) / (uint32_t)P;
        const uint32_t r_lo = min((uint32_t)part * per, range), r_hi = min(r_lo + per, range);
        bk_lo = r_lo << sub_log; bk_hi = r_hi << sub_log;
      } else {
        bk_lo = (uint32_t)part * 4096u / (uint32_t)P; bk_hi = (uint32_t)(part + 1) * 4096u / (uint32_t)P;
      }
      e_lo = bk_lo < (uint32_t)nbk ? (int)start[bk_lo] : B;
      e_hi = bk_hi < (uint32_t)nbk ? (int)start[bk_hi] : B;
#pragma unroll
      for (int j = 0; j < PERL; ++j) {
        const int b = tid + j * kKeyedThreads;
        if (b < B) {
          const uint32_t bk = bucket_of(kk[j], (uint32_t)b);
          if (bk >= bk_lo && bk < bk_hi) {
            const uint32_t pos = start[bk] + pp[j];
            w1[pos] = one_word ? (kk[j] << 13) | (uint32_t)b : kk[j];
            if (!one_word) s1[pos] = (uint16_t)b;
          }
        }
      }
      __syncthreads();
      // the thread's PERL elements rank themselves side by side: one loop to the longest of their buckets with PERL
      // independent LDS reads per trip (element by element the reads were one dependent chain each)
      uint32_t ek[PERL], ev[PERL], es[PERL], el[PERL], rk[PERL], maxlen = 0;
#pragma unroll
      for (int j = 0; j < PERL; ++j) {
        const int q = e_lo + tid + j * kKeyedThreads;
        const bool on = q < e_hi;
        const uint32_t w = on ? w1[q] : 0u;
        ek[j] = one_word ? w >> 13 : w;                 // row - lo
        ev[j] = one_word ? (w & 8191u) : (on ? (uint32_t)s1[q] : 0u);
        const uint32_t bk = on ? bucket_of(ek[j], ev[j]) : 0u;
        es[j] = on ? start[bk] : 0u;
        el[j] = on ? cnt[bk] : 0u;
        rk[j] = 0;
        maxlen = el[j] > maxlen ? el[j] : maxlen;
      }
      if (one_word) {
        for (uint32_t i = 0; i < maxlen; ++i) {
#pragma unroll
          for (int j = 0; j < PERL; ++j)
            rk[j] += (i < el[j] && w1[es[j] + (i < el[j] ? i : 0u)] < ((ek[j] << 13) | ev[j])) ? 1u : 0u;
        }
      } else {
        for (uint32_t i = 0; i < maxlen; ++i) {
#pragma unroll
          for (int j = 0; j < PERL; ++j) {
            const uint32_t at = es[j] + (i < el[j] ? i : 0u);
            const uint32_t ki = w1[at], vi = s1[at];
            rk[j] += (i < el[j] && (ki < ek[j] || (ki == ek[j] && vi < ev[j]))) ? 1u : 0u;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < PERL; ++j)
        if (e_lo + tid + j * kKeyedThreads < e_hi) {
          keys[0][es[j] + rk[j]] = ek[j] + lo;
          vals[0][es[j] + rk[j]] = (uint16_t)ev[j];
        }
      sorted_by_buckets = true;
    }
    __syncthreads();
    for (int d = tid; d < 16 * 256; d += kKeyedThreads) (&peers_mask[0][0])[d] = 0ull;     // `start` lived there
    __syncthreads();
  }
  if (!sorted_by_buckets) {                            // LSD passes: share 0 sorts the whole key, the others have nothing
    e_lo = 0;
    e_hi = part == 0 ? B : 0;
    if (part != 0) {
      if (tid == 0) { ucount[blockIdx.x] = 0; ubase[blockIdx.x] = ki * B; uend[blockIdx.x] = ki * B; }
      return;
    }
  }
  for (int p = 0; p < (sorted_by_buckets ? 0 : passes); ++p) {
    const int shift = dbits * p;
    for (int d = tid; d < 16 * 256; d += kKeyedThreads) (&whist[0][0])[d] = 0;
    uint32_t kreg[NB];
    uint16_t vreg[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {                      // the span's elements up front: one LDS round trip, not NB
      const int i = wlo + q * 64 + lane;
      kreg[q] = i < whi ? keys[cur][i] : 0u;
      vreg[q] = i < whi ? vals[cur][i] : (uint16_t)0;
    }
    // phase 1 -- who shares my digit in my batch: every lane ORs its bit into a per-(wave, digit) LDS word (the
    // result of an OR does not depend on the order the lanes are served in), reads the word back and clears it.
    // One 64-bit LDS atomic + one read per batch instead of eight ballots with per-lane 64-bit selects (that
    // loop was VALU-bound), and nothing in batch q+1 waits for batch q: the NB batches pipeline in the LDS queue.
    uint64_t peers[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const bool valid = wlo + q * 64 + lane < whi;
      const uint32_t d = ((kreg[q] - lo) >> shift) & dmask;
      if (valid) __hip_atomic_fetch_or(const_cast<unsigned long long*>(&pm[wave][d]), 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __builtin_amdgcn_wave_barrier();
      peers[q] = valid ? pm[wave][d] : 0ull;
      __builtin_amdgcn_wave_barrier();
      if (valid) pm[wave][d] = 0ull;                    // every peer writes the same zero: no leader needed yet
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                    // whist zeroed by everyone
    // per-(wave, digit) counts: one lane per distinct digit of a batch adds the batch's count
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const uint32_t d = ((kreg[q] - lo) >> shift) & dmask;
      const bool leader = peers[q] != 0ull && (peers[q] & lt_mask) == 0ull;
      if (leader) atomicAdd(&whist[wave][d], (uint32_t)__popcll(peers[q]));
    }
    __syncthreads();
    {
      // exclusive prefix over the 4096 (digit, wave) counters in digit-major order: thread t owns digit t/4,
      // waves 4*(t%4) .. +3; wave-level scan of the thread sums, then the 16 wave totals
      const int d = tid >> 2, w0 = (tid & 3) * 4;
      uint32_t c4[4], tot = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { c4[j] = whist[w0 + j][d]; tot += c4[j]; }
      uint32_t x = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
      }
      if (lane == 63) red[wave] = x;
      __syncthreads();
      uint32_t run = x - tot;
      for (int w = 0; w < wave; ++w) run += red[w];
#pragma unroll
      for (int j = 0; j < 4; ++j) { whist[w0 + j][d] = run; run += c4[j]; }
    }
    __syncthreads();
    // phase 2 -- positions: the batch's leader of a digit takes the running offset (one lane per address and
    // instruction, batches in program order => deterministic) and hands it to its peers with a bpermute
    uint32_t base[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const uint32_t d = ((kreg[q] - lo) >> shift) & dmask;
      const bool leader = peers[q] != 0ull && (peers[q] & lt_mask) == 0ull;
      base[q] = 0;
      if (leader) base[q] = atomicAdd(const_cast<uint32_t*>(&wh[wave][d]), (uint32_t)__popcll(peers[q]));
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const bool valid = peers[q] != 0ull;
      const int leader_lane = valid ? (int)__builtin_ctzll(peers[q]) : lane;
      const uint32_t b0 = (uint32_t)__builtin_amdgcn_ds_bpermute(leader_lane << 2, (int)base[q]);
      if (valid) {
        const uint32_t pos = b0 + (uint32_t)__popcll(peers[q] & lt_mask);
        keys[cur ^ 1][pos] = kreg[q];
        vals[cur ^ 1][pos] = vreg[q];
      }
    }
    __syncthreads();
    cur ^= 1;
  }
  // outputs: sorted slots, then the heads of this key (staged; compacted across keys by the second kernel)
  const int64_t gbase = (int64_t)ki * B;
  constexpr int PER = kKeyedB / kKeyedThreads;        // 8 contiguous elements per thread
  const int i_lo = e_lo + tid * PER;
  uint32_t flags = 0, c = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = i_lo + j;
    if (i < e_hi) {
      sorted_src[gbase + i] = sbase + (int)vals[cur][i] * K + k;
      if (i == e_lo || keys[cur][i] != keys[cur][i - 1]) { flags |= 1u << j; ++c; }   // (a share starts at a row boundary)
    }
  }
  uint32_t x = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  __syncthreads();
  if (lane == 63) red[wave] = x;
  __syncthreads();
  uint32_t u = x - c;
  for (int w = 0; w < wave; ++w) u += red[w];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    if (flags & (1u << j)) {                            // staged at the share's own positions: it has at most e_hi - e_lo heads
      uniq_stage[gbase + e_lo + u] = (int32_t)keys[cur][i_lo + j];
      seg_stage[gbase + e_lo + u] = (int32_t)(gbase + i_lo + j);
      ++u;
    }
  }
  if (tid == kKeyedThreads - 1) {
    ucount[blockIdx.x] = (int32_t)u;
    ubase[blockIdx.x] = (int32_t)(gbase + e_lo);
    uend[blockIdx.x] = (int32_t)(gbase + e_hi);           // where the share's last segment ends
  }
}

__global__ __launch_bounds__(kKeyedThreads) void keyed_compact_kernel(CompactRider cr) { compact_body(cr, blockIdx.x); }

// ------------------------------------------------------------------------------------------------
// a16: segmented gradient reduction.  A lane-group of LG lanes owns one distinct row and walks its
// segment in ascending slot order (4 independent loads in flight, added in order).
// ------------------------------------------------------------------------------------------------
// segments longer than this are split into kLongSeg-slot chunks (4 trips each) summed by their own lane groups.
// (16 = one trip per chunk was tried: at the bench's 6,200 rows of 17-64 slots the two same-address atomics per long
//  row and a workgroup-per-row finish cost more than the serial trips save: 25 + 8 + 11 us against 14 + 9 + 5.)
constexpr int kLongSeg = 64;
static_assert(kPlanLongSeg == kLongSeg, "the plan's compaction and the reduction must cut long rows into the same chunks");

struct GradWs {
  int32_t* counters;     // [0] chunks allocated, [1] long rows
  int32_t* long_row;     // [maxLong]   distinct-row index u
  int32_t* long_base;    // [maxLong]   first chunk of that row
  int32_t* chunk_lo;     // [maxChunks]
  int32_t* chunk_hi;
  float* chunk_partial;  // [maxChunks, E]
};

__global__ void zero_words_kernel(int32_t* __restrict__ p, int n) {
  if ((int)threadIdx.x < n) p[threadIdx.x] = 0;
}
// two single words at unrelated addresses (an empty plan's n_unique and seg_offsets[0]) in ONE launch -- never a pair of
// hipMemsetAsync calls: captured into a graph, two memset nodes are the pattern that faulted on replay
__global__ void zero_two_words_kernel(int32_t* __restrict__ a, int32_t* __restrict__ b) {
  if (threadIdx.x == 0) a[0] = 0;
  if (threadIdx.x == 1) b[0] = 0;
}

template <int VEC>
struct Acc {
  float v[VEC];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = 0.f;
  }
};

// Gradient source address of (slot, chunk).  Branch-free on purpose: the side's fields are picked with selects
// on kernel-argument scalars (indexing a.s[] with a per-lane index turns every field into a dependent memory
// load) and the element type is a template parameter (a per-slot dtype branch keeps the compiler from batching
// the loads of a trip: 0.4 us PER SLOT measured, 26 us for one 64-slot chunk).
template <int VEC, int ESZ>
__device__ __forceinline__ const char* grad_addr(const SideSet& a, uint32_t slot, uint32_t chunk) {
  const char* base = a.s[0].out;
  int64_t ld = a.s[0].ld;
  uint32_t sb = 0, K = (uint32_t)a.s[0].K, magic = a.s[0].magic;
#pragma unroll
  for (int i = 1; i < TT_MAX_SIDES; ++i) {
    const bool sel = i < a.n && slot >= a.s[i].slot_base;
    base = sel ? a.s[i].out : base;
    ld = sel ? a.s[i].ld : ld;
    sb = sel ? a.s[i].slot_base : sb;
    K = sel ? (uint32_t)a.s[i].K : K;
    magic = sel ? a.s[i].magic : magic;
  }
  const uint32_t local = slot - sb;
  uint32_t b = __umulhi(local, magic);       // floor(local / K) or one less (magic = floor(2^32 / K))
  uint32_t k = local - b * K;
  if (k >= K) { k -= K; ++b; }
  return base + ((int64_t)b * ld + (int64_t)(k * (uint32_t)a.E + chunk * VEC)) * ESZ;
}

template <int VEC, int DT>
__device__ __forceinline__ void load_grad_chunk(const SideSet& a, uint32_t slot, uint32_t chunk, float* o) {
  if (DT == TT_F32) {
    const float* p = reinterpret_cast<const float*>(grad_addr<VEC, 4>(a, slot, chunk));
    if (VEC == 4) {
      const float4 t = *reinterpret_cast<const float4*>(p);
      o[0] = t.x; o[1 % VEC] = t.y; o[2 % VEC] = t.z; o[3 % VEC] = t.w;
    } else {
      o[0] = p[0];
    }
  } else {
    const uint16_t* p = reinterpret_cast<const uint16_t*>(grad_addr<VEC, 2>(a, slot, chunk));
    if (VEC == 4) {
      const ushort4 t = *reinterpret_cast<const ushort4*>(p);
      o[0] = tt_bf2f(t.x); o[1 % VEC] = tt_bf2f(t.y); o[2 % VEC] = tt_bf2f(t.z); o[3 % VEC] = tt_bf2f(t.w);
    } else {
      o[0] = tt_bf2f(p[0]);
    }
  }
}

// ordered sum of slots sorted_src[lo..hi) for one chunk column.  The walk is a dependent chain of trips
// (index load -> decode -> gradient load), so the trip count and the instructions per trip -- not bandwidth --
// set the kernel time.  kBatch gradient loads are in flight per trip.
//   LGT > 0 (lane group of LGT = 4, 8 or 16 lanes, all active): the lanes split the trip's index loads and
//   decodes (kBatch / LGT each) and hand the addresses round with ds_bpermute, instead of all decoding all.
//   LGT == 0: every lane decodes every slot (any group width).
constexpr int kBatch = 16;
#define TT_GLOBAL __attribute__((address_space(1)))
using tt_u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using tt_u32x2 = __attribute__((ext_vector_type(2))) uint32_t;

__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int src, int width) {
  const uint32_t lo = __shfl((uint32_t)v, src, width), hi = __shfl((uint32_t)(v >> 32), src, width);
  return ((uint64_t)hi << 32) | lo;
}

template <int VEC, int DT, int LGT>
__device__ __forceinline__ void sum_range(const SideSet& a, const int32_t* __restrict__ sorted_src, int32_t lo, int32_t hi,
                                          uint32_t chunk, uint32_t lig, Acc<VEC>& acc) {
  constexpr int ESZ = DT == TT_F32 ? 4 : 2;
  if (LGT > 0) {
    constexpr int PER = LGT > 0 ? kBatch / (LGT > 0 ? LGT : 1) : 1;
    int32_t idx[PER];
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int32_t s = lo + p * LGT + (int32_t)lig;
      idx[p] = s < hi ? sorted_src[s] : 0;
    }
    for (int32_t i = lo; i < hi; i += kBatch) {
      const int32_t n = hi - i;                                   // group-uniform
      uint64_t mine[PER];
#pragma unroll
      for (int p = 0; p < PER; ++p) mine[p] = reinterpret_cast<uint64_t>(grad_addr<VEC, ESZ>(a, (uint32_t)idx[p], 0));
#pragma unroll
      for (int p = 0; p < PER; ++p) {                             // next trip's indices under this trip's loads
        const int32_t s = i + kBatch + p * LGT + (int32_t)lig;
        idx[p] = s < hi ? sorted_src[s] : 0;
      }
      // raw bits first, decode after the last load: a bf16 -> f32 convert inside the `j < n` branch makes the
      // compiler wait for each load where it stands (vmcnt(0) per slot: 27 us against 14 for the kernel)
      constexpr int RW = VEC * ESZ >= 4 ? VEC * ESZ / 4 : 1;
      uint32_t raw[kBatch][RW];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) {
        const uint64_t ptr = shfl_u64(mine[j / LGT], j % LGT, LGT) + (uint64_t)chunk * VEC * ESZ;
        if (j < n) {
          if (RW == 4) {
            const tt_u32x4 q = *reinterpret_cast<const TT_GLOBAL tt_u32x4*>(ptr);
            raw[j][0] = q.x; raw[j][1 % RW] = q.y; raw[j][2 % RW] = q.z; raw[j][3 % RW] = q.w;
          } else if (RW == 2) {
            const tt_u32x2 q = *reinterpret_cast<const TT_GLOBAL tt_u32x2*>(ptr);
            raw[j][0] = q.x; raw[j][1 % RW] = q.y;
          } else if (ESZ == 4) {
            raw[j][0] = *reinterpret_cast<const TT_GLOBAL uint32_t*>(ptr);
          } else {
            raw[j][0] = *reinterpret_cast<const TT_GLOBAL uint16_t*>(ptr);
          }
        } else {
#pragma unroll
          for (int e = 0; e < RW; ++e) raw[j][e] = 0u;
        }
      }
#pragma unroll
      for (int j = 0; j < kBatch; ++j)
        if (j < n) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            float f;
            if (DT == TT_F32) f = __uint_as_float(raw[j][e % RW]);
            else f = __uint_as_float((e & 1) ? (raw[j][(e / 2) % RW] & 0xffff0000u) : (raw[j][(e / 2) % RW] << 16));
            acc.v[e] += f;
          }
        }
    }
    return;
  }
  int32_t i = lo;
  if (i + kBatch <= hi) {
    int32_t idx[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) idx[j] = sorted_src[i + j];
    for (; i + kBatch <= hi; i += kBatch) {
      float t[kBatch][VEC];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) load_grad_chunk<VEC, DT>(a, (uint32_t)idx[j], chunk, t[j]);
      if (i + 2 * kBatch <= hi) {
#pragma unroll
        for (int j = 0; j < kBatch; ++j) idx[j] = sorted_src[i + kBatch + j];
      }
#pragma unroll
      for (int j = 0; j < kBatch; ++j)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc.v[e] += t[j][e];
    }
  }
  for (; i + 4 <= hi; i += 4) {
    float t[4][VEC];
#pragma unroll
    for (int j = 0; j < 4; ++j) load_grad_chunk<VEC, DT>(a, (uint32_t)sorted_src[i + j], chunk, t[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc.v[e] += t[j][e];
  }
  for (; i < hi; ++i) {
    float t[VEC];
    load_grad_chunk<VEC, DT>(a, (uint32_t)sorted_src[i], chunk, t);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc.v[e] += t[e];
  }
}

template <int VEC>
__device__ __forceinline__ void write_row(float* __restrict__ out, int64_t row, int32_t E, uint32_t chunk, const Acc<VEC>& acc,
                                          bool accumulate) {
  float* p = out + row * E + chunk * VEC;
  if (VEC == 4) {
    float4 t = make_float4(acc.v[0], acc.v[1 % VEC], acc.v[2 % VEC], acc.v[3 % VEC]);
    if (accumulate) {
      const float4 q = *reinterpret_cast<float4*>(p);
      t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
    }
    *reinterpret_cast<float4*>(p) = t;
  } else {
    p[0] = accumulate ? p[0] + acc.v[0] : acc.v[0];
  }
}

// planned: the long rows are already in the workspace's lists (built by the plan's compaction): nothing to register here
template <int VEC, int DT, int LGT>
__device__ __forceinline__ void seg_reduce_body(const SideSet& a, const int32_t* __restrict__ sorted_src,
                                                const int32_t* __restrict__ seg, const int32_t* __restrict__ unique_rows,
                                                const int32_t* __restrict__ n_unique, int32_t mode,
                                                float* __restrict__ out, const GradWs& ws, uint32_t LG, bool all_short, bool planned,
                                                uint32_t bid, uint32_t nblocks) {
  const uint32_t U = (uint32_t)*n_unique;
  const uint32_t gthread = bid * blockDim.x + threadIdx.x;
  const uint32_t lig = gthread % LG;
  const uint32_t ngroups = nblocks * blockDim.x / LG;
  for (uint32_t u = gthread / LG; u < U; u += ngroups) {
    const int32_t s0 = seg[u], s1 = seg[u + 1];
    if (planned && s1 - s0 > kLongSeg) continue;
    if (!all_short && s1 - s0 > kLongSeg) {
      const int32_t nch = (s1 - s0 + kLongSeg - 1) / kLongSeg;
      int32_t base = 0;
      if (lig == 0) {
        base = atomicAdd(&ws.counters[0], nch);
        const int32_t li = atomicAdd(&ws.counters[1], 1);
        ws.long_row[li] = (int32_t)u;
        ws.long_base[li] = base;
      }
      base = __shfl(base, 0, (int)LG);                     // the group's lanes write the chunk list together
      for (int32_t c = (int32_t)lig; c < nch; c += (int32_t)LG) {
        ws.chunk_lo[base + c] = s0 + c * kLongSeg;
        ws.chunk_hi[base + c] = min(s1, s0 + (c + 1) * kLongSeg);
      }
      continue;
    }
    const int64_t orow = mode == TT_GRAD_SPARSE ? (int64_t)u : (int64_t)unique_rows[u];
    for (uint32_t chunk = lig; chunk < a.C; chunk += LG) {
      Acc<VEC> acc;
      acc.zero();
      sum_range<VEC, DT, LGT>(a, sorted_src, s0, s1, chunk, lig, acc);
      write_row<VEC>(out, orow, a.E, chunk, acc, mode == TT_GRAD_DENSE_ACC);
    }
  }
}

template <int VEC, int DT, int LGT>
__global__ __launch_bounds__(kThreads) void seg_reduce_kernel(SideSet a, const int32_t* __restrict__ sorted_src,
                                                             const int32_t* __restrict__ seg, const int32_t* __restrict__ unique_rows,
                                                             const int32_t* __restrict__ n_unique, int32_t mode,
                                                             float* __restrict__ out, GradWs ws, uint32_t LG, bool all_short) {
  seg_reduce_body<VEC, DT, LGT>(a, sorted_src, seg, unique_rows, n_unique, mode, out, ws, LG, all_short, false, blockIdx.x, gridDim.x);
}

template <int VEC, int DT, int LGT>
__device__ __forceinline__ void seg_chunk_body(const SideSet& a, const int32_t* __restrict__ sorted_src, const GradWs& ws, uint32_t LG,
                                               uint32_t bid, uint32_t nblocks) {
  const uint32_t nchunks = (uint32_t)ws.counters[0];
  if (bid == 0 && threadIdx.x == 0) ws.counters[2] = ws.counters[1];      // snapshot for seg_long_finish_kernel
  const uint32_t gthread = bid * blockDim.x + threadIdx.x;
  const uint32_t lig = gthread % LG;
  const uint32_t ngroups = nblocks * blockDim.x / LG;
  for (uint32_t c = gthread / LG; c < nchunks; c += ngroups) {
    for (uint32_t chunk = lig; chunk < a.C; chunk += LG) {
      Acc<VEC> acc;
      acc.zero();
      sum_range<VEC, DT, LGT>(a, sorted_src, ws.chunk_lo[c], ws.chunk_hi[c], chunk, lig, acc);
      write_row<VEC>(ws.chunk_partial, (int64_t)c, a.E, chunk, acc, false);
    }
  }
}

template <int VEC, int DT, int LGT>
__global__ __launch_bounds__(kThreads) void seg_chunk_kernel(SideSet a, const int32_t* __restrict__ sorted_src, GradWs ws, uint32_t LG) {
  seg_chunk_body<VEC, DT, LGT>(a, sorted_src, ws, LG, blockIdx.x, gridDim.x);
}

// rows and chunks in ONE launch when the plan's compaction has already built the long-row list: workgroups [0, g1) take the
// rows (long ones skipped), workgroups [g1, g1 + g2) the chunks -- the chunk pass no longer waits for the row pass to register them
template <int VEC, int DT, int LGT>
__global__ __launch_bounds__(kThreads) void seg_reduce_chunk_kernel(SideSet a, const int32_t* __restrict__ sorted_src,
                                                                   const int32_t* __restrict__ seg, const int32_t* __restrict__ unique_rows,
                                                                   const int32_t* __restrict__ n_unique, int32_t mode,
                                                                   float* __restrict__ out, GradWs ws, uint32_t LG, uint32_t g1) {
  if (blockIdx.x < g1) seg_reduce_body<VEC, DT, LGT>(a, sorted_src, seg, unique_rows, n_unique, mode, out, ws, LG, false, true, blockIdx.x, g1);
  else seg_chunk_body<VEC, DT, LGT>(a, sorted_src, ws, LG, blockIdx.x - g1, gridDim.x - g1);
}

// measurement aid, compiled in with -DTT_SEG_STAMPS only (tools/r04_seg_stamps.py): start / end stamps and role of every workgroup
#ifdef TT_SEG_STAMPS
__device__ unsigned long long g_seg_stamps[4096 * 4];
#define TT_SEG_STAMP(i, role) do { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); if (threadIdx.x == 0 && blockIdx.x < 4096) { \
  g_seg_stamps[blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); g_seg_stamps[blockIdx.x * 4 + 2] = (role) + 1; } } while (0)
#else
#define TT_SEG_STAMP(i, role) do { } while (0)
#endif

// ... and, in front of both, the workgroups of a slab reduction tt_towers_mlp_bwd left queued in the context
// (TT_OPT_DEFER_SLAB_REDUCE): the weight gradients' split-K slabs and this reduction do not depend on each other
// Order of the roles in the flat grid (round 4, from per-workgroup stamps: tools/r04_seg_stamps.py).  The machine holds ~1800 of these
// workgroups at a time and hands them out in index order: what comes first starts at once, what comes last starts when slots free up.
//   [the first kChunkFirst chunk workgroups: the long rows' chunks, four dependent trips of gathers (9 us) -- they used to be dispatched
//    last, started 10 us in and ended the launch]
//   [the slab items: 2-4 us each on the still empty machine; ONE workgroup per projection-bias item instead of a row of idle ones]
//   [the rows: 3.5 us each, the plan's small-vocabulary keys (33-64-slot rows: 11 us) first]
//   [the remaining chunk workgroups: idle unless the batch is skewed]
// Rows (all, or half of them) in front of the slab items measured slower (profiles/NOTES.md: 20.3-20.9 against 18.5-19.4 us).
constexpr uint32_t kChunkFirst = 8;
#ifndef TT_SEG_THREADS
#define TT_SEG_THREADS 256
#endif
constexpr int kSegThreads = TT_SEG_THREADS;          // (measurement builds: 512 / 1024 -- is the dispatch rate per workgroup or per wave?)
#ifdef TT_SEG_WAVES                                  // (measurement builds: cap the waves per SIMD)
#define TT_SEG_OCC __attribute__((amdgpu_waves_per_eu(1, TT_SEG_WAVES)))
#else
#define TT_SEG_OCC
#endif
template <int VEC, int DT, int LGT>
__global__ __launch_bounds__(kSegThreads) TT_SEG_OCC void seg_reduce_chunk_slab_kernel(SideSet a, const int32_t* __restrict__ sorted_src,
                                                                        const int32_t* __restrict__ seg, const int32_t* __restrict__ unique_rows,
                                                                        const int32_t* __restrict__ n_unique, int32_t mode,
                                                                        float* __restrict__ out, GradWs ws, uint32_t LG, uint32_t g1,
                                                                        SlabBatch sb, uint32_t nsx, uint32_t n_items, uint32_t ns) {
  const uint32_t g2 = gridDim.x - g1 - ns;
  const uint32_t gc = g2 < kChunkFirst ? g2 : kChunkFirst;
#ifndef TT_SEG_LAYOUT
#define TT_SEG_LAYOUT 4
#endif
  // rows in front of the slab items: none (layout 4, shipping); half of them (2), all (3): measurement builds
  const uint32_t r1 = TT_SEG_LAYOUT == 2 ? (g1 + 1) / 2 : (TT_SEG_LAYOUT == 3 ? g1 : 0u);
  uint32_t b = blockIdx.x;
  if (b < gc) {
    TT_SEG_STAMP(0, 2);
    seg_chunk_body<VEC, DT, LGT>(a, sorted_src, ws, LG, b, g2);
    TT_SEG_STAMP(1, 2);
    return;
  }
  b -= gc;
  if (b < r1) {
    TT_SEG_STAMP(0, 1);
    seg_reduce_body<VEC, DT, LGT>(a, sorted_src, seg, unique_rows, n_unique, mode, out, ws, LG, false, true, b, g1);
    TT_SEG_STAMP(1, 1);
    return;
  }
  b -= r1;
  if (b < ns) {
    TT_SEG_STAMP(0, 0);
    int item, bx;
    if (slab_role_locate(sb, (int)n_items, (int)nsx, (int)b, item, bx)) slab_reduce_block(sb, bx, (int)nsx, item);
    TT_SEG_STAMP(1, 0);
    return;
  }
  b -= ns;
  if (b < g1 - r1) {
    TT_SEG_STAMP(0, 1);
    seg_reduce_body<VEC, DT, LGT>(a, sorted_src, seg, unique_rows, n_unique, mode, out, ws, LG, false, true, r1 + b, g1);
    TT_SEG_STAMP(1, 1);
    return;
  }
  b -= g1 - r1;
  TT_SEG_STAMP(0, 2);
  seg_chunk_body<VEC, DT, LGT>(a, sorted_src, ws, LG, gc + b, g2);
  TT_SEG_STAMP(1, 2);
}

// one WORKGROUP per long row: its lane groups sum contiguous ranges of the row's chunk partials (8 loads in flight,
// chunk order), the group sums are added in group order through LDS -- one or two trips however long the row is
// (a binary key at B = 8192 has ~4096-slot rows = 256 partials)
constexpr int kFinishMaxFloats = 4096;     // (kThreads / LG) * E floats of LDS
// emit(u, col, total): called once per (long row, column) by the thread that added the group sums
template <int VEC, typename EMIT>
__device__ __forceinline__ void long_rows_finish(int32_t E, uint32_t C, const int32_t* __restrict__ seg, const GradWs& ws, uint32_t LG,
                                                 uint32_t bid, uint32_t nblocks, float* __restrict__ part, EMIT&& emit) {
  const uint32_t nlong = (uint32_t)ws.counters[2];        // the long-row count as seg_chunk_body saw it
  const uint32_t grp = threadIdx.x / LG, lig = threadIdx.x % LG, ngrp = blockDim.x / LG;
  for (uint32_t li = bid; li < nlong; li += nblocks) {
    const int32_t u = ws.long_row[li], base = ws.long_base[li];
    const int32_t nch = (seg[u + 1] - seg[u] + kLongSeg - 1) / kLongSeg;
    const int32_t per = (nch + (int32_t)ngrp - 1) / (int32_t)ngrp;
    const int32_t c0 = min(nch, (int32_t)grp * per), c1 = min(nch, c0 + per);
    for (uint32_t chunk = lig; chunk < C; chunk += LG) {
      Acc<VEC> acc;
      acc.zero();
      const float* p0 = ws.chunk_partial + (int64_t)base * E + chunk * VEC;
      int32_t c = c0;
      for (; c + 8 <= c1; c += 8) {
        float t[8][VEC];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int e = 0; e < VEC; ++e) t[j][e] = p0[(int64_t)(c + j) * E + e];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc.v[e] += t[j][e];
      }
      for (; c < c1; ++c) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc.v[e] += p0[(int64_t)c * E + e];
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) part[grp * E + chunk * VEC + e] = acc.v[e];
    }
    __syncthreads();
    for (int32_t col = threadIdx.x; col < E; col += blockDim.x) {
      float tot = 0.f;
      for (uint32_t g = 0; g < ngrp; ++g) tot += part[g * E + col];
      emit(u, col, tot);
    }
    __syncthreads();
  }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void seg_long_finish_kernel(int32_t E, uint32_t C, const int32_t* __restrict__ seg,
                                                                  const int32_t* __restrict__ unique_rows, int32_t mode,
                                                                  float* __restrict__ out, GradWs ws, uint32_t LG) {
  __shared__ float part[kFinishMaxFloats];
  // nobody reads the live words [0] / [1] any more (counters[2] holds the snapshot), so one thread zeroes them here for the
  // next call -- a caller that keeps the words between calls needs no zeroing launch
  // (a "last workgroup done" atomic instead cost 35 us: 2048 same-address atomics with return serialise at ~17 ns each)
  if (blockIdx.x == 0 && threadIdx.x == 0) { ws.counters[0] = 0; ws.counters[1] = 0; }
  long_rows_finish<VEC>(E, C, seg, ws, LG, blockIdx.x, gridDim.x, part, [&](int32_t u, int32_t col, float tot) {
    const int64_t orow = mode == TT_GRAD_SPARSE ? (int64_t)u : (int64_t)unique_rows[u];
    float* o = out + orow * E + col;
    *o = mode == TT_GRAD_DENSE_ACC ? *o + tot : tot;
  });
}

// ------------------------------------------------------------------------------------------------
// a17: Adam
// ------------------------------------------------------------------------------------------------
struct AdamK {
  float lr_over_bc1, inv_sqrt_bc2, b1, b2, eps, wd;
  const float* dev;   // optional device copy of the six scalars above (graph replay)
};

__device__ __forceinline__ AdamK adam_resolve(const AdamK& k) {
  if (k.dev == nullptr) return k;
  AdamK r;
  r.lr_over_bc1 = k.dev[0]; r.inv_sqrt_bc2 = k.dev[1]; r.b1 = k.dev[2]; r.b2 = k.dev[3]; r.eps = k.dev[4]; r.wd = k.dev[5];
  r.dev = nullptr;
  return r;
}

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamK& k) {
  // explicit fused multiply-adds: left to the compiler, the contraction of a * b + c * d differs from one inlining context
  // to the next (the float4 row path and the scalar long-row path of adam_fused_kernel disagreed in the last bit)
  g = k.wd != 0.f ? __builtin_fmaf(k.wd, p, g) : g;
  m = __builtin_fmaf(k.b1, m, (1.f - k.b1) * g);
  v = __builtin_fmaf(k.b2, v, (1.f - k.b2) * g * g);
  const float denom = __builtin_fmaf(sqrtf(v), k.inv_sqrt_bc2, k.eps);
  p = __builtin_fmaf(-k.lr_over_bc1, m / denom, p);
}

__global__ __launch_bounds__(kThreads) void adam_dense_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                             float* __restrict__ m, float* __restrict__ v, int64_t n, AdamK k0) {
  const AdamK k = adam_resolve(k0);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pp = p[i], mm = m[i], vv = v[i];
    adam1(pp, g[i], mm, vv, k);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

__global__ __launch_bounds__(kThreads) void adam_dense_vec4_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                                  float4* __restrict__ m, float4* __restrict__ v, int64_t n4, AdamK k0) {
  const AdamK k = adam_resolve(k0);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = p[i], mm = m[i], vv = v[i];
    const float4 gg = g[i];
    adam1(pp.x, gg.x, mm.x, vv.x, k); adam1(pp.y, gg.y, mm.y, vv.y, k);
    adam1(pp.z, gg.z, mm.z, vv.z, k); adam1(pp.w, gg.w, mm.w, vv.w, k);
    p[i] = pp; m[i] = mm; v[i] = vv;
  }
}

constexpr int kAdamMulti = 32;
struct AdamMultiArgs {
  tt_adam_tensor t[kAdamMulti];
};

__global__ __launch_bounds__(kThreads) void adam_multi_kernel(AdamMultiArgs a, AdamK k0) {
  const AdamK k = adam_resolve(k0);
  const tt_adam_tensor& t = a.t[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t.n; i += stride) {
    float pp = t.p[i], mm = t.m[i], vv = t.v[i];
    adam1(pp, t.g[i], mm, vv, k);
    t.p[i] = pp; t.m[i] = mm; t.v[i] = vv;
  }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void adam_sparse_kernel(float* __restrict__ table, float* __restrict__ m, float* __restrict__ v,
                                                              int32_t E, uint32_t C, const int32_t* __restrict__ unique_rows,
                                                              const float* __restrict__ grad_rows, const int32_t* __restrict__ n_unique,
                                                              AdamK k0, uint32_t LG, int64_t table_rows) {
  const AdamK k = adam_resolve(k0);
  const uint32_t U = (uint32_t)*n_unique;
  const uint32_t gthread = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lig = gthread % LG;
  const uint32_t ngroups = gridDim.x * blockDim.x / LG;
  for (uint32_t u = gthread / LG; u < U; u += ngroups) {
    const int64_t row = unique_rows[u];
    if (row >= table_rows) continue;                   // routing pad (multi-GPU fixed-capacity buckets)
    for (uint32_t chunk = lig; chunk < C; chunk += LG) {
      const int64_t o = row * E + chunk * VEC;
      const float* gp = grad_rows + (int64_t)u * E + chunk * VEC;
      if (VEC == 4) {
        float4 pp = *reinterpret_cast<float4*>(table + o), mm = *reinterpret_cast<float4*>(m + o),
               vv = *reinterpret_cast<float4*>(v + o);
        const float4 gg = *reinterpret_cast<const float4*>(gp);
        adam1(pp.x, gg.x, mm.x, vv.x, k); adam1(pp.y, gg.y, mm.y, vv.y, k);
        adam1(pp.z, gg.z, mm.z, vv.z, k); adam1(pp.w, gg.w, mm.w, vv.w, k);
        *reinterpret_cast<float4*>(table + o) = pp;
        *reinterpret_cast<float4*>(m + o) = mm;
        *reinterpret_cast<float4*>(v + o) = vv;
      } else {
        float pp = table[o], mm = m[o], vv = v[o];
        adam1(pp, gp[0], mm, vv, k);
        table[o] = pp; m[o] = mm; v[o] = vv;
      }
    }
  }
}

// dense tensors + the looked-up table rows in ONE launch: blocks [0, nd) walk the dense tensors (prefix table),
// the rest are the row-sparse update (every small launch in the step's dependent chain costs ~5 us)
struct AdamFusedArgs {
  tt_adam_tensor t[kAdamMulti];
  int32_t blk0[kAdamMulti + 1];   // first block of dense tensor i; blk0[n] = nd
  int32_t n;
};

// LONG: the gradient reduction left its long rows unfinished (TT_GRAD_DEFER_FINISH): blocks [nd, nd + nlb) add a long row's
// chunk partials exactly as seg_long_finish_kernel does, store the sum into grad_rows and update that table row; the row
// blocks skip those rows.  One launch fewer in the step's dependent chain, the same values.
template <int VEC, bool LONG>
__global__ __launch_bounds__(kThreads) void adam_fused_kernel(AdamFusedArgs a, float* __restrict__ table, float* __restrict__ m,
                                                             float* __restrict__ v, int32_t E, uint32_t C,
                                                             const int32_t* __restrict__ unique_rows, float* __restrict__ grad_rows,
                                                             const int32_t* __restrict__ n_unique, AdamK k0, uint32_t LG, int64_t table_rows,
                                                             const int32_t* __restrict__ seg, GradWs ws, int nlb) {
  const AdamK k = adam_resolve(k0);
  const int nd = a.blk0[a.n];
  if ((int)blockIdx.x < nd) {
    int ti = 0;
    for (int i = 1; i < a.n; ++i)
      if ((int)blockIdx.x >= a.blk0[i]) ti = i;
    const tt_adam_tensor t = a.t[ti];
    const int64_t nb = a.blk0[ti + 1] - a.blk0[ti];
    const int64_t stride = nb * blockDim.x;
    for (int64_t i = (int64_t)((int)blockIdx.x - a.blk0[ti]) * blockDim.x + threadIdx.x; i < t.n; i += stride) {
      float pp = t.p[i], mm = t.m[i], vv = t.v[i];
      adam1(pp, t.g[i], mm, vv, k);
      t.p[i] = pp; t.m[i] = mm; t.v[i] = vv;
    }
    return;
  }
  const int first = nd + (LONG ? nlb : 0);
  if (LONG && (int)blockIdx.x < first) {
    __shared__ float part[kFinishMaxFloats];
    long_rows_finish<VEC>(E, C, seg, ws, LG, blockIdx.x - nd, (uint32_t)nlb, part, [&](int32_t u, int32_t col, float tot) {
      grad_rows[(int64_t)u * E + col] = tot;
      const int64_t row = unique_rows[u];
      if (row >= table_rows) return;
      const int64_t o = row * E + col;
      float pp = table[o], mm = m[o], vv = v[o];
      adam1(pp, tot, mm, vv, k);
      table[o] = pp; m[o] = mm; v[o] = vv;
    });
    return;
  }
  const uint32_t U = (uint32_t)*n_unique;
  const uint32_t gthread = (blockIdx.x - first) * blockDim.x + threadIdx.x;
  const uint32_t lig = gthread % LG;
  const uint32_t ngroups = (gridDim.x - first) * blockDim.x / LG;
  for (uint32_t u = gthread / LG; u < U; u += ngroups) {
    const int64_t row = unique_rows[u];
    if (row >= table_rows) continue;                   // routing pad (multi-GPU fixed-capacity buckets)
    if (LONG && seg[u + 1] - seg[u] > kLongSeg) continue;   // finished and applied by the long-row blocks
    for (uint32_t chunk = lig; chunk < C; chunk += LG) {
      const int64_t o = row * E + chunk * VEC;
      const float* gp = grad_rows + (int64_t)u * E + chunk * VEC;
      if (VEC == 4) {
        float4 pp = *reinterpret_cast<float4*>(table + o), mm = *reinterpret_cast<float4*>(m + o),
               vv = *reinterpret_cast<float4*>(v + o);
        const float4 gg = *reinterpret_cast<const float4*>(gp);
        adam1(pp.x, gg.x, mm.x, vv.x, k); adam1(pp.y, gg.y, mm.y, vv.y, k);
        adam1(pp.z, gg.z, mm.z, vv.z, k); adam1(pp.w, gg.w, mm.w, vv.w, k);
        *reinterpret_cast<float4*>(table + o) = pp;
        *reinterpret_cast<float4*>(m + o) = mm;
        *reinterpret_cast<float4*>(v + o) = vv;
      } else {
        float pp = table[o], mm = m[o], vv = v[o];
        adam1(pp, gp[0], mm, vv, k);
        table[o] = pp; m[o] = mm; v[o] = vv;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// a1/a2: device-side batch assembly
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void batch_gather_kernel(const int64_t* __restrict__ entity, uint32_t B,
                                                               const float* __restrict__ dense_store, uint32_t dense_dim,
                                                               const int64_t* __restrict__ cat_store, uint32_t K,
                                                               float* __restrict__ dense_out, int64_t* __restrict__ ids_out) {
  const uint32_t per = dense_dim + K;
  const uint32_t total = B * per;
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const uint32_t b = t / per, j = t - b * per;
    const int64_t e = entity[b];
    if (j < dense_dim) dense_out[(int64_t)b * dense_dim + j] = dense_store[e * dense_dim + j];
    else ids_out[(int64_t)b * K + (j - dense_dim)] = cat_store[e * K + (j - dense_dim)];
  }
}

struct CopyArgs {
  char* dst[TT_MAX_COPIES];
  const char* src[TT_MAX_COPIES];
  int64_t bytes[TT_MAX_COPIES];
};

__global__ __launch_bounds__(kThreads) void copy_multi_kernel(CopyArgs a) {
  const int seg = blockIdx.y;
  const int64_t n16 = a.bytes[seg] / 16, tail0 = n16 * 16;
  const float4* __restrict__ s = reinterpret_cast<const float4*>(a.src[seg]);
  float4* __restrict__ d = reinterpret_cast<float4*>(a.dst[seg]);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) d[i] = s[i];
  if (blockIdx.x == 0)
    for (int64_t i = tail0 + threadIdx.x; i < a.bytes[seg]; i += blockDim.x) a.dst[seg][i] = a.src[seg][i];
}

// Batch hand-over of a graph-replayed step: the copy segments of copy_multi_kernel plus, per side, the fused row of every id
// in KEY-MAJOR order (rows_km[side_base + k * B + b] = key_row_offset[k] + clamp(ids[b * K + k])) -- the input
// tt_dedup_plan_keyed_km sorts.  A key's rows sit one per K * 4 bytes in the lookup's sample-major array: gathered by the sort
// itself that is a 128-byte line per lane (6.5 of a share's 18 us); here a workgroup reads 64 samples' ids as one run, turns the
// tile in LDS and writes 256-byte runs per key.  Grid row 0: every side's 64-sample tiles, side by side (dispatched first: the
// few transposing workgroups must not queue behind the thousands of copy workgroups); row y >= 1: copy segment y - 1.
constexpr int kIngestMaxK = 64;
// f32 -> bf16 (RNE) copies riding in the hand-over launch (tt_cvt_list): the towers' bf16 weight shadows, refreshed every step
struct CvtDev {
  const float* src[TT_MAX_CVT];
  uint16_t* dst[TT_MAX_CVT];
  int64_t count[TT_MAX_CVT];
  int32_t n;
};
__device__ __forceinline__ void cvt_role(const CvtDev& v) {
  for (int seg = 0; seg < v.n; ++seg) {
    const int64_t n4 = v.count[seg] / 4, stride = (int64_t)gridDim.x * blockDim.x;
    const float4* __restrict__ s = reinterpret_cast<const float4*>(v.src[seg]);
    ushort4* __restrict__ d = reinterpret_cast<ushort4*>(v.dst[seg]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      const float4 x = s[i];
      ushort4 o;
      o.x = tt_f2bf(x.x); o.y = tt_f2bf(x.y); o.z = tt_f2bf(x.z); o.w = tt_f2bf(x.w);
      d[i] = o;
    }
    if (blockIdx.x == 0)
      for (int64_t i = 4 * n4 + threadIdx.x; i < v.count[seg]; i += blockDim.x) v.dst[seg][i] = tt_f2bf(v.src[seg][i]);
  }
}
struct IngestArgs {
  CopyArgs c;
  CvtDev v;
  int32_t n_copy, n_sides, B;
  const int64_t* ids[TT_MAX_SIDES];
  const int64_t* off[TT_MAX_SIDES];
  const int64_t* vocab[TT_MAX_SIDES];
  int32_t K[TT_MAX_SIDES];
  int32_t side_base[TT_MAX_SIDES];
  int32_t* rows_km;
  int32_t* rows_sm;    // optional: the same fused rows in SLOT order (side_base + b * K + k) for tt_embed_lookup_rows_fwd
  int32_t table_rows;  // > 0: the rows are checked against this table size where they are formed (row_in_table)
  uint32_t* dev_err;
};

__global__ __launch_bounds__(kThreads) void batch_ingest_kernel(IngestArgs a) {
  if ((int)blockIdx.y == a.n_copy + 1) { cvt_role(a.v); return; }        // (last grid row, present when a.v.n > 0)
  if (blockIdx.y >= 1) {
    const int seg = blockIdx.y - 1;
    const int64_t n16 = a.c.bytes[seg] / 16, tail0 = n16 * 16;
    const float4* __restrict__ s = reinterpret_cast<const float4*>(a.c.src[seg]);
    float4* __restrict__ d = reinterpret_cast<float4*>(a.c.dst[seg]);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) d[i] = s[i];
    if (blockIdx.x == 0)
      for (int64_t i = tail0 + threadIdx.x; i < a.c.bytes[seg]; i += blockDim.x) a.c.dst[seg][i] = a.c.src[seg][i];
    return;
  }
  // row 0 of the grid: every side's 64-sample tiles side by side (an own grid row per side meant a thousand empty workgroups each)
  const int B = a.B, tiles = (B + 63) / 64;
  const int si = (int)blockIdx.x / tiles, tile = (int)blockIdx.x % tiles;
  if (si >= a.n_sides) return;
  const int K = a.K[si];
  __shared__ int32_t tl[kIngestMaxK][65];
  __shared__ int64_t s_off[kIngestMaxK], s_hi[kIngestMaxK];
  const int64_t* __restrict__ ids = a.ids[si];
  if ((int)threadIdx.x < K) {
    s_off[threadIdx.x] = a.off[si][threadIdx.x];
    s_hi[threadIdx.x] = a.vocab[si][threadIdx.x] - 1;
  }
  __syncthreads();
  constexpr int PER = kIngestMaxK * 64 / kThreads;            // ids per thread and tile: all loads issued before the first use
  {
    const int b0 = tile * 64;
    const int nb = min(64, B - b0), n = nb * K;
    int64_t idv[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {                           // the tile's ids are one contiguous run
      const int e = threadIdx.x + u * kThreads;
      idv[u] = ids[(int64_t)b0 * K + (e < n ? e : 0)];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = threadIdx.x + u * kThreads;
      if (e < n) {
        const int bl = e / K, k = e - bl * K;
        int64_t id = idv[u];
        id = id < 0 ? 0 : (id > s_hi[k] ? s_hi[k] : id);      // clamp: cat_embed.py:117 (as the lookup)
        tl[k][bl] = (int32_t)row_in_table(s_off[k] + id, a.table_rows, a.dev_err);
        if (a.rows_sm) a.rows_sm[a.side_base[si] + (int64_t)b0 * K + e] = tl[k][bl];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < K * 64; e += kThreads) {
      const int k = e >> 6, bl = e & 63;
      if (bl < nb) a.rows_km[a.side_base[si] + (int64_t)k * B + b0 + bl] = tl[k][bl];
    }
  }
}

// The hand-over straight from the device-resident feature stores (tt_batch_ingest_store): grid row 0 = every side's 64-sample
// tiles (entity rows of the tile's pairs -> the tile's ids as one sample-major run + the key-major fused rows, turned in LDS as
// above); rows 1 .. n_sides = the dense feature rows of side y - 1 (16-byte pieces, a row's pieces on consecutive lanes);
// the rows after that = the copy segments.
struct StoreIngestArgs {
  IngestArgs g;                                  // ids[] unused
  const int64_t* order;
  const int64_t* entity[TT_MAX_SIDES];
  int64_t entity_stride[TT_MAX_SIDES];
  const float* dense_store[TT_MAX_SIDES];
  const int64_t* cat_store[TT_MAX_SIDES];
  float* dense_out[TT_MAX_SIDES];
  int64_t* ids_out[TT_MAX_SIDES];
  int32_t dense_dim[TT_MAX_SIDES];
  int32_t n_rows[TT_MAX_SIDES];                  // entity rows of the store (0 = unchecked): indices are clamped into [0, n_rows)
};

template <bool VEC>
__global__ __launch_bounds__(kThreads) void batch_ingest_store_kernel(StoreIngestArgs a) {
  const int B = a.g.B;
  if ((int)blockIdx.y == 1 + a.g.n_sides + a.g.n_copy) { cvt_role(a.g.v); return; }
  if ((int)blockIdx.y > a.g.n_sides) {
    const int seg = blockIdx.y - 1 - a.g.n_sides;
    const int64_t n16 = a.g.c.bytes[seg] / 16, tail0 = n16 * 16;
    const float4* __restrict__ s = reinterpret_cast<const float4*>(a.g.c.src[seg]);
    float4* __restrict__ d = reinterpret_cast<float4*>(a.g.c.dst[seg]);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) d[i] = s[i];
    if (blockIdx.x == 0)
      for (int64_t i = tail0 + threadIdx.x; i < a.g.c.bytes[seg]; i += blockDim.x) a.g.c.dst[seg][i] = a.g.c.src[seg][i];
    return;
  }
  if (blockIdx.y >= 1) {                                       // dense feature rows of one side
    const int si = blockIdx.y - 1;
    const int dd = a.dense_dim[si];
    if (dd == 0) return;
    const int64_t* __restrict__ ent = a.entity[si];
    const int64_t es = a.entity_stride[si];
    const int per = VEC ? dd / 4 : dd;                         // pieces per row
    const int64_t total = (int64_t)B * per, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
      const int b = (int)(t / per), j = (int)(t - (int64_t)b * per);
      const int64_t o = a.order ? a.order[b] : b;
      int64_t e = ent[o * es];
      if (a.n_rows[si] > 0) e = e < 0 ? 0 : (e >= a.n_rows[si] ? a.n_rows[si] - 1 : e);
      if (VEC) reinterpret_cast<float4*>(a.dense_out[si] + (int64_t)b * dd)[j] = reinterpret_cast<const float4*>(a.dense_store[si] + e * dd)[j];
      else a.dense_out[si][(int64_t)b * dd + j] = a.dense_store[si][e * dd + j];
    }
    return;
  }
  const int tiles = (B + 63) / 64;
  const int si = (int)blockIdx.x / tiles, tile = (int)blockIdx.x % tiles;
  if (si >= a.g.n_sides) return;
  const int K = a.g.K[si];
  __shared__ int32_t tl[kIngestMaxK][65];
  __shared__ int64_t s_off[kIngestMaxK], s_hi[kIngestMaxK], s_ent[64];
  const int b0 = tile * 64;
  const int nb = min(64, B - b0), n = nb * K;
  if ((int)threadIdx.x < K) {
    s_off[threadIdx.x] = a.g.off[si][threadIdx.x];
    s_hi[threadIdx.x] = a.g.vocab[si][threadIdx.x] - 1;
  }
  if ((int)threadIdx.x >= 64 && (int)threadIdx.x < 64 + nb) {
    const int bl = threadIdx.x - 64;
    const int64_t o = a.order ? a.order[b0 + bl] : b0 + bl;
    int64_t e = a.entity[si][o * a.entity_stride[si]];
    if (a.n_rows[si] > 0) e = e < 0 ? 0 : (e >= a.n_rows[si] ? a.n_rows[si] - 1 : e);
    s_ent[bl] = e;
  }
  __syncthreads();
  constexpr int PER = kIngestMaxK * 64 / kThreads;
  const int64_t* __restrict__ cat = a.cat_store[si];
  int64_t idv[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {                              // a sample's K ids are one contiguous run of its entity row
    const int e = threadIdx.x + u * kThreads;
    const int ec = e < n ? e : 0;
    const int bl = ec / K, k = ec - bl * K;
    idv[u] = cat[s_ent[bl] * K + k];
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int e = threadIdx.x + u * kThreads;
    if (e < n) {
      const int bl = e / K, k = e - bl * K;
      int64_t id = idv[u];
      a.ids_out[si][(int64_t)b0 * K + e] = id;                 // sample-major: the KJT values() of the batch
      id = id < 0 ? 0 : (id > s_hi[k] ? s_hi[k] : id);         // clamp: cat_embed.py:117 (as the lookup)
      tl[k][bl] = (int32_t)row_in_table(s_off[k] + id, a.g.table_rows, a.g.dev_err);
      if (a.g.rows_sm) a.g.rows_sm[a.g.side_base[si] + (int64_t)b0 * K + e] = tl[k][bl];
    }
  }
  if (a.g.rows_km == nullptr) return;
  __syncthreads();
  for (int e = threadIdx.x; e < K * 64; e += kThreads) {
    const int k = e >> 6, bl = e & 63;
    if (bl < nb) a.g.rows_km[a.g.side_base[si] + (int64_t)k * B + b0 + bl] = tl[k][bl];
  }
}

// ------------------------------------------------------------------------------------------------
// Hand-over AND lookup in one launch (tt_batch_ingest_lookup / tt_batch_ingest_store_lookup; round 4).  The hand-over's tile
// workgroups already hold the batch's clamped fused rows in LDS; here they gather the table rows themselves and write them
// into the towers' input x (cat_embed.py:157-178 + base_tower.py:139), so the separate lookup launch -- its boundary, its
// re-read of the ids and its dispatch ramp -- is gone, and the copies of the dense features run beside the gathers.
//   tile = TS samples of one side, TS = 2^ts_shift chosen so that TS * K <= 512 slots: a 256-thread workgroup decodes two
//   slots per thread, a wave gathers up to two 64-slot chunks (every load issued before the first store: 16 x 16 B per lane
//   in flight, the standalone kernel's 19 waves per CU x 8 become ~10 x 16);
//   a lane moves 32 B of a row (LPR = E / 8 lanes per row): two 16-B loads, one 16-B bf16 store (two for f32 rows).
// Grid row 0 = every side's tiles (dispatched first); then, from the stores, one row per side for the dense features; then
// the copy segments.  Bit-identical to the hand-over followed by tt_embed_lookup_fwd (test).
// ------------------------------------------------------------------------------------------------
struct LookupPart {
  const float* table;
  char* out[TT_MAX_SIDES];
  int64_t ld[TT_MAX_SIDES];          // elements
  int32_t dtype[TT_MAX_SIDES];
  int32_t ts_shift[TT_MAX_SIDES];
  int32_t tile_base[TT_MAX_SIDES + 1];
  int32_t E;
  unsigned long long* ring;          // measurement: per-workgroup stamps of the tile role (tt_embed_lookup_set_profile)
  int32_t ring_slots;
  int32_t nt;                        // TT_OPT_LOOKUP_NT: bf16 rows leave by non-temporal stores
  int32_t table_rows;                // row_in_table
  uint32_t* dev_err;
};
constexpr int kTileSlots = 512;

template <int LPR, bool FROM_STORE, bool VEC>
__device__ __forceinline__ void ingest_lookup_body(const StoreIngestArgs& a, const LookupPart& lp) {
  const int B = a.g.B;
  const int first_copy_row = 1 + (FROM_STORE ? a.g.n_sides : 0);
  if ((int)blockIdx.y == first_copy_row + a.g.n_copy) { cvt_role(a.g.v); return; }
  if ((int)blockIdx.y >= first_copy_row) {                     // copy segments
    const int seg = blockIdx.y - first_copy_row;
    const int64_t n16 = a.g.c.bytes[seg] / 16, tail0 = n16 * 16;
    const float4* __restrict__ s = reinterpret_cast<const float4*>(a.g.c.src[seg]);
    float4* __restrict__ d = reinterpret_cast<float4*>(a.g.c.dst[seg]);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) d[i] = s[i];
    if (blockIdx.x == 0)
      for (int64_t i = tail0 + threadIdx.x; i < a.g.c.bytes[seg]; i += blockDim.x) a.g.c.dst[seg][i] = a.g.c.src[seg][i];
    return;
  }
  if (FROM_STORE && blockIdx.y >= 1) {                         // dense feature rows of one side
    const int si = blockIdx.y - 1;
    const int dd = a.dense_dim[si];
    if (dd == 0) return;
    const int64_t* __restrict__ ent = a.entity[si];
    const int64_t es = a.entity_stride[si];
    const int64_t nr = a.n_rows[si];
    const int per = VEC ? dd / 4 : dd;
    const int64_t total = (int64_t)B * per, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
      const int b = (int)(t / per), j = (int)(t - (int64_t)b * per);
      const int64_t o = a.order ? a.order[b] : b;
      int64_t e = ent[o * es];
      if (nr > 0) e = e < 0 ? 0 : (e >= nr ? nr - 1 : e);
      if (VEC) reinterpret_cast<float4*>(a.dense_out[si] + (int64_t)b * dd)[j] = reinterpret_cast<const float4*>(a.dense_store[si] + e * dd)[j];
      else a.dense_out[si][(int64_t)b * dd + j] = a.dense_store[si][e * dd + j];
    }
    return;
  }
  // ---- row 0: tiles ----
  int si = 0;
#pragma unroll
  for (int i = 1; i < TT_MAX_SIDES; ++i)
    if (i < a.g.n_sides && (int)blockIdx.x >= lp.tile_base[i]) si = i;
  if ((int)blockIdx.x >= lp.tile_base[a.g.n_sides]) return;
  const int tile = (int)blockIdx.x - lp.tile_base[si];
  const int K = a.g.K[si], KP = K | 1, sh = lp.ts_shift[si], TS = 1 << sh;
  const int b0 = tile << sh;
  const int nb = min(TS, B - b0), n = nb * K;                  // <= kTileSlots
  // A wave works alone up to its gathers: it decodes its two 64-slot chunks (w and w + 4) itself -- id, key offset and vocabulary
  // straight from memory, like the stand-alone lookup -- parks {row, destination} in wave-private LDS and issues every row load;
  // only the key-major read-out of the tile's rows (tl) needs the other waves, and by then the rows are in flight.
  __shared__ int32_t rows_w[kThreads / 64][2][64];             // fused row of the wave's slot (the gather's order)
  __shared__ int64_t dst_w[kThreads / 64][2][64];              // byte offset of the slot's destination row in x (-1: no slot)
  __shared__ int32_t tl[kTileSlots + 64];                      // the tile's rows at bl * KP + k (odd stride: the key-major read-out)
  const int esz = lp.dtype[si] == TT_BF16 ? 2 : 4;
  constexpr int RPI = 64 / LPR;                                  // rows per wave-instruction
  constexpr int NIT = LPR;                                       // wave-instructions per chunk
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane / LPR, part = lane % LPR;
  const int64_t* __restrict__ offp = a.g.off[si];
  const int64_t* __restrict__ vocp = a.g.vocab[si];
  float4 v[2][NIT][2];
  int64_t dv[2][NIT];
  // decode both chunks first (their id loads are independent and issued together), then every row load of both
  int64_t idv[2], hiv[2], ofv[2];
  int blv[2], kv[2];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int e = (wave + cc * (kThreads / 64)) * 64 + lane;
    const int ec = e < n ? e : 0;
    const int bl = ec / K, k = ec - bl * K;
    blv[cc] = bl; kv[cc] = k;
    if (FROM_STORE) {
      const int64_t o = a.order ? a.order[b0 + bl] : b0 + bl;
      int64_t en = a.entity[si][o * a.entity_stride[si]];
      const int64_t nr = a.n_rows[si];
      if (nr > 0) en = en < 0 ? 0 : (en >= nr ? nr - 1 : en);
      idv[cc] = a.cat_store[si][en * K + k];
    } else {
      idv[cc] = a.g.ids[si][(int64_t)b0 * K + ec];
    }
    hiv[cc] = vocp[k] - 1;
    ofv[cc] = offp[k];
  }
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int e = (wave + cc * (kThreads / 64)) * 64 + lane;
    int32_t row = 0;
    int64_t dst = -1;
    if (e < n) {
      int64_t id = idv[cc];
      if (FROM_STORE) a.ids_out[si][(int64_t)b0 * K + e] = id;  // sample-major: the KJT values() of the batch
      id = id < 0 ? 0 : (id > hiv[cc] ? hiv[cc] : id);           // clamp: cat_embed.py:117
      row = (int32_t)row_in_table(ofv[cc] + id, lp.table_rows, lp.dev_err);
      dst = ((int64_t)(b0 + blv[cc]) * lp.ld[si] + (int64_t)kv[cc] * lp.E) * esz;
      tl[blv[cc] * KP + kv[cc]] = row;
    }
    rows_w[wave][cc][lane] = row;
    dst_w[wave][cc][lane] = dst;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const int sl = j * RPI + sub;
      dv[cc][j] = dst_w[wave][cc][sl];
      v[cc][j][0] = v[cc][j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (dv[cc][j] >= 0) {
        const float* src = lp.table + (int64_t)rows_w[wave][cc][sl] * lp.E + part * 8;
        v[cc][j][0] = *reinterpret_cast<const float4*>(src);
        v[cc][j][1] = *reinterpret_cast<const float4*>(src + 4);
      }
    }
  }
  // while the rows fly: the key-major rows for the duplicate-row plan
  if (a.g.rows_km != nullptr) {
    __syncthreads();
    for (int e = threadIdx.x; e < (K << sh); e += kThreads) {
      const int k = e >> sh, bl = e & (TS - 1);
      if (bl < nb) a.g.rows_km[a.g.side_base[si] + (int64_t)k * B + b0 + bl] = tl[bl * KP + k];
    }
  }
  char* __restrict__ out = lp.out[si];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      if (dv[cc][j] < 0) continue;
      if (esz == 4) {                                          // non-temporal: the rows are read next by another kernel
        f32x4n t0, t1;
        t0[0] = v[cc][j][0].x; t0[1] = v[cc][j][0].y; t0[2] = v[cc][j][0].z; t0[3] = v[cc][j][0].w;
        t1[0] = v[cc][j][1].x; t1[1] = v[cc][j][1].y; t1[2] = v[cc][j][1].z; t1[3] = v[cc][j][1].w;
        f32x4n* d = reinterpret_cast<f32x4n*>(out + dv[cc][j] + part * 32);
        __builtin_nontemporal_store(t0, d);
        __builtin_nontemporal_store(t1, d + 1);
      } else {
        uint4 o;
        o.x = (uint32_t)tt_f2bf(v[cc][j][0].x) | ((uint32_t)tt_f2bf(v[cc][j][0].y) << 16);
        o.y = (uint32_t)tt_f2bf(v[cc][j][0].z) | ((uint32_t)tt_f2bf(v[cc][j][0].w) << 16);
        o.z = (uint32_t)tt_f2bf(v[cc][j][1].x) | ((uint32_t)tt_f2bf(v[cc][j][1].y) << 16);
        o.w = (uint32_t)tt_f2bf(v[cc][j][1].z) | ((uint32_t)tt_f2bf(v[cc][j][1].w) << 16);
        if (lp.nt) {
          using u32x4n = __attribute__((ext_vector_type(4))) unsigned int;
          u32x4n t; t[0] = o.x; t[1] = o.y; t[2] = o.z; t[3] = o.w;
          __builtin_nontemporal_store(t, reinterpret_cast<u32x4n*>(out + dv[cc][j] + part * 16));
        } else {
          *reinterpret_cast<uint4*>(out + dv[cc][j] + part * 16) = o;
        }
      }
    }
  }
}

template <int LPR, bool FROM_STORE, bool VEC>
__global__ __launch_bounds__(kThreads) void ingest_lookup_kernel(StoreIngestArgs a, LookupPart lp) {
  // measurement only (tt_embed_lookup_set_profile): start / end stamps of EVERY workgroup, column = its linear index -- the tiles
  // (gather phase) come first, the copy roles after them; the host reduces either set
  const uint32_t wg = blockIdx.y * gridDim.x + blockIdx.x;
  const bool stamp = lp.ring && wg < (uint32_t)kProfileMaxWg && threadIdx.x == 0;
  unsigned long long t_start = 0, n_launch = 0;
  if (stamp) {
    t_start = __builtin_amdgcn_s_memrealtime();
    n_launch = lp.ring[wg];
  }
  ingest_lookup_body<LPR, FROM_STORE, VEC>(a, lp);
  if (lp.ring) {                                               // wait for this workgroup's stores
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (stamp) {
      unsigned long long* pair = lp.ring + kProfileMaxWg + ((n_launch % (unsigned long long)lp.ring_slots) * kProfileMaxWg + wg) * 2;
      pair[0] = t_start;
      pair[1] = __builtin_amdgcn_s_memrealtime();
      lp.ring[wg] = n_launch + 1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// multi-GPU routing: distinct rows -> fixed-capacity owner buckets (stable, no host sync)
//   a workgroup of 4 waves covers 2048 consecutive plan rows, a wave 512 of them in 8 batches of 64
// ------------------------------------------------------------------------------------------------
constexpr int kRouteChunk = 2048, kRouteWaves = 4;

struct RoutePads { int32_t id[TT_MAX_RANKS]; };

__global__ __launch_bounds__(kThreads) void route_count_kernel(const int32_t* __restrict__ unique_rows, const int32_t* __restrict__ n_unique,
                                                              uint32_t G, uint32_t* __restrict__ seg_counts) {
  __shared__ uint32_t cnt[kRouteWaves][TT_MAX_RANKS];
  const uint32_t U = (uint32_t)*n_unique, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane < G) cnt[wave][lane] = 0;
  __builtin_amdgcn_wave_barrier();
  const uint32_t u0 = blockIdx.x * kRouteChunk + wave * (kRouteChunk / kRouteWaves);
#pragma unroll
  for (int q = 0; q < kRouteChunk / kRouteWaves / 64; ++q) {
    const uint32_t u = u0 + q * 64 + lane;
    if (u < U) atomicAdd(&cnt[wave][(uint32_t)unique_rows[u] % G], 1u);
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < G) seg_counts[((size_t)blockIdx.x * kRouteWaves + wave) * G + lane] = cnt[wave][lane];
}

// Every workgroup adds up the (block, wave) segment counts itself -- all of them for the totals (pads, counts, overflow flag), those
// in front of its own block for its starting offsets: nseg * G words from L2 per workgroup instead of a one-workgroup scan launch
// between the two passes (round 3: 6.4 us of the sharded step for 152 x 4 numbers).  Chaining the three passes through a ready-flag
// buffer in ONE launch was measured too: 19.7 us against 19.0 for the three -- the last workgroup's serial tail ate the launches saved.
__global__ __launch_bounds__(kThreads) void route_scatter_kernel(const int32_t* __restrict__ unique_rows, const int32_t* __restrict__ n_unique,
                                                                uint32_t G, uint32_t C, const uint32_t* __restrict__ seg_counts,
                                                                int32_t* __restrict__ counts, int32_t* __restrict__ overflow, RoutePads pads,
                                                                int32_t pad_u, int32_t* __restrict__ send_ids, int32_t* __restrict__ send_u,
                                                                int32_t* __restrict__ pos_u) {
  __shared__ uint32_t tot[TT_MAX_RANKS], pre[TT_MAX_RANKS];
  __shared__ uint32_t off[kRouteWaves][TT_MAX_RANKS];
  __shared__ unsigned long long pm[kRouteWaves][TT_MAX_RANKS];
  const uint32_t U = (uint32_t)*n_unique, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t nseg = gridDim.x * kRouteWaves, my_first = blockIdx.x * kRouteWaves;
  if (threadIdx.x < G) { tot[threadIdx.x] = 0; pre[threadIdx.x] = 0; }
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < nseg * G; e += kThreads) {
    const uint32_t v = seg_counts[e];
    if (v) {
      atomicAdd(&tot[e % G], v);
      if (e / G < my_first) atomicAdd(&pre[e % G], v);
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < G) {
    counts[threadIdx.x] = (int32_t)tot[threadIdx.x];
    if (tot[threadIdx.x] > C) overflow[0] = 1;          // sticky: the caller owns (and clears) the flag
  }
  // unused tail of every bucket: pad entries, written by the whole grid
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < G * C; i += gridDim.x * blockDim.x) {
    const uint32_t g = i / C, p = i - g * C;
    if (p >= tot[g]) { send_ids[i] = pads.id[g]; send_u[i] = pad_u; }
  }
  if (lane < G) {
    uint32_t o = pre[lane];
    for (uint32_t w = 0; w < wave; ++w) o += seg_counts[((size_t)my_first + w) * G + lane];
    off[wave][lane] = o;
    pm[wave][lane] = 0ull;
  }
  __builtin_amdgcn_wave_barrier();
  volatile uint32_t(*vo)[TT_MAX_RANKS] = off;
  volatile unsigned long long(*vp)[TT_MAX_RANKS] = pm;
  const uint64_t lt_mask = (1ull << lane) - 1ull;
  const uint32_t u0 = blockIdx.x * kRouteChunk + wave * (kRouteChunk / kRouteWaves);
#pragma unroll 1
  for (int q = 0; q < kRouteChunk / kRouteWaves / 64; ++q) {
    const uint32_t u = u0 + q * 64 + lane;
    const bool valid = u < U;
    const uint32_t row = valid ? (uint32_t)unique_rows[u] : 0u;
    const uint32_t g = row % G;
    // rank among the lanes of this batch with the same owner: OR-ed lane set (order-independent), as in the keyed sort
    if (valid) __hip_atomic_fetch_or(const_cast<unsigned long long*>(&vp[wave][g]), 1ull << lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __builtin_amdgcn_wave_barrier();
    const uint64_t peers = valid ? vp[wave][g] : 0ull;
    __builtin_amdgcn_wave_barrier();
    if (valid) vp[wave][g] = 0ull;
    __builtin_amdgcn_wave_barrier();
    const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
    uint32_t pos = 0;
    if (valid) pos = vo[wave][g] + rank;
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) vo[wave][g] = pos + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      if (pos < C) {
        send_ids[(size_t)g * C + pos] = (int32_t)(row / G);
        send_u[(size_t)g * C + pos] = (int32_t)u;
        pos_u[u] = (int32_t)(g * C + pos);
      } else {
        pos_u[u] = (int32_t)(G * C);                    // did not fit (flagged in the prologue): the row AFTER the buckets,
      }                                                 // which the caller keeps all-zero -- never another row's embedding
    }
  }
}

// out[i, :] = rows[i] < 0 ? 0 : table[min(rows[i], R - 1), :]   (16-byte lanes; the owner-side gather / gradient hand-over)
// OUT_BF16: rows leave as bf16 (RNE) -- what the bf16 tower input would round them to anyway, at half the wire bytes
template <bool OUT_BF16>
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(const float* __restrict__ table, const int32_t* __restrict__ rows, uint32_t n,
                                                              int32_t R, uint32_t C4, void* __restrict__ out) {
  const uint64_t total = (uint64_t)n * C4;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t i = (uint32_t)(t / C4), c = (uint32_t)(t - (uint64_t)i * C4);
    int32_t r = rows[i];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                  // negative index: a zero row (unused bucket entries)
    if (r >= 0) v = reinterpret_cast<const float4*>(table)[(uint64_t)(r >= R ? R - 1 : r) * C4 + c];
    if (OUT_BF16) {
      ushort4 o;
      o.x = tt_f2bf(v.x); o.y = tt_f2bf(v.y); o.z = tt_f2bf(v.z); o.w = tt_f2bf(v.w);
      reinterpret_cast<ushort4*>(out)[(uint64_t)i * C4 + c] = o;
    } else {
      reinterpret_cast<float4*>(out)[(uint64_t)i * C4 + c] = v;
    }
  }
}

// idx_slot[slot] = pos_u[u] for the slots of plan row u: an 8-lane group per row
__global__ __launch_bounds__(kThreads) void route_expand_kernel(const int32_t* __restrict__ sorted_src, const int32_t* __restrict__ seg,
                                                               const int32_t* __restrict__ n_unique, const int32_t* __restrict__ pos_u,
                                                               int64_t* __restrict__ idx_slot) {
  const uint32_t U = (uint32_t)*n_unique;
  const uint32_t gthread = blockIdx.x * blockDim.x + threadIdx.x, lig = gthread & 7u;
  const uint32_t ngroups = gridDim.x * blockDim.x / 8;
  for (uint32_t u = gthread / 8; u < U; u += ngroups) {
    const int64_t v = pos_u[u];
    for (int32_t p = seg[u] + (int32_t)lig; p < seg[u + 1]; p += 8) idx_slot[sorted_src[p]] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// host helpers
// ------------------------------------------------------------------------------------------------
inline uint32_t pow2_at_least(uint32_t x) {
  uint32_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

inline int grid_for(const tt_ctx* ctx, int64_t threads_needed) {
  const int64_t cap = (int64_t)ctx->num_cus * 8;
  int64_t b = tt_cdiv(threads_needed, kThreads);
  if (b < 1) b = 1;
  return (int)(b < cap ? b : cap);
}

struct DedupWs {
  uint32_t *keysA, *keysB, *valsB, *hist, *total, *blockcount;
  size_t bytes;
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

inline DedupWs dedup_layout(char* base, int64_t M) {
  DedupWs w;
  const uint32_t nblk = (uint32_t)tt_cdiv(M > 0 ? M : 1, kSortTile);
  size_t o = 0;
  auto take = [&](size_t n) { char* p = base ? base + o : nullptr; o += align256(n); return p; };
  w.keysA = reinterpret_cast<uint32_t*>(take(sizeof(uint32_t) * (size_t)(M + 1)));
  w.keysB = reinterpret_cast<uint32_t*>(take(sizeof(uint32_t) * (size_t)(M + 1)));
  w.valsB = reinterpret_cast<uint32_t*>(take(sizeof(uint32_t) * (size_t)(M + 1)));
  w.hist = reinterpret_cast<uint32_t*>(take(sizeof(uint32_t) * (size_t)2048 * nblk));
  w.total = reinterpret_cast<uint32_t*>(take(sizeof(uint32_t) * (size_t)2048));
  w.blockcount = reinterpret_cast<uint32_t*>(take(sizeof(uint32_t) * (size_t)(nblk + 1)));
  w.bytes = o;
  return w;
}

// segment heads of the sorted keys: unique rows, segment offsets, their number -- one chained launch, or count + write
static int launch_heads(tt_ctx* ctx, hipStream_t st, const DedupWs& w, int64_t M, uint32_t nblk, int32_t* n_unique, int32_t* unique_rows,
                        int32_t* seg_offsets, uint32_t drop_from) {
  uint32_t* chain = (int64_t)nblk + 1 <= ctx->chain_words && nblk <= 2048 ? tt_chain_for(ctx, st) : nullptr;
  if (chain) {
    head_chained_kernel<<<nblk, kThreads, 0, st>>>(w.keysA, (uint32_t)M, chain, n_unique, unique_rows, seg_offsets, drop_from, ctx->dev_err,
                                                   ctx->chain_spin);
    TT_LAUNCH_CHECK();
    return TT_OK;
  }
  head_count_kernel<<<nblk, kThreads, 0, st>>>(w.keysA, (uint32_t)M, w.blockcount);
  TT_LAUNCH_CHECK();
  head_write_kernel<<<nblk, kThreads, 0, st>>>(w.keysA, (uint32_t)M, w.blockcount, n_unique, unique_rows, seg_offsets, drop_from);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

struct GradLayout {
  GradWs ws;
  size_t bytes;
  int64_t max_long, max_chunks;
};

inline GradLayout grad_layout(char* base, int64_t M, int32_t E) {
  GradLayout g;
  g.max_long = M / kLongSeg + 1;
  g.max_chunks = M / kLongSeg + g.max_long + 1;
  size_t o = 0;
  auto take = [&](size_t n) { char* p = base ? base + o : nullptr; o += align256(n); return p; };
  g.ws.counters = reinterpret_cast<int32_t*>(take(sizeof(int32_t) * 3));     // chunks allocated, long rows, snapshot of the long-row count
  g.ws.long_row = reinterpret_cast<int32_t*>(take(sizeof(int32_t) * (size_t)g.max_long));
  g.ws.long_base = reinterpret_cast<int32_t*>(take(sizeof(int32_t) * (size_t)g.max_long));
  g.ws.chunk_lo = reinterpret_cast<int32_t*>(take(sizeof(int32_t) * (size_t)g.max_chunks));
  g.ws.chunk_hi = reinterpret_cast<int32_t*>(take(sizeof(int32_t) * (size_t)g.max_chunks));
  g.ws.chunk_partial = reinterpret_cast<float*>(take(sizeof(float) * (size_t)g.max_chunks * (size_t)E));
  g.bytes = o;
  return g;
}

template <int BITS>
int sort_pass(hipStream_t st, const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, uint32_t M, int shift,
              uint32_t* hist, uint32_t* total, uint32_t nblk) {
  sort_hist_kernel<BITS><<<nblk, kThreads, 0, st>>>(kin, M, shift, hist, nblk);
  TT_LAUNCH_CHECK();
  sort_colscan_kernel<BITS><<<(1u << BITS) / 64, 64 * kScanSeg, 0, st>>>(hist, nblk, total);
  TT_LAUNCH_CHECK();
  sort_scatter_kernel<BITS><<<nblk, kThreads, 0, st>>>(kin, vin, kout, vout, M, shift, hist, total);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

AdamK make_adam(int64_t step, float lr, float b1, float b2, float eps, float wd, const float* dev) {
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  AdamK k;
  k.lr_over_bc1 = (float)((double)lr / bc1);
  k.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  k.b1 = b1; k.b2 = b2; k.eps = eps; k.wd = wd;
  k.dev = dev;
  return k;
}

// checks a tt_cvt_list and fills the device form; returns the number of extra grid rows (0 or 1) or < 0
static int fill_cvt(const char* who, const tt_cvt_list* cvt, CvtDev* v, int64_t* widest) {
  v->n = 0;
  if (!cvt || cvt->n == 0) return 0;
  if (cvt->n < 0 || cvt->n > TT_MAX_CVT) {
    tt_set_error("%s: cvt->n = %d not in [0, %d]", who, cvt->n, TT_MAX_CVT);
    return TT_ERR_INVALID_ARG;
  }
  for (int i = 0; i < cvt->n; ++i) {
    if (!(cvt->count[i] >= 0 && (cvt->count[i] == 0 || (cvt->src[i] && cvt->dst[i]))) || !tt_aligned(cvt->src[i], 16) || !tt_aligned(cvt->dst[i], 8)) {
      tt_set_error("%s: conversion %d NULL / misaligned (f32 source 16-byte, bf16 destination 8-byte aligned)", who, i);
      return TT_ERR_INVALID_ARG;
    }
    v->src[i] = cvt->src[i];
    v->dst[i] = reinterpret_cast<uint16_t*>(cvt->dst[i]);
    v->count[i] = cvt->count[i];
    if (cvt->count[i] * 4 > *widest) *widest = cvt->count[i] * 4;       // (bytes of f32 read: sizes the grid like a copy segment)
  }
  v->n = cvt->n;
  return 1;
}

}  // namespace

extern "C" {

#ifdef TT_SEG_STAMPS
int tt_debug_seg_stamps(int clear, unsigned long long* host_out) {
  if (clear) {
    static unsigned long long zeros[4096 * 4];
    return hipMemcpyToSymbol(HIP_SYMBOL(g_seg_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : 1;
  }
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_seg_stamps), sizeof(unsigned long long) * 4096 * 4) == hipSuccess ? 0 : 1;
}
#endif

int tt_embed_lookup_set_profile(tt_ctx* ctx, uint64_t* ring_dev, int32_t n_slots) {
  TT_CHECK_ARG(ctx, "tt_embed_lookup_set_profile: ctx NULL");
  TT_CHECK_ARG(ring_dev == nullptr || n_slots >= 1, "tt_embed_lookup_set_profile: n_slots must be >= 1");
  ctx->lookup_stamps = reinterpret_cast<unsigned long long*>(ring_dev);
  ctx->lookup_stamp_slots = ring_dev ? n_slots : 0;
  return TT_OK;
}

static int lookup_fwd_impl(tt_ctx* ctx, const float* table, int64_t table_rows, int32_t E, const tt_embed_side* sides, int32_t n_sides, int64_t B,
                           int32_t* rows_out, const int32_t* rows_in, tt_stream stream);

int tt_embed_lookup_fwd(tt_ctx* ctx, const float* table, int64_t table_rows, int32_t E, const tt_embed_side* sides,
                        int32_t n_sides, int64_t B, int32_t* rows_out, tt_stream stream) {
  return lookup_fwd_impl(ctx, table, table_rows, E, sides, n_sides, B, rows_out, nullptr, stream);
}

int tt_embed_lookup_rows_fwd(tt_ctx* ctx, const float* table, int64_t table_rows, int32_t E, const tt_embed_side* sides,
                             int32_t n_sides, int64_t B, const int32_t* rows, tt_stream stream) {
  TT_CHECK_ARG(table && rows, "tt_embed_lookup_rows_fwd: NULL table / rows");
  TT_CHECK_ARG(E % 4 == 0 && E / 4 <= 64 && ((E / 4) & (E / 4 - 1)) == 0 && tt_aligned(table, 16),
               "tt_embed_lookup_rows_fwd: E=%d must be 4 x a power of two <= 256 and the table 16-byte aligned", E);
  return lookup_fwd_impl(ctx, table, table_rows, E, sides, n_sides, B, nullptr, rows, stream);
}

static int lookup_fwd_impl(tt_ctx* ctx, const float* table, int64_t table_rows, int32_t E, const tt_embed_side* sides, int32_t n_sides, int64_t B,
                           int32_t* rows_out, const int32_t* rows_in, tt_stream stream) {
  TT_CHECK_ARG(ctx && sides && (table || rows_out), "tt_embed_lookup_fwd: NULL argument");
  TT_CHECK_ARG(n_sides >= 1 && n_sides <= TT_MAX_SIDES, "tt_embed_lookup_fwd: n_sides=%d not in [1,%d]", n_sides, TT_MAX_SIDES);
  TT_CHECK_ARG(E >= 1 && B >= 0 && table_rows >= 1, "tt_embed_lookup_fwd: bad E=%d B=%lld rows=%lld", E, (long long)B, (long long)table_rows);
  TT_CHECK_ARG(table_rows <= INT32_MAX, "tt_embed_lookup_fwd: table_rows %lld exceeds int32 row index", (long long)table_rows);
  SideSet a{};
  a.n = n_sides;
  a.E = E;
  bool vec4 = (E % 4 == 0) && tt_aligned(table, 16);
  int64_t slots = 0;
  for (int i = 0; i < n_sides; ++i) {
    const tt_embed_side& s = sides[i];
    TT_CHECK_ARG(s.K >= 0 && (s.K == 0 || (((s.ids && s.key_row_offset && s.key_vocab) || rows_in) && (s.out || !table))), "tt_embed_lookup_fwd: side %d has NULL pointers", i);
    TT_CHECK_ARG(s.out_dtype == TT_F32 || s.out_dtype == TT_BF16, "tt_embed_lookup_fwd: side %d bad out_dtype %d", i, s.out_dtype);
    TT_CHECK_ARG(s.ld_out >= (int64_t)s.K * E, "tt_embed_lookup_fwd: side %d ld_out %lld < K*E", i, (long long)s.ld_out);
    a.s[i] = SideDev{s.ids, s.key_row_offset, s.key_vocab, reinterpret_cast<char*>(s.out), s.ld_out, (uint32_t)slots, s.K, s.out_dtype, 0u};
    const size_t esz = s.out_dtype == TT_BF16 ? 2 : 4;
    vec4 = vec4 && (s.ld_out % 4 == 0) && tt_aligned(s.out, 4 * esz);
    slots += B * s.K;
  }
  const int64_t C = table ? (vec4 ? E / 4 : E) : 1;
  TT_CHECK_ARG(slots * C < (int64_t)1 << 31, "tt_embed_lookup_fwd: %lld slots x %lld chunks exceeds 2^31 tasks", (long long)slots, (long long)C);
  if (slots == 0) return TT_OK;
  a.C = (uint32_t)C;
  a.total_slots = (uint32_t)slots;
  a.table_rows = (int32_t)table_rows;
  a.dev_err = ctx->dev_err;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  constexpr int U = 4;
  const bool pow2c = vec4 && table && C <= 64 && (C & (C - 1)) == 0;
  if (pow2c) {
    const int spw = 64;                                    // slots per wave pass (32 and 16 measured slower at every C)
    int64_t wg = tt_cdiv(tt_cdiv(slots, spw), kThreads / 64);
    const int64_t cap = (int64_t)ctx->num_cus * 16;
    const int grid = (int)(wg < cap ? wg : cap);
    // two 16-byte pieces per lane when the rows and the outputs allow 32-byte lanes
    bool wide = C >= 2;
    for (int i = 0; i < n_sides; ++i) {
      const size_t esz = sides[i].out_dtype == TT_BF16 ? 2 : 4;
      wide = wide && (sides[i].ld_out % 8 == 0) && tt_aligned(sides[i].out, 8 * esz);
    }
#define TT_LK1(CV, WV) do { if (rows_in) lookup_wave_kernel<CV, spw, WV, true><<<grid, kThreads, 0, st>>>(a, table, rows_out, rows_in, ctx->lookup_stamps, ctx->lookup_stamp_slots); \
                            else lookup_wave_kernel<CV, spw, WV, false><<<grid, kThreads, 0, st>>>(a, table, rows_out, rows_in, ctx->lookup_stamps, ctx->lookup_stamp_slots); } while (0)
#define TT_LK(CV) do { if (wide) TT_LK1(CV, (CV >= 2 ? 2 : 1)); else TT_LK1(CV, 1); } while (0)
    switch (C) {
      case 1: TT_LK1(1, 1); break;
      case 2: TT_LK(2); break;
      case 4: TT_LK(4); break;
      case 8: TT_LK(8); break;
      case 16: TT_LK(16); break;
      case 32: TT_LK(32); break;
      default: TT_LK(64); break;
    }
#undef TT_LK
#undef TT_LK1
    TT_LAUNCH_CHECK();
    return TT_OK;
  }
  if (rows_in) {
    tt_set_error("tt_embed_lookup_rows_fwd: outputs must be 4-element aligned (ld_out and base) for the precomputed-row form");
    return TT_ERR_UNSUPPORTED;
  }
  const int grid = grid_for(ctx, tt_cdiv(slots * C, U));
  if (vec4 && table) lookup_kernel<4, U><<<grid, kThreads, 0, st>>>(a, table, rows_out);
  else lookup_kernel<1, U><<<grid, kThreads, 0, st>>>(a, table, rows_out);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

size_t tt_dedup_workspace_bytes(int64_t M) { return dedup_layout(nullptr, M).bytes; }

int tt_dedup_plan(tt_ctx* ctx, const int32_t* rows, int64_t M, int64_t table_rows, int32_t* sorted_src, int32_t* unique_rows,
                  int32_t* seg_offsets, int32_t* n_unique, void* workspace, size_t workspace_bytes, tt_stream stream) {
  TT_CHECK_ARG(ctx && sorted_src && unique_rows && seg_offsets && n_unique, "tt_dedup_plan: NULL output");
  TT_CHECK_ARG(M >= 0 && M < ((int64_t)1 << 31) - kSortTile, "tt_dedup_plan: M=%lld out of range", (long long)M);
  TT_CHECK_ARG(table_rows >= 1 && table_rows <= INT32_MAX, "tt_dedup_plan: table_rows=%lld out of range", (long long)table_rows);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (M == 0) {
    zero_two_words_kernel<<<1, 64, 0, st>>>(n_unique, seg_offsets);
    TT_LAUNCH_CHECK();
    return TT_OK;
  }
  TT_CHECK_ARG(rows && workspace, "tt_dedup_plan: NULL rows/workspace");
  if (workspace_bytes < tt_dedup_workspace_bytes(M)) {
    tt_set_error("tt_dedup_plan: workspace %zu < required %zu", workspace_bytes, tt_dedup_workspace_bytes(M));
    return TT_ERR_WORKSPACE;
  }
  DedupWs w = dedup_layout(reinterpret_cast<char*>(workspace), M);
  const uint32_t nblk = (uint32_t)tt_cdiv(M, kSortTile);
  int bits = 1;
  while (((int64_t)1 << bits) < table_rows) ++bits;
  // digit plan: fewest passes with 8- or 11-bit digits
  int digit = 8, passes = (bits + 7) / 8;
  if ((bits + 10) / 11 < passes) { digit = 11; passes = (bits + 10) / 11; }
  // ping-pong so that the LAST pass lands values in sorted_src and keys in keysA
  const uint32_t* kin = reinterpret_cast<const uint32_t*>(rows);
  const uint32_t* vin = nullptr;
  for (int p = 0; p < passes; ++p) {
    const bool last_to_A = ((passes - 1 - p) % 2) == 0;
    uint32_t* kout = last_to_A ? w.keysA : w.keysB;
    uint32_t* vout = last_to_A ? reinterpret_cast<uint32_t*>(sorted_src) : w.valsB;
    int rc = digit == 8 ? sort_pass<8>(st, kin, vin, kout, vout, (uint32_t)M, p * digit, w.hist, w.total, nblk)
                        : sort_pass<11>(st, kin, vin, kout, vout, (uint32_t)M, p * digit, w.hist, w.total, nblk);
    if (rc != TT_OK) return rc;
    kin = kout;
    vin = vout;
  }
  return launch_heads(ctx, st, w, M, nblk, n_unique, unique_rows, seg_offsets, 0xFFFFFFFFu);
}

int tt_dedup_plan_runs(tt_ctx* ctx, const int32_t* rows, int32_t G, int64_t C, int64_t row_limit, int32_t* sorted_src,
                       int32_t* unique_rows, int32_t* seg_offsets, int32_t* n_unique, void* workspace, size_t workspace_bytes,
                       tt_stream stream) {
  TT_CHECK_ARG(ctx && rows && sorted_src && unique_rows && seg_offsets && n_unique && workspace, "tt_dedup_plan_runs: NULL argument");
  TT_CHECK_ARG(G >= 1 && G <= TT_MAX_RANKS && C >= 1 && (int64_t)G * C < ((int64_t)1 << 31) - kSortTile, "tt_dedup_plan_runs: bad G / C");
  const int64_t M = (int64_t)G * C;
  if (workspace_bytes < tt_dedup_workspace_bytes(M)) {
    tt_set_error("tt_dedup_plan_runs: workspace %zu < required %zu", workspace_bytes, tt_dedup_workspace_bytes(M));
    return TT_ERR_WORKSPACE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DedupWs w = dedup_layout(reinterpret_cast<char*>(workspace), M);
  const uint32_t nblk = (uint32_t)tt_cdiv(M, kSortTile);
  merge_runs_kernel<<<grid_for(ctx, M), kThreads, 0, st>>>(rows, (uint32_t)G, (uint32_t)C, w.keysA, sorted_src);
  TT_LAUNCH_CHECK();
  return launch_heads(ctx, st, w, M, nblk, n_unique, unique_rows, seg_offsets,
                      row_limit > 0 && row_limit < (int64_t)0xFFFFFFFFll ? (uint32_t)row_limit : 0xFFFFFFFFu);
}

size_t tt_dedup_keyed_workspace_bytes(int64_t M, int32_t n_keys) {
  return align256(sizeof(int32_t) * (size_t)(M > 0 ? M : 1)) * 2 +
         3 * align256(sizeof(int32_t) * (size_t)(n_keys > 0 ? n_keys : 1) * kKeyedMaxParts);      // + head counts, stage bases and ends per (key, share)
}

static int dedup_plan_keyed_impl(tt_ctx* ctx, const int32_t* rows, const int32_t* side_K, int32_t n_sides, int64_t B, int32_t* sorted_src,
                                 int32_t* unique_rows, int32_t* seg_offsets, int32_t* n_unique, void* workspace, size_t workspace_bytes,
                                 tt_stream stream, bool key_major, int32_t E = 0, void* grad_ws = nullptr, size_t grad_ws_bytes = 0) {
  TT_CHECK_ARG(ctx && rows && side_K && sorted_src && unique_rows && seg_offsets && n_unique && workspace, "tt_dedup_plan_keyed: NULL argument");
  TT_CHECK_ARG(n_sides >= 1 && n_sides <= TT_MAX_SIDES, "tt_dedup_plan_keyed: n_sides=%d", n_sides);
  if (B < 1 || B > kKeyedB) {
    tt_set_error("tt_dedup_plan_keyed: B=%lld not in [1, %d]; use tt_dedup_plan", (long long)B, kKeyedB);
    return TT_ERR_UNSUPPORTED;
  }
  KeyedArgs a{};
  a.n_sides = n_sides;
  a.B = (int32_t)B;
  int n_keys = 0;
  int64_t slots = 0;
  for (int i = 0; i < n_sides; ++i) {
    TT_CHECK_ARG(side_K[i] >= 0, "tt_dedup_plan_keyed: negative K");
    a.side_base[i] = (int32_t)slots;
    a.key_base[i] = n_keys;
    a.K[i] = side_K[i];
    n_keys += side_K[i];
    slots += B * side_K[i];
  }
  a.side_base[n_sides] = (int32_t)slots;
  a.key_base[n_sides] = n_keys;
  TT_CHECK_ARG(slots < ((int64_t)1 << 31), "tt_dedup_plan_keyed: too many slots");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n_keys == 0) {
    zero_two_words_kernel<<<1, 64, 0, st>>>(n_unique, seg_offsets);
    TT_LAUNCH_CHECK();
    return TT_OK;
  }
  if (workspace_bytes < tt_dedup_keyed_workspace_bytes(slots, n_keys)) {
    tt_set_error("tt_dedup_plan_keyed: workspace %zu < required %zu", workspace_bytes, tt_dedup_keyed_workspace_bytes(slots, n_keys));
    return TT_ERR_WORKSPACE;
  }
  char* w = reinterpret_cast<char*>(workspace);
  int32_t* uniq_stage = reinterpret_cast<int32_t*>(w);
  int32_t* seg_stage = reinterpret_cast<int32_t*>(w + align256(sizeof(int32_t) * (size_t)slots));
  int32_t* ucount = reinterpret_cast<int32_t*>(w + 2 * align256(sizeof(int32_t) * (size_t)slots));
  int32_t* ubase = reinterpret_cast<int32_t*>(w + 2 * align256(sizeof(int32_t) * (size_t)slots) +
                                              align256(sizeof(int32_t) * (size_t)n_keys * kKeyedMaxParts));
  int32_t* uend = reinterpret_cast<int32_t*>(w + 2 * align256(sizeof(int32_t) * (size_t)slots) +
                                             2 * align256(sizeof(int32_t) * (size_t)n_keys * kKeyedMaxParts));
  PlanLong pl{};
  if (grad_ws) {                                         // the gradient reduction's own workspace layout: its lists are filled here
    if (E < 1 || grad_ws_bytes < tt_embed_grad_workspace_bytes(slots, E)) {
      tt_set_error("tt_dedup_plan_keyed_long: gradient workspace %zu < required %zu", grad_ws_bytes, tt_embed_grad_workspace_bytes(slots, E));
      return TT_ERR_WORKSPACE;
    }
    const GradLayout gl = grad_layout(reinterpret_cast<char*>(grad_ws), slots, E);
    pl = PlanLong{gl.ws.counters, gl.ws.long_row, gl.ws.long_base, gl.ws.chunk_lo, gl.ws.chunk_hi};
  }
  // workgroups per key (TT_OPT_KEYED_PARTS overrides): the sort's scatter, ranking and output phases split P ways, the load and
  // histogram phases are repeated by every share; small batches are launch-bound anyway
  int parts = ctx->keyed_parts > 0 ? ctx->keyed_parts : (B >= 2048 ? 6 : 1);                        // (38 keys: 4 shares 17.9 us, 5: 17.4, 6: 17.2)
  if (parts > kKeyedMaxParts) parts = kKeyedMaxParts;
  while (parts > 1 && (int64_t)n_keys * parts > (int64_t)ctx->num_cus) --parts;        // a workgroup needs a CU of its own (144 KB of LDS)
  a.parts = parts;
  keyed_sort_kernel<<<n_keys * parts, kKeyedThreads, 0, st>>>(a, rows, sorted_src, uniq_stage, seg_stage, ucount, ubase, uend, key_major,
                                                              pl.counters);
  TT_LAUNCH_CHECK();
  const CompactRider cr{uniq_stage, seg_stage, ucount, ubase, uend, pl, n_keys * parts, slots, unique_rows, seg_offsets, n_unique};
  if (ctx->defer_riders & 1) {                           // rides beside the towers' tail_fwd (tt_riders.h); a second plan before that
    if (ctx->riders->c_wg > 0)                           // launch takes the queue's place, the older one is launched now
      if (int rc = tt_riders_flush(ctx, st)) return rc;
    ctx->riders->c = cr;
    ctx->riders->c_wg = n_keys * parts;
    return TT_OK;
  }
  keyed_compact_kernel<<<n_keys * parts, kKeyedThreads, 0, st>>>(cr);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

int tt_dedup_plan_keyed(tt_ctx* ctx, const int32_t* rows, const int32_t* side_K, int32_t n_sides, int64_t B, int32_t* sorted_src,
                        int32_t* unique_rows, int32_t* seg_offsets, int32_t* n_unique, void* workspace, size_t workspace_bytes,
                        tt_stream stream) {
  return dedup_plan_keyed_impl(ctx, rows, side_K, n_sides, B, sorted_src, unique_rows, seg_offsets, n_unique, workspace, workspace_bytes,
                               stream, false);
}

int tt_dedup_plan_keyed_long(tt_ctx* ctx, const int32_t* rows, int32_t rows_key_major, const int32_t* side_K, int32_t n_sides, int64_t B,
                             int32_t E, int32_t* sorted_src, int32_t* unique_rows, int32_t* seg_offsets, int32_t* n_unique,
                             void* grad_workspace, size_t grad_workspace_bytes, void* workspace, size_t workspace_bytes, tt_stream stream) {
  TT_CHECK_ARG(grad_workspace, "tt_dedup_plan_keyed_long: NULL gradient workspace");
  return dedup_plan_keyed_impl(ctx, rows, side_K, n_sides, B, sorted_src, unique_rows, seg_offsets, n_unique, workspace, workspace_bytes,
                               stream, rows_key_major != 0, E, grad_workspace, grad_workspace_bytes);
}

int tt_dedup_plan_keyed_km(tt_ctx* ctx, const int32_t* rows_km, const int32_t* side_K, int32_t n_sides, int64_t B, int32_t* sorted_src,
                           int32_t* unique_rows, int32_t* seg_offsets, int32_t* n_unique, void* workspace, size_t workspace_bytes,
                           tt_stream stream) {
  return dedup_plan_keyed_impl(ctx, rows_km, side_K, n_sides, B, sorted_src, unique_rows, seg_offsets, n_unique, workspace, workspace_bytes,
                               stream, true);
}

size_t tt_embed_grad_workspace_bytes(int64_t M, int32_t E) { return grad_layout(nullptr, M > 0 ? M : 1, E > 0 ? E : 1).bytes; }

int tt_embed_grad_bwd(tt_ctx* ctx, const tt_grad_src* srcs, int32_t n_srcs, int64_t B, int32_t E, const int32_t* sorted_src,
                      const int32_t* seg_offsets, const int32_t* unique_rows, const int32_t* n_unique, int64_t M, int32_t mode,
                      float* out, int32_t* counters, void* workspace, size_t workspace_bytes, tt_stream stream) {
  TT_CHECK_ARG(ctx && srcs && out, "tt_embed_grad_bwd: NULL argument");
  TT_CHECK_ARG(n_srcs >= 1 && n_srcs <= TT_MAX_SIDES, "tt_embed_grad_bwd: n_srcs=%d", n_srcs);
  // TT_GRAD_SHORT_SEGMENTS: every segment is summed by its own lane group whatever its length -- the right choice (and
  // three launches fewer) when the caller knows no segment is long, e.g. the owner side of the row exchange, where a
  // row arrives at most once per rank.  Results do not depend on the flag.
  const bool all_short = (mode & TT_GRAD_SHORT_SEGMENTS) != 0;
  const bool planned = (mode & TT_GRAD_PLANNED) != 0 && !all_short;      // the workspace holds the plan's long-row list and counters
  // TT_GRAD_DEFER_FINISH: the long rows' chunk partials are left unadded -- tt_adam_fused_step_finish (or tt_embed_grad_finish)
  // on the same workspace completes `out`; sparse mode on a planned workspace only
  const bool defer = (mode & TT_GRAD_DEFER_FINISH) != 0;
  mode &= ~(TT_GRAD_SHORT_SEGMENTS | TT_GRAD_PLANNED | TT_GRAD_DEFER_FINISH);
  TT_CHECK_ARG(!defer || (planned && mode == TT_GRAD_SPARSE), "tt_embed_grad_bwd: TT_GRAD_DEFER_FINISH needs TT_GRAD_PLANNED | TT_GRAD_SPARSE");
  TT_CHECK_ARG(mode >= TT_GRAD_SPARSE && mode <= TT_GRAD_DENSE_ACC, "tt_embed_grad_bwd: bad mode %d", mode);
  TT_CHECK_ARG(E >= 1 && B >= 0, "tt_embed_grad_bwd: bad E/B");
  if (M == 0) return TT_OK;
  TT_CHECK_ARG(sorted_src && seg_offsets && unique_rows && n_unique && workspace, "tt_embed_grad_bwd: NULL plan/workspace");
  if (workspace_bytes < tt_embed_grad_workspace_bytes(M, E)) {
    tt_set_error("tt_embed_grad_bwd: workspace %zu < required %zu", workspace_bytes, tt_embed_grad_workspace_bytes(M, E));
    return TT_ERR_WORKSPACE;
  }
  SideSet a{};
  a.n = n_srcs;
  a.E = E;
  bool vec4 = (E % 4 == 0) && tt_aligned(out, 16);
  int64_t slots = 0;
  for (int i = 0; i < n_srcs; ++i) {
    const tt_grad_src& s = srcs[i];
    TT_CHECK_ARG(s.K == 0 || s.d_out, "tt_embed_grad_bwd: src %d NULL", i);
    TT_CHECK_ARG(s.dtype == TT_F32 || s.dtype == TT_BF16, "tt_embed_grad_bwd: src %d bad dtype", i);
    TT_CHECK_ARG(s.dtype == srcs[0].dtype, "tt_embed_grad_bwd: all sources must share one dtype (src %d differs)", i);
    a.s[i] = SideDev{nullptr, nullptr, nullptr, const_cast<char*>(reinterpret_cast<const char*>(s.d_out)), s.ld, (uint32_t)slots, s.K, s.dtype, s.K > 1 ? (uint32_t)(0x100000000ull / (uint64_t)s.K) : 0xFFFFFFFFu};   // K = 1: 2^32 - 1, the fix-up step covers it
    const size_t esz = s.dtype == TT_BF16 ? 2 : 4;
    vec4 = vec4 && (s.ld % 4 == 0) && tt_aligned(s.d_out, 4 * esz);
    slots += B * s.K;
  }
  TT_CHECK_ARG(slots == M, "tt_embed_grad_bwd: sum(B*K)=%lld != M=%lld", (long long)slots, (long long)M);
  a.C = (uint32_t)(vec4 ? E / 4 : E);
  a.total_slots = (uint32_t)slots;
  const uint32_t LG = pow2_at_least(a.C) > 64 ? 64 : pow2_at_least(a.C);
  const int dt = srcs[0].dtype;
  GradLayout gl = grad_layout(reinterpret_cast<char*>(workspace), M, E);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (int rc = tt_riders_flush(ctx, st)) return rc;      // a plan compaction nobody hosted: this reduction reads its output
  if (planned) {
    // counters and lists were written into THIS workspace by tt_dedup_plan_keyed_long: nothing to zero, nothing to register
  } else if (counters) {
    gl.ws.counters = counters;      // caller-kept, zero on entry: the finish kernel re-zeroes them (one launch fewer per step)
  } else if (!all_short) {
    zero_words_kernel<<<1, 64, 0, st>>>(gl.ws.counters, 2);    // (a kernel, not a memset node: see graph notes in DESIGN.md)
    TT_LAUNCH_CHECK();
  }
  // a slab reduction the tower backward left in the context: inside this launch when the workspace is the plan's own (nothing
  // here then writes the shared scratch the slabs live in), launched on its own first otherwise
  TnPending* slabs = (ctx->deferred && ctx->deferred->n > 0) ? ctx->deferred : nullptr;
  if (slabs && !planned) {
    if (int rc = tt_gemm_deferred_flush(ctx, st)) return rc;
    slabs = nullptr;
  }
  const int nsx = slabs ? tt_slab_role_blocks_x(slabs) : 0;
  const int g1 = grid_for(ctx, M * LG);
  const int g2 = grid_for(ctx, gl.max_chunks * LG);
  const int g3 = (int)(gl.max_long < (int64_t)ctx->num_cus * 8 ? gl.max_long : (int64_t)ctx->num_cus * 8);   // a workgroup per long row
  if ((int64_t)(kThreads / LG) * E > kFinishMaxFloats) {
    tt_set_error("tt_embed_grad_bwd: E=%d too wide for the long-row finish (max %d)", E, kFinishMaxFloats * (int)LG / kThreads);
    return TT_ERR_UNSUPPORTED;
  }
  // all sources share one element type (checked above): it is a template parameter of the kernels, and so is
  // the lane-group width when every lane of a group owns exactly one chunk (shared decode, see sum_range)
#define TT_SEG_LAUNCH(V, D, G)                                                                                                  \
  do {                                                                                                                          \
    if (planned && slabs) {                                                                                                     \
      const int sc = kSegThreads / kThreads;                                                                                  \
      const int g1s = (int)tt_cdiv(g1, sc), g2s = (int)tt_cdiv(g2, sc), nsxs = (int)tt_cdiv(nsx, sc);                           \
      const int nss = tt_slab_role_blocks(slabs, nsxs);                                                                         \
      seg_reduce_chunk_slab_kernel<V, D, G><<<nss + g1s + g2s, kSegThreads, 0, st>>>(                                           \
          a, sorted_src, seg_offsets, unique_rows, n_unique, mode, out, gl.ws, LG, (uint32_t)g1s, slabs->sb, (uint32_t)nsxs,    \
          (uint32_t)slabs->n, (uint32_t)nss);                                                                                   \
      TT_LAUNCH_CHECK();                                                                                                        \
      slabs->n = 0;                                                                                                             \
      slabs->maxtotal = 1;                                                                                                      \
      if (!defer) seg_long_finish_kernel<V><<<g3, kThreads, 0, st>>>(E, a.C, seg_offsets, unique_rows, mode, out, gl.ws, LG);   \
      break;                                                                                                                    \
    }                                                                                                                           \
    if (planned) {                                                                                                              \
      seg_reduce_chunk_kernel<V, D, G><<<g1 + g2, kThreads, 0, st>>>(a, sorted_src, seg_offsets, unique_rows, n_unique, mode,   \
                                                                     out, gl.ws, LG, (uint32_t)g1);                             \
      TT_LAUNCH_CHECK();                                                                                                        \
      if (!defer) seg_long_finish_kernel<V><<<g3, kThreads, 0, st>>>(E, a.C, seg_offsets, unique_rows, mode, out, gl.ws, LG);   \
      break;                                                                                                                    \
    }                                                                                                                           \
    seg_reduce_kernel<V, D, G><<<g1, kThreads, 0, st>>>(a, sorted_src, seg_offsets, unique_rows, n_unique, mode, out, gl.ws, LG, \
                                                        all_short);                                                             \
    TT_LAUNCH_CHECK();                                                                                                          \
    if (all_short) break;                                                                                                       \
    seg_chunk_kernel<V, D, G><<<g2, kThreads, 0, st>>>(a, sorted_src, gl.ws, LG);                                                \
    TT_LAUNCH_CHECK();                                                                                                          \
    seg_long_finish_kernel<V><<<g3, kThreads, 0, st>>>(E, a.C, seg_offsets, unique_rows, mode, out, gl.ws, LG);                 \
  } while (0)
#define TT_SEG_LAUNCH_G(V, D)                              \
  do {                                                     \
    if (lgt == 8) TT_SEG_LAUNCH(V, D, 8);                  \
    else if (lgt == 16) TT_SEG_LAUNCH(V, D, 16);           \
    else if (lgt == 4) TT_SEG_LAUNCH(V, D, 4);             \
    else TT_SEG_LAUNCH(V, D, 0);                           \
  } while (0)
  const int lgt = (vec4 && a.C == LG && (LG == 4 || LG == 8 || LG == 16)) ? (int)LG : 0;
  if (vec4 && dt == TT_F32) TT_SEG_LAUNCH_G(4, TT_F32);
  else if (vec4) TT_SEG_LAUNCH_G(4, TT_BF16);
  else if (dt == TT_F32) TT_SEG_LAUNCH(1, TT_F32, 0);
  else TT_SEG_LAUNCH(1, TT_BF16, 0);
#undef TT_SEG_LAUNCH_G
#undef TT_SEG_LAUNCH
  TT_LAUNCH_CHECK();
  return TT_OK;
}

void tt_adam_hparams(int64_t step, float lr, float beta1, float beta2, float eps, float weight_decay, float out6[6]) {
  const AdamK k = make_adam(step < 1 ? 1 : step, lr, beta1, beta2, eps, weight_decay, nullptr);
  out6[0] = k.lr_over_bc1; out6[1] = k.inv_sqrt_bc2; out6[2] = k.b1; out6[3] = k.b2; out6[4] = k.eps; out6[5] = k.wd;
}

int tt_adam_dense_step(tt_ctx* ctx, float* p, const float* g, float* m, float* v, int64_t n, int64_t step, float lr, float beta1,
                       float beta2, float eps, float weight_decay, const float* hparams_dev, tt_stream stream) {
  if (int rc = tt_riders_flush(ctx, reinterpret_cast<hipStream_t>(stream))) return rc;       // (a queued plan compaction: its rows are read here)
  if (ctx && ctx->deferred && ctx->deferred->n > 0)      // a queued slab reduction: the gradients are not complete before it
    if (int rc = tt_gemm_deferred_flush(ctx, reinterpret_cast<hipStream_t>(stream))) return rc;
  TT_CHECK_ARG(ctx && (n == 0 || (p && g && m && v)), "tt_adam_dense_step: NULL argument");
  TT_CHECK_ARG(step >= 1 && n >= 0, "tt_adam_dense_step: step must be >= 1");
  if (n == 0) return TT_OK;
  const AdamK k = make_adam(step, lr, beta1, beta2, eps, weight_decay, hparams_dev);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (n % 4 == 0 && tt_aligned(p, 16) && tt_aligned(g, 16) && tt_aligned(m, 16) && tt_aligned(v, 16)) {
    adam_dense_vec4_kernel<<<grid_for(ctx, n / 4), kThreads, 0, st>>>(reinterpret_cast<float4*>(p), reinterpret_cast<const float4*>(g),
                                                                      reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v), n / 4, k);
  } else {
    adam_dense_kernel<<<grid_for(ctx, n), kThreads, 0, st>>>(p, g, m, v, n, k);
  }
  TT_LAUNCH_CHECK();
  return TT_OK;
}

int tt_adam_multi_step(tt_ctx* ctx, const tt_adam_tensor* tensors, int32_t n_tensors, int64_t step, float lr, float beta1,
                       float beta2, float eps, float weight_decay, const float* hparams_dev, tt_stream stream) {
  if (int rc = tt_riders_flush(ctx, reinterpret_cast<hipStream_t>(stream))) return rc;       // (a queued plan compaction: its rows are read here)
  if (ctx && ctx->deferred && ctx->deferred->n > 0)      // a queued slab reduction: the gradients are not complete before it
    if (int rc = tt_gemm_deferred_flush(ctx, reinterpret_cast<hipStream_t>(stream))) return rc;
  TT_CHECK_ARG(ctx && (n_tensors == 0 || tensors), "tt_adam_multi_step: NULL argument");
  TT_CHECK_ARG(step >= 1 && n_tensors >= 0, "tt_adam_multi_step: step must be >= 1");
  const AdamK k = make_adam(step, lr, beta1, beta2, eps, weight_decay, hparams_dev);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int base = 0; base < n_tensors; base += kAdamMulti) {
    AdamMultiArgs a{};
    const int cnt = n_tensors - base < kAdamMulti ? n_tensors - base : kAdamMulti;
    int64_t nmax = 1;
    for (int i = 0; i < cnt; ++i) {
      const tt_adam_tensor& t = tensors[base + i];
      TT_CHECK_ARG(t.n >= 0 && (t.n == 0 || (t.p && t.g && t.m && t.v)), "tt_adam_multi_step: tensor %d has NULL pointers", base + i);
      a.t[i] = t;
      nmax = t.n > nmax ? t.n : nmax;
    }
    int64_t gx = tt_cdiv(nmax, kThreads);
    if (gx > 64) gx = 64;
    adam_multi_kernel<<<dim3((unsigned)gx, (unsigned)cnt), kThreads, 0, st>>>(a, k);
    TT_LAUNCH_CHECK();
  }
  return TT_OK;
}

int tt_sparse_adam_step(tt_ctx* ctx, float* table, float* m, float* v, int64_t table_rows, int32_t E, const int32_t* unique_rows,
                        const float* grad_rows, const int32_t* n_unique, int64_t M, int64_t step, float lr, float beta1, float beta2, float eps,
                        float weight_decay, const float* hparams_dev, tt_stream stream) {
  if (int rc = tt_riders_flush(ctx, reinterpret_cast<hipStream_t>(stream))) return rc;       // (a queued plan compaction: its rows are read here)
  if (ctx && ctx->deferred && ctx->deferred->n > 0)      // a queued slab reduction: the gradients are not complete before it
    if (int rc = tt_gemm_deferred_flush(ctx, reinterpret_cast<hipStream_t>(stream))) return rc;
  TT_CHECK_ARG(ctx && table && m && v, "tt_sparse_adam_step: NULL state");
  TT_CHECK_ARG(step >= 1 && E >= 1 && M >= 0 && table_rows >= 1, "tt_sparse_adam_step: bad step/E/M/table_rows");
  if (M == 0) return TT_OK;
  TT_CHECK_ARG(unique_rows && grad_rows && n_unique, "tt_sparse_adam_step: NULL plan");
  const AdamK k = make_adam(step, lr, beta1, beta2, eps, weight_decay, hparams_dev);
  const bool vec4 = (E % 4 == 0) && tt_aligned(table, 16) && tt_aligned(m, 16) && tt_aligned(v, 16) && tt_aligned(grad_rows, 16);
  const uint32_t C = vec4 ? E / 4 : E;
  const uint32_t LG = pow2_at_least(C) > 64 ? 64 : pow2_at_least(C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for(ctx, M * LG);
  if (vec4) adam_sparse_kernel<4><<<grid, kThreads, 0, st>>>(table, m, v, E, C, unique_rows, grad_rows, n_unique, k, LG, table_rows);
  else adam_sparse_kernel<1><<<grid, kThreads, 0, st>>>(table, m, v, E, C, unique_rows, grad_rows, n_unique, k, LG, table_rows);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

static int adam_fused_impl(tt_ctx* ctx, const tt_adam_tensor* tensors, int32_t n_tensors, float* table, float* m, float* v,
                           int64_t table_rows, int32_t E, const int32_t* unique_rows, float* grad_rows, const int32_t* n_unique, int64_t M,
                           const int32_t* seg_offsets, void* grad_workspace, size_t grad_workspace_bytes, int64_t step, float lr,
                           float beta1, float beta2, float eps, float weight_decay, const float* hparams_dev, tt_stream stream,
                           const char* who) {
  if (int rc = tt_riders_flush(ctx, reinterpret_cast<hipStream_t>(stream))) return rc;       // (a queued plan compaction: its rows are read here)
  if (ctx && ctx->deferred && ctx->deferred->n > 0)      // a queued slab reduction: the gradients are not complete before it
    if (int rc = tt_gemm_deferred_flush(ctx, reinterpret_cast<hipStream_t>(stream))) return rc;
  TT_CHECK_ARG(ctx && tensors && table && m && v && unique_rows && grad_rows && n_unique, "%s: NULL argument", who);
  TT_CHECK_ARG(n_tensors >= 1 && n_tensors <= kAdamMulti, "%s: n_tensors=%d not in [1,%d]", who, n_tensors, kAdamMulti);
  TT_CHECK_ARG(step >= 1 && E >= 1 && M >= 1 && table_rows >= 1, "%s: bad step/E/M/table_rows", who);
  AdamFusedArgs a{};
  a.n = n_tensors;
  int nd = 0;
  for (int i = 0; i < n_tensors; ++i) {
    const tt_adam_tensor& t = tensors[i];
    TT_CHECK_ARG(t.n >= 0 && (t.n == 0 || (t.p && t.g && t.m && t.v)), "%s: tensor %d has NULL pointers", who, i);
    a.t[i] = t;
    a.blk0[i] = nd;
    int64_t nb = tt_cdiv(t.n > 0 ? t.n : 1, kThreads);
    nd += (int)(nb > 64 ? 64 : nb);
  }
  a.blk0[n_tensors] = nd;
  const AdamK k = make_adam(step, lr, beta1, beta2, eps, weight_decay, hparams_dev);
  const bool vec4 = (E % 4 == 0) && tt_aligned(table, 16) && tt_aligned(m, 16) && tt_aligned(v, 16) && tt_aligned(grad_rows, 16);
  const uint32_t C = vec4 ? E / 4 : E;
  const uint32_t LG = pow2_at_least(C) > 64 ? 64 : pow2_at_least(C);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (grad_workspace) {
    // the reduction's deferred finish (tt_embed_grad_bwd with TT_GRAD_DEFER_FINISH) rides in this launch
    TT_CHECK_ARG(seg_offsets, "%s: NULL seg_offsets", who);
    if (grad_workspace_bytes < tt_embed_grad_workspace_bytes(M, E)) {
      tt_set_error("%s: gradient workspace %zu < required %zu", who, grad_workspace_bytes, tt_embed_grad_workspace_bytes(M, E));
      return TT_ERR_WORKSPACE;
    }
    if ((int64_t)(kThreads / LG) * E > kFinishMaxFloats) {
      tt_set_error("%s: E=%d too wide for the long-row finish (max %d)", who, E, kFinishMaxFloats * (int)LG / kThreads);
      return TT_ERR_UNSUPPORTED;
    }
    const GradLayout gl = grad_layout(reinterpret_cast<char*>(grad_workspace), M, E);
    const int nlb = (int)(gl.max_long < (int64_t)ctx->num_cus * 8 ? gl.max_long : (int64_t)ctx->num_cus * 8);
    const int grid = nd + nlb + grid_for(ctx, M * LG);
    if (vec4) adam_fused_kernel<4, true><<<grid, kThreads, 0, st>>>(a, table, m, v, E, C, unique_rows, grad_rows, n_unique, k, LG, table_rows, seg_offsets, gl.ws, nlb);
    else adam_fused_kernel<1, true><<<grid, kThreads, 0, st>>>(a, table, m, v, E, C, unique_rows, grad_rows, n_unique, k, LG, table_rows, seg_offsets, gl.ws, nlb);
    TT_LAUNCH_CHECK();
    return TT_OK;
  }
  const int grid = nd + grid_for(ctx, M * LG);
  if (vec4) adam_fused_kernel<4, false><<<grid, kThreads, 0, st>>>(a, table, m, v, E, C, unique_rows, grad_rows, n_unique, k, LG, table_rows, nullptr, GradWs{}, 0);
  else adam_fused_kernel<1, false><<<grid, kThreads, 0, st>>>(a, table, m, v, E, C, unique_rows, grad_rows, n_unique, k, LG, table_rows, nullptr, GradWs{}, 0);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

int tt_adam_fused_step(tt_ctx* ctx, const tt_adam_tensor* tensors, int32_t n_tensors, float* table, float* m, float* v, int64_t table_rows,
                       int32_t E,
                       const int32_t* unique_rows, const float* grad_rows, const int32_t* n_unique, int64_t M, int64_t step, float lr,
                       float beta1, float beta2, float eps, float weight_decay, const float* hparams_dev, tt_stream stream) {
  return adam_fused_impl(ctx, tensors, n_tensors, table, m, v, table_rows, E, unique_rows, const_cast<float*>(grad_rows), n_unique, M,
                         nullptr, nullptr, 0, step, lr, beta1, beta2, eps, weight_decay, hparams_dev, stream, "tt_adam_fused_step");
}

int tt_adam_fused_step_finish(tt_ctx* ctx, const tt_adam_tensor* tensors, int32_t n_tensors, float* table, float* m, float* v,
                              int64_t table_rows, int32_t E, const int32_t* unique_rows, float* grad_rows, const int32_t* n_unique,
                              int64_t M, const int32_t* seg_offsets, void* grad_workspace, size_t grad_workspace_bytes, int64_t step,
                              float lr, float beta1, float beta2, float eps, float weight_decay, const float* hparams_dev,
                              tt_stream stream) {
  TT_CHECK_ARG(grad_workspace, "tt_adam_fused_step_finish: NULL gradient workspace");
  return adam_fused_impl(ctx, tensors, n_tensors, table, m, v, table_rows, E, unique_rows, grad_rows, n_unique, M, seg_offsets,
                         grad_workspace, grad_workspace_bytes, step, lr, beta1, beta2, eps, weight_decay, hparams_dev, stream,
                         "tt_adam_fused_step_finish");
}

int tt_embed_grad_finish(tt_ctx* ctx, int32_t E, const int32_t* seg_offsets, int64_t M, float* out, void* workspace,
                         size_t workspace_bytes, tt_stream stream) {
  TT_CHECK_ARG(ctx && seg_offsets && out && workspace, "tt_embed_grad_finish: NULL argument");
  TT_CHECK_ARG(E >= 1 && M >= 1, "tt_embed_grad_finish: bad E/M");
  if (workspace_bytes < tt_embed_grad_workspace_bytes(M, E)) {
    tt_set_error("tt_embed_grad_finish: workspace %zu < required %zu", workspace_bytes, tt_embed_grad_workspace_bytes(M, E));
    return TT_ERR_WORKSPACE;
  }
  const bool vec4 = (E % 4 == 0) && tt_aligned(out, 16);
  const uint32_t C = vec4 ? E / 4 : E;
  const uint32_t LG = pow2_at_least(C) > 64 ? 64 : pow2_at_least(C);
  const GradLayout gl = grad_layout(reinterpret_cast<char*>(workspace), M, E);
  const int g3 = (int)(gl.max_long < (int64_t)ctx->num_cus * 8 ? gl.max_long : (int64_t)ctx->num_cus * 8);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (vec4) seg_long_finish_kernel<4><<<g3, kThreads, 0, st>>>(E, C, seg_offsets, nullptr, TT_GRAD_SPARSE, out, gl.ws, LG);
  else seg_long_finish_kernel<1><<<g3, kThreads, 0, st>>>(E, C, seg_offsets, nullptr, TT_GRAD_SPARSE, out, gl.ws, LG);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

size_t tt_route_workspace_bytes(int64_t M, int32_t G) {
  const int64_t nb = tt_cdiv(M > 0 ? M : 1, kRouteChunk);
  return align256(sizeof(uint32_t) * (size_t)nb * kRouteWaves * (size_t)(G > 0 ? G : 1));
}

static int route_bucket_impl(tt_ctx* ctx, const int32_t* unique_rows, const int32_t* n_unique, int64_t M, int32_t G, int32_t C,
                             const int32_t* pad_id, int32_t pad_u, int32_t* send_ids, int32_t* send_u, int32_t* pos_u, int32_t* counts,
                             int32_t* overflow, void* workspace, size_t workspace_bytes, const int32_t* sorted_src,
                             const int32_t* seg_offsets, int64_t* idx_slot, tt_stream stream) {
  TT_CHECK_ARG(ctx && unique_rows && n_unique && pad_id && send_ids && send_u && pos_u && counts && overflow && workspace,
               "tt_route_bucket: NULL argument");
  TT_CHECK_ARG(M >= 1 && M < ((int64_t)1 << 31) && G >= 1 && G <= TT_MAX_RANKS && C >= 1 && (int64_t)G * C < ((int64_t)1 << 31),
               "tt_route_bucket: bad M / G / C");
  TT_CHECK_ARG(idx_slot == nullptr || (sorted_src && seg_offsets), "tt_route_bucket_expand: NULL plan arrays");
  if (workspace_bytes < tt_route_workspace_bytes(M, G)) {
    tt_set_error("tt_route_bucket: workspace %zu < required %zu", workspace_bytes, tt_route_workspace_bytes(M, G));
    return TT_ERR_WORKSPACE;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const unsigned nb = (unsigned)tt_cdiv(M, kRouteChunk);
  uint32_t* seg = reinterpret_cast<uint32_t*>(workspace);
  RoutePads pads{};
  for (int g = 0; g < G; ++g) pads.id[g] = pad_id[g];
  route_count_kernel<<<nb, kThreads, 0, st>>>(unique_rows, n_unique, (uint32_t)G, seg);
  TT_LAUNCH_CHECK();
  route_scatter_kernel<<<nb, kThreads, 0, st>>>(unique_rows, n_unique, (uint32_t)G, (uint32_t)C, seg, counts, overflow, pads, pad_u, send_ids,
                                                send_u, pos_u);
  TT_LAUNCH_CHECK();
  if (idx_slot != nullptr) {
    route_expand_kernel<<<grid_for(ctx, M * 8), kThreads, 0, st>>>(sorted_src, seg_offsets, n_unique, pos_u, idx_slot);
    TT_LAUNCH_CHECK();
  }
  return TT_OK;
}

int tt_route_bucket(tt_ctx* ctx, const int32_t* unique_rows, const int32_t* n_unique, int64_t M, int32_t G, int32_t C,
                    const int32_t* pad_id, int32_t pad_u, int32_t* send_ids, int32_t* send_u, int32_t* pos_u, int32_t* counts,
                    int32_t* overflow, void* workspace, size_t workspace_bytes, tt_stream stream) {
  return route_bucket_impl(ctx, unique_rows, n_unique, M, G, C, pad_id, pad_u, send_ids, send_u, pos_u, counts, overflow, workspace,
                           workspace_bytes, nullptr, nullptr, nullptr, stream);
}

int tt_route_bucket_expand(tt_ctx* ctx, const int32_t* unique_rows, const int32_t* n_unique, int64_t M, int32_t G, int32_t C,
                           const int32_t* pad_id, int32_t pad_u, int32_t* send_ids, int32_t* send_u, int32_t* pos_u, int32_t* counts,
                           int32_t* overflow, void* workspace, size_t workspace_bytes, const int32_t* sorted_src,
                           const int32_t* seg_offsets, int64_t* idx_slot, tt_stream stream) {
  TT_CHECK_ARG(idx_slot != nullptr, "tt_route_bucket_expand: NULL idx_slot");
  return route_bucket_impl(ctx, unique_rows, n_unique, M, G, C, pad_id, pad_u, send_ids, send_u, pos_u, counts, overflow, workspace,
                           workspace_bytes, sorted_src, seg_offsets, idx_slot, stream);
}

int tt_gather_rows(tt_ctx* ctx, const float* table, int64_t table_rows, int32_t E, const int32_t* rows, int64_t n, void* out,
                   int32_t out_dtype, tt_stream stream) {
  TT_CHECK_ARG(ctx && table && rows && out, "tt_gather_rows: NULL argument");
  TT_CHECK_ARG(table_rows >= 1 && table_rows <= INT32_MAX && n >= 0 && n < ((int64_t)1 << 31) && E >= 4 && E % 4 == 0,
               "tt_gather_rows: bad shape (E must be a multiple of 4)");
  TT_CHECK_ARG(out_dtype == TT_F32 || out_dtype == TT_BF16, "tt_gather_rows: bad out_dtype");
  TT_CHECK_ARG(tt_aligned(table, 16) && tt_aligned(out, out_dtype == TT_BF16 ? 8 : 16), "tt_gather_rows: table / out alignment");
  if (n == 0) return TT_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int grid = grid_for(ctx, n * (E / 4));
  if (out_dtype == TT_BF16) gather_rows_kernel<true><<<grid, kThreads, 0, st>>>(table, rows, (uint32_t)n, (int32_t)table_rows, (uint32_t)(E / 4), out);
  else gather_rows_kernel<false><<<grid, kThreads, 0, st>>>(table, rows, (uint32_t)n, (int32_t)table_rows, (uint32_t)(E / 4), out);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

int tt_route_expand(tt_ctx* ctx, const int32_t* sorted_src, const int32_t* seg_offsets, const int32_t* n_unique, const int32_t* pos_u,
                    int64_t M, int64_t* idx_slot, tt_stream stream) {
  TT_CHECK_ARG(ctx && sorted_src && seg_offsets && n_unique && pos_u && idx_slot, "tt_route_expand: NULL argument");
  TT_CHECK_ARG(M >= 1, "tt_route_expand: M < 1");
  route_expand_kernel<<<grid_for(ctx, M * 8), kThreads, 0, reinterpret_cast<hipStream_t>(stream)>>>(sorted_src, seg_offsets, n_unique, pos_u,
                                                                                                      idx_slot);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

int tt_copy_multi(tt_ctx* ctx, int32_t n, void* const* dst, const void* const* src, const int64_t* bytes, tt_stream stream) {
  TT_CHECK_ARG(ctx && n >= 0 && n <= TT_MAX_COPIES && (n == 0 || (dst && src && bytes)), "tt_copy_multi: bad arguments");
  if (n == 0) return TT_OK;
  CopyArgs a{};
  int64_t mx = 0;
  for (int i = 0; i < n; ++i) {
    TT_CHECK_ARG(bytes[i] >= 0 && (bytes[i] == 0 || (dst[i] && src[i])), "tt_copy_multi: segment %d NULL", i);
    TT_CHECK_ARG(tt_aligned(dst[i], 16) && tt_aligned(src[i], 16), "tt_copy_multi: segment %d not 16-byte aligned", i);
    a.dst[i] = reinterpret_cast<char*>(dst[i]);
    a.src[i] = reinterpret_cast<const char*>(src[i]);
    a.bytes[i] = bytes[i];
    mx = bytes[i] > mx ? bytes[i] : mx;
  }
  int64_t gx = tt_cdiv(mx / 16 + 1, kThreads);
  const int64_t cap = (int64_t)ctx->num_cus * 4;
  if (gx > cap) gx = cap;
  copy_multi_kernel<<<dim3((unsigned)gx, (unsigned)n), kThreads, 0, reinterpret_cast<hipStream_t>(stream)>>>(a);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

// The hand-over kernels leave by this door: an ordinary launch -- remembering, when the stream is being captured, the graph node the
// launch became -- or, while the context is retargeting (tt_handover_retarget), no launch at all: the node of the executable graph
// is re-pointed at this call's function, grid and arguments (hipGraphExecKernelNodeSetParams; launches of the graph already in flight
// keep the arguments they were enqueued with: tools/probe/graph_setparams.hip).
static int handover_launch(tt_ctx* ctx, const void* fn, dim3 grid, void** args, hipStream_t st) {
  if (ctx->ho_exec) {
    hipKernelNodeParams p{};
    p.func = const_cast<void*>(fn);
    p.gridDim = grid;
    p.blockDim = dim3(kThreads);
    p.kernelParams = args;
    TT_HIP(hipGraphExecKernelNodeSetParams(reinterpret_cast<hipGraphExec_t>(ctx->ho_exec), reinterpret_cast<hipGraphNode_t>(ctx->ho_node), &p));
    return TT_OK;
  }
  TT_HIP(hipLaunchKernel(fn, grid, dim3(kThreads), args, 0, st));
  TT_LAUNCH_CHECK();
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const hipGraphNode_t* deps = nullptr;
  size_t n_deps = 0;
  if (hipStreamGetCaptureInfo_v2(st, &cs, nullptr, nullptr, &deps, &n_deps) == hipSuccess && cs == hipStreamCaptureStatusActive && n_deps == 1)
    ctx->ho_last = const_cast<void*>(reinterpret_cast<const void*>(deps[0]));
  else
    (void)hipGetLastError();
  return TT_OK;
}

int tt_batch_ingest(tt_ctx* ctx, int32_t n, void* const* dst, const void* const* src, const int64_t* bytes, const tt_embed_side* sides,
                    int32_t n_sides, int64_t B, int32_t* rows_km, int32_t* rows_sm, int64_t table_rows, const tt_cvt_list* cvt, tt_stream stream) {
  TT_CHECK_ARG(ctx && n >= 0 && n <= TT_MAX_COPIES && (n == 0 || (dst && src && bytes)), "tt_batch_ingest: bad copy arguments");
  TT_CHECK_ARG(sides && rows_km && n_sides >= 1 && n_sides <= TT_MAX_SIDES && B >= 1, "tt_batch_ingest: bad side arguments");
  TT_CHECK_ARG(table_rows >= 0 && table_rows <= INT32_MAX, "tt_batch_ingest: table_rows %lld out of range", (long long)table_rows);
  IngestArgs a{};
  int64_t mx = 0, slots = 0;
  for (int i = 0; i < n; ++i) {
    TT_CHECK_ARG(bytes[i] >= 0 && (bytes[i] == 0 || (dst[i] && src[i])), "tt_batch_ingest: segment %d NULL", i);
    TT_CHECK_ARG(tt_aligned(dst[i], 16) && tt_aligned(src[i], 16), "tt_batch_ingest: segment %d not 16-byte aligned", i);
    a.c.dst[i] = reinterpret_cast<char*>(dst[i]);
    a.c.src[i] = reinterpret_cast<const char*>(src[i]);
    a.c.bytes[i] = bytes[i];
    mx = bytes[i] > mx ? bytes[i] : mx;
  }
  a.n_copy = n;
  a.n_sides = n_sides;
  a.B = (int32_t)B;
  a.rows_km = rows_km;
  a.rows_sm = rows_sm;
  a.table_rows = (int32_t)table_rows;
  a.dev_err = ctx->dev_err;
  for (int i = 0; i < n_sides; ++i) {
    const tt_embed_side& s = sides[i];
    TT_CHECK_ARG(s.K >= 1 && s.ids && s.key_row_offset && s.key_vocab, "tt_batch_ingest: side %d NULL / no keys", i);
    if (s.K > kIngestMaxK) {
      tt_set_error("tt_batch_ingest: side %d has %d keys (max %d)", i, s.K, kIngestMaxK);
      return TT_ERR_UNSUPPORTED;
    }
    a.ids[i] = s.ids; a.off[i] = s.key_row_offset; a.vocab[i] = s.key_vocab; a.K[i] = s.K;
    a.side_base[i] = (int32_t)slots;
    slots += B * s.K;
  }
  TT_CHECK_ARG(slots < ((int64_t)1 << 31), "tt_batch_ingest: too many slots");
  const int cvt_rows = fill_cvt("tt_batch_ingest", cvt, &a.v, &mx);
  if (cvt_rows < 0) return cvt_rows;
  int64_t gx = tt_cdiv(mx / 16 + 1, kThreads);
  const int64_t cap = (int64_t)ctx->num_cus * 4;
  if (gx > cap) gx = cap;
  const int64_t tiles = tt_cdiv(B, 64) * n_sides;        // row 0 holds every tile (the copy rows stride over their segments)
  if (tiles > gx) gx = tiles;
  void* args[] = {&a};
  return handover_launch(ctx, reinterpret_cast<const void*>(batch_ingest_kernel), dim3((unsigned)gx, (unsigned)(n + 1 + cvt_rows)), args,
                         reinterpret_cast<hipStream_t>(stream));
}

int tt_batch_ingest_store(tt_ctx* ctx, int32_t n, void* const* dst, const void* const* src, const int64_t* bytes, const tt_embed_side* sides,
                          const tt_store_side* stores, int32_t n_sides, int64_t B, const int64_t* order, int32_t* rows_km, int32_t* rows_sm,
                          int64_t table_rows, const tt_cvt_list* cvt, tt_stream stream) {
  TT_CHECK_ARG(ctx && n >= 0 && n <= TT_MAX_COPIES && (n == 0 || (dst && src && bytes)), "tt_batch_ingest_store: bad copy arguments");
  TT_CHECK_ARG(table_rows >= 0 && table_rows <= INT32_MAX, "tt_batch_ingest_store: table_rows %lld out of range", (long long)table_rows);
  TT_CHECK_ARG(sides && stores && n_sides >= 1 && n_sides <= TT_MAX_SIDES && B >= 1, "tt_batch_ingest_store: bad side arguments");
  StoreIngestArgs a{};
  int64_t mx = 0, slots = 0, dense_pieces = 0;
  bool vec = true;
  for (int i = 0; i < n; ++i) {
    TT_CHECK_ARG(bytes[i] >= 0 && (bytes[i] == 0 || (dst[i] && src[i])), "tt_batch_ingest_store: segment %d NULL", i);
    TT_CHECK_ARG(tt_aligned(dst[i], 16) && tt_aligned(src[i], 16), "tt_batch_ingest_store: segment %d not 16-byte aligned", i);
    a.g.c.dst[i] = reinterpret_cast<char*>(dst[i]);
    a.g.c.src[i] = reinterpret_cast<const char*>(src[i]);
    a.g.c.bytes[i] = bytes[i];
    mx = bytes[i] > mx ? bytes[i] : mx;
  }
  a.g.n_copy = n;
  a.g.n_sides = n_sides;
  a.g.B = (int32_t)B;
  a.g.rows_km = rows_km;
  a.g.rows_sm = rows_sm;
  a.g.table_rows = (int32_t)table_rows;
  a.g.dev_err = ctx->dev_err;
  a.order = order;
  for (int i = 0; i < n_sides; ++i) {
    const tt_embed_side& s = sides[i];
    const tt_store_side& t = stores[i];
    TT_CHECK_ARG(s.K >= 1 && s.key_row_offset && s.key_vocab, "tt_batch_ingest_store: side %d NULL / no keys", i);
    TT_CHECK_ARG(t.entity && t.entity_stride >= 1 && t.cat_store && t.ids_out && t.dense_dim >= 0 &&
                 (t.dense_dim == 0 || (t.dense_store && t.dense_out)), "tt_batch_ingest_store: store %d NULL / bad shape", i);
    if (s.K > kIngestMaxK) {
      tt_set_error("tt_batch_ingest_store: side %d has %d keys (max %d)", i, s.K, kIngestMaxK);
      return TT_ERR_UNSUPPORTED;
    }
    a.g.off[i] = s.key_row_offset; a.g.vocab[i] = s.key_vocab; a.g.K[i] = s.K;
    a.g.side_base[i] = (int32_t)slots;
    slots += B * s.K;
    a.entity[i] = t.entity; a.entity_stride[i] = t.entity_stride; a.dense_store[i] = t.dense_store; a.cat_store[i] = t.cat_store;
    a.dense_out[i] = t.dense_out; a.ids_out[i] = t.ids_out; a.dense_dim[i] = t.dense_dim;
    a.n_rows[i] = t.n_rows > 0 ? t.n_rows : 0;
    vec = vec && t.dense_dim % 4 == 0 && tt_aligned(t.dense_store, 16) && tt_aligned(t.dense_out, 16);
    const int64_t p = B * (int64_t)t.dense_dim;
    dense_pieces = p > dense_pieces ? p : dense_pieces;
  }
  TT_CHECK_ARG(slots < ((int64_t)1 << 31), "tt_batch_ingest_store: too many slots");
  const int cvt_rows = fill_cvt("tt_batch_ingest_store", cvt, &a.g.v, &mx);
  if (cvt_rows < 0) return cvt_rows;
  int64_t gx = tt_cdiv(mx / 16 + 1, kThreads);
  const int64_t rows_wg = tt_cdiv(vec ? dense_pieces / 4 : dense_pieces, kThreads);
  if (rows_wg > gx) gx = rows_wg;
  const int64_t cap = (int64_t)ctx->num_cus * 8;
  if (gx > cap) gx = cap;
  const int64_t tiles = tt_cdiv(B, 64) * n_sides;        // row 0 holds every tile (the other rows stride over their work)
  if (tiles > gx) gx = tiles;
  const dim3 grid((unsigned)gx, (unsigned)(1 + n_sides + n + cvt_rows));
  void* args[] = {&a};
  return handover_launch(ctx, vec ? reinterpret_cast<const void*>(batch_ingest_store_kernel<true>) : reinterpret_cast<const void*>(batch_ingest_store_kernel<false>),
                         grid, args, reinterpret_cast<hipStream_t>(stream));
}

// shared by the two fused hand-over + lookup entries: checks the lookup half and fills LookupPart; returns the tile count or < 0
static int64_t fill_lookup_part(tt_ctx* ctx, const char* who, const tt_embed_side* sides, int32_t n_sides, int64_t B, const tt_ingest_lookup* lk,
                                LookupPart* lp) {
  if (!(lk && lk->table && lk->table_rows >= 1 && lk->table_rows <= INT32_MAX)) {
    tt_set_error("%s: lookup part NULL / bad table", who);
    return TT_ERR_INVALID_ARG;
  }
  const int E = lk->E;
  if (!(E == 8 || E == 16 || E == 32 || E == 64) || !tt_aligned(lk->table, 16)) {
    tt_set_error("%s: E=%d not in {8, 16, 32, 64} or table not 16-byte aligned (use the separate hand-over and tt_embed_lookup_fwd)", who, E);
    return TT_ERR_UNSUPPORTED;
  }
  lp->table = lk->table;
  lp->E = E;
  lp->ring = ctx->lookup_stamps;
  lp->ring_slots = ctx->lookup_stamp_slots;
  lp->nt = ctx->lookup_nt;
  lp->table_rows = (int32_t)lk->table_rows;
  lp->dev_err = ctx->dev_err;
  int64_t tiles = 0;
  for (int i = 0; i < n_sides; ++i) {
    const tt_embed_side& s = sides[i];
    if (!(s.out && (s.out_dtype == TT_F32 || s.out_dtype == TT_BF16) && s.ld_out >= (int64_t)s.K * E)) {
      tt_set_error("%s: side %d needs an output (out, ld_out >= K*E, out_dtype f32 | bf16)", who, i);
      return TT_ERR_INVALID_ARG;
    }
    const size_t esz = s.out_dtype == TT_BF16 ? 2 : 4;
    if (!tt_aligned(s.out, 8 * esz) || (s.ld_out * esz) % (8 * esz) != 0) {
      tt_set_error("%s: side %d output not aligned to 8 elements", who, i);
      return TT_ERR_UNSUPPORTED;
    }
    int sh = 6;                                              // TS = 64 samples, halved until TS * K <= 512 slots (K <= 64: TS >= 8)
    while (sh > 3 && ((int64_t)s.K << sh) > kTileSlots) --sh;
    lp->out[i] = reinterpret_cast<char*>(s.out);
    lp->ld[i] = s.ld_out;
    lp->dtype[i] = s.out_dtype;
    lp->ts_shift[i] = sh;
    lp->tile_base[i] = (int32_t)tiles;
    tiles += tt_cdiv(B, (int64_t)1 << sh);
  }
  lp->tile_base[n_sides] = (int32_t)tiles;
  for (int i = n_sides + 1; i <= TT_MAX_SIDES; ++i) lp->tile_base[i] = (int32_t)tiles;
  if (tiles >= ((int64_t)1 << 30)) {
    tt_set_error("%s: too many tiles", who);
    return TT_ERR_INVALID_ARG;
  }
  return tiles;
}

#define TT_INGEST_LOOKUP_FN(FROM_STORE, VEC)                                                                              \
  (lp.E == 8 ? reinterpret_cast<const void*>(ingest_lookup_kernel<1, FROM_STORE, VEC>)                                    \
   : lp.E == 16 ? reinterpret_cast<const void*>(ingest_lookup_kernel<2, FROM_STORE, VEC>)                                 \
   : lp.E == 32 ? reinterpret_cast<const void*>(ingest_lookup_kernel<4, FROM_STORE, VEC>)                                 \
                : reinterpret_cast<const void*>(ingest_lookup_kernel<8, FROM_STORE, VEC>))

int tt_batch_ingest_lookup(tt_ctx* ctx, int32_t n, void* const* dst, const void* const* src, const int64_t* bytes, const tt_embed_side* sides,
                           int32_t n_sides, int64_t B, int32_t* rows_km, const tt_ingest_lookup* lk, const tt_cvt_list* cvt, tt_stream stream) {
  TT_CHECK_ARG(ctx && n >= 0 && n <= TT_MAX_COPIES && (n == 0 || (dst && src && bytes)), "tt_batch_ingest_lookup: bad copy arguments");
  TT_CHECK_ARG(sides && n_sides >= 1 && n_sides <= TT_MAX_SIDES && B >= 1, "tt_batch_ingest_lookup: bad side arguments");
  StoreIngestArgs a{};
  LookupPart lp{};
  int64_t mx = 0, slots = 0;
  for (int i = 0; i < n; ++i) {
    TT_CHECK_ARG(bytes[i] >= 0 && (bytes[i] == 0 || (dst[i] && src[i])), "tt_batch_ingest_lookup: segment %d NULL", i);
    TT_CHECK_ARG(tt_aligned(dst[i], 16) && tt_aligned(src[i], 16), "tt_batch_ingest_lookup: segment %d not 16-byte aligned", i);
    a.g.c.dst[i] = reinterpret_cast<char*>(dst[i]);
    a.g.c.src[i] = reinterpret_cast<const char*>(src[i]);
    a.g.c.bytes[i] = bytes[i];
    mx = bytes[i] > mx ? bytes[i] : mx;
  }
  a.g.n_copy = n;
  a.g.n_sides = n_sides;
  a.g.B = (int32_t)B;
  a.g.rows_km = rows_km;
  for (int i = 0; i < n_sides; ++i) {
    const tt_embed_side& s = sides[i];
    TT_CHECK_ARG(s.K >= 1 && s.ids && s.key_row_offset && s.key_vocab, "tt_batch_ingest_lookup: side %d NULL / no keys", i);
    if (s.K > kIngestMaxK) {
      tt_set_error("tt_batch_ingest_lookup: side %d has %d keys (max %d)", i, s.K, kIngestMaxK);
      return TT_ERR_UNSUPPORTED;
    }
    a.g.ids[i] = s.ids; a.g.off[i] = s.key_row_offset; a.g.vocab[i] = s.key_vocab; a.g.K[i] = s.K;
    a.g.side_base[i] = (int32_t)slots;
    slots += B * s.K;
  }
  TT_CHECK_ARG(slots < ((int64_t)1 << 31), "tt_batch_ingest_lookup: too many slots");
  const int64_t tiles = fill_lookup_part(ctx, "tt_batch_ingest_lookup", sides, n_sides, B, lk, &lp);
  if (tiles < 0) return (int)tiles;
  const int cvt_rows = fill_cvt("tt_batch_ingest_lookup", cvt, &a.g.v, &mx);
  if (cvt_rows < 0) return cvt_rows;
  int64_t gx = tt_cdiv(mx / 16 + 1, kThreads);
  const int64_t cap = (int64_t)ctx->num_cus * 4;
  if (gx > cap) gx = cap;
  if (tiles > gx) gx = tiles;
  const dim3 grid((unsigned)gx, (unsigned)(n + 1 + cvt_rows));
  void* args[] = {&a, &lp};
  return handover_launch(ctx, TT_INGEST_LOOKUP_FN(false, true), grid, args, reinterpret_cast<hipStream_t>(stream));
}

int tt_batch_ingest_store_lookup(tt_ctx* ctx, int32_t n, void* const* dst, const void* const* src, const int64_t* bytes,
                                 const tt_embed_side* sides, const tt_store_side* stores, int32_t n_sides, int64_t B, const int64_t* order,
                                 int32_t* rows_km, const tt_ingest_lookup* lk, const tt_cvt_list* cvt, tt_stream stream) {
  TT_CHECK_ARG(ctx && n >= 0 && n <= TT_MAX_COPIES && (n == 0 || (dst && src && bytes)), "tt_batch_ingest_store_lookup: bad copy arguments");
  TT_CHECK_ARG(sides && stores && n_sides >= 1 && n_sides <= TT_MAX_SIDES && B >= 1, "tt_batch_ingest_store_lookup: bad side arguments");
  StoreIngestArgs a{};
  LookupPart lp{};
  int64_t mx = 0, slots = 0, dense_pieces = 0;
  bool vec = true;
  for (int i = 0; i < n; ++i) {
    TT_CHECK_ARG(bytes[i] >= 0 && (bytes[i] == 0 || (dst[i] && src[i])), "tt_batch_ingest_store_lookup: segment %d NULL", i);
    TT_CHECK_ARG(tt_aligned(dst[i], 16) && tt_aligned(src[i], 16), "tt_batch_ingest_store_lookup: segment %d not 16-byte aligned", i);
    a.g.c.dst[i] = reinterpret_cast<char*>(dst[i]);
    a.g.c.src[i] = reinterpret_cast<const char*>(src[i]);
    a.g.c.bytes[i] = bytes[i];
    mx = bytes[i] > mx ? bytes[i] : mx;
  }
  a.g.n_copy = n;
  a.g.n_sides = n_sides;
  a.g.B = (int32_t)B;
  a.g.rows_km = rows_km;
  a.order = order;
  for (int i = 0; i < n_sides; ++i) {
    const tt_embed_side& s = sides[i];
    const tt_store_side& t = stores[i];
    TT_CHECK_ARG(s.K >= 1 && s.key_row_offset && s.key_vocab, "tt_batch_ingest_store_lookup: side %d NULL / no keys", i);
    TT_CHECK_ARG(t.entity && t.entity_stride >= 1 && t.cat_store && t.ids_out && t.dense_dim >= 0 && t.n_rows >= 0 &&
                 (t.dense_dim == 0 || (t.dense_store && t.dense_out)), "tt_batch_ingest_store_lookup: store %d NULL / bad shape", i);
    if (s.K > kIngestMaxK) {
      tt_set_error("tt_batch_ingest_store_lookup: side %d has %d keys (max %d)", i, s.K, kIngestMaxK);
      return TT_ERR_UNSUPPORTED;
    }
    a.g.off[i] = s.key_row_offset; a.g.vocab[i] = s.key_vocab; a.g.K[i] = s.K;
    a.g.side_base[i] = (int32_t)slots;
    slots += B * s.K;
    a.entity[i] = t.entity; a.entity_stride[i] = t.entity_stride; a.dense_store[i] = t.dense_store; a.cat_store[i] = t.cat_store;
    a.dense_out[i] = t.dense_out; a.ids_out[i] = t.ids_out; a.dense_dim[i] = t.dense_dim;
    a.n_rows[i] = t.n_rows > 0 ? t.n_rows : 0;
    vec = vec && t.dense_dim % 4 == 0 && tt_aligned(t.dense_store, 16) && tt_aligned(t.dense_out, 16);
    const int64_t p = B * (int64_t)t.dense_dim;
    dense_pieces = p > dense_pieces ? p : dense_pieces;
  }
  TT_CHECK_ARG(slots < ((int64_t)1 << 31), "tt_batch_ingest_store_lookup: too many slots");
  const int64_t tiles = fill_lookup_part(ctx, "tt_batch_ingest_store_lookup", sides, n_sides, B, lk, &lp);
  if (tiles < 0) return (int)tiles;
  const int cvt_rows = fill_cvt("tt_batch_ingest_store_lookup", cvt, &a.g.v, &mx);
  if (cvt_rows < 0) return cvt_rows;
  int64_t gx = tt_cdiv(mx / 16 + 1, kThreads);
  const int64_t rows_wg = tt_cdiv(vec ? dense_pieces / 4 : dense_pieces, kThreads);
  if (rows_wg > gx) gx = rows_wg;
  const int64_t cap = (int64_t)ctx->num_cus * 8;
  if (gx > cap) gx = cap;
  if (tiles > gx) gx = tiles;
  const dim3 grid((unsigned)gx, (unsigned)(1 + n_sides + n + cvt_rows));
  void* args[] = {&a, &lp};
  return handover_launch(ctx, vec ? TT_INGEST_LOOKUP_FN(true, true) : TT_INGEST_LOOKUP_FN(true, false), grid, args, reinterpret_cast<hipStream_t>(stream));
}

int tt_batch_gather(tt_ctx* ctx, const int64_t* entity, int64_t B, const float* dense_store, int32_t dense_dim,
                    const int64_t* cat_store, int32_t K, float* dense_out, int64_t* ids_out, tt_stream stream) {
  TT_CHECK_ARG(ctx && entity, "tt_batch_gather: NULL argument");
  TT_CHECK_ARG(B >= 0 && dense_dim >= 0 && K >= 0, "tt_batch_gather: negative size");
  TT_CHECK_ARG(dense_dim == 0 || (dense_store && dense_out), "tt_batch_gather: NULL dense buffers");
  TT_CHECK_ARG(K == 0 || (cat_store && ids_out), "tt_batch_gather: NULL id buffers");
  const int64_t total = B * ((int64_t)dense_dim + K);
  TT_CHECK_ARG(total < ((int64_t)1 << 31), "tt_batch_gather: too many elements");
  if (total == 0) return TT_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  batch_gather_kernel<<<grid_for(ctx, total), kThreads, 0, st>>>(entity, (uint32_t)B, dense_store, (uint32_t)dense_dim, cat_store,
                                                                  (uint32_t)K, dense_out, ids_out);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

}  // extern "C"
