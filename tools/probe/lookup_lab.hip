// Lab for the embedding lookup on tables the caches cannot hold (BASELINE configs[2]: 100 M + 10 M rows of 128 B on one GPU).
// The shipping kernel's structure (tt_embed.hip: lookup_wave_kernel) with compile-time variants, timed the way a replayed step
// sees it: N launches back to back on one stream (HIP events around the run => mean LAUNCH time incl. the kernel boundary), and
// per-workgroup s_memrealtime stamps (=> body = min start .. max end).  Every variant's output is compared with variant 0's.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probe/lookup_lab tools/probe/lookup_lab.hip
//   tools/probe/lookup_lab [rows_notice rows_company [B [launches]]]
//
// Variant knobs: ST = how the bf16 rows are stored (0 plain, 1 nontemporal, 2 system-scope write-through "sc0 sc1");
//                CPW = 64-slot chunks per wave (1 = one pass per wave, the shipping form; > 1 = software-pipelined passes);
//                PAIR = a lane moves 32 B of a row (two loads, one 16-B store; 4 lanes per row) instead of 16 B;
//                SRC = 0 int64 ids + clamp + key offset (cat_embed.py:103-121), 1 int32 fused rows precomputed slot-major.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kThreads = 256;
constexpr int kE = 32;
constexpr int kMaxWg = 8192;

struct Side {
  const int64_t* ids;
  const int64_t* off;
  const int64_t* vocab;
  const int32_t* rows;   // SRC = 1: fused rows, slot-major
  char* out;             // bf16 [B, ld]
  int64_t ld;            // elements
  uint32_t slot_base;
  int32_t K;
};
struct Args {
  Side s[2];
  uint32_t total_slots;
  const float* table;
  unsigned long long* stamps;   // [2 * kMaxWg] or null
};

__device__ __forceinline__ uint16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <int ST>
__device__ __forceinline__ void store8(char* p, ushort4 o) {
  if (ST == 0) {
    *reinterpret_cast<ushort4*>(p) = o;
  } else if (ST == 1) {
    using u16x4 = __attribute__((ext_vector_type(4))) unsigned short;
    u16x4 t; t[0] = o.x; t[1] = o.y; t[2] = o.z; t[3] = o.w;
    __builtin_nontemporal_store(t, reinterpret_cast<u16x4*>(p));
  } else {
    const uint64_t v = (uint64_t)o.x | ((uint64_t)o.y << 16) | ((uint64_t)o.z << 32) | ((uint64_t)o.w << 48);
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  }
}
template <int ST>
__device__ __forceinline__ void store16(char* p, uint4 v) {
  if (ST == 0) {
    *reinterpret_cast<uint4*>(p) = v;
  } else if (ST == 1) {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    u32x4 t; t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
  } else {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    u32x4 t; t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(t) : "memory");
  }
}

struct Rec { const float* src; char* dst; };

template <int SRC>
__device__ __forceinline__ Rec decode(const Args& a, uint32_t slot) {
  Rec r{nullptr, nullptr};
  if (slot >= a.total_slots) return r;
  const int si = slot >= a.s[1].slot_base ? 1 : 0;
  const Side& s = a.s[si];
  const uint32_t local = slot - s.slot_base;
  const uint32_t b = local / (uint32_t)s.K, k = local - b * (uint32_t)s.K;
  int64_t row;
  if (SRC == 0) {
    int64_t id = s.ids[local];
    const int64_t hi = s.vocab[k] - 1;
    id = id < 0 ? 0 : (id > hi ? hi : id);
    row = s.off[k] + id;
  } else {
    row = s.rows[local];
  }
  r.src = a.table + row * kE;
  r.dst = s.out + ((int64_t)b * s.ld + (int64_t)k * kE) * 2;
  return r;
}

// ---- the shipping form and its variants: a wave owns CPW chunks of 64 consecutive slots ----
template <int ST, int CPW, int PAIR, int SRC, int TPB = kThreads>
__global__ __launch_bounds__(TPB) void lab_kernel(Args a) {
  const bool stamp = a.stamps && blockIdx.x < kMaxWg && threadIdx.x == 0;
  unsigned long long t0 = 0;
  if (stamp) t0 = __builtin_amdgcn_s_memrealtime();
  constexpr int LPR = PAIR ? 4 : 8;          // lanes per row
  constexpr int RPI = 64 / LPR;              // rows per wave-instruction
  constexpr int NIT = 64 / RPI;              // wave-instructions per chunk
  __shared__ Rec recs[TPB / 64][2][64];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t nchunks = (a.total_slots + 63) / 64;
  const uint32_t nwaves = gridDim.x * (TPB / 64);
  const uint32_t w = blockIdx.x * (TPB / 64) + wave;
  const uint32_t sub = lane / LPR, part = lane % LPR;
  float4 v[2][NIT][PAIR ? 2 : 1];
  // chunk c of this wave = w + c * nwaves (grid-stride); software pipeline: loads of chunk c+1 are issued before chunk c is stored
  auto issue = [&](int buf, uint32_t chunk) {
    recs[wave][buf][lane] = decode<SRC>(a, chunk * 64 + lane);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const Rec r = recs[wave][buf][j * RPI + sub];
#pragma unroll
      for (int h = 0; h < (PAIR ? 2 : 1); ++h) {
        v[buf][j][h] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r.src) v[buf][j][h] = *reinterpret_cast<const float4*>(r.src + (part * (PAIR ? 2 : 1) + h) * 4);
      }
    }
  };
  auto drain = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NIT; ++j) {
      const Rec r = recs[wave][buf][j * RPI + sub];
      if (!r.dst) continue;
      if (PAIR) {
        uint4 o;
        o.x = f2bf(v[buf][j][0].x) | ((uint32_t)f2bf(v[buf][j][0].y) << 16);
        o.y = f2bf(v[buf][j][0].z) | ((uint32_t)f2bf(v[buf][j][0].w) << 16);
        o.z = f2bf(v[buf][j][PAIR ? 1 : 0].x) | ((uint32_t)f2bf(v[buf][j][PAIR ? 1 : 0].y) << 16);
        o.w = f2bf(v[buf][j][PAIR ? 1 : 0].z) | ((uint32_t)f2bf(v[buf][j][PAIR ? 1 : 0].w) << 16);
        store16<ST>(r.dst + part * 16, o);
      } else {
        ushort4 o;
        o.x = f2bf(v[buf][j][0].x); o.y = f2bf(v[buf][j][0].y); o.z = f2bf(v[buf][j][0].z); o.w = f2bf(v[buf][j][0].w);
        store8<ST>(r.dst + part * 8, o);
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  if (CPW == 1) {
    for (uint32_t chunk = w; chunk < nchunks; chunk += nwaves) {
      issue(0, chunk);
      drain(0);
    }
  } else {
    uint32_t chunk = w;
    if (chunk < nchunks) issue(0, chunk);
    int buf = 0;
#pragma unroll 1
    for (; chunk < nchunks; chunk += nwaves) {
      const uint32_t nxt = chunk + nwaves;
      if (buf == 0) {
        if (nxt < nchunks) issue(1, nxt);
        drain(0);
      } else {
        if (nxt < nchunks) issue(0, nxt);
        drain(1);
      }
      buf ^= 1;
    }
  }
  if (a.stamps) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (stamp) {
      a.stamps[2 * blockIdx.x] = t0;
      a.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

__global__ void fill_table(float* t, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u + (uint32_t)(i >> 32) * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    t[i] = (float)(int32_t)h * (1.0f / 2147483648.0f);
  }
}
__global__ void rows_of(Args a, int32_t* rows) {
  for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < a.total_slots; slot += gridDim.x * blockDim.x) {
    const Rec r = decode<0>(a, slot);
    rows[slot] = (int32_t)((r.src - a.table) / kE);
  }
}
__global__ void diff_count(const uint16_t* x, const uint16_t* y, size_t n, unsigned long long* cnt) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += x[i] != y[i];
  if (c) atomicAdd(cnt, c);
}
__global__ void stream_copy(const float4* s, float4* d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

static const int kVocN[32] = {12, 12, 28993, 29920, 18, 22, 23, 12, 13, 12, 12, 13, 13, 248, 25, 23, 12, 12, 12, 12, 12, 25, 137, 12, 332, 12, 12, 12, 12, 12, 12, 15};
static const int kVocC[6] = {315, 3680, 65, 12, 33, 12};

static std::vector<int64_t> scale_vocabs(const int* v, int n, int64_t total) {   // synthetic.scale_vocabs
  double sum = 0;
  for (int i = 0; i < n; ++i) sum += v[i];
  const double f = (double)total / sum;
  std::vector<int64_t> out(n);
  int64_t s = 0; int big = 0;
  for (int i = 0; i < n; ++i) { out[i] = std::max<int64_t>(2, (int64_t)std::nearbyint(v[i] * f)); s += out[i]; if (out[i] > out[big]) big = i; }
  out[big] += total - s;
  return out;
}

int main(int argc, char** argv) {
  const int64_t rows_n = argc > 1 ? atoll(argv[1]) : 100000000, rows_c = argc > 2 ? atoll(argv[2]) : 10000000;
  const int B = argc > 3 ? atoi(argv[3]) : 8192, launches = argc > 4 ? atoi(argv[4]) : 64;
  const int small_giants = argc > 5 ? atoi(argv[5]) : 0;     // 1: the three giant keys draw their ids from 4096 rows only (no HBM / TLB misses)
  constexpr int POOL = 8;
  auto vn = scale_vocabs(kVocN, 32, rows_n), vc = scale_vocabs(kVocC, 6, rows_c);
  const int64_t R = rows_n + rows_c;
  float* table;
  CK(hipMalloc(&table, (size_t)R * kE * 4));
  fill_table<<<4096, 256>>>(table, (size_t)R * kE);
  std::vector<int64_t> offn(32), offc(6);
  { int64_t o = 0; for (int i = 0; i < 32; ++i) { offn[i] = o; o += vn[i]; } for (int i = 0; i < 6; ++i) { offc[i] = o; o += vc[i]; } }
  int64_t *d_offn, *d_offc, *d_vn, *d_vc;
  CK(hipMalloc(&d_offn, 32 * 8)); CK(hipMalloc(&d_offc, 6 * 8)); CK(hipMalloc(&d_vn, 32 * 8)); CK(hipMalloc(&d_vc, 6 * 8));
  CK(hipMemcpy(d_offn, offn.data(), 32 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_offc, offc.data(), 6 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_vn, vn.data(), 32 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_vc, vc.data(), 6 * 8, hipMemcpyHostToDevice));
  // id pool: uniform per key (xorshift64*), sample-major
  int64_t* d_ids_n[POOL]; int64_t* d_ids_c[POOL]; int32_t* d_rows[POOL];
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { st ^= st >> 12; st ^= st << 25; st ^= st >> 27; return st * 2685821657736338717ull; };
  for (int p = 0; p < POOL; ++p) {
    std::vector<int64_t> hn((size_t)B * 32), hc((size_t)B * 6);
    for (int b = 0; b < B; ++b) {
      for (int k = 0; k < 32; ++k) { int64_t v = vn[k]; if (small_giants && v > 4096) v = 4096; hn[(size_t)b * 32 + k] = (int64_t)(rnd() % (uint64_t)v); }
      for (int k = 0; k < 6; ++k) { int64_t v = vc[k]; if (small_giants && v > 4096) v = 4096; hc[(size_t)b * 6 + k] = (int64_t)(rnd() % (uint64_t)v); }
    }
    CK(hipMalloc(&d_ids_n[p], hn.size() * 8)); CK(hipMalloc(&d_ids_c[p], hc.size() * 8));
    CK(hipMemcpy(d_ids_n[p], hn.data(), hn.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ids_c[p], hc.data(), hc.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_rows[p], (size_t)B * 38 * 4));
  }
  const int64_t ldn = 128 + 32 * kE, ldc = 128 + 6 * kE;
  uint16_t *xn, *xc, *xn0, *xc0;
  CK(hipMalloc(&xn, (size_t)B * ldn * 2)); CK(hipMalloc(&xc, (size_t)B * ldc * 2));
  CK(hipMalloc(&xn0, (size_t)B * ldn * 2)); CK(hipMalloc(&xc0, (size_t)B * ldc * 2));
  CK(hipMemset(xn0, 0, (size_t)B * ldn * 2)); CK(hipMemset(xc0, 0, (size_t)B * ldc * 2));   // (columns [0, 128) are never written)
  unsigned long long *stamps, *cnt;
  CK(hipMalloc(&stamps, 2 * kMaxWg * 8)); CK(hipMalloc(&cnt, 8));
  float4 *fl_s, *fl_d;
  const size_t FL = (size_t)512 << 20;
  CK(hipMalloc(&fl_s, FL)); CK(hipMalloc(&fl_d, FL));
  auto args = [&](int p, uint16_t* on, uint16_t* oc, bool stamp) {
    Args a{};
    a.s[0] = Side{d_ids_n[p], d_offn, d_vn, d_rows[p], reinterpret_cast<char*>(on + 128), ldn, 0u, 32};
    a.s[1] = Side{d_ids_c[p], d_offc, d_vc, d_rows[p] + (size_t)B * 32, reinterpret_cast<char*>(oc + 128), ldc, (uint32_t)(B * 32), 6};
    a.total_slots = (uint32_t)(B * 38);
    a.table = table;
    a.stamps = stamp ? stamps : nullptr;
    return a;
  };
  for (int p = 0; p < POOL; ++p) rows_of<<<1024, 256>>>(args(p, xn, xc, false), d_rows[p]);
  CK(hipDeviceSynchronize());
  const uint32_t nchunks = (uint32_t)((B * 38 + 63) / 64);
  struct V { const char* name; void (*k)(Args); int cpw; int tpb = kThreads; };
  const V vs[] = {
      {"ST0 plain stores          (shipping)", lab_kernel<0, 1, 0, 0>, 1},
      {"ST0 PAIR SRC1", lab_kernel<0, 1, 1, 1>, 1},
      {"ST1 PAIR SRC1", lab_kernel<1, 1, 1, 1>, 1},
      {"ST2 PAIR SRC1", lab_kernel<2, 1, 1, 1>, 1},
      {"ST1 PAIR", lab_kernel<1, 1, 1, 0>, 1},
      {"ST0 PAIR 128 threads", lab_kernel<0, 1, 1, 0, 128>, 1, 128},
      {"ST0 PAIR 512 threads", lab_kernel<0, 1, 1, 0, 512>, 1, 512},
      {"ST0 PAIR 1024 threads", lab_kernel<0, 1, 1, 0, 1024>, 1, 1024},
      {"ST1 nontemporal stores", lab_kernel<1, 1, 0, 0>, 1},
      {"ST2 sc0 sc1 stores", lab_kernel<2, 1, 0, 0>, 1},
      {"ST0 PAIR (32 B per lane)", lab_kernel<0, 1, 1, 0>, 1},
      {"ST2 PAIR", lab_kernel<2, 1, 1, 0>, 1},
      {"ST0 rows precomputed (SRC1)", lab_kernel<0, 1, 0, 1>, 1},
      {"ST2 rows precomputed (SRC1)", lab_kernel<2, 1, 0, 1>, 1},
      {"ST0 CPW2 pipelined", lab_kernel<0, 2, 0, 0>, 2},
      {"ST2 CPW2 pipelined", lab_kernel<2, 2, 0, 0>, 2},
      {"ST1 CPW2 pipelined", lab_kernel<1, 2, 0, 0>, 2},
  };
  printf("tables %lld + %lld rows (%.2f GB), B = %d, %u chunks, %d launches per timing, small_giants %d\n", (long long)rows_n, (long long)rows_c,
         (double)R * 128 / 1e9, B, nchunks, launches, small_giants);
  printf("%-40s %8s %8s %8s %8s | %s\n", "variant", "launch", "coldbody", "instep", "frac", "workgroup anatomy (us): start p90, duration mean / p90, end p50");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double algo = (double)B * 38 * (128 + 8 + 64);
  for (size_t vi = 0; vi < sizeof(vs) / sizeof(vs[0]); ++vi) {
    const V& v = vs[vi];
    const int wpb = v.tpb / 64;
    const int grid = (int)((nchunks + wpb * v.cpw - 1) / (wpb * v.cpw));
    // correctness against variant 0 on pool batch 3
    CK(hipMemset(xn, 0, (size_t)B * ldn * 2)); CK(hipMemset(xc, 0, (size_t)B * ldc * 2));
    v.k<<<grid, v.tpb>>>(args(3, vi == 0 ? xn0 : xn, vi == 0 ? xc0 : xc, false));
    CK(hipGetLastError());
    unsigned long long bad = 0;
    if (vi) {
      CK(hipMemset(cnt, 0, 8));
      diff_count<<<1024, 256>>>(xn + 0, xn0 + 0, (size_t)B * ldn, cnt);
      diff_count<<<256, 256>>>(xc, xc0, (size_t)B * ldc, cnt);
      CK(hipMemcpy(&bad, cnt, 8, hipMemcpyDeviceToHost));
      // the first 128 columns are never written: both hold their fill (xn0 is unwritten memory there) -- compare written columns only
    }
    // back-to-back launches
    for (int i = 0; i < 8; ++i) v.k<<<grid, v.tpb>>>(args(i % POOL, xn, xc, false));
    stream_copy<<<2048, 256>>>(fl_s, fl_d, FL / 16);
    double best = 1e9, sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < launches; ++i) v.k<<<grid, v.tpb>>>(args(i % POOL, xn, xc, false));
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / launches;
      best = std::min(best, us); sum += us;
    }
    const double launch_us = sum / 5;
    // in-step-like: every lookup behind a 2 x 192 MB streaming copy (what the other kernels of a step do to the caches); the
    // copy alone is timed the same way and subtracted
    double pair_us = 0;
    {
      const size_t cp = ((size_t)192 << 20) / 16;
      const int NP = 24;
      float ms_c = 0, ms_p = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < NP; ++i) stream_copy<<<2048, 256>>>(fl_s, fl_d, cp);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_c, e0, e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < NP; ++i) { stream_copy<<<2048, 256>>>(fl_s, fl_d, cp); v.k<<<grid, v.tpb>>>(args(i % POOL, xn, xc, false)); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_p, e0, e1));
      }
      pair_us = (ms_p - ms_c) * 1e3 / NP;
    }
    // body: stamped launches (each behind a cache-evicting stream copy, as inside a training step)
    double body = 0; int nb = 0;
    std::vector<unsigned long long> h(2 * kMaxWg);
    double a_start90 = 0, a_dur = 0, a_dur90 = 0, a_end50 = 0;
    for (int i = 0; i < 6; ++i) {
      stream_copy<<<2048, 256>>>(fl_s, fl_d, FL / 16);
      CK(hipMemset(stamps, 0, 2 * kMaxWg * 8));
      v.k<<<grid, v.tpb>>>(args(i % POOL, xn, xc, true));
      CK(hipMemcpy(h.data(), stamps, 2 * kMaxWg * 8, hipMemcpyDeviceToHost));
      const int n = std::min(grid, kMaxWg);
      unsigned long long lo = ~0ull, hi = 0;
      std::vector<double> stv, dur, en;
      for (int g = 0; g < n; ++g) { lo = std::min(lo, h[2 * g]); hi = std::max(hi, h[2 * g + 1]); }
      for (int g = 0; g < n; ++g) { stv.push_back((h[2 * g] - lo) * 0.01); dur.push_back((h[2 * g + 1] - h[2 * g]) * 0.01); en.push_back((h[2 * g + 1] - lo) * 0.01); }
      std::sort(stv.begin(), stv.end()); std::sort(en.begin(), en.end());
      double dm = 0; for (double d : dur) dm += d; dm /= n;
      std::sort(dur.begin(), dur.end());
      if (i) { body += (hi - lo) * 0.01; ++nb; a_start90 += stv[(size_t)(0.9 * n)]; a_dur += dm; a_dur90 += dur[(size_t)(0.9 * n)]; a_end50 += en[n / 2]; }
    }
    body /= nb;
    printf("%-40s %8.2f %8.2f %8.2f %8.3f | %.2f, %.2f / %.2f, %.2f   grid %d  min-of-5 %.2f  %s\n", v.name, launch_us, body, pair_us,
           algo / (launch_us * 1e-6) / 8e12, a_start90 / nb, a_dur / nb, a_dur90 / nb, a_end50 / nb, grid, best, bad ? "MISMATCH" : "ok");
    fflush(stdout);
  }
  return 0;
}
