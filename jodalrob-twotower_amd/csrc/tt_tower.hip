// Tower MLP (BaseTower.forward after the lookup, and its backward): Linear layers on the f32 MFMA
// GEMM tiles of tt_gemm.hip, BatchNorm1d statistics as deterministic two-stage column reductions,
// ReLU / BN / dropout / L2-normalise as fused elementwise and row-wise kernels.
#include "tt_gemm.h"
#include "tt_riders.h"

namespace {

constexpr int kThreads = 256;
constexpr float kBnEps = 1e-5f, kBnMomentum = 0.1f, kNormEps = 1e-12f;
constexpr int kMaxChunks = 128;  // row chunks of the two-stage column reductions

__device__ __forceinline__ float dropout_scale(bool on, float p, uint64_t seed, uint64_t idx) {
  if (!on) return 1.f;
  return tt_uniform01(seed, idx) >= p ? 1.f / (1.f - p) : 0.f;
}
__device__ __forceinline__ uint64_t seed_of(uint64_t seed, const uint64_t* seed_dev) { return seed_dev ? seed + seed_dev[0] : seed; }

// ---- column statistics of relu(pre): Welford per thread, Chan combine ------------------------
struct Wf {
  float n, mean, m2;
};
__device__ __forceinline__ Wf wf_combine(Wf a, Wf b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  Wf o;
  o.n = a.n + b.n;
  const float d = b.mean - a.mean, w = b.n / o.n;      // one division per combine (the chains of 32 are latency-bound)
  o.mean = a.mean + d * w;
  o.m2 = a.m2 + b.m2 + d * d * (a.n * w);
  return o;
}

// Canonical merge order of the chunk statistics (round 4; every finish -- bn_stats_finish_kernel, the fused tails, the SyncBN hand-over
// -- uses it, so their results stay bit-identical to each other).  Chunk lane jl owns chunks jl, jl + 4, jl + 8, ... in four SUB-CHAINS of
// kMaxChunks / 16 chunks each:   lane(jl) = ((C0 + C1) + C2) + C3,  C_s = ((v[8 s] + v[8 s + 1]) + ...) + v[8 s + 7];
// the four lanes then merge as (lane0 + lane1) + (lane2 + lane3).  Sub-chains exist so that the fused forward tail can give each one to
// its own thread (16 per column: 8 triples in flight per thread instead of 32 -- 52 registers, two workgroups per CU) without changing
// the association; before, a lane was one chain of 32.
constexpr int kSubChain = kMaxChunks / 16;
__device__ __forceinline__ Wf wf_lane_merge(const Wf* v /* [kMaxChunks / 4] */) {
  Wf o{0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    Wf cs{0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kSubChain; ++i) cs = wf_combine(cs, v[kSubChain * s + i]);
    o = wf_combine(o, cs);
  }
  return o;
}

template <typename A>
struct Batch {
  A a[TT_MAX_SIDES];
};

// grid (colblocks of 64, nchunks, towers); thread = (column, row-lane of 4)
struct BnStatArgs {
  const float* pre; int B, H, rows_per_chunk, nchunks; float* partial; float* mean; float* rstd; float* rm; float* rv; int64_t* nbt;
  int64_t pstride = 0;   // floats between chunks when READING partial (0 = 3 * H); chunks are always written densely
};

__global__ __launch_bounds__(kThreads) void bn_stats_partial_kernel(Batch<BnStatArgs> batch) {
  const BnStatArgs& a = batch.a[blockIdx.z];
  const int H = a.H;
  if ((int)blockIdx.x * 64 >= H || (int)blockIdx.y >= a.nchunks) return;
  __shared__ Wf sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * a.rows_per_chunk, r1 = min(a.B, r0 + a.rows_per_chunk);
  // per thread: sums of (x - x0) and (x - x0)^2 over its <= a few dozen rows, x0 = its first value (no division
  // and no dependent chain per element; the shift keeps the tiny-sample variance free of cancellation), turned
  // into (n, mean, M2) once and merged with Chan's formula from there on
  Wf w{0.f, 0.f, 0.f};
  if (c < H && r0 + rl < r1) {
    const float x0 = fmaxf(a.pre[(int64_t)(r0 + rl) * H + c], 0.f);
    float s = 0.f, q = 0.f, cnt = 0.f;
#pragma unroll 16
    for (int r = r0 + rl; r < r1; r += 4) {
      const float d = fmaxf(a.pre[(int64_t)r * H + c], 0.f) - x0;
      s += d;
      q += d * d;
      cnt += 1.f;
    }
    w.n = cnt;
    w.mean = x0 + s / cnt;
    w.m2 = fmaxf(q - s * (s / cnt), 0.f);
  }
  sh[rl][threadIdx.x & 63] = w;
  __syncthreads();
  if (rl == 0 && c < H) {
    Wf o = sh[0][threadIdx.x];
    o = wf_combine(o, sh[1][threadIdx.x]);
    o = wf_combine(o, sh[2][threadIdx.x]);
    o = wf_combine(o, sh[3][threadIdx.x]);
    float* p = a.partial + (int64_t)blockIdx.y * 3 * H;
    p[c] = o.n; p[H + c] = o.mean; p[2 * H + c] = o.m2;
  }
}

// finish: workgroup = 64 columns x 4 chunk-lanes; lane j combines chunks j, j+4, ... (all loads first), then the
// four partial results are combined in lane order (fixed order => reproducible)
__global__ __launch_bounds__(kThreads) void bn_stats_finish_kernel(Batch<BnStatArgs> batch) {
  const BnStatArgs& a = batch.a[blockIdx.y];
  const int H = a.H;
  if ((int)blockIdx.x * 64 >= H) return;
  __shared__ Wf sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), jl = threadIdx.x >> 6;
  Wf o{0.f, 0.f, 0.f};
  if (c < H) {
    Wf v[kMaxChunks / 4];
#pragma unroll
    for (int i = 0; i < kMaxChunks / 4; ++i) {
      const int k = jl + 4 * i;
      const float* p = a.partial + (int64_t)k * (a.pstride ? a.pstride : 3 * H);
      v[i] = k < a.nchunks ? Wf{p[c], p[H + c], p[2 * H + c]} : Wf{0.f, 0.f, 0.f};
    }
    o = wf_lane_merge(v);
  }
  sh[jl][threadIdx.x & 63] = o;
  __syncthreads();
  if (jl != 0 || c >= H) return;
  o = wf_combine(wf_combine(sh[0][threadIdx.x], sh[1][threadIdx.x]), wf_combine(sh[2][threadIdx.x], sh[3][threadIdx.x]));
  const float var = o.n > 0.f ? o.m2 / o.n : 0.f;
  a.mean[c] = o.mean;
  a.rstd[c] = 1.f / sqrtf(var + kBnEps);
  if (a.rm) {   // nn.BatchNorm1d: momentum 0.1, unbiased variance in the running estimate
    a.rm[c] = (1.f - kBnMomentum) * a.rm[c] + kBnMomentum * o.mean;
    a.rv[c] = (1.f - kBnMomentum) * a.rv[c] + kBnMomentum * (o.n > 1.f ? o.m2 / (o.n - 1.f) : var);
  }
  if (a.nbt && c == 0) a.nbt[0] += 1;
}

__global__ void bn_eval_prepare_kernel(Batch<BnStatArgs> batch) {
  const BnStatArgs& a = batch.a[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < a.H) {
    a.mean[c] = a.rm[c];
    a.rstd[c] = 1.f / sqrtf(a.rv[c] + kBnEps);
  }
}

struct BnApplyArgs {
  const float* pre; int64_t total; int H; const float* mean; const float* rstd; const float* g; const float* b;
  uint64_t salt; float* act;
};

__global__ __launch_bounds__(kThreads) void bn_apply_kernel(Batch<BnApplyArgs> batch, bool drop, float p, uint64_t seed0,
                                                           const uint64_t* __restrict__ seed_dev) {
  const BnApplyArgs& a = batch.a[blockIdx.y];
  const uint64_t seed = drop ? seed_of(seed0, seed_dev) : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int H = a.H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.total; i += stride) {
    const int c = (int)(i % H);
    const float x = fmaxf(a.pre[i], 0.f);
    const float y = (x - a.mean[c]) * a.rstd[c] * a.g[c] + a.b[c];
    a.act[i] = y * dropout_scale(drop, p, seed, a.salt + (uint64_t)i);
  }
}

// ---- deterministic two-stage column sums of the BatchNorm backward ------------------------------
//   da = d_act * dropscale ; xhat from pre ; S1 = sum da, S2 = sum da * xhat
struct ColArgs {
  const float* x; int64_t ldx;
  const float* pre; const float* mean; const float* rstd;
  uint64_t salt;
  int B, H, rows_per_chunk, nchunks;
  float* partial; float* out0; float* out1;
  int64_t pstride = 0;   // floats between chunks when READING partial (0 = 2 * H)
  float out_scale = 1.f; // colsum_finish_kernel: S1 / S2 are stored times this (1 / ranks under SyncBN); the raw sums go to sums_raw
  float* sums_raw = nullptr;   // optional [2 * H]: the unscaled S1 | S2 for bn_bwd_apply_kernel
};

__global__ __launch_bounds__(kThreads) void colsum_partial_kernel(Batch<ColArgs> batch, bool drop, float p, uint64_t seed0,
                                                                 const uint64_t* __restrict__ seed_dev) {
  const ColArgs& a = batch.a[blockIdx.z];
  const int H = a.H;
  if ((int)blockIdx.x * 64 >= H || (int)blockIdx.y >= a.nchunks) return;
  __shared__ float sh[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * a.rows_per_chunk, r1 = min(a.B, r0 + a.rows_per_chunk);
  float s0 = 0.f, s1 = 0.f;
  const uint64_t seed = drop ? seed_of(seed0, seed_dev) : 0;
  if (c < H) {
#pragma unroll 16
    for (int r = r0 + rl; r < r1; r += 4) {
      const int64_t i = (int64_t)r * H + c;
      const float da = a.x[(int64_t)r * a.ldx + c] * dropout_scale(drop, p, seed, a.salt + (uint64_t)i);
      const float xh = (fmaxf(a.pre[i], 0.f) - a.mean[c]) * a.rstd[c];
      s0 += da;
      s1 += da * xh;
    }
  }
  sh[0][rl][threadIdx.x & 63] = s0;
  sh[1][rl][threadIdx.x & 63] = s1;
  __syncthreads();
  if (rl == 0 && c < H) {
    float* q = a.partial + (int64_t)blockIdx.y * 2 * H;
    q[c] = ((sh[0][0][threadIdx.x] + sh[0][1][threadIdx.x]) + sh[0][2][threadIdx.x]) + sh[0][3][threadIdx.x];
    q[H + c] = ((sh[1][0][threadIdx.x] + sh[1][1][threadIdx.x]) + sh[1][2][threadIdx.x]) + sh[1][3][threadIdx.x];
  }
}

__global__ __launch_bounds__(kThreads) void colsum_finish_kernel(Batch<ColArgs> batch) {
  const ColArgs& a = batch.a[blockIdx.y];
  const int H = a.H;
  if ((int)blockIdx.x * 64 >= H) return;
  __shared__ float sh[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), jl = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (c < H) {
    float v0[kMaxChunks / 4], v1[kMaxChunks / 4];
#pragma unroll
    for (int i = 0; i < kMaxChunks / 4; ++i) {
      const int k = jl + 4 * i;
      const float* q = a.partial + (int64_t)k * (a.pstride ? a.pstride : 2 * H);
      v0[i] = k < a.nchunks ? q[c] : 0.f;
      v1[i] = k < a.nchunks ? q[H + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < kMaxChunks / 4; ++i) { s0 += v0[i]; s1 += v1[i]; }
  }
  sh[0][jl][threadIdx.x & 63] = s0;
  sh[1][jl][threadIdx.x & 63] = s1;
  __syncthreads();
  if (jl != 0 || c >= H) return;
  const float t0 = (sh[0][0][threadIdx.x] + sh[0][1][threadIdx.x]) + (sh[0][2][threadIdx.x] + sh[0][3][threadIdx.x]);
  const float t1 = (sh[1][0][threadIdx.x] + sh[1][1][threadIdx.x]) + (sh[1][2][threadIdx.x] + sh[1][3][threadIdx.x]);
  a.out0[c] = t0 * a.out_scale;
  a.out1[c] = t1 * a.out_scale;
  if (a.sums_raw) { a.sums_raw[c] = t0; a.sums_raw[H + c] = t1; }
}

// BN backward apply, in place on the gradient buffer:  d_act -> d_pre
//   train: d_a = g rstd (da - S1/B - xhat S2/B)   eval: d_a = da g rstd ;   d_pre = d_a [pre > 0]
struct BnBwdArgs {
  float* d; const float* pre; int64_t total; int H; float invB;
  const float* mean; const float* rstd; const float* g; const float* S1; const float* S2; uint64_t salt;
};

__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(Batch<BnBwdArgs> batch, bool train, bool drop, float p, uint64_t seed0,
                                                               const uint64_t* __restrict__ seed_dev) {
  const BnBwdArgs& a = batch.a[blockIdx.y];
  const uint64_t seed = drop ? seed_of(seed0, seed_dev) : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int H = a.H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.total; i += stride) {
    const int c = (int)(i % H);
    const float pr = a.pre[i];
    const float da = a.d[i] * dropout_scale(drop, p, seed, a.salt + (uint64_t)i);
    float dv;
    if (train) {
      const float xh = (fmaxf(pr, 0.f) - a.mean[c]) * a.rstd[c];
      dv = a.g[c] * a.rstd[c] * (da - a.S1[c] * a.invB - xh * (a.S2[c] * a.invB));
    } else {
      dv = da * a.g[c] * a.rstd[c];
    }
    a.d[i] = pr > 0.f ? dv : 0.f;
  }
}

// ---- row-wise L2 normalise: one wave per row ---------------------------------------------------
// a product that is rounded on its own (HIP's __fmul_rn is a plain `*`, which the compiler still contracts)
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

struct NormArgs { const float* y; const float* emb_in; const float* d_emb; int B, D; float* out; };

__global__ __launch_bounds__(kThreads) void l2norm_fwd_kernel(Batch<NormArgs> batch) {
  const NormArgs& a = batch.a[blockIdx.y];
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, D = a.D;
  if (row >= a.B) return;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float v = a.y[(int64_t)row * D + c];
    ss += v * v;
  }
  const float den = fmaxf(sqrtf(wave_sum(ss)), kNormEps);
  for (int c = lane; c < D; c += 64) a.out[(int64_t)row * D + c] = a.y[(int64_t)row * D + c] / den;
}

__global__ __launch_bounds__(kThreads) void l2norm_bwd_kernel(Batch<NormArgs> batch) {
  const NormArgs& a = batch.a[blockIdx.y];
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, D = a.D;
  if (row >= a.B) return;
  float ss = 0.f, dot = 0.f;
  for (int c = lane; c < D; c += 64) {
    const int64_t i = (int64_t)row * D + c;
    ss += a.y[i] * a.y[i];
    dot += a.emb_in[i] * a.d_emb[i];
  }
  const float nrm = sqrtf(wave_sum(ss));
  dot = wave_sum(dot);
  const float den = fmaxf(nrm, kNormEps);
  for (int c = lane; c < D; c += 64) {
    const int64_t i = (int64_t)row * D + c;
    a.out[i] = nrm > kNormEps ? (a.d_emb[i] - a.emb_in[i] * dot) / den : a.d_emb[i] / den;
  }
}

// The same arithmetic (same per-lane column sets, same fma chains, same butterflies: bit-identical) for widths that are a
// multiple of 64 up to 256, with RW rows per wave whose loads are all issued before the first reduction -- the one-row-at-a-time
// kernels above keep 1-4 loads in flight per lane and stream at 1.9 TB/s (B = 65536, D = 256: 72 / 107 us per tower).
template <int NC, int RW>
__global__ __launch_bounds__(kThreads) void l2norm_fwd_fast_kernel(Batch<NormArgs> batch) {
  const NormArgs& a = batch.a[blockIdx.y];
  const int lane = threadIdx.x & 63, D = 64 * NC;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= a.B) return;
  float v[RW][NC];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int j = 0; j < NC; ++j) v[r][j] = row0 + r < a.B ? a.y[(int64_t)(row0 + r) * D + lane + 64 * j] : 0.f;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    float ss = mul_rn(v[r][0], v[r][0]);
#pragma unroll
    for (int j = 1; j < NC; ++j) ss = __builtin_fmaf(v[r][j], v[r][j], ss);
    const float den = fmaxf(sqrtf(wave_sum(ss)), kNormEps);
    if (row0 + r < a.B) {
#pragma unroll
      for (int j = 0; j < NC; ++j) a.out[(int64_t)(row0 + r) * D + lane + 64 * j] = v[r][j] / den;
    }
  }
}

template <int NC, int RW>
__global__ __launch_bounds__(kThreads) void l2norm_bwd_fast_kernel(Batch<NormArgs> batch) {
  const NormArgs& a = batch.a[blockIdx.y];
  const int lane = threadIdx.x & 63, D = 64 * NC;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= a.B) return;
  float y[RW][NC], e[RW][NC], d[RW][NC];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const bool ok = row0 + r < a.B;
      const int64_t i = (int64_t)(row0 + r) * D + lane + 64 * j;
      y[r][j] = ok ? a.y[i] : 0.f;
      e[r][j] = ok ? a.emb_in[i] : 0.f;
      d[r][j] = ok ? a.d_emb[i] : 0.f;
    }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    float ss = mul_rn(y[r][0], y[r][0]), dot = mul_rn(e[r][0], d[r][0]);
#pragma unroll
    for (int j = 1; j < NC; ++j) {
      ss = __builtin_fmaf(y[r][j], y[r][j], ss);
      dot = __builtin_fmaf(e[r][j], d[r][j], dot);
    }
    const float nrm = sqrtf(wave_sum(ss));
    dot = wave_sum(dot);
    const float den = fmaxf(nrm, kNormEps);
    if (row0 + r < a.B) {
#pragma unroll
      for (int j = 0; j < NC; ++j)
        a.out[(int64_t)(row0 + r) * D + lane + 64 * j] = nrm > kNormEps ? (d[r][j] - e[r][j] * dot) / den : d[r][j] / den;
    }
  }
}

// widths of all towers of the batch equal and in {64, 128, 192, 256}: the multi-row kernels
template <bool BWD>
static bool launch_l2norm_fast(hipStream_t st, const Batch<NormArgs>& na, int n, int64_t B) {
  const int D = na.a[0].D;
  for (int t = 1; t < n; ++t)
    if (na.a[t].D != D || na.a[t].B != na.a[0].B) return false;
  if (D % 64 != 0 || D < 64 || D > 256) return false;
  constexpr int RW = BWD ? 2 : 4;
  const dim3 grid((unsigned)tt_cdiv(B, 4 * RW), (unsigned)n);
#define TT_L2(NC)                                                                              \
  do {                                                                                         \
    if (BWD) l2norm_bwd_fast_kernel<NC, RW><<<grid, kThreads, 0, st>>>(na);                    \
    else l2norm_fwd_fast_kernel<NC, RW><<<grid, kThreads, 0, st>>>(na);                        \
  } while (0)
  switch (D / 64) {
    case 1: TT_L2(1); break;
    case 2: TT_L2(2); break;
    case 3: TT_L2(3); break;
    default: TT_L2(4); break;
  }
#undef TT_L2
  return true;
}

// ---- fused narrow tail -------------------------------------------------------------------------
// When the last hidden block and the output are at most 64 wide (the [128, 64] towers of the reference's config)
// everything after the block's Linear is a few MB of data in a chain of ~5 us launches: slab reduction, BN
// statistics (2), BN apply, the 64x64 output Linear, L2 normalise -- and the mirror image in the backward pass.
// Training with bf16 GEMM operands, that chain runs as two kernels per pass (the split is the batch-wide reduction
// of the BN statistics).  Arithmetic and summation order follow the unfused kernels above step for step (the forward
// and the BN gradients come out bit-identical); only the output-layer weight / bias gradients are summed over
// different row chunks.  Workgroups of 1024 threads cover 64 rows: thread (c, rq) = column c, rows rq + 16 j; the
// ordered column reductions run on the first 256 threads (c, rl) exactly as in the unfused kernels.
using tl_f32x16 = __attribute__((ext_vector_type(16))) float;
using tl_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
constexpr int kTailThreads = 1024;
constexpr int kTailLd = 72;      // bf16 elements per LDS row of a 64-wide operand tile (144 B: aligned 16-B fragments)
constexpr int kTailLdF = 65;     // f32 row stride of the staging tiles

// (A) finish the split-K block GEMM (slabs + bias -> pre) and take the per-chunk BN statistics of relu(pre) in the same
// pass over the rows: slab_reduce_kernel + bn_stats_partial_kernel.  grid (nchunks, towers)
struct HeadArgs { const float* slabs; int64_t slab_stride; int splits; const float* bias; float* pre; BnStatArgs s; };

__global__ __launch_bounds__(kTailThreads) void tail_head_kernel(Batch<HeadArgs> batch) {
  const HeadArgs& h = batch.a[blockIdx.y];
  const BnStatArgs& a = h.s;
  const int H = a.H;
  if ((int)blockIdx.x >= a.nchunks) return;
  __shared__ float X[64 * kTailLdF];
  __shared__ Wf sh[4][64];
  const int t = threadIdx.x, c = t & 63, rq = t >> 6;
  const int r0 = blockIdx.x * a.rows_per_chunk, r1 = min(a.B, r0 + a.rows_per_chunk);
  const float bias = (h.bias && c < H) ? h.bias[c] : 0.f;
  float x0 = 0.f, s = 0.f, q = 0.f, cnt = 0.f;
  for (int b0 = r0; b0 < r1; b0 += 64) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < H) {
#pragma unroll 4
      for (int z = 0; z < h.splits; ++z) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = b0 + rq + 16 * j;
          if (r < r1) v[j] += h.slabs[(int64_t)z * h.slab_stride + (int64_t)r * H + c];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = rq + 16 * j, r = b0 + row;
        if (r < r1) {
          const float o = v[j] + bias;
          h.pre[(int64_t)r * H + c] = o;
          X[row * kTailLdF + c] = fmaxf(o, 0.f);
        }
      }
    }
    __syncthreads();
    if (t < 256 && c < H) {               // rq = row lane: rows r0 + rq, + 4, ... in bn_stats_partial_kernel's order
      const int nrow = min(64, r1 - b0);
      for (int row = rq; row < nrow; row += 4) {
        const float xv = X[row * kTailLdF + c];
        if (cnt == 0.f) x0 = xv;
        const float d = xv - x0;
        s += d;
        q += d * d;
        cnt += 1.f;
      }
    }
    __syncthreads();
  }
  if (t < 256) {
    Wf w{0.f, 0.f, 0.f};
    if (c < H && cnt > 0.f) {
      w.n = cnt;
      w.mean = x0 + s / cnt;
      w.m2 = fmaxf(q - s * (s / cnt), 0.f);
    }
    sh[rq][c] = w;
  }
  __syncthreads();
  if (t < 64 && c < H) {
    Wf o = sh[0][c];
    o = wf_combine(o, sh[1][c]);
    o = wf_combine(o, sh[2][c]);
    o = wf_combine(o, sh[3][c]);
    float* p = a.partial + (int64_t)blockIdx.x * 3 * H;
    p[c] = o.n; p[H + c] = o.mean; p[2 * H + c] = o.m2;
  }
}

// (A') the whole front of a one-block tower in ONE launch: dense projection (x[:, 0:h0] = dense . W_proj^T + b_proj, stored
// bf16), the block's Linear over x = [projection | looked-up rows] and the chunk BN statistics of relu(pre) -- the
// projection GEMM, the split-K block GEMM and tail_head_kernel above (3 dependent launches, 12.6 + 12.6 + 9.4 us at
// B = 8192: 77 MB of traffic of which 34 MB are split-K slabs written and read back).  One 512-thread workgroup per
// 64-row block (the BN chunks).  Operands go through LDS in 128-wide k stages: every thread loads 16-byte pieces of whole
// rows (coalesced: a fragment-shaped load straight from global memory -- one row per lane -- was tried first and ran at a
// quarter of the texture-address rate, 50 us), rounds f32 to bf16 on the way in, and the loads run THREE stages ahead of
// the MFMAs in registers -- all unconditional (a piece past the end re-reads the thread's own first piece and is zeroed),
// so the compiler can count them.  Bound: a CU draws L2-resident bytes at ~70 GB/s (MI355X_MICROARCH.md) and every
// workgroup re-reads its tower's f32 weights -- 426 KB of the 637 KB a notice workgroup moves: ~9 us.  The block GEMM's first stages are requested before the projection is computed.  K is split over the
// waves of the block GEMM (4 tiles x 2 k-halves; the projection's 8 tiles take a wave each), the partial tiles are added
// through LDS in a fixed order: reproducible, but not bit-identical to the split-K GEMM path.  The projection tile goes to x AND stays in
// LDS as the block GEMM's operand for k < h0.
// Needs: din <= 256, din and h0 + K E multiples of 64, h0 a multiple of 32 and <= 128, H <= 64.  grid (nchunks, towers)
struct FrontArgs {
  const float* dense; int64_t ld_dense; int din;
  const float* w_proj; const float* b_proj; int h0;
  uint16_t* x; int64_t ldx; int kx;
  const float* w; const float* bias; float* pre;
  BnStatArgs s;
  const uint16_t* w_proj16; const uint16_t* w16;   // W16: bf16 shadows of w_proj / w (tt_tower_params.w_proj_bf16 / w_bf16)
};
constexpr int kFS = 136;                                         // bf16 elements per LDS row of a 128-wide k stage
constexpr int kFrontThreads = 512;                               // 8 waves at up to 256 registers: three stages of loads in flight
constexpr int kFrontA = 64 * kFS * 2, kFrontB = 128 * kFS * 2;   // bytes of one A / B stage buffer
constexpr int kFrontLds = 2 * kFrontA + 2 * kFrontB + kFrontA + 4 * 64 * 12;

using tl_bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
__device__ __forceinline__ void front_put4(__bf16* dst, const float4& v, bool live) {
  tl_bf16x4 o;
  o[0] = (__bf16)(live ? v.x : 0.f); o[1] = (__bf16)(live ? v.y : 0.f);
  o[2] = (__bf16)(live ? v.z : 0.f); o[3] = (__bf16)(live ? v.w : 0.f);
  *reinterpret_cast<tl_bf16x4*>(dst) = o;
}

// uniform base + 32-bit byte offset: the global_load "saddr" form, one VGPR of address per load instead of two
template <typename T>
__device__ __forceinline__ T front_ld(const void* base, uint32_t off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + off);
}

// W16 (round 4): the weights come as bf16 shadows -- pieces of 8 elements laid out like the x pieces (row (t >> 4) + 32 pass,
// k = 8 (t & 15)), copied into LDS as they are: 213 of the 637 KB a notice workgroup moved were the f32 halves of values it rounded
// to bf16 anyway, and the 112 registers of weight pieces in flight become 56 (no scratch).  Same bits in LDS, same results.
template <bool W16>
__global__ __launch_bounds__(kFrontThreads) void tower_front_kernel(Batch<FrontArgs> batch) {
  const FrontArgs& f = batch.a[blockIdx.y];
  const BnStatArgs& a = f.s;
  const int H = a.H, B = a.B, h0 = f.h0, din = f.din, kx = f.kx;
  if ((int)blockIdx.x >= a.nchunks) return;
  extern __shared__ __attribute__((aligned(16))) char front_smem[];
  __bf16* bufA = reinterpret_cast<__bf16*>(front_smem);                              // [2][64][kFS]
  __bf16* bufB = reinterpret_cast<__bf16*>(front_smem + 2 * kFrontA);                // [2][128][kFS]
  __bf16* Pt = reinterpret_cast<__bf16*>(front_smem + 2 * kFrontA + 2 * kFrontB);    // [64][kFS] projection tile
  Wf (*sh)[64] = reinterpret_cast<Wf (*)[64]>(front_smem + 3 * kFrontA + 2 * kFrontB);
  float* part = reinterpret_cast<float*>(bufB);                                      // partial tiles (after the MFMAs): 32 KB
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 31, lh = lane >> 5;
  const int c = t & 63, rq = t >> 6;
  const int r0 = blockIdx.x * a.rows_per_chunk, r1 = min(B, r0 + a.rows_per_chunk);
  const float bias = (f.bias && c < H) ? f.bias[c] : 0.f;
  // piece of an f32 stage: row (t >> 5) + 16 pass, k = 4 (t & 31); piece of a bf16 stage: row (t >> 4) + 32 pass, k = 8 (t & 15)
  const int frow = t >> 5, fk = 4 * (t & 31), xrow = t >> 4, xk = 8 * (t & 15);
  // ADDRESS columns of a thread's first piece: a 64-wide operand (din = 64, kx = 64) fills only half of the 128-wide first stage,
  // and the lanes of the other half must re-read columns that exist -- unclamped, their piece of the LAST row lay up to 256 bytes
  // behind the end of the buffer (harmless for the result: those columns are zeroed on the way into LDS; a fault when the
  // buffer ends a mapping)
  const int fkp = din >= 128 ? fk : (fk & 63), fkb = kx >= 128 ? fk : (fk & 63), xkb = kx >= 128 ? xk : (xk & 63);
  const int xkp = din >= 128 ? xk : (xk & 63);
  const int ns2 = 3 * ((kx + 383) / 384);                       // block stages, padded to the unroll of 3 (pad stages are zeros)
  const int pn = (wave >> 1) * 32 + li;                         // this lane's projection column
  const float bp = pn < h0 ? f.b_proj[pn] : 0.f;
  float x0 = 0.f, s = 0.f, q = 0.f, cnt = 0.f;
  for (int b0 = r0; b0 < r1; b0 += 64) {
    // ---- everything that can be requested now: the projection's two stages, the block GEMM's first three ----
    constexpr int NPB = W16 ? 4 : 8, NQB = W16 ? 2 : 4;     // weight pieces per thread and stage: projection / block
    float4 pa[2][4], pb[2][NPB];                            // (W16: the float4 holds 8 bf16 -- moved, never computed on)
    uint32_t oa[4], ob[NPB];                                // byte offsets of this thread's rows (k = fk)
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) oa[ps] = (uint32_t)(min(b0 + frow + 16 * ps, B - 1) * (int)f.ld_dense + fkp) * 4u;
#pragma unroll
    for (int ps = 0; ps < NPB; ++ps)
      ob[ps] = W16 ? (uint32_t)(min(xrow + 32 * ps, h0 - 1) * din + xkp) * 2u : (uint32_t)(min(frow + 16 * ps, h0 - 1) * din + fkp) * 4u;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      // a piece past the end re-reads this thread's own first piece (discarded): never ONE address for the whole grid
      const uint32_t ko = 128 * st + fk < din ? 512u * st : 0u;
      const uint32_t ko16 = 128 * st + xk < din ? 256u * st : 0u;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) pa[st][ps] = front_ld<float4>(f.dense, oa[ps] + ko);
#pragma unroll
      for (int ps = 0; ps < NPB; ++ps) pb[st][ps] = W16 ? front_ld<float4>(f.w_proj16, ob[ps] + ko16) : front_ld<float4>(f.w_proj, ob[ps] + ko);
    }
    uint4 qa[3][2];
    float4 qb[3][NQB];
    uint32_t ox[2], ow[NQB];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) ox[ps] = (uint32_t)(min(b0 + xrow + 32 * ps, B - 1) * (int)f.ldx + xkb) * 2u;
#pragma unroll
    for (int ps = 0; ps < NQB; ++ps)
      ow[ps] = W16 ? (uint32_t)(min(xrow + 32 * ps, H - 1) * kx + xkb) * 2u : (uint32_t)(min(frow + 16 * ps, H - 1) * kx + fkb) * 4u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int ka = 128 * u + xk, kb = 128 * u + fk;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) qa[u][ps] = front_ld<uint4>(f.x, ox[ps] + (ka < kx ? 256u * u : 0u));
#pragma unroll
      for (int ps = 0; ps < NQB; ++ps)
        qb[u][ps] = W16 ? front_ld<float4>(f.w16, ow[ps] + (ka < kx ? 256u * u : 0u)) : front_ld<float4>(f.w, ow[ps] + (kb < kx ? 512u * u : 0u));
    }
    __builtin_amdgcn_sched_barrier(0);                      // 36 loads in flight before the first one is waited for
    // ---- projection: wave = tile (rt, ct) of the 64 x h0 block, whole K ----
    {
      const int rt = wave & 1, ct = wave >> 1;
      tl_f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bool live = 128 * st + fk < din, live16 = 128 * st + xk < din;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) front_put4(bufA + st * 64 * kFS + (frow + 16 * ps) * kFS + fk, pa[st][ps], live);
#pragma unroll
        for (int ps = 0; ps < NPB; ++ps) {
          if (W16) *reinterpret_cast<float4*>(bufB + st * 128 * kFS + (xrow + 32 * ps) * kFS + xk) = live16 ? pb[st][ps] : float4{0.f, 0.f, 0.f, 0.f};
          else front_put4(bufB + st * 128 * kFS + (frow + 16 * ps) * kFS + fk, pb[st][ps], live);
        }
      }
      {                                                     // third block stage, into the registers the projection just freed
        const int ka = 256 + xk, kb = 256 + fk;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) qa[2][ps] = front_ld<uint4>(f.x, ox[ps] + (ka < kx ? 512u : 0u));
#pragma unroll
        for (int ps = 0; ps < NQB; ++ps)
          qb[2][ps] = W16 ? front_ld<float4>(f.w16, ow[ps] + (ka < kx ? 512u : 0u)) : front_ld<float4>(f.w, ow[ps] + (kb < kx ? 1024u : 0u));
      }
      __syncthreads();
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const __bf16* A = bufA + st * 64 * kFS;
        const __bf16* Bm = bufB + st * 128 * kFS;
        if (128 * st < din) {
#pragma unroll
          for (int s2 = 0; s2 < 8; ++s2) {
            const int k = 16 * s2 + 8 * lh;
            const tl_bf16x8 av = *reinterpret_cast<const tl_bf16x8*>(A + (rt * 32 + li) * kFS + k);
            const tl_bf16x8 bv = *reinterpret_cast<const tl_bf16x8*>(Bm + (ct * 32 + li) * kFS + k);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (pn < h0) Pt[m * kFS + pn] = (__bf16)(acc[r] + bp);
      }
      __syncthreads();                                      // Pt complete; the stage buffers are free again
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)                        // the projection tile -> x[:, 0:h0], 16 bytes per piece
        if (xk < h0 && b0 + xrow + 32 * ps < r1)
          *reinterpret_cast<uint4*>(f.x + (int64_t)(b0 + xrow + 32 * ps) * f.ldx + xk) =
              *reinterpret_cast<const uint4*>(Pt + (xrow + 32 * ps) * kFS + xk);
    }
    // ---- block Linear: tile (rt, nh) of the 64 x 64 block, k-half kq = MFMA k-steps 4 kq .. 4 kq + 3 of each stage ----
    {
      const int tile = wave & 3, kq = wave >> 2, rt = tile & 1, nh = tile >> 1;
      tl_f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      for (int s0 = 0; s0 < ns2; s0 += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int st = s0 + u;
          __bf16* A = bufA + (st & 1) * 64 * kFS;
          __bf16* Bm = bufB + (st & 1) * 128 * kFS;
#pragma unroll
          for (int ps = 0; ps < 2; ++ps)
            *reinterpret_cast<uint4*>(A + (xrow + 32 * ps) * kFS + xk) = (128 * st + xk < kx) ? qa[u][ps] : uint4{0u, 0u, 0u, 0u};
#pragma unroll
          for (int ps = 0; ps < NQB; ++ps) {
            if (W16) *reinterpret_cast<float4*>(Bm + (xrow + 32 * ps) * kFS + xk) = (128 * st + xk < kx) ? qb[u][ps] : float4{0.f, 0.f, 0.f, 0.f};
            else front_put4(Bm + (frow + 16 * ps) * kFS + fk, qb[u][ps], 128 * st + fk < kx);
          }
          __syncthreads();
          {                                                 // three stages ahead, into the registers just stored
            const int ka = 128 * (st + 3) + xk, kb = 128 * (st + 3) + fk;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) qa[u][ps] = front_ld<uint4>(f.x, ox[ps] + (ka < kx ? 256u * (uint32_t)(st + 3) : 0u));
#pragma unroll
            for (int ps = 0; ps < NQB; ++ps)
              qb[u][ps] = W16 ? front_ld<float4>(f.w16, ow[ps] + (ka < kx ? 256u * (uint32_t)(st + 3) : 0u))
                              : front_ld<float4>(f.w, ow[ps] + (kb < kx ? 512u * (uint32_t)(st + 3) : 0u));
          }
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) {
            const int kl = 16 * (4 * kq + s2), kglob = 128 * st + kl;
            const __bf16* asrc = kglob < h0 ? Pt + (rt * 32 + li) * kFS + kglob + 8 * lh : A + (rt * 32 + li) * kFS + kl + 8 * lh;
            const tl_bf16x8 av = *reinterpret_cast<const tl_bf16x8*>(asrc);
            const tl_bf16x8 bv = *reinterpret_cast<const tl_bf16x8*>(Bm + (nh * 32 + li) * kFS + kl + 8 * lh);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
          }
        }
      }
      __syncthreads();                                    // last stage read: bufB becomes the partial-tile area [2][64][64]
      const int n = nh * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) part[(kq * 64 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + n] = acc[r];
    }
    __syncthreads();
    // ---- k-halves added in order + bias -> pre; relu tile (aliases part[0]: each thread overwrites what it read) ----
    if (c < H) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = rq + 8 * j;
        v[j] = (part[row * 64 + c] + part[(64 + row) * 64 + c]) + bias;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = rq + 8 * j, r = b0 + row;
        if (r < r1) f.pre[(int64_t)r * H + c] = v[j];
        part[row * 64 + c] = fmaxf(v[j], 0.f);
      }
    }
    __syncthreads();
    if (t < 256 && c < H) {               // as tail_head_kernel / bn_stats_partial_kernel: rows rq, rq + 4, ... of the block
      const int nrow = min(64, r1 - b0);
      for (int row = rq; row < nrow; row += 4) {
        const float xv = part[row * 64 + c];
        if (cnt == 0.f) x0 = xv;
        const float d = xv - x0;
        s += d;
        q += d * d;
        cnt += 1.f;
      }
    }
    __syncthreads();
  }
  if (t < 256) {
    Wf w{0.f, 0.f, 0.f};
    if (c < H && cnt > 0.f) {
      w.n = cnt;
      w.mean = x0 + s / cnt;
      w.m2 = fmaxf(q - s * (s / cnt), 0.f);
    }
    sh[rq][c] = w;
  }
  __syncthreads();
  if (t < 64 && c < H) {
    Wf o = sh[0][c];
    o = wf_combine(o, sh[1][c]);
    o = wf_combine(o, sh[2][c]);
    o = wf_combine(o, sh[3][c]);
    float* p = a.partial + (int64_t)blockIdx.x * 3 * H;
    p[c] = o.n; p[H + c] = o.mean; p[2 * H + c] = o.m2;
  }
}

// (B) BN statistics finish (every workgroup, same order as bn_stats_finish_kernel) + BN apply + dropout + output Linear
// (one 64 x 64 x 64 bf16 MFMA tile) + L2 normalise, for 64 rows per workgroup.  grid (cdiv(B, 64), towers)
struct TailFwdArgs {
  BnStatArgs s;
  const float* g; const float* b; uint64_t salt; float* act;
  const float* w_out; const float* b_out; int D; float* y; float* emb;
  __bf16* pk_rows; __bf16* pk_frag; int Dp; float pk_scale;   // optional: the score kernels' two operand images of emb (tt_score_pack_bf16)
};

static_assert(kTailThreads == kRiderThreads, "the riders run in the tail kernels' workgroups");
// (cr_wg > 0: the FIRST grid row -- dispatched first: not every workgroup of the launch is resident at once -- is the keyed plan's
// compaction riding in this launch: tt_riders.h)
// Round 4: at most 64 registers, so that TWO of these 1024-thread workgroups share a CU: the launch is 256 tail workgroups plus the
// riding compaction's grid row, and at 116 registers (one workgroup per CU) the riders' row -- dispatched first -- made half the
// tail workgroups wait for a CU (18.6 us for a 10-us chain).  The 96 registers were the chunk statistics' partials, all in flight at
// once: every thread now fetches and merges ONE sub-chain of eight chunks (wf_lane_merge's order), all 1024 threads taking part.
// measurement aid, compiled in with -DTT_TAIL_STAMPS only (tools/r04_tail_stamps.py): phase stamps of the tail kernels' workgroups
#ifdef TT_TAIL_STAMPS
__device__ unsigned long long g_tail_stamps[2][512 * 8];
#define TT_TAIL_STAMP(k, i) do { if (threadIdx.x == 0) g_tail_stamps[k][((blockIdx.y * gridDim.x + blockIdx.x) & 511) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TT_TAIL_STAMP(k, i) do { } while (0)
#endif

__global__ __launch_bounds__(kTailThreads) __attribute__((amdgpu_waves_per_eu(8, 8)))
void tail_fwd_kernel(Batch<TailFwdArgs> batch, bool drop, float p, uint64_t seed0,
                     const uint64_t* __restrict__ seed_dev, CompactRider cr, int cr_wg) {
  if (cr_wg > 0 && blockIdx.y == 0) {
    if ((int)blockIdx.x < cr_wg) compact_body(cr, blockIdx.x);
    return;
  }
  const TailFwdArgs& f = batch.a[blockIdx.y - (cr_wg > 0 ? 1 : 0)];
  const BnStatArgs& a = f.s;
  const int H = a.H, D = f.D, B = a.B;
  const int m0 = blockIdx.x * 64;
  if (m0 >= B) return;
  TT_TAIL_STAMP(0, 0);
  __shared__ Wf sh[16][64];
  __shared__ float s_mean[64], s_rstd[64];
  __shared__ __attribute__((aligned(16))) __bf16 As[64 * kTailLd];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[64 * kTailLd];
  __shared__ float Y[64 * kTailLdF];
  const int t = threadIdx.x, c = t & 63, rq = t >> 6;
  // the rows' pre-activations are in flight while the statistics settle
  float pre[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = m0 + rq + 16 * j;
    const float v = a.pre[(int64_t)min(r, B - 1) * H + min(c, H - 1)];     // (clamped address, no branch around the load)
    pre[j] = (c < H && r < B) ? v : 0.f;
  }
  {                                                      // thread (c, rq): sub-chain rq >> 2 of chunk lane rq & 3 (wf_lane_merge's order)
    Wf cs{0.f, 0.f, 0.f};
    if (c < H) {
      const int jl = rq & 3, sc = rq >> 2;
      Wf v[kSubChain];
      const int64_t ps = a.pstride ? a.pstride : 3 * H;
#pragma unroll
      for (int i = 0; i < kSubChain; ++i) {
        const int k = jl + 4 * (kSubChain * sc + i);
        const float* q = a.partial + (int64_t)min(k, a.nchunks - 1) * ps;      // (clamped address, no branch around the loads)
        const float x0 = q[c], x1 = q[H + c], x2 = q[2 * H + c];
        v[i] = k < a.nchunks ? Wf{x0, x1, x2} : Wf{0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < kSubChain; ++i) cs = wf_combine(cs, v[i]);
    }
    sh[rq][c] = cs;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {                          // W_out [D, H] -> Bs[n][k]
    const int n = rq + 16 * j;
    const float w = f.w_out[(int64_t)min(n, D - 1) * H + min(c, H - 1)];
    Bs[n * kTailLd + c] = (__bf16)((n < D && c < H) ? w : 0.f);
  }
  __syncthreads();
  TT_TAIL_STAMP(0, 1);
  if (t < 64 && c < H) {
    Wf ln[4];
#pragma unroll
    for (int jl = 0; jl < 4; ++jl) {                       // lane(jl) = ((C0 + C1) + C2) + C3, starting from the empty statistic
      Wf o{0.f, 0.f, 0.f};
#pragma unroll
      for (int sc = 0; sc < 4; ++sc) o = wf_combine(o, sh[4 * sc + jl][c]);
      ln[jl] = o;
    }
    const Wf o = wf_combine(wf_combine(ln[0], ln[1]), wf_combine(ln[2], ln[3]));
    const float var = o.n > 0.f ? o.m2 / o.n : 0.f;
    const float rstd = 1.f / sqrtf(var + kBnEps);
    s_mean[c] = o.mean;
    s_rstd[c] = rstd;
    if (blockIdx.x == 0) {
      a.mean[c] = o.mean;
      a.rstd[c] = rstd;
      if (a.rm) {
        a.rm[c] = (1.f - kBnMomentum) * a.rm[c] + kBnMomentum * o.mean;
        a.rv[c] = (1.f - kBnMomentum) * a.rv[c] + kBnMomentum * (o.n > 1.f ? o.m2 / (o.n - 1.f) : var);
      }
      if (a.nbt && c == 0) a.nbt[0] += 1;
    }
  }
  __syncthreads();
  TT_TAIL_STAMP(0, 2);
  {
    const uint64_t seed = drop ? seed_of(seed0, seed_dev) : 0;
    const float mean = c < H ? s_mean[c] : 0.f, rstd = c < H ? s_rstd[c] : 0.f;
    const float g = c < H ? f.g[c] : 0.f, bb = c < H ? f.b[c] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = rq + 16 * j, r = m0 + row;
      float v = 0.f;
      if (c < H && r < B) {
        const int64_t i = (int64_t)r * H + c;
        const float x = fmaxf(pre[j], 0.f);
        const float yv = (x - mean) * rstd * g + bb;
        v = yv * dropout_scale(drop, p, seed, f.salt + (uint64_t)i);
        f.act[i] = v;
      }
      As[row * kTailLd + c] = (__bf16)v;
    }
  }
  __syncthreads();
  TT_TAIL_STAMP(0, 3);
  const int lane = t & 63, wave = t >> 6;
  if (wave < 4) {
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
    tl_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const tl_bf16x8 av = *reinterpret_cast<const tl_bf16x8*>(As + (wr * 32 + li) * kTailLd + 16 * s2 + 8 * lh);
      const tl_bf16x8 bv = *reinterpret_cast<const tl_bf16x8*>(Bs + (wc * 32 + li) * kTailLd + 16 * s2 + 8 * lh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    }
    const int n = wc * 32 + li;
    const float bv = n < D ? f.b_out[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      Y[m * kTailLdF + n] = acc[r] + bv;
    }
  }
  __syncthreads();
  TT_TAIL_STAMP(0, 4);
  // one wave per row as l2norm_fwd_kernel, four independent rows per wave in flight.  mul_rn: a plain v * v is
  // contracted into the first butterfly add (fma(v, v, partner's square)), which leaves the two partners -- and so the
  // lanes of one row -- with differently rounded sums
  float v[4], den[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = lane < D ? Y[(wave * 4 + j) * kTailLdF + lane] : 0.f;
    den[j] = mul_rn(v[j], v[j]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) den[j] += __shfl_xor(den[j], o);
  }
  float e[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = m0 + wave * 4 + j;
    const float d = fmaxf(sqrtf(den[j]), kNormEps);
    e[j] = 0.f;
    if (r < B && lane < D) {
      e[j] = v[j] / d;
      f.y[(int64_t)r * D + lane] = v[j];
      f.emb[(int64_t)r * D + lane] = e[j];
    }
  }
  if (f.pk_rows && lane < f.Dp) {
    // the unit rows straight into the score kernels' operand images (layouts: pack_bf16_kernel in tt_score_bf16.hip; padding
    // rows and columns are written as zeros, this workgroup owns rows m0 .. m0 + 63 of both images):
    //   rows image  [tile][k-step][half][row in tile][8]  -- element (row, d): chunk ((t * Dp/16 + d/16) * 2 + (d%16)/8) * 32 + row%32
    //   frag image  [tile][s][h][d][8]                    -- element (row, d): 8 rows of one column; this wave's 4 rows are 4 adjacent slots
    const int Dp = f.Dp, d = lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = m0 + wave * 4 + j;
      const int64_t c = (((int64_t)(row >> 5) * (Dp >> 4) + (d >> 4)) * 2 + ((d >> 3) & 1)) * 32 + (row & 31);
      f.pk_rows[c * 8 + (d & 7)] = (__bf16)(e[j] * f.pk_scale);
    }
    const int row0 = m0 + wave * 4, rr = row0 & 31, q = rr & 15;
    const int64_t fi = (((int64_t)(row0 >> 5) * 2 + (rr >> 4)) * 2 + ((q & 7) >> 2)) * Dp + d;
    using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (__bf16)(e[j] * f.pk_scale);
    *reinterpret_cast<bf16x4*>(f.pk_frag + fi * 8 + (q >> 3) * 4) = o;
  }
#ifdef TT_TAIL_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  TT_TAIL_STAMP(0, 5);
#endif
}

// (B') the same kernel for wider towers: last hidden width H <= 256, output D <= 128 (scripts/train.py's own [512, 256] -> 128).
// 64 rows per workgroup as above; the column-wise parts run over H / 64 column blocks, the output Linear is a [64 x H] . [H x D]
// product on 8 waves (one 32 x 32 tile each, the whole K in one MFMA chain: the order of gemm_bf16_kernel without split-K),
// a lane of the L2-normalise owns D / 64 columns.  Statistics finish, BN apply and the normalise keep the separate kernels'
// arithmetic and order.  Dynamic LDS: A tile [64][H + 8] bf16 | W_out [D][H + 8] bf16 | Y [64][D + 1] f32 | statistics.
constexpr int kWideH = 256, kWideD = 128;
constexpr int kWideLds = 64 * (kWideH + 8) * 2 + kWideD * (kWideH + 8) * 2 + 64 * (kWideD + 1) * 4 + 4 * 4 * 64 * 12 + 2 * kWideH * 4;

__global__ __launch_bounds__(kTailThreads) void tail_fwd_wide_kernel(Batch<TailFwdArgs> batch, bool drop, float p, uint64_t seed0,
                                                                    const uint64_t* __restrict__ seed_dev) {
  const TailFwdArgs& f = batch.a[blockIdx.y];
  const BnStatArgs& a = f.s;
  const int H = a.H, D = f.D, B = a.B;
  const int m0 = blockIdx.x * 64;
  if (m0 >= B) return;
  const int Hp = (H + 15) & ~15;                             // K of the output product: H padded to the MFMA's 16 (pad columns are zeros)
  const int ldA = Hp + 8;                                    // bf16 elements per LDS row of the A tile and of W_out
  extern __shared__ __attribute__((aligned(16))) char wide_smem[];
  __bf16* As = reinterpret_cast<__bf16*>(wide_smem);                                   // [64][ldA]
  __bf16* Bs = As + 64 * (kWideH + 8);                                                  // [D][ldA]
  float* Y = reinterpret_cast<float*>(Bs + kWideD * (kWideH + 8));                      // [64][D + 1]
  Wf (*sh)[4][64] = reinterpret_cast<Wf (*)[4][64]>(Y + 64 * (kWideD + 1));             // [4 column blocks][4 chunk lanes][64]
  float* s_mean = reinterpret_cast<float*>(sh + 4);
  float* s_rstd = s_mean + kWideH;
  const int t = threadIdx.x, c = t & 63, rq = t >> 6;
  const int HB = (H + 63) / 64;
  // statistics: thread = (column block t >> 8, chunk lane (t >> 6) & 3, column): the order of bn_stats_finish_kernel
  {
    const int cb = t >> 8, jl = (t >> 6) & 3, col = cb * 64 + c;
    Wf o{0.f, 0.f, 0.f};
    if (cb < HB && col < H) {
      Wf v[kMaxChunks / 4];
      const int64_t ps = a.pstride ? a.pstride : 3 * H;
#pragma unroll
      for (int i = 0; i < kMaxChunks / 4; ++i) {
        const int k = jl + 4 * i;
        const float* q = a.partial + (int64_t)k * ps;
        v[i] = k < a.nchunks ? Wf{q[col], q[H + col], q[2 * H + col]} : Wf{0.f, 0.f, 0.f};
      }
      o = wf_lane_merge(v);
    }
    sh[cb][jl][c] = o;
  }
  // W_out [D, H] -> Bs[n][k] (coalesced along k); columns H .. Hp - 1 of both operand tiles are zeros
  for (int e = t; e < D * H; e += kTailThreads) {
    const int n = e / H, k = e - n * H;
    Bs[n * ldA + k] = (__bf16)f.w_out[e];
  }
  for (int e = t; e < (D + 64) * (Hp - H); e += kTailThreads) {
    const int r = e / (Hp - H), k = H + e % (Hp - H);
    if (r < D) Bs[r * ldA + k] = (__bf16)0.f;
    else As[(r - D) * ldA + k] = (__bf16)0.f;
  }
  __syncthreads();
  if (t < 256) {
    const int col = t;                                        // one thread per column finishes its four chunk lanes
    if (col < H) {
      const int cb = col >> 6, cc = col & 63;
      const Wf o = wf_combine(wf_combine(sh[cb][0][cc], sh[cb][1][cc]), wf_combine(sh[cb][2][cc], sh[cb][3][cc]));
      const float var = o.n > 0.f ? o.m2 / o.n : 0.f;
      const float rstd = 1.f / sqrtf(var + kBnEps);
      s_mean[col] = o.mean;
      s_rstd[col] = rstd;
      if (blockIdx.x == 0) {
        a.mean[col] = o.mean;
        a.rstd[col] = rstd;
        if (a.rm) {
          a.rm[col] = (1.f - kBnMomentum) * a.rm[col] + kBnMomentum * o.mean;
          a.rv[col] = (1.f - kBnMomentum) * a.rv[col] + kBnMomentum * (o.n > 1.f ? o.m2 / (o.n - 1.f) : var);
        }
        if (a.nbt && col == 0) a.nbt[0] += 1;
      }
    }
  }
  __syncthreads();
  {
    const uint64_t seed = drop ? seed_of(seed0, seed_dev) : 0;
    for (int cb = 0; cb < HB; ++cb) {
      const int col = cb * 64 + c;
      const bool on = col < H;
      const float mean = on ? s_mean[col] : 0.f, rstd = on ? s_rstd[col] : 0.f;
      const float g = on ? f.g[col] : 0.f, bb = on ? f.b[col] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = rq + 16 * j, r = m0 + row;
        float v = 0.f;
        if (on && r < B) {
          const int64_t i = (int64_t)r * H + col;
          const float x = fmaxf(a.pre[i], 0.f);
          const float yv = (x - mean) * rstd * g + bb;
          v = yv * dropout_scale(drop, p, seed, f.salt + (uint64_t)i);
          f.act[i] = v;
        }
        if (on) As[row * ldA + col] = (__bf16)v;
      }
    }
  }
  __syncthreads();
  const int lane = t & 63, wave = t >> 6;
  const int DT = (D + 31) / 32;                               // 32-column tiles of the output (<= 4)
  if (wave < 2 * DT) {
    const int wr = wave & 1, wc = wave >> 1, li = lane & 31, lh = lane >> 5;
    tl_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nrow = min(wc * 32 + li, D - 1);                // (columns past D: a valid row of W_out, result discarded)
    for (int k0 = 0; k0 < Hp; k0 += 16) {
      const tl_bf16x8 av = *reinterpret_cast<const tl_bf16x8*>(As + (wr * 32 + li) * ldA + k0 + 8 * lh);
      const tl_bf16x8 bv = *reinterpret_cast<const tl_bf16x8*>(Bs + nrow * ldA + k0 + 8 * lh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    }
    const int n = wc * 32 + li;
    const float bo = n < D ? f.b_out[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (n < D) Y[m * (kWideD + 1) + n] = acc[r] + bo;
    }
  }
  __syncthreads();
  // one wave per row as l2norm_fwd_kernel, four rows per wave in flight; a lane owns columns lane and lane + 64
  float v[4][2], den[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float ssq = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int col = lane + 64 * u;
      v[j][u] = col < D ? Y[(wave * 4 + j) * (kWideD + 1) + col] : 0.f;
    }
    // per-lane partial in column order (lane, lane + 64), then the butterfly: l2norm_fwd_kernel's order
    ssq = __builtin_fmaf(v[j][1], v[j][1], mul_rn(v[j][0], v[j][0]));    // l2norm_fwd_kernel's contracted `ss += v * v` over lane, lane + 64
    den[j] = ssq;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) den[j] += __shfl_xor(den[j], o);
  }
  float e[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = m0 + wave * 4 + j;
    const float d = fmaxf(sqrtf(den[j]), kNormEps);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int col = lane + 64 * u;
      e[j][u] = 0.f;
      if (r < B && col < D) {
        e[j][u] = v[j][u] / d;
        f.y[(int64_t)r * D + col] = v[j][u];
        f.emb[(int64_t)r * D + col] = e[j][u];
      }
    }
  }
  if (f.pk_rows) {
    const int Dp = f.Dp;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int d = lane + 64 * u;
      if (d >= Dp) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = m0 + wave * 4 + j;
        const int64_t cidx = (((int64_t)(row >> 5) * (Dp >> 4) + (d >> 4)) * 2 + ((d >> 3) & 1)) * 32 + (row & 31);
        f.pk_rows[cidx * 8 + (d & 7)] = (__bf16)(e[j][u] * f.pk_scale);
      }
      const int row0 = m0 + wave * 4, rr = row0 & 31, q = rr & 15;
      const int64_t fi = (((int64_t)(row0 >> 5) * 2 + (rr >> 4)) * 2 + ((q & 7) >> 2)) * Dp + d;
      using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
      bf16x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (__bf16)(e[j][u] * f.pk_scale);
      *reinterpret_cast<bf16x4*>(f.pk_frag + fi * 8 + (q >> 3) * 4) = o;
    }
  }
}

// (C) backward head: L2-normalise backward -> d_y; d_act = d_y . W_out; the chunk's share of the output-layer weight /
// bias gradients (d_y^T . act, column sums of d_y) into slabs; per-chunk BN column sums S1 / S2.  One workgroup per
// row chunk (the chunks of colsum_partial_kernel), 64 rows at a time.  d_act leaves this kernel already multiplied by
// the dropout scale (tail_bwd_apply_kernel does not regenerate the mask).  grid (nchunks, towers)
struct TailBwdArgs {
  const float* y; const float* emb; const float* d_emb; float* d_y; int D;
  const float* w_out; const float* act; float* d_act;
  ColArgs col;                       // x / ldx unused: d_act is taken from the tile
  float* w_slab; float* b_slab;      // [nchunks][D * H], [nchunks][D]
};

// (fr_on: the FIRST grid row is one workgroup running the symmetric score forward's loss reduction -- tt_riders.h)
__global__ __launch_bounds__(kTailThreads) void tail_bwd_kernel(Batch<TailBwdArgs> batch, bool drop, float p, uint64_t seed0,
                                                               const uint64_t* __restrict__ seed_dev, Finish2Rider fr, int fr_on) {
  if (fr_on && blockIdx.y == 0) {
    if (blockIdx.x == 0) finish2_body(fr);
    return;
  }
  const TailBwdArgs& f = batch.a[blockIdx.y - fr_on];
  const ColArgs& a = f.col;
  const int H = a.H, D = f.D;
  if ((int)blockIdx.x >= a.nchunks) return;
  // dyA [row][d] | dyT [d][row] | (XH f32 tile aliases these two once the MFMAs are done) ; Wn [h][d] ; actT [h][row]
  __shared__ __attribute__((aligned(16))) __bf16 dy2[2 * 64 * kTailLd];
  __shared__ __attribute__((aligned(16))) __bf16 Wn[64 * kTailLd];
  __shared__ __attribute__((aligned(16))) __bf16 actT[64 * kTailLd];
  __shared__ float DY[64 * kTailLdF];                                   // d_y, then d_act, of the 64 rows
  __shared__ float sh[3][4][64];
  static_assert(sizeof(float) * 64 * kTailLdF <= sizeof(__bf16) * 2 * 64 * kTailLd, "XH must fit over dyA | dyT");
  __bf16* dyA = dy2;
  __bf16* dyT = dy2 + 64 * kTailLd;
  float* XH = reinterpret_cast<float*>(dy2);
  const int t = threadIdx.x, c = t & 63, rq = t >> 6;
  const int lane = c, wave = rq;
  const int r0 = blockIdx.x * a.rows_per_chunk, r1 = min(a.B, r0 + a.rows_per_chunk);
  const uint64_t seed = drop ? seed_of(seed0, seed_dev) : 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {                         // W_out [D, H]: element (k = d, n = h) -> Wn[h][d]
    const int d = rq + 16 * j;
    const float w = f.w_out[(int64_t)min(d, D - 1) * H + min(c, H - 1)];
    Wn[c * kTailLd + d] = (__bf16)((d < D && c < H) ? w : 0.f);
  }
  const float mean = c < H ? a.mean[c] : 0.f, rstd = c < H ? a.rstd[c] : 0.f;
  tl_f32x16 accw;
#pragma unroll
  for (int i = 0; i < 16; ++i) accw[i] = 0.f;
  float s0 = 0.f, s1 = 0.f, cs = 0.f;
  for (int b0 = r0; b0 < r1; b0 += 64) {
    // 1. d_y of rows b0 .. b0 + 63: one wave per row as l2norm_bwd_kernel, four independent rows per wave in flight
    float yv[4], e[4], de[4], ss[4], dot[4], xh[4], sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = b0 + wave * 4 + j;
      const bool ok = r < r1 && lane < D;                 // (unconditional loads at clamped addresses: a load under a per-lane
      const int64_t i = (int64_t)min(r, r1 - 1) * D + min(lane, D - 1);   //  condition is waited for before the next is issued)
      const float v0 = f.y[i], v1 = f.emb[i], v2 = f.d_emb[i];
      yv[j] = ok ? v0 : 0.f;
      e[j] = ok ? v1 : 0.f;
      de[j] = ok ? v2 : 0.f;
    }
    float av[4], pr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                       // thread (c, rq): rows rq + 16 j of act / pre
      const int r = b0 + rq + 16 * j;
      const bool ok = c < H && r < r1;
      const int64_t i = (int64_t)min(r, r1 - 1) * H + min(c, H - 1);
      const float v0 = f.act[i], v1 = a.pre[i];
      av[j] = ok ? v0 : 0.f;
      pr[j] = ok ? v1 : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ss[j] = mul_rn(yv[j], yv[j]);
      dot[j] = mul_rn(e[j], de[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ss[j] += __shfl_xor(ss[j], o);
        dot[j] += __shfl_xor(dot[j], o);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wave * 4 + j, r = b0 + row;
      const float nrm = sqrtf(ss[j]);
      const float den = fmaxf(nrm, kNormEps);
      float out = nrm > kNormEps ? (de[j] - e[j] * dot[j]) / den : de[j] / den;
      if (r < r1 && lane < D) f.d_y[(int64_t)r * D + lane] = out;
      else out = 0.f;
      DY[row * kTailLdF + lane] = out;
      dyA[row * kTailLd + lane] = (__bf16)out;
      dyT[lane * kTailLd + row] = (__bf16)out;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = rq + 16 * j, r = b0 + row;
      actT[c * kTailLd + row] = (__bf16)av[j];
      xh[j] = (fmaxf(pr[j], 0.f) - mean) * rstd;
      sc[j] = (c < H && r < r1) ? dropout_scale(drop, p, seed, a.salt + (uint64_t)((int64_t)r * H + c)) : 0.f;
    }
    __syncthreads();
    // 2. bias-gradient column sums of d_y, data gradient d_act = d_y . W_out, weight-gradient tile += d_y^T . act
    const int wr = (wave >> 1) & 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
    tl_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    if (wave < 4) {
      if (c < D) {
#pragma unroll
        for (int j = 0; j < 16; ++j) cs += DY[(rq + 4 * j) * kTailLdF + c];
      }
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const int ko = 16 * s2 + 8 * lh;
        const tl_bf16x8 a1 = *reinterpret_cast<const tl_bf16x8*>(dyA + (wr * 32 + li) * kTailLd + ko);
        const tl_bf16x8 b1 = *reinterpret_cast<const tl_bf16x8*>(Wn + (wc * 32 + li) * kTailLd + ko);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
        const tl_bf16x8 a2 = *reinterpret_cast<const tl_bf16x8*>(dyT + (wr * 32 + li) * kTailLd + ko);
        const tl_bf16x8 b2 = *reinterpret_cast<const tl_bf16x8*>(actT + (wc * 32 + li) * kTailLd + ko);
        accw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, accw, 0, 0, 0);
      }
    }
    __syncthreads();                                     // DY (as d_y), dyA and dyT have been read
    if (wave < 4) {
      const int n = wc * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) DY[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * kTailLdF + n] = acc[r];
    }
    __syncthreads();
    // 3. d_act (times the dropout scale) out; da and xhat staged for the ordered column sums
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = rq + 16 * j, r = b0 + row;
      const float da = DY[row * kTailLdF + c] * sc[j];
      if (c < H && r < r1) f.d_act[(int64_t)r * H + c] = da;
      DY[row * kTailLdF + c] = da;
      XH[row * kTailLdF + c] = xh[j];
    }
    __syncthreads();
    if (t < 256 && c < H) {                              // colsum_partial_kernel's order: rows r0 + rq, + 4, ...
      const int nrow = min(64, r1 - b0);
      for (int row = rq; row < nrow; row += 4) {
        const float da = DY[row * kTailLdF + c];
        s0 += da;
        s1 += da * XH[row * kTailLdF + c];
      }
    }
    __syncthreads();
  }
  if (t < 256) {
    sh[0][rq][c] = s0;
    sh[1][rq][c] = s1;
    sh[2][rq][c] = cs;
  }
  __syncthreads();
  if (t < 64) {
    if (c < H) {
      float* q = a.partial + (int64_t)blockIdx.x * 2 * H;
      q[c] = ((sh[0][0][c] + sh[0][1][c]) + sh[0][2][c]) + sh[0][3][c];
      q[H + c] = ((sh[1][0][c] + sh[1][1][c]) + sh[1][2][c]) + sh[1][3][c];
    }
    if (c < D) f.b_slab[(int64_t)blockIdx.x * D + c] = ((sh[2][0][c] + sh[2][1][c]) + sh[2][2][c]) + sh[2][3][c];
  }
  if (wave < 4) {
    const int wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
    float* ws = f.w_slab + (int64_t)blockIdx.x * D * H;
    const int n = wc * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < D && n < H) ws[(int64_t)m * H + n] = accw[r];
    }
  }
}

// (D) S1 / S2 finish (every workgroup, colsum_finish_kernel's order) + BN backward apply in place on the (already
// dropout-scaled) d_act, 64 rows per workgroup.  grid (cdiv(B, 64), towers)
struct TailApplyArgs {
  ColArgs col; BnBwdArgs bn;
  const float* w_slab; const float* b_slab; int D; float* g_w_out; float* g_b_out;   // [slab_chunks][D * H], [slab_chunks][D] -> [D, H], [D]
  int slab_chunks;      // row chunks of tail_bwd_kernel on THIS rank (col.nchunks counts ranks when the sums come from all ranks)
  float out_scale;      // BN weight / bias gradients are stored times this (1 / ranks under SyncBN: see twotower.h)
};

__global__ __launch_bounds__(kTailThreads) void tail_bwd_apply_kernel(Batch<TailApplyArgs> batch) {
  const TailApplyArgs& ta = batch.a[blockIdx.y];
  const ColArgs& a = ta.col;
  const BnBwdArgs& b = ta.bn;
  const int H = a.H, B = a.B;
  const int t = threadIdx.x, c = t & 63, rq = t >> 6;
  const int m0 = blockIdx.x * 64;
  if (m0 >= B) return;
  __shared__ float sh[2][4][64];
  __shared__ float S[2][64];
  __shared__ float red[12][64];
  // output-layer weight / bias gradients: the per-chunk slabs of tail_bwd_kernel summed in a fixed order, 64 elements
  // per workgroup, by the 768 threads that are idle while the first 256 finish S1 / S2 -- thread (e, zl): slabs zl,
  // zl + 12, ...; then the 12 partial sums in order (wave 1)
  const int DH = ta.D * H, total = DH + ta.D, nseg = (total + 63) / 64;
  auto slab_partial = [&](int seg) {
    const int e = seg * 64 + c, zl = rq - 4;
    float v[(kMaxChunks + 11) / 12];
#pragma unroll
    for (int i = 0; i < (kMaxChunks + 11) / 12; ++i) {
      const int z = zl + 12 * i;
      v[i] = (e < total && z < ta.slab_chunks) ? (e < DH ? ta.w_slab[(int64_t)z * DH + e] : ta.b_slab[(int64_t)z * ta.D + (e - DH)]) : 0.f;
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < (kMaxChunks + 11) / 12; ++i) sum += v[i];
    red[zl][c] = sum;
  };
  auto slab_final = [&](int seg) {
    const int e = seg * 64 + c;
    if (e < total) {
      float o = 0.f;
#pragma unroll
      for (int k = 0; k < 12; ++k) o += red[k][c];
      if (e < DH) ta.g_w_out[e] = o;
      else ta.g_b_out[e - DH] = o;
    }
  };
  float pr[4], da[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = m0 + rq + 16 * j;
    const bool ok = c < H && r < B;
    const int64_t i = (int64_t)min(r, B - 1) * H + min(c, H - 1);          // (clamped address, no branch around the loads)
    const float v0 = b.pre[i], v1 = b.d[i];
    pr[j] = ok ? v0 : 0.f;
    da[j] = ok ? v1 : 0.f;
  }
  if (t < 256) {
    float s0 = 0.f, s1 = 0.f;
    if (c < H) {
      float v0[kMaxChunks / 4], v1[kMaxChunks / 4];
      const int64_t ps = a.pstride ? a.pstride : 2 * H;
#pragma unroll
      for (int i = 0; i < kMaxChunks / 4; ++i) {
        const int k = rq + 4 * i;
        const float* q = a.partial + (int64_t)k * ps;
        v0[i] = k < a.nchunks ? q[c] : 0.f;
        v1[i] = k < a.nchunks ? q[H + c] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < kMaxChunks / 4; ++i) { s0 += v0[i]; s1 += v1[i]; }
    }
    sh[0][rq][c] = s0;
    sh[1][rq][c] = s1;
  } else if ((int)blockIdx.x < nseg) {
    slab_partial(blockIdx.x);
  }
  __syncthreads();
  if (t < 64 && c < H) {
    const float t0 = (sh[0][0][c] + sh[0][1][c]) + (sh[0][2][c] + sh[0][3][c]);
    const float t1 = (sh[1][0][c] + sh[1][1][c]) + (sh[1][2][c] + sh[1][3][c]);
    S[0][c] = t0;
    S[1][c] = t1;
    if (blockIdx.x == 0) { a.out0[c] = t0 * ta.out_scale; a.out1[c] = t1 * ta.out_scale; }
  } else if (rq == 1 && (int)blockIdx.x < nseg) {
    slab_final(blockIdx.x);
  }
  __syncthreads();
  if (c < H) {
    const float mean = b.mean[c], rstd = b.rstd[c], g = b.g[c], S1 = S[0][c], S2 = S[1][c];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = m0 + rq + 16 * j;
      if (r < B) {
        const float xh = (fmaxf(pr[j], 0.f) - mean) * rstd;
        const float dv = g * rstd * (da[j] - S1 * b.invB - xh * (S2 * b.invB));
        b.d[(int64_t)r * H + c] = pr[j] > 0.f ? dv : 0.f;
      }
    }
  }
  // fewer row blocks than slab segments (B < 64 * nseg): the remaining segments in further rounds
  for (int seg = blockIdx.x + gridDim.x; seg < nseg; seg += gridDim.x) {
    if (t >= 256) slab_partial(seg);
    __syncthreads();
    if (rq == 1) slab_final(seg);
    __syncthreads();
  }
}

// (C') wide backward tail (last hidden width <= kWideH, output <= kWideD): L2-normalise backward, the data gradient
// d_act = d_y . W_out on MFMA (dropout scale applied) and the per-chunk BN column sums S1 / S2 taken from the accumulators, in
// one launch -- l2norm_bwd_kernel + the data-gradient GEMM + colsum_partial_kernel.  The weight gradient d_y^T . act stays the
// split-K GEMM (per-chunk slabs of [D x H] would be 17 MB per tower here).  At most 128 chunks per tower exist, so a workgroup
// has its CU to itself: 8 waves with up to 256 registers each.  Wave w owns columns 32 w .. 32 w + 31 of every 64-row block
// (two 32 x 32 accumulator tiles); its W_out operand fragments are loaded straight from global memory -- a lane's 8 k-values are
// 8 rows of the row-major [D, H] matrix, each load coalesced over the 32 columns -- converted once and kept in registers for
// the whole chunk, so W_out is read once per workgroup and never staged.  Only d_y goes through LDS.  grid (nchunks, towers)
constexpr int kWbLd = kWideD + 8;                                      // bf16 per LDS row of the d_y tile (K = D)
constexpr int kWideBwdThreads = 512;
static_assert(kWideH == 32 * (kWideBwdThreads / 64), "one 32-column strip per wave");

__global__ __launch_bounds__(kWideBwdThreads) void tail_bwd_wide_kernel(Batch<TailBwdArgs> batch, bool drop, float p, uint64_t seed0,
                                                                       const uint64_t* __restrict__ seed_dev) {
  const TailBwdArgs& f = batch.a[blockIdx.y];
  const ColArgs& a = f.col;
  const int H = a.H, D = f.D;
  if ((int)blockIdx.x >= a.nchunks) return;
  __shared__ __attribute__((aligned(16))) __bf16 dyA[64 * kWbLd];      // d_y rows of the current block (k = d)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int Dp = (D + 15) & ~15;                                        // K zero-padded to whole MFMA steps
  const int r0 = blockIdx.x * a.rows_per_chunk, r1 = min(a.B, r0 + a.rows_per_chunk);
  const uint64_t seed = drop ? seed_of(seed0, seed_dev) : 0;
  const int li = lane & 31, lh = lane >> 5;
  const int n = wave * 32 + li;
  const bool col_ok = n < H, wave_on = wave * 32 < H;
  const int ncl = min(n, H - 1);
  tl_bf16x8 bw[kWideD / 16];                                            // B operand: element (k = d, n = h) = W_out[d][h]
  if (wave_on) {
    float wq[kWideD / 16][8];
#pragma unroll
    for (int ks = 0; ks < kWideD / 16; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = 16 * ks + 8 * lh + e;
        wq[ks][e] = f.w_out[(int64_t)min(d, D - 1) * H + ncl];          // (clamped address, masked below: no branch around a load)
      }
#pragma unroll
    for (int ks = 0; ks < kWideD / 16; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = 16 * ks + 8 * lh + e;
        bw[ks][e] = (__bf16)((d < D && col_ok) ? wq[ks][e] : 0.f);
      }
  }
  const float mean = a.mean[ncl], rstd = a.rstd[ncl];
  float s0 = 0.f, s1 = 0.f;
  for (int b0 = r0; b0 < r1; b0 += 64) {
    // 1. d_y of rows b0 .. b0 + 63 (l2norm_bwd_kernel's arithmetic): wave w rows 8 w + j, lane columns lane, lane + 64
    float yv[8][2], e[8][2], de[8][2], pr[2][16];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int r = b0 + wave * 8 + j, col = lane + 64 * u;
        const bool ok = r < r1 && col < D;
        const int64_t i = (int64_t)min(r, r1 - 1) * D + min(col, D - 1);
        const float v0 = f.y[i], v1 = f.emb[i], v2 = f.d_emb[i];
        yv[j][u] = ok ? v0 : 0.f;
        e[j][u] = ok ? v1 : 0.f;
        de[j][u] = ok ? v2 : 0.f;
      }
    if (wave_on) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {                  // pre at this lane's accumulator positions
          const int row = b0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          pr[rt][r] = a.pre[(int64_t)min(row, r1 - 1) * H + ncl];       // (rows / columns past the end never reach a sum: da = 0)
        }
    }
    float ss[8], dot[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ss[j] = fmaf(yv[j][1], yv[j][1], mul_rn(yv[j][0], yv[j][0]));
      dot[j] = fmaf(e[j][1], de[j][1], mul_rn(e[j][0], de[j][0]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        ss[j] += __shfl_xor(ss[j], o);
        dot[j] += __shfl_xor(dot[j], o);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = wave * 8 + j, r = b0 + row;
      const float nrm = sqrtf(ss[j]);
      const float den = fmaxf(nrm, kNormEps);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int col = lane + 64 * u;
        float out = nrm > kNormEps ? (de[j][u] - e[j][u] * dot[j]) / den : de[j][u] / den;
        if (r < r1 && col < D) f.d_y[(int64_t)r * D + col] = out;
        else out = 0.f;
        if (col < Dp) dyA[row * kWbLd + col] = (__bf16)out;
      }
    }
    __syncthreads();
    // 2. d_act tiles on MFMA; dropout scale, store, column sums straight from the accumulators (rows in accumulator order)
    if (wave_on) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        tl_f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < kWideD / 16; ++ks)
          if (16 * ks < Dp) {
            const tl_bf16x8 a1 = *reinterpret_cast<const tl_bf16x8*>(dyA + (rt * 32 + li) * kWbLd + 16 * ks + 8 * lh);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw[ks], acc, 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = b0 + rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const bool ok = col_ok && row < r1;
          const int64_t i = (int64_t)row * H + n;
          const float da = ok ? acc[r] * dropout_scale(drop, p, seed, a.salt + (uint64_t)i) : 0.f;
          if (ok) f.d_act[i] = da;
          s0 += da;
          s1 += da * ((fmaxf(pr[rt][r], 0.f) - mean) * rstd);
        }
      }
    }
    __syncthreads();                                     // dyA is rewritten by the next block
  }
  s0 += __shfl_xor(s0, 32);
  s1 += __shfl_xor(s1, 32);
  if (lh == 0 && col_ok) {
    float* q = a.partial + (int64_t)blockIdx.x * 2 * H;
    q[n] = s0;
    q[H + n] = s1;
  }
}

// (D') wide backward apply: S1 / S2 finish (every workgroup, colsum_finish_kernel's order) + BN backward apply in place on the
// already dropout-scaled d_act, 64 rows per workgroup: thread (c, rq) = column c < 256, rows rq + 4 j.  grid (cdiv(B, 64), towers)
__global__ __launch_bounds__(kTailThreads) void tail_bwd_apply_wide_kernel(Batch<TailApplyArgs> batch) {
  const TailApplyArgs& ta = batch.a[blockIdx.y];
  const ColArgs& a = ta.col;
  const BnBwdArgs& b = ta.bn;
  const int H = a.H, B = a.B;
  const int t = threadIdx.x, c = t & 255, rq = t >> 8;
  const int m0 = blockIdx.x * 64;
  if (m0 >= B) return;
  __shared__ float sh[2][4][kWideH];
  float pr[16], da[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int64_t i = (int64_t)min(m0 + rq + 4 * j, B - 1) * H + min(c, H - 1);      // (clamped: stores are guarded below)
    pr[j] = b.pre[i];
    da[j] = b.d[i];
  }
  float s0 = 0.f, s1 = 0.f;
  {
    const int64_t ps = a.pstride ? a.pstride : 2 * H;
    const int cc = min(c, H - 1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {              // chunks rq, rq + 4, ... in order, 32 loads in flight at a time
      float v0[kMaxChunks / 8], v1[kMaxChunks / 8];
#pragma unroll
      for (int i = 0; i < kMaxChunks / 8; ++i) {
        const int k = rq + 4 * (i + half * (kMaxChunks / 8));
        const float* q = a.partial + (int64_t)min(k, a.nchunks - 1) * ps;
        const float x0 = q[cc], x1 = q[H + cc];
        v0[i] = k < a.nchunks ? x0 : 0.f;
        v1[i] = k < a.nchunks ? x1 : 0.f;
      }
#pragma unroll
      for (int i = 0; i < kMaxChunks / 8; ++i) { s0 += v0[i]; s1 += v1[i]; }
    }
  }
  sh[0][rq][c] = s0;
  sh[1][rq][c] = s1;
  __syncthreads();
  if (c < H) {
    const float S1 = (sh[0][0][c] + sh[0][1][c]) + (sh[0][2][c] + sh[0][3][c]);
    const float S2 = (sh[1][0][c] + sh[1][1][c]) + (sh[1][2][c] + sh[1][3][c]);
    if (blockIdx.x == 0 && rq == 0) { a.out0[c] = S1 * ta.out_scale; a.out1[c] = S2 * ta.out_scale; }
    const float mean = b.mean[c], rstd = b.rstd[c], g = b.g[c];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int r = m0 + rq + 4 * j;
      if (r < B) {
        const float xh = (fmaxf(pr[j], 0.f) - mean) * rstd;
        const float dv = g * rstd * (da[j] - S1 * b.invB - xh * (S2 * b.invB));
        b.d[(int64_t)r * H + c] = pr[j] > 0.f ? dv : 0.f;
      }
    }
  }
}

// SyncBN phase 1: this rank's chunk statistics merged (bn_stats_finish_kernel's order) into ONE (n, mean, M2) triple per
// column, and its column sums S1 / S2 into one pair -- what the caller exchanges between the ranks.  grid (1, towers)
struct LocalArgs { const float* partial; int nchunks, H; float* out; };

__global__ __launch_bounds__(kThreads) void tail_local_stats_kernel(Batch<LocalArgs> batch) {
  const LocalArgs& a = batch.a[blockIdx.y];
  const int H = a.H;
  if ((int)blockIdx.x * 64 >= H) return;
  __shared__ Wf sh[4][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, jl = threadIdx.x >> 6;
  Wf o{0.f, 0.f, 0.f};
  if (c < H) {
    Wf v[kMaxChunks / 4];
#pragma unroll
    for (int i = 0; i < kMaxChunks / 4; ++i) {
      const int k = jl + 4 * i;
      const float* q = a.partial + (int64_t)k * 3 * H;
      v[i] = k < a.nchunks ? Wf{q[c], q[H + c], q[2 * H + c]} : Wf{0.f, 0.f, 0.f};
    }
    o = wf_lane_merge(v);
  }
  sh[jl][cl] = o;
  __syncthreads();
  if (jl == 0 && c < H) {
    o = wf_combine(wf_combine(sh[0][cl], sh[1][cl]), wf_combine(sh[2][cl], sh[3][cl]));
    a.out[c] = o.n; a.out[H + c] = o.mean; a.out[2 * H + c] = o.m2;
  }
}

__global__ __launch_bounds__(kThreads) void tail_local_colsum_kernel(Batch<LocalArgs> batch) {
  const LocalArgs& a = batch.a[blockIdx.y];
  const int H = a.H;
  if ((int)blockIdx.x * 64 >= H) return;
  __shared__ float sh[2][4][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, jl = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (c < H) {
    float v0[kMaxChunks / 4], v1[kMaxChunks / 4];
#pragma unroll
    for (int i = 0; i < kMaxChunks / 4; ++i) {
      const int k = jl + 4 * i;
      const float* q = a.partial + (int64_t)k * 2 * H;
      v0[i] = k < a.nchunks ? q[c] : 0.f;
      v1[i] = k < a.nchunks ? q[H + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < kMaxChunks / 4; ++i) { s0 += v0[i]; s1 += v1[i]; }
  }
  sh[0][jl][cl] = s0;
  sh[1][jl][cl] = s1;
  __syncthreads();
  if (jl == 0 && c < H) {
    a.out[c] = (sh[0][0][cl] + sh[0][1][cl]) + (sh[0][2][cl] + sh[0][3][cl]);
    a.out[H + c] = (sh[1][0][cl] + sh[1][1][cl]) + (sh[1][2][cl] + sh[1][3][cl]);
  }
}

// ---- host side ---------------------------------------------------------------------------------
inline int ew_grid(const tt_ctx* ctx, int64_t n, int towers) {
  const int64_t cap = (int64_t)ctx->num_cus * 8 / (towers > 0 ? towers : 1);
  int64_t b = tt_cdiv(n, kThreads);
  return (int)(b < 1 ? 1 : (b < cap ? b : cap));
}

inline int chunks_for(int64_t B, int H) {
  int64_t n = 512 / tt_cdiv(H, 64);
  if (n > kMaxChunks) n = kMaxChunks;
  const int64_t mx = tt_cdiv(B, 64);
  if (n > mx) n = mx;
  return (int)(n < 1 ? 1 : n);
}

struct WsLayout {
  char* gemm;
  size_t gemm_bytes;
  char* tn[TT_MAX_HIDDEN + 2];        // own slab region per weight-gradient GEMM (0 = projection, 1 + i = block i, last = output):
  size_t tn_bytes[TT_MAX_HIDDEN + 2]; // their reductions are deferred into one launch at the end of the backward pass
  float* col;
  size_t col_bytes;
  size_t total;
};

inline int in_width(const tt_tower_params* p, int i) { return i == 0 ? p->h0 + p->kcat_e : p->hidden[i - 1]; }
inline int last_width(const tt_tower_params* p) { return p->n_hidden == 0 ? p->h0 + p->kcat_e : p->hidden[p->n_hidden - 1]; }

// the fused narrow tail (tail_*_kernel) applies to a training pass with bf16 GEMM operands whose last hidden block and
// output are at most 64 wide
inline bool tail_shape_ok(const tt_tower_params* p) {
  return p->n_hidden >= 1 && p->hidden[p->n_hidden - 1] <= 64 && p->d_out <= 64;
}
inline bool tail_fusable(int n, const tt_tower_params* const* P, int train) {
  if (!train || P[0]->compute_dtype != TT_BF16) return false;
  for (int t = 0; t < n; ++t)
    if (!tail_shape_ok(P[t]) || (P[t]->flags & TT_TOWER_UNFUSED_TAIL)) return false;
  return true;
}
// the one-launch front (tower_front_kernel): fused tail + ONE hidden block + bf16 tower input, projection width a multiple of
// 32 and at most 128, 16-byte-aligned rows everywhere
inline bool front_fusable(int n, const tt_tower_params* const* P, const tt_tower_acts* const* A) {
  for (int t = 0; t < n; ++t) {
    const tt_tower_params* p = P[t];
    const int kx = p->h0 + p->kcat_e;
    if (p->n_hidden != 1 || p->x_dtype != TT_BF16 || (p->flags & TT_TOWER_UNFUSED_FRONT)) return false;
    if (p->h0 % 32 != 0 || p->h0 > 128 || p->din % 64 != 0 || p->din > 256 || kx % 64 != 0) return false;
    if (!tt_aligned(A[t]->dense, 16) || !tt_aligned(A[t]->x, 16) || !tt_aligned(p->w_proj, 16) || !tt_aligned(p->w[0], 16)) return false;
  }
  return true;
}
// the wide forward tail (tail_fwd_wide_kernel): training, bf16 operands, last hidden width <= 256, output <= 128 -- shapes the
// narrow fused tail does not take
inline bool wide_tail_ok(int n, const tt_tower_params* const* P, int train) {
  if (!train || P[0]->compute_dtype != TT_BF16) return false;
  for (int t = 0; t < n; ++t) {
    const tt_tower_params* p = P[t];
    if (p->n_hidden < 1 || p->hidden[p->n_hidden - 1] > kWideH || p->d_out > kWideD || (p->flags & TT_TOWER_UNFUSED_TAIL)) return false;
  }
  return true;
}
inline size_t tail_slab_bytes(const tt_tower_params* p) {   // per-chunk output-layer gradient slabs of tail_bwd_kernel
  return sizeof(float) * (size_t)kMaxChunks * ((size_t)p->d_out * p->hidden[p->n_hidden - 1] + (size_t)p->d_out) + 256;
}

inline WsLayout ws_layout(const tt_tower_params* p, int64_t B, char* base) {
  size_t g = tt_gemm_tn_workspace_bytes(p->h0, p->din, B);
  int hmax = p->d_out > p->h0 ? p->d_out : p->h0;
  for (int i = 0; i < p->n_hidden; ++i) {
    const size_t b = tt_gemm_tn_workspace_bytes(p->hidden[i], in_width(p, i), B);
    g = b > g ? b : g;
    hmax = p->hidden[i] > hmax ? p->hidden[i] : hmax;
  }
  const size_t b = tt_gemm_tn_workspace_bytes(p->d_out, last_width(p), B);
  g = b > g ? b : g;
  {                                                    // split-K slabs of the forward GEMMs (largest output)
    int nmax = p->h0 > p->d_out ? p->h0 : p->d_out;
    for (int i = 0; i < p->n_hidden; ++i) nmax = p->hidden[i] > nmax ? p->hidden[i] : nmax;
    const size_t f = tt_gemm_nt_workspace_bytes(B, nmax, 0);
    g = f > g ? f : g;
  }
  WsLayout w;
  w.gemm_bytes = (g + 255) & ~size_t(255);
  w.col_bytes = sizeof(float) * 3 * (size_t)hmax * (size_t)kMaxChunks + 256;
  w.gemm = base;
  w.col = reinterpret_cast<float*>(base ? base + w.gemm_bytes : nullptr);
  size_t o = w.gemm_bytes + ((w.col_bytes + 255) & ~size_t(255));
  for (int l = 0; l < p->n_hidden + 2; ++l) {
    size_t nb = l == 0 ? tt_gemm_tn_workspace_bytes(p->h0, p->din, B)
              : l == p->n_hidden + 1 ? tt_gemm_tn_workspace_bytes(p->d_out, last_width(p), B)
                                     : tt_gemm_tn_workspace_bytes(p->hidden[l - 1], in_width(p, l - 1), B);
    if (l == p->n_hidden + 1 && tail_shape_ok(p) && tail_slab_bytes(p) > nb) nb = tail_slab_bytes(p);
    if (l == 0 && p->n_hidden >= 1) {                   // the one-launch first-block backward keeps its G slabs here
      const size_t gb = tt_gemm_back_g_workspace_bytes(p->hidden[0], p->h0, p->din, B);
      nb = gb > nb ? gb : nb;
    }
    w.tn[l] = base ? base + o : nullptr;
    w.tn_bytes[l] = (nb + 255) & ~size_t(255);
    o += w.tn_bytes[l];
  }
  w.total = o;
  return w;
}

int check_params(const tt_tower_params* p, const char* who) {
  TT_CHECK_ARG(p, "%s: params NULL", who);
  TT_CHECK_ARG(p->n_hidden >= 0 && p->n_hidden <= TT_MAX_HIDDEN, "%s: n_hidden=%d not in [0,%d]", who, p->n_hidden, TT_MAX_HIDDEN);
  TT_CHECK_ARG(p->din >= 1 && p->h0 >= 1 && p->kcat_e >= 0 && p->d_out >= 1, "%s: bad dims", who);
  TT_CHECK_ARG(p->w_proj && p->b_proj && p->w_out && p->b_out, "%s: NULL projection/output weights", who);
  for (int i = 0; i < p->n_hidden; ++i) {
    TT_CHECK_ARG(p->hidden[i] >= 1, "%s: hidden[%d] < 1", who, i);
    TT_CHECK_ARG(p->w[i] && p->b[i] && p->bn_w[i] && p->bn_b[i] && p->bn_rm[i] && p->bn_rv[i], "%s: NULL block %d params", who, i);
  }
  return TT_OK;
}

int check_batch(int32_t n, const tt_tower_params* const* P, int64_t B, float dropout_p, void* const* ws, const size_t* wsb,
                const char* who) {
  TT_CHECK_ARG(n >= 1 && n <= TT_MAX_SIDES && P && ws && wsb, "%s: need 1..%d towers", who, TT_MAX_SIDES);
  TT_CHECK_ARG(B >= 0 && B < ((int64_t)1 << 24), "%s: B=%lld out of range", who, (long long)B);
  TT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "%s: dropout_p=%f not in [0,1)", who, dropout_p);
  for (int t = 0; t < n; ++t) {
    if (int rc = check_params(P[t], who)) return rc;
    TT_CHECK_ARG(P[t]->n_hidden == P[0]->n_hidden, "%s: towers fused into one launch must have the same number of hidden blocks", who);
    TT_CHECK_ARG(P[t]->compute_dtype == P[0]->compute_dtype, "%s: towers fused into one launch must share compute_dtype", who);
    TT_CHECK_ARG((P[t]->x_dtype == TT_F32 && P[t]->dx_dtype == TT_F32) || P[t]->compute_dtype == TT_BF16,
                 "%s: bf16 x / d_x need compute_dtype TT_BF16", who);
    TT_CHECK_ARG((P[t]->x_dtype == TT_F32 || P[t]->x_dtype == TT_BF16) && (P[t]->dx_dtype == TT_F32 || P[t]->dx_dtype == TT_BF16),
                 "%s: bad x_dtype / dx_dtype", who);
    TT_CHECK_ARG(P[t]->sync_phase == P[0]->sync_phase && P[t]->sync_phase >= 0 && P[t]->sync_phase <= 2, "%s: bad / mixed sync_phase", who);
    TT_CHECK_ARG(P[t]->sync_phase == 0 || (P[t]->sync_ranks >= 1 && P[t]->sync_ranks <= kMaxChunks), "%s: sync_ranks=%d not in [1, %d]", who,
                 P[t]->sync_ranks, kMaxChunks);
    if (B > 0 && (!ws[t] || wsb[t] < tt_tower_workspace_bytes(P[t], B))) {
      tt_set_error("%s: workspace of tower %d: %zu < required %zu", who, t, wsb[t], tt_tower_workspace_bytes(P[t], B));
      return TT_ERR_WORKSPACE;
    }
  }
  return TT_OK;
}

}  // namespace

extern "C" {

size_t tt_tower_workspace_bytes(const tt_tower_params* p, int64_t B) {
  if (!p || p->n_hidden < 0 || p->n_hidden > TT_MAX_HIDDEN) return 0;
  return ws_layout(p, B, nullptr).total;
}

int tt_towers_mlp_fwd(tt_ctx* ctx, int32_t n, const tt_tower_params* const* P, const tt_tower_acts* const* A, int64_t B,
                      int32_t train, float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* const* workspaces,
                      const size_t* workspace_bytes, tt_stream stream) {
  TT_CHECK_ARG(ctx && A, "tt_towers_mlp_fwd: NULL argument");
  if (int rc = check_batch(n, P, B, dropout_p, workspaces, workspace_bytes, "tt_towers_mlp_fwd")) return rc;
  if (B == 0) return TT_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  WsLayout ws[TT_MAX_SIDES];
  const float* in[TT_MAX_SIDES];
  int in_w[TT_MAX_SIDES];
  GemmNT nt[TT_MAX_SIDES];
  for (int t = 0; t < n; ++t) {
    TT_CHECK_ARG(A[t] && A[t]->dense && A[t]->x && A[t]->y && A[t]->emb, "tt_towers_mlp_fwd: NULL activation buffers");
    ws[t] = ws_layout(P[t], B, reinterpret_cast<char*>(workspaces[t]));
    const int wx = P[t]->h0 + P[t]->kcat_e;
    // x[:, 0:h0] = dense W_proj^T + b_proj        (base_tower.py:133)
    nt[t] = GemmNT{A[t]->dense, P[t]->din, P[t]->w_proj, P[t]->din, P[t]->b_proj, reinterpret_cast<float*>(A[t]->x), wx, B, P[t]->h0,
                   P[t]->din, false, 1.f};
    nt[t].c_bf16 = P[t]->x_dtype == TT_BF16;
    in[t] = reinterpret_cast<const float*>(A[t]->x);
    in_w[t] = wx;
  }
  const bool drop = train && dropout_p > 0.f;
  const int nh = P[0]->n_hidden;
  const bool fused = tail_fusable(n, P, train);
  const int phase = P[0]->sync_phase;
  if (phase != 0) {
    if (nh != 1 || !train) {
      tt_set_error("tt_towers_mlp_fwd: sync_phase %d needs a training pass over towers with exactly one hidden block (the pass is cut at its BatchNorm)", phase);
      return TT_ERR_UNSUPPORTED;
    }
    for (int t = 0; t < n; ++t)
      TT_CHECK_ARG(phase == 1 ? A[t]->bn_sync_local != nullptr : A[t]->bn_sync_all != nullptr, "tt_towers_mlp_fwd: NULL SyncBN buffer");
  }
  for (int t = 0; t < n; ++t) { nt[t].bf16 = P[0]->compute_dtype == TT_BF16; nt[t].workspace = ws[t].gemm; nt[t].workspace_bytes = ws[t].gemm_bytes; }
  const bool front = fused && front_fusable(n, P, A);    // projection + block Linear + chunk statistics in one launch
  if (phase != 2 && !front)
    if (int rc = tt_gemm_nt_batched(st, nt, n)) return rc;
  NtDeferred nd[TT_MAX_SIDES];
  for (int i = 0; i < nh; ++i) {
    const bool tail = fused && i == nh - 1;
    Batch<BnStatArgs> bs{};
    Batch<BnApplyArgs> ba{};
    int hmax = 1, cmax = 1;
    int64_t tmax = 1;
    for (int t = 0; t < n; ++t) {
      const int H = P[t]->hidden[i];
      TT_CHECK_ARG(A[t]->pre[i] && A[t]->act[i] && A[t]->mean[i] && A[t]->rstd[i], "tt_towers_mlp_fwd: NULL buffers of block %d", i);
      nt[t] = GemmNT{in[t], in_w[t], P[t]->w[i], in_w[t], P[t]->b[i], A[t]->pre[i], H, B, H, in_w[t], false, 1.f};
      nt[t].a_bf16 = i == 0 && P[t]->x_dtype == TT_BF16;
      const int nchunks = chunks_for(B, H);
      bs.a[t] = BnStatArgs{A[t]->pre[i], (int)B, H, (int)tt_cdiv(B, nchunks), nchunks, ws[t].col, A[t]->mean[i], A[t]->rstd[i],
                           P[t]->bn_rm[i], P[t]->bn_rv[i], P[t]->bn_nbt[i]};
      // dropout element index = GLOBAL row * H + column: the rank's first row enters through the salt
      ba.a[t] = BnApplyArgs{A[t]->pre[i], B * H, H, A[t]->mean[i], A[t]->rstd[i], P[t]->bn_w[i], P[t]->bn_b[i],
                            (((uint64_t)(i + 1) << 40) ^ ((uint64_t)t << 52)) + (uint64_t)P[t]->rng_row_offset * (uint64_t)H, A[t]->act[i]};
      hmax = H > hmax ? H : hmax;
      cmax = nchunks > cmax ? nchunks : cmax;
      tmax = B * H > tmax ? B * H : tmax;
    }
    for (int t = 0; t < n; ++t) {
      nt[t].bf16 = P[0]->compute_dtype == TT_BF16; nt[t].workspace = ws[t].gemm; nt[t].workspace_bytes = ws[t].gemm_bytes;
      nt[t].defer = tail ? &nd[t] : nullptr;
    }
    if (phase != 2 && !(front && tail))
      if (int rc = tt_gemm_nt_batched(st, nt, n)) return rc;
    if (tail) {
      // slabs (+ bias) -> pre with the chunk statistics in the same pass, then everything up to the unit rows in one kernel
      if (phase != 2) {
        if (front) {
          Batch<FrontArgs> fb{};
          bool w16 = true;                                // every tower brings bf16 shadows of both weights (16-byte aligned)
          for (int t = 0; t < n; ++t) {
            fb.a[t] = FrontArgs{A[t]->dense, P[t]->din, P[t]->din, P[t]->w_proj, P[t]->b_proj, P[t]->h0,
                                reinterpret_cast<uint16_t*>(A[t]->x), in_w[t], in_w[t], P[t]->w[i], P[t]->b[i], A[t]->pre[i], bs.a[t],
                                reinterpret_cast<const uint16_t*>(P[t]->w_proj_bf16), reinterpret_cast<const uint16_t*>(P[t]->w_bf16[i])};
            w16 = w16 && P[t]->w_proj_bf16 && P[t]->w_bf16[i] && tt_aligned(P[t]->w_proj_bf16, 16) && tt_aligned(P[t]->w_bf16[i], 16);
          }
          if (w16) {
            TT_LDS_ONCE(kFrontLds, tower_front_kernel<true>);
            tower_front_kernel<true><<<dim3((unsigned)cmax, (unsigned)n), kFrontThreads, kFrontLds, st>>>(fb);
          } else {
            TT_LDS_ONCE(kFrontLds, tower_front_kernel<false>);
            tower_front_kernel<false><<<dim3((unsigned)cmax, (unsigned)n), kFrontThreads, kFrontLds, st>>>(fb);
          }
        } else if (nd[0].splits > 0) {
          Batch<HeadArgs> hb{};
          for (int t = 0; t < n; ++t) hb.a[t] = HeadArgs{nd[t].slabs, nd[t].slab_stride, nd[t].splits, P[t]->b[i], A[t]->pre[i], bs.a[t]};
          tail_head_kernel<<<dim3((unsigned)cmax, (unsigned)n), kTailThreads, 0, st>>>(hb);
        } else {
          bn_stats_partial_kernel<<<dim3(1, (unsigned)cmax, (unsigned)n), kThreads, 0, st>>>(bs);
        }
        TT_LAUNCH_CHECK();
      }
      if (phase == 1) {                                 // SyncBN: hand this rank's statistics to the caller and stop
        Batch<LocalArgs> la{};
        for (int t = 0; t < n; ++t) la.a[t] = LocalArgs{bs.a[t].partial, bs.a[t].nchunks, bs.a[t].H, A[t]->bn_sync_local};
        tail_local_stats_kernel<<<dim3((unsigned)tt_cdiv(hmax, 64), (unsigned)n), kThreads, 0, st>>>(la);
        TT_LAUNCH_CHECK();
        return TT_OK;
      }
      if (phase == 2)                                   // ... and continue with every rank's triple as one "chunk"
        for (int t = 0; t < n; ++t) {
          bs.a[t].partial = const_cast<float*>(A[t]->bn_sync_all); bs.a[t].nchunks = P[t]->sync_ranks; bs.a[t].pstride = A[t]->bn_sync_stride;
        }
      Batch<TailFwdArgs> tf{};
      for (int t = 0; t < n; ++t)
        tf.a[t] = TailFwdArgs{bs.a[t], P[t]->bn_w[i], P[t]->bn_b[i], ba.a[t].salt, A[t]->act[i], P[t]->w_out, P[t]->b_out, P[t]->d_out,
                              A[t]->y, A[t]->emb, nullptr, nullptr, 0, 1.f};
      for (int t = 0; t < n; ++t)
        if (A[t]->emb_packed) {
          const int Dp = P[t]->d_out <= 32 ? 32 : 64;
          __bf16* base = reinterpret_cast<__bf16*>(A[t]->emb_packed);
          tf.a[t].pk_rows = base;
          tf.a[t].pk_frag = base + tt_cdiv(B, 64) * 64 * Dp;
          tf.a[t].Dp = Dp;
          tf.a[t].pk_scale = A[t]->emb_pack_scale == 0.f ? 1.f : A[t]->emb_pack_scale;
        }
      const int cr_wg = ctx->riders->c_wg;               // a queued plan compaction rides in one extra grid row
      const int64_t gx = tt_cdiv(B, 64) > cr_wg ? tt_cdiv(B, 64) : cr_wg;
      tail_fwd_kernel<<<dim3((unsigned)gx, (unsigned)(n + (cr_wg > 0 ? 1 : 0))), kTailThreads, 0, st>>>(tf, drop, dropout_p, seed, seed_dev,
                                                                                                       ctx->riders->c, cr_wg);
      ctx->riders->c_wg = 0;
      TT_LAUNCH_CHECK();
      return TT_OK;
    }
    if (train) {
      if (phase != 2) {
        bn_stats_partial_kernel<<<dim3((unsigned)tt_cdiv(hmax, 64), (unsigned)cmax, (unsigned)n), kThreads, 0, st>>>(bs);
        TT_LAUNCH_CHECK();
      }
      if (phase == 1) {                                 // SyncBN on the separate kernels (any width): hand this rank's statistics out and stop
        Batch<LocalArgs> la{};
        for (int t = 0; t < n; ++t) la.a[t] = LocalArgs{bs.a[t].partial, bs.a[t].nchunks, bs.a[t].H, A[t]->bn_sync_local};
        tail_local_stats_kernel<<<dim3((unsigned)tt_cdiv(hmax, 64), (unsigned)n), kThreads, 0, st>>>(la);
        TT_LAUNCH_CHECK();
        return TT_OK;
      }
      if (phase == 2)                                   // ... and continue with every rank's triple as one "chunk"
        for (int t = 0; t < n; ++t) {
          bs.a[t].partial = const_cast<float*>(A[t]->bn_sync_all); bs.a[t].nchunks = P[t]->sync_ranks; bs.a[t].pstride = A[t]->bn_sync_stride;
        }
      if (i == nh - 1 && !fused && wide_tail_ok(n, P, train)) {
        // statistics finish + BN apply + dropout + output Linear + L2 normalise + the score kernels' operand images in one launch
        TT_LDS_ONCE(kWideLds, tail_fwd_wide_kernel);
        Batch<TailFwdArgs> tf{};
        for (int t = 0; t < n; ++t) {
          tf.a[t] = TailFwdArgs{bs.a[t], P[t]->bn_w[i], P[t]->bn_b[i], ba.a[t].salt, A[t]->act[i], P[t]->w_out, P[t]->b_out, P[t]->d_out,
                                A[t]->y, A[t]->emb, nullptr, nullptr, 0, 1.f};
          if (A[t]->emb_packed) {
            const int Dp = P[t]->d_out <= 32 ? 32 : (P[t]->d_out <= 64 ? 64 : 128);
            __bf16* base = reinterpret_cast<__bf16*>(A[t]->emb_packed);
            tf.a[t].pk_rows = base;
            tf.a[t].pk_frag = base + tt_cdiv(B, 64) * 64 * Dp;
            tf.a[t].Dp = Dp;
            tf.a[t].pk_scale = A[t]->emb_pack_scale == 0.f ? 1.f : A[t]->emb_pack_scale;
          }
        }
        tail_fwd_wide_kernel<<<dim3((unsigned)tt_cdiv(B, 64), (unsigned)n), kTailThreads, kWideLds, st>>>(tf, drop, dropout_p, seed, seed_dev);
        TT_LAUNCH_CHECK();
        return TT_OK;
      }
      bn_stats_finish_kernel<<<dim3((unsigned)tt_cdiv(hmax, 64), (unsigned)n), kThreads, 0, st>>>(bs);
      TT_LAUNCH_CHECK();
    } else {
      bn_eval_prepare_kernel<<<dim3((unsigned)tt_cdiv(hmax, 256), (unsigned)n), 256, 0, st>>>(bs);
      TT_LAUNCH_CHECK();
    }
    bn_apply_kernel<<<dim3((unsigned)ew_grid(ctx, tmax, n), (unsigned)n), kThreads, 0, st>>>(ba, drop, dropout_p, seed, seed_dev);
    TT_LAUNCH_CHECK();
    for (int t = 0; t < n; ++t) {
      in[t] = A[t]->act[i];
      in_w[t] = P[t]->hidden[i];
    }
  }
  Batch<NormArgs> na{};
  int dmax = 1;
  for (int t = 0; t < n; ++t) {
    nt[t] = GemmNT{in[t], in_w[t], P[t]->w_out, in_w[t], P[t]->b_out, A[t]->y, P[t]->d_out, B, P[t]->d_out, in_w[t], false, 1.f};
    nt[t].a_bf16 = nh == 0 && P[t]->x_dtype == TT_BF16;
    na.a[t] = NormArgs{A[t]->y, nullptr, nullptr, (int)B, P[t]->d_out, A[t]->emb};
    dmax = P[t]->d_out > dmax ? P[t]->d_out : dmax;
  }
  for (int t = 0; t < n; ++t) { nt[t].bf16 = P[0]->compute_dtype == TT_BF16; nt[t].workspace = ws[t].gemm; nt[t].workspace_bytes = ws[t].gemm_bytes; }
  if (int rc = tt_gemm_nt_batched(st, nt, n)) return rc;
  if (!launch_l2norm_fast<false>(st, na, n, B)) l2norm_fwd_kernel<<<dim3((unsigned)tt_cdiv(B, 4), (unsigned)n), kThreads, 0, st>>>(na);
  TT_LAUNCH_CHECK();
  (void)dmax;
  for (int t = 0; t < n; ++t)                            // emb_packed is honoured on every path: here by the pack kernel
    if (A[t]->emb_packed)
      if (int rc = tt_score_pack_bf16(ctx, A[t]->emb, B, P[t]->d_out, A[t]->emb_pack_scale, A[t]->emb_packed, stream)) return rc;
  return TT_OK;
}

int tt_towers_mlp_bwd(tt_ctx* ctx, int32_t n, const tt_tower_params* const* P, const tt_tower_acts* const* A,
                      const float* const* d_emb, const tt_tower_grads* const* G, int64_t B, int32_t train, float dropout_p,
                      uint64_t seed, const uint64_t* seed_dev, void* const* workspaces, const size_t* workspace_bytes,
                      tt_stream stream) {
  TT_CHECK_ARG(ctx && A && G && d_emb, "tt_towers_mlp_bwd: NULL argument");
  if (int rc = check_batch(n, P, B, dropout_p, workspaces, workspace_bytes, "tt_towers_mlp_bwd")) return rc;
  TT_CHECK_ARG(B >= 1, "tt_towers_mlp_bwd: B < 1");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool drop = train && dropout_p > 0.f;
  const int nh = P[0]->n_hidden;
  WsLayout ws[TT_MAX_SIDES];
  float* dcur[TT_MAX_SIDES];
  Batch<NormArgs> na{};
  GemmTN tn[TT_MAX_SIDES];
  GemmNN nn[TT_MAX_SIDES];
  for (int t = 0; t < n; ++t) {
    const tt_tower_grads* g = G[t];
    TT_CHECK_ARG(A[t] && g && d_emb[t] && g->w_proj && g->b_proj && g->w_out && g->b_out && g->d_x && g->d_y,
                 "tt_towers_mlp_bwd: NULL gradient buffers");
    ws[t] = ws_layout(P[t], B, reinterpret_cast<char*>(workspaces[t]));
    na.a[t] = NormArgs{A[t]->y, A[t]->emb, d_emb[t], (int)B, P[t]->d_out, g->d_y};
  }
  const bool fused = tail_fusable(n, P, train);
  const bool wide = !fused && wide_tail_ok(n, P, train);  // tail_bwd_wide_kernel + tail_bwd_apply_wide_kernel
  const int phase = P[0]->sync_phase;
  if (phase != 0) {
    if (nh != 1 || !train) {
      tt_set_error("tt_towers_mlp_bwd: sync_phase %d needs a training pass over towers with exactly one hidden block (the pass is cut at its BatchNorm)", phase);
      return TT_ERR_UNSUPPORTED;
    }
    for (int t = 0; t < n; ++t)
      TT_CHECK_ARG(G[t] && (phase == 1 ? G[t]->s_sync_local != nullptr : G[t]->s_sync_all != nullptr), "tt_towers_mlp_bwd: NULL SyncBN buffer");
  }
  if (!fused && !wide && phase != 2) {
    if (!launch_l2norm_fast<true>(st, na, n, B)) l2norm_bwd_kernel<<<dim3((unsigned)tt_cdiv(B, 4), (unsigned)n), kThreads, 0, st>>>(na);
    TT_LAUNCH_CHECK();
  }
  for (int t = 0; t < n; ++t) {
    const tt_tower_grads* g = G[t];
    const float* in_last = nh == 0 ? reinterpret_cast<const float*>(A[t]->x) : A[t]->act[nh - 1];
    const int lw = last_width(P[t]);
    tn[t] = GemmTN{g->d_y, P[t]->d_out, in_last, lw, g->w_out, lw, P[t]->d_out, lw, B, ws[t].tn[nh + 1], ws[t].tn_bytes[nh + 1], g->b_out};
    dcur[t] = nh == 0 ? reinterpret_cast<float*>(g->d_x) : g->scratch[nh - 1];
    TT_CHECK_ARG(dcur[t], "tt_towers_mlp_bwd: NULL scratch buffer");
    nn[t] = GemmNN{g->d_y, P[t]->d_out, P[t]->w_out, lw, dcur[t], lw, B, lw, P[t]->d_out};
    tn[t].b_bf16 = nh == 0 && P[t]->x_dtype == TT_BF16;
    nn[t].c_bf16 = nh == 0 && P[t]->dx_dtype == TT_BF16;
  }
  for (int t = 0; t < n; ++t) tn[t].bf16 = nn[t].bf16 = P[0]->compute_dtype == TT_BF16;
  // the slab reductions of all weight-gradient GEMMs run as ONE launch at the end (nothing in this pass reads them)
  struct PendingGuard {
    TnPending* p = tt_gemm_tn_pending_create();
    ~PendingGuard() { tt_gemm_tn_pending_destroy(p); }
  } pend;
  if (fused) {
    // L2-normalise backward, both output-layer GEMMs and the BN column sums of the last block in one kernel; the BN
    // backward apply (after the batch-wide S1 / S2) in a second
    const int i = nh - 1;
    Batch<TailBwdArgs> tb{};
    Batch<TailApplyArgs> tp{};
    int cmax = 1;
    for (int t = 0; t < n; ++t) {
      const tt_tower_grads* g = G[t];
      const int H = P[t]->hidden[i], D = P[t]->d_out;
      TT_CHECK_ARG(g->w[i] && g->b[i] && g->bn_w[i] && g->bn_b[i], "tt_towers_mlp_bwd: NULL gradient buffers of block %d", i);
      const int nchunks = chunks_for(B, H);
      const uint64_t salt = (((uint64_t)(i + 1) << 40) ^ ((uint64_t)t << 52)) + (uint64_t)P[t]->rng_row_offset * (uint64_t)H;
      const ColArgs col{nullptr, 0, A[t]->pre[i], A[t]->mean[i], A[t]->rstd[i], salt, (int)B, H, (int)tt_cdiv(B, nchunks), nchunks,
                        ws[t].col, g->bn_b[i], g->bn_w[i]};
      float* w_slab = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws[t].tn[nh + 1]) + 255) & ~uintptr_t(255));
      float* b_slab = w_slab + (size_t)nchunks * D * H;
      tb.a[t] = TailBwdArgs{A[t]->y, A[t]->emb, d_emb[t], g->d_y, D, P[t]->w_out, A[t]->act[i], dcur[t], col, w_slab, b_slab};
      ColArgs colD = col;                               // SyncBN: S1 / S2 of every rank, one "chunk" each, over ranks * B rows
      const int ranks = phase == 2 ? P[t]->sync_ranks : 1;
      if (phase == 2) { colD.partial = const_cast<float*>(g->s_sync_all); colD.nchunks = ranks; colD.pstride = g->s_sync_stride; }
      tp.a[t] = TailApplyArgs{colD, BnBwdArgs{dcur[t], A[t]->pre[i], B * H, H, 1.f / ((float)B * (float)ranks), A[t]->mean[i], A[t]->rstd[i],
                                              P[t]->bn_w[i], g->bn_b[i], g->bn_w[i], salt},
                              w_slab, b_slab, D, g->w_out, g->b_out, nchunks, 1.f / (float)ranks};
      cmax = nchunks > cmax ? nchunks : cmax;
    }
    if (phase != 2) {
      const int fr_on = ctx->riders->f_wg > 0 ? 1 : 0;    // a queued loss reduction rides in one extra grid row
      tail_bwd_kernel<<<dim3((unsigned)cmax, (unsigned)(n + fr_on)), kTailThreads, 0, st>>>(tb, drop, dropout_p, seed, seed_dev, ctx->riders->f,
                                                                                           fr_on);
      ctx->riders->f_wg = 0;
      TT_LAUNCH_CHECK();
    }
    if (phase == 1) {                                   // SyncBN: hand this rank's column sums to the caller and stop
      Batch<LocalArgs> la{};
      for (int t = 0; t < n; ++t) la.a[t] = LocalArgs{tb.a[t].col.partial, tb.a[t].col.nchunks, tb.a[t].col.H, G[t]->s_sync_local};
      tail_local_colsum_kernel<<<dim3(1, (unsigned)n), kThreads, 0, st>>>(la);                  // (fused tail: H <= 64)
      TT_LAUNCH_CHECK();
      return TT_OK;
    }
    tail_bwd_apply_kernel<<<dim3((unsigned)tt_cdiv(B, 64), (unsigned)n), kTailThreads, 0, st>>>(tp);
    TT_LAUNCH_CHECK();
  } else if (wide) {
    // L2-normalise backward, the output layer's data-gradient GEMM and the BN column sums of the last block in one kernel (the
    // weight-gradient GEMM stays the split-K one); column-sum finish + BN backward apply in a second, below
    const int i = nh - 1;
    if (phase != 2) {
      Batch<TailBwdArgs> tb{};
      int cmax = 1;
      for (int t = 0; t < n; ++t) {
        const int H = P[t]->hidden[i];
        const int nchunks = chunks_for(B, H);
        const uint64_t salt = (((uint64_t)(i + 1) << 40) ^ ((uint64_t)t << 52)) + (uint64_t)P[t]->rng_row_offset * (uint64_t)H;
        const ColArgs col{nullptr, 0, A[t]->pre[i], A[t]->mean[i], A[t]->rstd[i], salt, (int)B, H, (int)tt_cdiv(B, nchunks), nchunks,
                          ws[t].col, G[t]->bn_b[i], G[t]->bn_w[i]};
        tb.a[t] = TailBwdArgs{A[t]->y, A[t]->emb, d_emb[t], G[t]->d_y, P[t]->d_out, P[t]->w_out, nullptr, dcur[t], col, nullptr, nullptr};
        cmax = nchunks > cmax ? nchunks : cmax;
      }
      tail_bwd_wide_kernel<<<dim3((unsigned)cmax, (unsigned)n), kWideBwdThreads, 0, st>>>(tb, drop, dropout_p, seed, seed_dev);
      TT_LAUNCH_CHECK();
      if (int rc = tt_gemm_tn_batched(st, tn, n, pend.p)) return rc;
    }
  } else if (phase != 2) {
    if (int rc = tt_gemm_tn_batched(st, tn, n, pend.p)) return rc;
    if (int rc = tt_gemm_nn_batched(st, nn, n)) return rc;
  }
  for (int i = nh - 1; i >= 0; --i) {
    Batch<ColArgs> cb{};
    Batch<BnBwdArgs> bb{};
    int hmax = 1, cmax = 1;
    int64_t tmax = 1;
    for (int t = 0; t < n; ++t) {
      const tt_tower_grads* g = G[t];
      const int H = P[t]->hidden[i];
      TT_CHECK_ARG(g->w[i] && g->b[i] && g->bn_w[i] && g->bn_b[i], "tt_towers_mlp_bwd: NULL gradient buffers of block %d", i);
      const int nchunks = chunks_for(B, H);
      const uint64_t salt = (((uint64_t)(i + 1) << 40) ^ ((uint64_t)t << 52)) + (uint64_t)P[t]->rng_row_offset * (uint64_t)H;
      // S1 = sum da -> bn bias grad ; S2 = sum da*xhat -> bn weight grad
      cb.a[t] = ColArgs{dcur[t], H, A[t]->pre[i], A[t]->mean[i], A[t]->rstd[i], salt, (int)B, H, (int)tt_cdiv(B, nchunks), nchunks,
                        ws[t].col, g->bn_b[i], g->bn_w[i]};
      bb.a[t] = BnBwdArgs{dcur[t], A[t]->pre[i], B * H, H, 1.f / (float)B, A[t]->mean[i], A[t]->rstd[i], P[t]->bn_w[i], g->bn_b[i],
                          g->bn_w[i], salt};
      hmax = H > hmax ? H : hmax;
      cmax = nchunks > cmax ? nchunks : cmax;
      tmax = B * H > tmax ? B * H : tmax;
    }
    if (wide && i == nh - 1 && phase != 1) {
      // the column sums of tail_bwd_wide_kernel (phase 2: of every rank, one "chunk" each) finished and applied in one launch
      Batch<TailApplyArgs> tp{};
      for (int t = 0; t < n; ++t) {
        ColArgs colD = cb.a[t];
        const int ranks = phase == 2 ? P[t]->sync_ranks : 1;
        if (phase == 2) { colD.partial = const_cast<float*>(G[t]->s_sync_all); colD.nchunks = ranks; colD.pstride = G[t]->s_sync_stride; }
        BnBwdArgs bn = bb.a[t];
        bn.invB = 1.f / ((float)B * (float)ranks);
        tp.a[t] = TailApplyArgs{colD, bn, nullptr, nullptr, P[t]->d_out, nullptr, nullptr, 0, 1.f / (float)ranks};
      }
      tail_bwd_apply_wide_kernel<<<dim3((unsigned)tt_cdiv(B, 64), (unsigned)n), kTailThreads, 0, st>>>(tp);
      TT_LAUNCH_CHECK();
    } else if (!(fused && i == nh - 1)) {
      if (phase != 2 && !(wide && i == nh - 1)) {
        colsum_partial_kernel<<<dim3((unsigned)tt_cdiv(hmax, 64), (unsigned)cmax, (unsigned)n), kThreads, 0, st>>>(cb, drop, dropout_p, seed, seed_dev);
        TT_LAUNCH_CHECK();
      }
      if (phase == 1) {                                 // SyncBN on the separate kernels: hand this rank's column sums out and stop
        if (int rc = tt_gemm_tn_flush(st, pend.p)) return rc;     // (the output layer's weight-gradient slabs)
        Batch<LocalArgs> la{};
        for (int t = 0; t < n; ++t) la.a[t] = LocalArgs{cb.a[t].partial, cb.a[t].nchunks, cb.a[t].H, G[t]->s_sync_local};
        tail_local_colsum_kernel<<<dim3((unsigned)tt_cdiv(hmax, 64), (unsigned)n), kThreads, 0, st>>>(la);
        TT_LAUNCH_CHECK();
        return TT_OK;
      }
      if (phase == 2)                                   // S1 / S2 of every rank, one "chunk" each, over ranks * B rows
        for (int t = 0; t < n; ++t) {
          const int ranks = P[t]->sync_ranks, H = P[t]->hidden[i];
          float* raw = ws[t].col + 2 * (size_t)hmax * kMaxChunks;         // (the third third of the column workspace is free here)
          cb.a[t].partial = const_cast<float*>(G[t]->s_sync_all); cb.a[t].nchunks = ranks; cb.a[t].pstride = G[t]->s_sync_stride;
          cb.a[t].out_scale = 1.f / (float)ranks; cb.a[t].sums_raw = raw;
          bb.a[t].invB = 1.f / ((float)B * (float)ranks); bb.a[t].S1 = raw; bb.a[t].S2 = raw + H;
        }
      colsum_finish_kernel<<<dim3((unsigned)tt_cdiv(hmax, 64), (unsigned)n), kThreads, 0, st>>>(cb);
      TT_LAUNCH_CHECK();
      bn_bwd_apply_kernel<<<dim3((unsigned)ew_grid(ctx, tmax, n), (unsigned)n), kThreads, 0, st>>>(bb, train != 0, drop, dropout_p, seed, seed_dev);
      TT_LAUNCH_CHECK();
    }
    for (int t = 0; t < n; ++t) {
      const tt_tower_grads* g = G[t];
      const int H = P[t]->hidden[i];
      const int iw = in_width(P[t], i);
      const float* in_i = i == 0 ? reinterpret_cast<const float*>(A[t]->x) : A[t]->act[i - 1];
      tn[t] = GemmTN{dcur[t], H, in_i, iw, g->w[i], iw, H, iw, B, ws[t].tn[1 + i], ws[t].tn_bytes[1 + i], g->b[i]};
      tn[t].b_bf16 = i == 0 && P[t]->x_dtype == TT_BF16;
      float* dnext = i == 0 ? reinterpret_cast<float*>(g->d_x) : g->scratch[i - 1];
      TT_CHECK_ARG(dnext, "tt_towers_mlp_bwd: NULL scratch buffer");
      nn[t] = GemmNN{dcur[t], H, P[t]->w[i], iw, dnext, iw, B, iw, H};
      nn[t].c_bf16 = i == 0 && P[t]->dx_dtype == TT_BF16;
      dcur[t] = dnext;
    }
    for (int t = 0; t < n; ++t) tn[t].bf16 = nn[t].bf16 = P[0]->compute_dtype == TT_BF16;
    if (i == 0 && P[0]->compute_dtype == TT_BF16) {
      // first block + projection: three independent products of d_pre in ONE launch (tt_gemm.h: GemmBack); the projection's
      // gradients come out of the slab-reduction launch, d_x[:, 0:h0] is not materialised
      GemmBack gb[TT_MAX_SIDES];
      bool ok = true;
      for (int t = 0; t < n; ++t) {
        const tt_tower_grads* g = G[t];
        const int wx = P[t]->h0 + P[t]->kcat_e;
        ok = ok && !(P[t]->flags & TT_TOWER_UNFUSED_BACK);
        gb[t] = GemmBack{tn[t].A, P[t]->hidden[0], A[t]->x, wx, wx, P[t]->x_dtype == TT_BF16, A[t]->dense, P[t]->din, P[t]->din,
                         P[t]->w[0], P[t]->h0, reinterpret_cast<float*>(g->d_x), wx, P[t]->dx_dtype == TT_BF16, g->w[0], g->b[0],
                         g->w_proj, g->b_proj, ws[t].tn[1], ws[t].tn_bytes[1], ws[t].tn[0], ws[t].tn_bytes[0], B};
        gb[t].w16 = P[t]->w_bf16[0];
      }
      if (ok && tt_gemm_back_supported(gb, n)) {
        if (int rc = tt_gemm_back_batched(st, gb, n, pend.p)) return rc;
        if (ctx->defer_slab_reduce) return tt_gemm_tn_defer(ctx, pend.p);     // rides in tt_embed_grad_bwd's launch
        return tt_gemm_tn_flush(st, pend.p);
      }
    }
    if (int rc = tt_gemm_tn_batched(st, tn, n, pend.p)) return rc;
    if (int rc = tt_gemm_nn_batched(st, nn, n)) return rc;
  }
  // dense projection: d_x[:, 0:h0]
  for (int t = 0; t < n; ++t) {
    const tt_tower_grads* g = G[t];
    const int wx = P[t]->h0 + P[t]->kcat_e;
    tn[t] = GemmTN{reinterpret_cast<const float*>(g->d_x), wx, A[t]->dense, P[t]->din, g->w_proj, P[t]->din, P[t]->h0, P[t]->din, B,
                   ws[t].tn[0], ws[t].tn_bytes[0], g->b_proj};
    tn[t].a_bf16 = P[t]->dx_dtype == TT_BF16;
  }
  for (int t = 0; t < n; ++t) tn[t].bf16 = P[0]->compute_dtype == TT_BF16;
  if (int rc = tt_gemm_tn_batched(st, tn, n, pend.p)) return rc;
  // (not deferred like the one-launch first-block backward above: at these widths the slabs are HBM traffic, not launch
  //  latency -- measured 0.508 ms per step either way at scripts/train.py's [512, 256] -> 128)
  return tt_gemm_tn_flush(st, pend.p);
}

int tt_tower_mlp_fwd(tt_ctx* ctx, const tt_tower_params* p, const tt_tower_acts* a, int64_t B, int32_t train, float dropout_p,
                     uint64_t seed, const uint64_t* seed_dev, void* workspace, size_t workspace_bytes, tt_stream stream) {
  return tt_towers_mlp_fwd(ctx, 1, &p, &a, B, train, dropout_p, seed, seed_dev, &workspace, &workspace_bytes, stream);
}

int tt_tower_mlp_bwd(tt_ctx* ctx, const tt_tower_params* p, const tt_tower_acts* a, const float* d_emb, const tt_tower_grads* g,
                     int64_t B, int32_t train, float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* workspace,
                     size_t workspace_bytes, tt_stream stream) {
  return tt_towers_mlp_bwd(ctx, 1, &p, &a, &d_emb, &g, B, train, dropout_p, seed, seed_dev, &workspace, &workspace_bytes, stream);
}

int tt_linear_fwd(tt_ctx* ctx, const float* X, int64_t ldx, const float* W, const float* bias, float* Y, int64_t ldy, int64_t M,
                  int32_t N, int32_t K, int32_t relu, tt_stream stream) {
  TT_CHECK_ARG(ctx && (M == 0 || (X && W && Y)), "tt_linear_fwd: NULL argument");
  TT_CHECK_ARG(M >= 0 && N >= 1 && K >= 1 && ldx >= K && ldy >= N, "tt_linear_fwd: bad shape");
  TT_CHECK_ARG(M < ((int64_t)1 << 31), "tt_linear_fwd: M too large");
  return tt_gemm_nt(reinterpret_cast<hipStream_t>(stream), X, ldx, W, K, bias, Y, ldy, M, N, K, relu != 0);
}

}  // extern "C"

#ifdef TT_TAIL_STAMPS
extern "C" int tt_debug_tail_stamps(int which, unsigned long long* host_out /* [512 * 8] */) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tail_stamps), sizeof(unsigned long long) * 512 * 8,
                             sizeof(unsigned long long) * 512 * 8 * (size_t)which, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
