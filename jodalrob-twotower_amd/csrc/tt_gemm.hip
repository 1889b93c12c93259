// f32 MFMA GEMM tiles for the tower MLP (tall-skinny: M = batch, N,K = layer widths).
// 64x64 output tile per 256-thread workgroup, 4 waves x one 32x32 accumulator
// (v_mfma_f32_32x32x2_f32), K staged 16 deep through LDS in k-major order so that both MFMA
// operand reads are conflict-free ds_read_b32 (lanes 0-31 -> k, lanes 32-63 -> k+1).
#include "tt_gemm.h"

#include <stdlib.h>

namespace {

constexpr int BM = 64, BN = 64, BK = 16, PAD = 4, THREADS = 256;
using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GemmArgs {
  const float* A; int64_t lda; const float* B; int64_t ldb; float* C; int64_t ldc; int64_t slab_stride;
  int M, N, K, kchunk, splits;
  const float* bias; int relu; float alpha; float* colsum_slab;
  int a_bf16 = 0, b_bf16 = 0, c_bf16 = 0;   // bf16 kernel only: element type of A / B / C in memory
  // gemm_back_kernel role 1 only: the tile G_z [64 rows of H, 64 columns] leaves as proj_w[those rows, 0:proj_h0]^T . G_z
  const float* proj_w = nullptr; int proj_ldw = 0, proj_h0 = 0;
};
struct GemmBatch { GemmArgs a[TT_MAX_SIDES]; };

// MODE 0: operand is K-contiguous  : elem(x, k) = P[x*ld + k]   (x = m for A, n for B)
// MODE 1: operand is X-contiguous  : elem(x, k) = P[k*ld + x]
template <int MODE, bool VEC>
struct TileLoader {
  float r[4];
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int x0, int X, int k0, int kend, int t) {
    if (MODE == 0) {
      const int row = t >> 2, kq = (t & 3) * 4;
      const int x = x0 + row, k = k0 + kq;
      if (VEC && x < X && k + 3 < kend) {
        const float4 v = *reinterpret_cast<const float4*>(P + (int64_t)x * ld + k);
        r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (x < X && k + j < kend) ? P[(int64_t)x * ld + k + j] : 0.f;
      }
    } else {
      const int k = k0 + (t >> 4), xq = (t & 15) * 4;
      const int x = x0 + xq;
      if (VEC && k < kend && x + 3 < X) {
        const float4 v = *reinterpret_cast<const float4*>(P + (int64_t)k * ld + x);
        r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (k < kend && x + j < X) ? P[(int64_t)k * ld + x + j] : 0.f;
      }
    }
  }
  __device__ __forceinline__ void store(float (*S)[BM + PAD], int t) const {
    if (MODE == 0) {
      const int row = t >> 2, kq = (t & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) S[kq + j][row] = r[j];
    } else {
      const int k = t >> 4, xq = (t & 15) * 4;
      *reinterpret_cast<float4*>(&S[k][xq]) = make_float4(r[0], r[1], r[2], r[3]);
    }
  }
};

// COLSUM (TN mode only): additionally accumulate sum_k A[k][m] (the bias gradient sum_batch dY) of this
// workgroup's k-range into colsum_slab[split][m] -- the A tiles are in registers anyway.
// Horizontal batching: up to TT_MAX_SIDES independent problems (the towers) share ONE launch;
// blockIdx.z = problem * zsplits + split, workgroups outside a problem's own grid exit at once.
template <int MODE_A, int MODE_B, bool VEC, bool COLSUM = false>
__global__ __launch_bounds__(THREADS) void gemm_kernel(GemmBatch batch, int zsplits) {
  const GemmArgs& g = batch.a[blockIdx.z / zsplits];
  const int split = blockIdx.z % zsplits;
  const int M = g.M, N = g.N, K = g.K, kchunk = g.kchunk;
  if ((int)blockIdx.x * BM >= M || (int)blockIdx.y * BN >= N || split >= g.splits) return;
  const float* __restrict__ A = g.A;
  const float* __restrict__ B = g.B;
  float* __restrict__ C = g.C;
  const int64_t lda = g.lda, ldb = g.ldb, ldc = g.ldc, slab_stride = g.slab_stride;
  const float* __restrict__ bias = g.bias;
  const int relu = g.relu;
  const float alpha = g.alpha;
  float* __restrict__ colsum_slab = g.colsum_slab;
  __shared__ __attribute__((aligned(16))) float As[BK][BM + PAD];
  __shared__ __attribute__((aligned(16))) float Bs[BK][BN + PAD];
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kbeg = split * kchunk;
  const int kend = min(K, kbeg + kchunk);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  TileLoader<MODE_A, VEC> la;
  TileLoader<MODE_B, VEC> lb;
  if (kbeg < kend) {
    la.load(A, lda, m0, M, kbeg, kend, t);
    lb.load(B, ldb, n0, N, kbeg, kend, t);
  }
  const int li = lane & 31, lh = lane >> 5;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    la.store(As, t);
    lb.store(Bs, t);
    if (COLSUM) {
#pragma unroll
      for (int j = 0; j < 4; ++j) cs[j] += la.r[j];
    }
    __syncthreads();
    if (k0 + BK < kend) {
      la.load(A, lda, m0, M, k0 + BK, kend, t);
      lb.load(B, ldb, n0, N, k0 + BK, kend, t);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[kk + lh][wr * 32 + li];
      const float b = Bs[kk + lh][wc * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  if (COLSUM && colsum_slab && blockIdx.y == 0) {          // 16 k-lanes per column quad -> fixed-order sum through LDS
    float(*red)[BM + PAD] = As;                            // (As is free: the loop ended with a barrier)
    *reinterpret_cast<float4*>(&red[t >> 4][(t & 15) * 4]) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    __syncthreads();
    if (t < BM && m0 + t < M) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += red[q][t];
      colsum_slab[(int64_t)split * M + m0 + t] = sum;
    }
  }
  float* Cz = C + (int64_t)split * slab_stride;
  const int n = n0 + wc * 32 + li;
  const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m < M && n < N) {
      float v = acc[r] * alpha + bv;
      if (relu) v = fmaxf(v, 0.f);
      Cz[(int64_t)m * ldc + n] = v;
    }
  }
}

// (SlabArgs / SlabBatch: tt_gemm.h)

// ---- bf16-operand variant: f32 tensors in memory, rounded to bf16 (RNE) on the way into LDS, f32 accumulate on
// v_mfma_f32_32x32x16_bf16 (16x fewer matrix cycles than the exact-f32 form).  LDS tiles are [x][k] with k
// contiguous (80-B rows: conflict-free ds_read_b128 fragments of 8 consecutive k).
using bf16x8g = __attribute__((ext_vector_type(8))) __bf16;
// (BK = 64 per step measured 1-2 us SLOWER per GEMM than 32: these loops are HBM-bandwidth-bound -- 3.5-3.7 TB/s
// counting the split-K slabs -- not latency-bound.)
constexpr int BK16 = 32, LDS16 = 40, KR = BK16 / 4;

template <int MODE>
struct TileLoader16 {
  float r[KR];
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int x0, int X, int k0, int kend, int t, bool vec) {
    if (MODE == 0) {                       // K-contiguous: KR consecutive k of one row per thread
      const int x = x0 + (t >> 2), k = k0 + (t & 3) * KR;
      if (vec && x < X && k + KR - 1 < kend) {
#pragma unroll
        for (int q = 0; q < KR / 4; ++q) {
          const float4 a = *reinterpret_cast<const float4*>(P + (int64_t)x * ld + k + 4 * q);
          r[4 * q] = a.x; r[4 * q + 1] = a.y; r[4 * q + 2] = a.z; r[4 * q + 3] = a.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < KR; ++j) r[j] = (x < X && k + j < kend) ? P[(int64_t)x * ld + k + j] : 0.f;
      }
    } else {                               // X-contiguous: one x, KR consecutive k (coalesced dword loads across x)
      const int x = x0 + (t & 63), k = k0 + (t >> 6) * KR;
#pragma unroll
      for (int j = 0; j < KR; ++j) r[j] = (x < X && k + j < kend) ? P[(int64_t)(k + j) * ld + x] : 0.f;
    }
  }
  // same tile from a bf16 tensor (held widened in r[]: the conversion back in store() is exact)
  __device__ __forceinline__ void load_bf16(const uint16_t* __restrict__ P, int64_t ld, int x0, int X, int k0, int kend, int t, bool vec,
                                            bool allow_tr) {
    if (MODE == 0) {
      const int x = x0 + (t >> 2), k = k0 + (t & 3) * KR;
      if (vec && x < X && k + KR - 1 < kend) {
#pragma unroll
        for (int q = 0; q < KR / 8; ++q) {
          const uint4 a = *reinterpret_cast<const uint4*>(P + (int64_t)x * ld + k + 8 * q);
          const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            r[8 * q + 2 * j] = __builtin_bit_cast(float, w[j] << 16);
            r[8 * q + 2 * j + 1] = __builtin_bit_cast(float, w[j] & 0xFFFF0000u);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < KR; ++j) r[j] = (x < X && k + j < kend) ? tt_bf2f(P[(int64_t)x * ld + k + j]) : 0.f;
      }
    } else if (vec && allow_tr && KR == 8) {
      // X-contiguous bf16 (the weight gradient reads x [batch, 1152] bf16 with k = batch row): ONE 16-byte load of 8
      // adjacent x at one k per thread, transposed on the way into LDS (8 ds_write_b16) -- the 2-byte-load form issued
      // 8 global loads per thread and ran at 24 us against 19 us when x was f32.  r[j] = element (x + j, k).
      const int x = x0 + 8 * (t & 7), k = k0 + (t >> 3);
      if (k < kend && x + 7 < X) {
        const uint4 a = *reinterpret_cast<const uint4*>(P + (int64_t)k * ld + x);
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          r[2 * j] = __builtin_bit_cast(float, w[j] << 16);
          r[2 * j + 1] = __builtin_bit_cast(float, w[j] & 0xFFFF0000u);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (k < kend && x + j < X) ? tt_bf2f(P[(int64_t)k * ld + x + j]) : 0.f;
      }
      transposed = true;
    } else {
      const int x = x0 + (t & 63), k = k0 + (t >> 6) * KR;
#pragma unroll
      for (int j = 0; j < KR; ++j) r[j] = (x < X && k + j < kend) ? tt_bf2f(P[(int64_t)(k + j) * ld + x]) : 0.f;
    }
  }
  __device__ __forceinline__ void load_any(const float* __restrict__ P, int is_bf16, int64_t ld, int x0, int X, int k0, int kend, int t,
                                           bool vec, bool allow_tr = false) {     // allow_tr: B operand only (COLSUM reads la.r per x)
    if (is_bf16) load_bf16(reinterpret_cast<const uint16_t*>(P), ld, x0, X, k0, kend, t, vec, allow_tr);
    else load(P, ld, x0, X, k0, kend, t, vec);
  }
  bool transposed = false;   // r[] holds 8 x at ONE k (16-byte load of an X-contiguous bf16 source)
  __device__ __forceinline__ void store(__bf16* __restrict__ S, int t) const {
    if (MODE == 1 && transposed) {
      const int row = 8 * (t & 7), k = t >> 3;
#pragma unroll
      for (int j = 0; j < 8; ++j) S[(row + j) * LDS16 + k] = (__bf16)r[j];
      return;
    }
    const int row = MODE == 0 ? (t >> 2) : (t & 63), kq = MODE == 0 ? (t & 3) * KR : (t >> 6) * KR;
#pragma unroll
    for (int q = 0; q < KR / 8; ++q) {
      bf16x8g v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)r[8 * q + j];
      *reinterpret_cast<bf16x8g*>(S + row * LDS16 + kq + 8 * q) = v;
    }
  }
};

template <int MODE_A, int MODE_B, bool COLSUM>
__global__ __launch_bounds__(THREADS) void gemm_bf16_kernel(GemmBatch batch, int zsplits, int vec) {
  const GemmArgs& g = batch.a[blockIdx.z / zsplits];
  const int split = blockIdx.z % zsplits;
  const int M = g.M, N = g.N, K = g.K, kchunk = g.kchunk;
  if ((int)blockIdx.x * BM >= M || (int)blockIdx.y * BN >= N || split >= g.splits) return;
  __shared__ __attribute__((aligned(16))) __bf16 As[BM * LDS16];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[BN * LDS16];
  __shared__ float red[4][64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kbeg = split * kchunk, kend = min(K, kbeg + kchunk);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  TileLoader16<MODE_A> la;
  TileLoader16<MODE_B> lb;
  float cs = 0.f;
  if (kbeg < kend) {
    la.load_any(g.A, g.a_bf16, g.lda, m0, M, kbeg, kend, t, vec);
    lb.load_any(g.B, g.b_bf16, g.ldb, n0, N, kbeg, kend, t, vec, true);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BK16) {
    la.store(As, t);
    lb.store(Bs, t);
    if (COLSUM) {                          // MODE_A == 1 there: this thread holds column (t & 63), 8 batch rows
#pragma unroll
      for (int j = 0; j < KR; ++j) cs += la.r[j];
    }
    __syncthreads();
    if (k0 + BK16 < kend) {
      la.load_any(g.A, g.a_bf16, g.lda, m0, M, k0 + BK16, kend, t, vec);
      lb.load_any(g.B, g.b_bf16, g.ldb, n0, N, k0 + BK16, kend, t, vec, true);
    }
#pragma unroll
    for (int s2 = 0; s2 < BK16 / 16; ++s2) {
      const bf16x8g a = *reinterpret_cast<const bf16x8g*>(As + (wr * 32 + li) * LDS16 + 16 * s2 + 8 * lh);
      const bf16x8g b = *reinterpret_cast<const bf16x8g*>(Bs + (wc * 32 + li) * LDS16 + 16 * s2 + 8 * lh);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  if (COLSUM && g.colsum_slab && blockIdx.y == 0) {
    red[t >> 6][t & 63] = cs;
    __syncthreads();
    if (t < BM && m0 + t < M) g.colsum_slab[(int64_t)split * M + m0 + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  }
  float* Cz = g.C + (int64_t)split * g.slab_stride;
  const int n = n0 + wc * 32 + li;
  const float bv = (g.bias && n < N) ? g.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m < M && n < N) {
      float v = acc[r] * g.alpha + bv;
      if (g.relu) v = fmaxf(v, 0.f);
      if (g.c_bf16) reinterpret_cast<uint16_t*>(Cz)[(int64_t)m * g.ldc + n] = tt_f2bf(v);
      else Cz[(int64_t)m * g.ldc + n] = v;
    }
  }
}

// ---- the same tile with the loads THREE k-steps ahead ------------------------------------------------------------------
// gemm_bf16_kernel keeps one k-step of loads in flight: a workgroup's time is (k-steps) x (one global-load round trip) --
// 10 dependent round trips for a weight-gradient workgroup at B = 8192 (18 us for 30 MB).  This form holds three k-steps of
// operand pieces in registers and double-buffers the LDS tiles (one barrier per step).  Every load is unconditional and in
// a fixed order (hipcc then waits with vmcnt(N), not vmcnt(0)): it takes the shapes that need no predicates -- M, N
// multiples of 64, every k-range a multiple of 32, 16-byte-aligned rows -- and element types as template parameters; the
// general kernel above takes the rest.  Same k order inside a workgroup, same split-K layout: results are bit-identical.
// LDS tiles: a K-contiguous operand (MODE 0) is staged [x][k] (LDS16-element rows), read as one 16-byte fragment per lane.  An
// X-contiguous operand (MODE 1: the batch-major matrices of the weight-gradient products, W in the data-gradient product) is
// staged AS LOADED, [k][x] with LDT1-element rows -- one 16-byte store per thread -- and the MFMA fragments come out of
// ds_read_b64_tr_b16 (a 16-lane group reads a 4 k x 16 x block, lane i receives column i): the first version transposed on the
// way in with eight 2-byte stores per thread, 8- to 16-way bank conflicts each, and loaded f32 operands one dword at a time.
constexpr int LDT1 = 72;
static_assert(BK16 * LDT1 <= BM * LDS16, "a [k][x] tile fits the stage buffer of an [x][k] one");
using s16x4g = __attribute__((ext_vector_type(4))) short;
using s16x8g = __attribute__((ext_vector_type(8))) short;

template <int MODE, bool BF16>
struct FastLoader16 {
  // raw pieces exactly as loaded (bf16: one 16-byte piece in q; f32: 8 values in r): every ALU op on them sits in store(),
  // behind the barrier of the previous step -- an unpack in load() is scheduled early and waits for the whole pipeline
  float r[BF16 ? 1 : KR];
  uint4 q;
  __device__ __forceinline__ void load(const float* __restrict__ Pf, int64_t ld, int x0, int k0, int t) {
    // MODE 0: row x0 + t / 4, 8 consecutive k ; MODE 1: row k0 + t / 8, 8 consecutive x -- 16 (bf16) or 32 (f32) contiguous bytes
    const int64_t at = MODE == 0 ? (int64_t)(x0 + (t >> 2)) * ld + k0 + (t & 3) * KR : (int64_t)(k0 + (t >> 3)) * ld + x0 + 8 * (t & 7);
    if (BF16) {
      q = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(Pf) + at);
    } else {
      const float4 a = *reinterpret_cast<const float4*>(Pf + at);
      const float4 b = *reinterpret_cast<const float4*>(Pf + at + 4);
      r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = a.w; r[4] = b.x; r[5] = b.y; r[6] = b.z; r[7] = b.w;
    }
  }
  // MODE 1, f32: this thread's 8 values are columns x0 + 8 (t & 7) + j of batch row k0 + t / 8
  __device__ __forceinline__ void colsum(float (&cs)[KR]) const {
#pragma unroll
    for (int j = 0; j < (BF16 ? 1 : KR); ++j) cs[j] += r[j];
  }
  __device__ __forceinline__ void store(__bf16* __restrict__ S, int t) {
    __bf16* at = MODE == 0 ? S + (t >> 2) * LDS16 + (t & 3) * KR : S + (t >> 3) * LDT1 + 8 * (t & 7);
    if (BF16) {
      asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z), "+v"(q.w));
      *reinterpret_cast<uint4*>(at) = q;
      return;
    }
#pragma unroll
    for (int j = 0; j < (BF16 ? 1 : KR); ++j) asm volatile("" : "+v"(r[j]));
    bf16x8g v;
#pragma unroll
    for (int j = 0; j < (BF16 ? 1 : KR); ++j) v[j] = (__bf16)r[j];
    *reinterpret_cast<bf16x8g*>(at) = v;
  }
};

// one MFMA operand fragment of k-step s2 (k = 16 s2 + 8 lh + 0..7) for the 32 rows / columns w32 of a staged tile
template <int MODE>
__device__ __forceinline__ bf16x8g fast_frag(const __bf16* __restrict__ S, int w32, int s2, int lane) {
  const int li = lane & 31, lh = lane >> 5;
  if (MODE == 0) return *reinterpret_cast<const bf16x8g*>(S + (w32 * 32 + li) * LDS16 + 16 * s2 + 8 * lh);
  // lane 4 q + p of a 16-lane group supplies the address of block row q, columns 4 p .. 4 p + 3; lane i receives column i
  const int g16 = lane & 15, cg = (lane >> 4) & 1;
  const __bf16* at = S + (16 * s2 + 8 * lh + (g16 >> 2)) * LDT1 + w32 * 32 + 16 * cg + 4 * (g16 & 3);
  const s16x4g lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4g*)(at));
  const s16x4g hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4g*)(at + 4 * LDT1));
  const s16x8g v{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
  return __builtin_bit_cast(bf16x8g, v);
}

struct FastSmem {
  __bf16 As[2][BM * LDS16];
  __bf16 Bs[2][BN * LDS16];
  float red[4][64];
};
struct BackSmem : FastSmem {
  __bf16 lo[64 * 72];                      // low halves of the G_z tile (PROJ epilogue)
};

// one 64 x 64 tile (bx, by) of problem g, k-range `split`
template <int MODE_A, int MODE_B, bool COLSUM, bool A_BF16, bool B_BF16, bool PROJ = false>
__device__ __forceinline__ void gemm_fast_tile(const GemmArgs& g, int split, int bx, int by, FastSmem& sm) {
  static_assert(!(MODE_A == 1 && A_BF16), "COLSUM reads the A pieces per column");
  const int M = g.M, K = g.K, kchunk = g.kchunk;
  __bf16 (&As)[2][BM * LDS16] = sm.As;
  __bf16 (&Bs)[2][BN * LDS16] = sm.Bs;
  float (&red)[4][64] = sm.red;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
  const int m0 = bx * BM, n0 = by * BN;
  const int kbeg = split * kchunk, kend = min(K, kbeg + kchunk);
  const int nsteps = kend > kbeg ? (kend - kbeg) / BK16 : 0;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float cs[KR];
#pragma unroll
  for (int j = 0; j < KR; ++j) cs[j] = 0.f;
  if (nsteps > 0) {
    FastLoader16<MODE_A, A_BF16> la[3];
    FastLoader16<MODE_B, B_BF16> lb[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {                            // a step past the end re-reads the last one (discarded)
      const int k = kbeg + min(u, nsteps - 1) * BK16;
      la[u].load(g.A, g.lda, m0, k, t);
      lb[u].load(g.B, g.ldb, n0, k, t);
    }
    __builtin_amdgcn_sched_barrier(0);
    auto step = [&](int i, FastLoader16<MODE_A, A_BF16>& a, FastLoader16<MODE_B, B_BF16>& b) {
      __bf16* A = As[i & 1];
      __bf16* Bm = Bs[i & 1];
      a.store(A, t);
      b.store(Bm, t);
      if (COLSUM) a.colsum(cs);            // MODE_A == 1 there: columns 8 (t & 7) + j of one batch row per step
      __syncthreads();
      const int k = kbeg + min(i + 3, nsteps - 1) * BK16;
      a.load(g.A, g.lda, m0, k, t);
      b.load(g.B, g.ldb, n0, k, t);
#pragma unroll
      for (int s2 = 0; s2 < BK16 / 16; ++s2) {
        const bf16x8g av = fast_frag<MODE_A>(A, wr, s2, lane);
        const bf16x8g bv = fast_frag<MODE_B>(Bm, wc, s2, lane);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
      }
    };
    const int full = nsteps - nsteps % 3;
    for (int i0 = 0; i0 < full; i0 += 3) {                  // whole groups of three: no exit inside the unrolled body
      step(i0, la[0], lb[0]);
      step(i0 + 1, la[1], lb[1]);
      step(i0 + 2, la[2], lb[2]);
    }
    if (nsteps - full >= 1) step(full, la[0], lb[0]);
    if (nsteps - full >= 2) step(full + 1, la[1], lb[1]);
  }
  if (COLSUM && g.colsum_slab && by == 0) {
    // thread (k lane t / 8, column group t & 7) holds 8 column sums over its batch rows: the 32 k lanes are added in order
    float* red32 = reinterpret_cast<float*>(&As[0][0]);    // [32][64] f32 = 8 KB over the two A stage buffers
    static_assert(sizeof(float) * 32 * 64 <= sizeof(__bf16) * 2 * BM * LDS16, "column-sum staging fits the A stage buffers");
    __syncthreads();                        // every wave is done with the stage buffers
#pragma unroll
    for (int j = 0; j < KR; ++j) red32[(t >> 3) * 64 + 8 * (t & 7) + j] = cs[j];
    __syncthreads();
    if (t < BM) {
      float sum = 0.f;
#pragma unroll 8
      for (int kl = 0; kl < 32; ++kl) sum += red32[kl * 64 + t];
      g.colsum_slab[(int64_t)split * M + m0 + t] = sum;
    }
  }
  (void)red;
  float* Cz = g.C + (int64_t)split * g.slab_stride;
  const int n = n0 + wc * 32 + li;
  if (PROJ) {
    Cz = g.C + ((int64_t)split * (M / BM) + bx) * g.slab_stride;      // one slab per (split, 64-row block of h)
    // this split's G_z tile [64 h of block bx, 64 columns] -> slab of P = W[block bx, 0:h0]^T . G_z  [h0, 64 columns]: summed over
    // the splits AND the H / 64 row blocks by the slab reduction, that is the projection's weight gradient.  Both factors go through LDS as
    // bf16 ([column][h] and [i][h], 72-element rows), 64 rows of i at a time
    static_assert(LDT1 == 72 && 64 * LDT1 <= 2 * BM * LDS16, "a 64-deep [k][x] tile fits the two stage buffers of an operand");
    __bf16* Gt = &As[0][0];                 // [h][column]: 64 x 72 bf16 = 9216 B <= the two A stage buffers
    __bf16* Gl = static_cast<BackSmem&>(sm).lo;
    __bf16* Wt = &Bs[0][0];                 // [h][i]
    __syncthreads();                        // every wave is done with the stage buffers
    // G_z enters the second product as hi + lo (two bf16 terms: 16 significant bits): a split's partial sums can be much
    // larger than the total they cancel to, and rounding them to 8 bits cost 2e-3 of the finished gradient.  Both factors of
    // that product have the contraction index h as their ROW index in memory (W[h][i], G_z[h][column]): they are staged as
    // they lie -- k-major, contiguous stores -- and the fragments come out of transposing LDS reads (fast_frag<1>); the first
    // version transposed both on the way in with 2-byte stores (W: 16 per thread and 64-column block, 4-way conflicts and worse)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int at = (wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDT1 + wc * 32 + li;
      const __bf16 hi = (__bf16)acc[r];
      Gt[at] = hi;
      Gl[at] = (__bf16)(acc[r] - (float)hi);
    }
    for (int i0 = 0; i0 < g.proj_h0; i0 += 64) {
      {                                     // W[h][i0 .. i0 + 63] -> Wt[h][i]: thread = (h = t >> 2, 16 columns), two 16-byte stores
        const int h = t >> 2, iq = (t & 3) * 16;
        const float* src = g.proj_w + (int64_t)(m0 + h) * g.proj_ldw + i0 + iq;
        float4 w[4];
#pragma unroll
        for (int v4 = 0; v4 < 4; ++v4) w[v4] = *reinterpret_cast<const float4*>(src + 4 * v4);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          bf16x8g v;
          v[0] = (__bf16)w[2 * hf].x; v[1] = (__bf16)w[2 * hf].y; v[2] = (__bf16)w[2 * hf].z; v[3] = (__bf16)w[2 * hf].w;
          v[4] = (__bf16)w[2 * hf + 1].x; v[5] = (__bf16)w[2 * hf + 1].y; v[6] = (__bf16)w[2 * hf + 1].z; v[7] = (__bf16)w[2 * hf + 1].w;
          *reinterpret_cast<bf16x8g*>(Wt + h * LDT1 + iq + 8 * hf) = v;
        }
      }
      __syncthreads();
      f32x16 o;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const bf16x8g av = fast_frag<1>(Wt, wr, s2, lane);
        const bf16x8g bv2 = fast_frag<1>(Gt, wc, s2, lane);
        const bf16x8g bv3 = fast_frag<1>(Gl, wc, s2, lane);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv3, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv2, o, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) Cz[(int64_t)(i0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * g.ldc + n] = o[r];
      __syncthreads();
    }
    return;
  }
  const float bv = g.bias ? g.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    float v = acc[r] * g.alpha + bv;
    if (g.relu) v = fmaxf(v, 0.f);
    if (g.c_bf16) reinterpret_cast<uint16_t*>(Cz)[(int64_t)m * g.ldc + n] = tt_f2bf(v);
    else Cz[(int64_t)m * g.ldc + n] = v;
  }
}

template <int MODE_A, int MODE_B, bool COLSUM, bool A_BF16, bool B_BF16>
__global__ __launch_bounds__(THREADS, 4) void gemm_bf16_fast_kernel(GemmBatch batch, int zsplits) {
  const GemmArgs& g = batch.a[blockIdx.z / zsplits];
  const int split = blockIdx.z % zsplits;
  if ((int)blockIdx.x * BM >= g.M || (int)blockIdx.y * BN >= g.N || split >= g.splits) return;
  __shared__ __attribute__((aligned(16))) FastSmem sm;
  gemm_fast_tile<MODE_A, MODE_B, COLSUM, A_BF16, B_BF16>(g, split, blockIdx.x, blockIdx.y, sm);
}

// ---- first-block backward of all towers in ONE launch ---------------------------------------------------------------
// Per tower three independent products of the block's incoming gradient d_pre [B, H]:
//   role 0  dW  = d_pre^T . x       (+ column sums of d_pre = the bias gradient)      split over the batch, slabs
//   role 1  G   = d_pre^T . dense                                                     split over the batch, slabs
//   role 2  d_x[:, h0:] = d_pre . W[:, h0:]                                           (the looked-up rows' gradients)
// The projection's weight gradient needs d_proj = d_x[:, 0:h0] as an operand when computed directly
// (dW_proj = d_proj^T . dense) -- a dependency that costs a launch; since d_proj = d_pre . W[:, 0:h0],
// dW_proj = W[:, 0:h0]^T . G = sum over the batch splits z of W[:, 0:h0]^T . G_z: a role-1 workgroup multiplies its own G_z
// tile from the left before it writes the slab (one more 64-deep MFMA pass: W rounded to bf16 like every Linear operand, G_z
// as hi + lo bf16 pairs), the ordinary slab
// reduction finishes it; db_proj = W[:, 0:h0]^T . db is a 64 x h0 matrix-vector product in the slab-reduction launch.
// d_x[:, 0:h0] is never materialised.  Flat grid: workgroup ->
// (problem, split, tile) through a prefix table, long problems first.
// role 2 of gemm_back_kernel with K = 64: C[64 rows, nb x 64 columns] = A[64, 64] . W[64, columns].  The row tile of A (d_pre)
// is staged in LDS ONCE for up to kNnCols column tiles -- as one 64 x 64 tile per workgroup every tile re-read its 16 KB of A
// next to 16 KB of W for 16 KB of output (78 MB of L2 reads for the 40 MB of d_x at B = 8192).
// WB16 (round 4): W arrives as its bf16 shadow (tt_tower_params.w_bf16): half the bytes of the operand every workgroup re-reads.
constexpr int kNnCols = 4;
template <bool WB16>
__device__ __forceinline__ void gemm_nn_k64_tiles(const GemmArgs& g, int bx, int by0, int nb, FastSmem& sm) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1, li = lane & 31, lh = lane >> 5;
  const int m0 = bx * BM;
  FastLoader16<0, false> la[2];
  FastLoader16<1, WB16> lb[2];
  la[0].load(g.A, g.lda, m0, 0, t);
  la[1].load(g.A, g.lda, m0, BK16, t);
  lb[0].load(g.B, g.ldb, by0 * BN, 0, t);
  lb[1].load(g.B, g.ldb, by0 * BN, BK16, t);
  la[0].store(sm.As[0], t);
  la[1].store(sm.As[1], t);
  for (int j = 0; j < nb; ++j) {
    const int n0 = (by0 + j) * BN;
    lb[0].store(sm.Bs[0], t);
    lb[1].store(sm.Bs[1], t);
    __syncthreads();
    const int nn = by0 + (j + 1 < nb ? j + 1 : j);          // next column tile (the last one again: discarded)
    lb[0].load(g.B, g.ldb, nn * BN, 0, t);
    lb[1].load(g.B, g.ldb, nn * BN, BK16, t);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int s2 = 0; s2 < BK16 / 16; ++s2) {
        const bf16x8g av = fast_frag<0>(sm.As[st], wr, s2, lane);
        const bf16x8g bv = fast_frag<1>(sm.Bs[st], wc, s2, lane);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
      }
    const int n = n0 + wc * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (g.c_bf16) reinterpret_cast<uint16_t*>(g.C)[(int64_t)m * g.ldc + n] = tt_f2bf(acc[r]);
      else g.C[(int64_t)m * g.ldc + n] = acc[r];
    }
    __syncthreads();                                       // the B buffers are free for the next column tile
  }
}

constexpr int kBackProbs = 3 * TT_MAX_SIDES;
struct BackBatch {
  GemmArgs g[kBackProbs];
  int role[kBackProbs], tiles_m[kBackProbs], tiles[kBackProbs], wg_end[kBackProbs];
  int n;
};

template <bool X_BF16>
__global__ __launch_bounds__(THREADS, 4) void gemm_back_kernel(BackBatch b) {
  int p = 0;
  while (p + 1 < b.n && (int)blockIdx.x >= b.wg_end[p]) ++p;
  const int local = (int)blockIdx.x - (p ? b.wg_end[p - 1] : 0);
  const int split = local / b.tiles[p], tile = local % b.tiles[p];
  const int bx = tile % b.tiles_m[p], by = tile / b.tiles_m[p];
  __shared__ __attribute__((aligned(16))) BackSmem sm;
  const GemmArgs& g = b.g[p];
  if (b.role[p] == 0) gemm_fast_tile<1, 1, true, false, X_BF16>(g, split, bx, by, sm);
  else if (b.role[p] == 1) gemm_fast_tile<1, 1, false, false, false, true>(g, split, bx, by, sm);
  else if (g.K == 64) {                                    // by = group of kNnCols column tiles
    const int nt_cols = g.N / BN;
    if (g.b_bf16) gemm_nn_k64_tiles<true>(g, bx, by * kNnCols, min(kNnCols, nt_cols - by * kNnCols), sm);
    else gemm_nn_k64_tiles<false>(g, bx, by * kNnCols, min(kNnCols, nt_cols - by * kNnCols), sm);
  } else gemm_fast_tile<0, 1, false, false, false>(g, split, bx, by, sm);
}

__global__ __launch_bounds__(THREADS) void slab_reduce_kernel(SlabBatch batch) {
  slab_reduce_block(batch, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

inline bool vec_ok(const float* p, int64_t ld, bool bf16_elems = false) {
  return tt_aligned(p, 16) && (ld % (bf16_elems ? 8 : 4) == 0);
}

inline int tn_splits(int64_t M, int64_t N, int64_t R) {
  const int64_t tiles = tt_cdiv(M, BM) * tt_cdiv(N, BN);
  // ~2 workgroups per CU (TT_GEMM_TN_WGS: A/B runs).  More splits shorten each workgroup's batch range but every split
  // writes a 64 x 64 f32 slab: at 1024 workgroups the slabs of the block weight gradient (16.5 MB) were as large as its
  // operands; 512 measured 5 us per step faster than 1024, 384 and 256 slower again
  const int64_t target = 512;
  int64_t s = target / (tiles > 0 ? tiles : 1);
  int64_t maxs = tt_cdiv(R, 128);
  if (maxs > 64) maxs = 64;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace

template <int MA, int MB>
static int launch_gemm(hipStream_t st, const GemmBatch& b, int n, int zsplits, bool vec, bool colsum, bool bf16 = false) {
  int mt = 1, nt = 1;
  for (int i = 0; i < n; ++i) {
    mt = (int)tt_cdiv(b.a[i].M, BM) > mt ? (int)tt_cdiv(b.a[i].M, BM) : mt;
    nt = (int)tt_cdiv(b.a[i].N, BN) > nt ? (int)tt_cdiv(b.a[i].N, BN) : nt;
  }
  dim3 grid((unsigned)mt, (unsigned)nt, (unsigned)(n * zsplits));
  if (bf16) {
    // shapes without edges and one element type per operand across the batch: the three-steps-ahead kernel
    bool fast = vec;
    for (int i = 0; i < n && fast; ++i) {
      const GemmArgs& g = b.a[i];
      fast = g.M % BM == 0 && g.N % BN == 0 && g.K % BK16 == 0 && g.kchunk % BK16 == 0 && g.a_bf16 == b.a[0].a_bf16 &&
             g.b_bf16 == b.a[0].b_bf16 && !(MA == 1 && g.a_bf16) && !(MA == 0 && MB == 0 && g.b_bf16) && !(MB == 1 && MA == 0 && (g.a_bf16 || g.b_bf16));
    }
    if (fast) {
      const bool ab = b.a[0].a_bf16, bb = b.a[0].b_bf16;
      if (MA == 0 && MB == 0) {
        if (ab) gemm_bf16_fast_kernel<0, 0, false, true, false><<<grid, THREADS, 0, st>>>(b, zsplits);
        else gemm_bf16_fast_kernel<0, 0, false, false, false><<<grid, THREADS, 0, st>>>(b, zsplits);
      } else if (MA == 0 && MB == 1) {
        gemm_bf16_fast_kernel<0, 1, false, false, false><<<grid, THREADS, 0, st>>>(b, zsplits);
      } else {
        if (colsum) {
          if (bb) gemm_bf16_fast_kernel<1, 1, true, false, true><<<grid, THREADS, 0, st>>>(b, zsplits);
          else gemm_bf16_fast_kernel<1, 1, true, false, false><<<grid, THREADS, 0, st>>>(b, zsplits);
        } else {
          if (bb) gemm_bf16_fast_kernel<1, 1, false, false, true><<<grid, THREADS, 0, st>>>(b, zsplits);
          else gemm_bf16_fast_kernel<1, 1, false, false, false><<<grid, THREADS, 0, st>>>(b, zsplits);
        }
      }
      TT_LAUNCH_CHECK();
      return TT_OK;
    }
    if (colsum) gemm_bf16_kernel<MA, MB, true><<<grid, THREADS, 0, st>>>(b, zsplits, vec ? 1 : 0);
    else gemm_bf16_kernel<MA, MB, false><<<grid, THREADS, 0, st>>>(b, zsplits, vec ? 1 : 0);
    TT_LAUNCH_CHECK();
    return TT_OK;
  }
  if (colsum) {
    if (vec) gemm_kernel<MA, MB, true, true><<<grid, THREADS, 0, st>>>(b, zsplits);
    else gemm_kernel<MA, MB, false, true><<<grid, THREADS, 0, st>>>(b, zsplits);
  } else {
    if (vec) gemm_kernel<MA, MB, true, false><<<grid, THREADS, 0, st>>>(b, zsplits);
    else gemm_kernel<MA, MB, false, false><<<grid, THREADS, 0, st>>>(b, zsplits);
  }
  TT_LAUNCH_CHECK();
  return TT_OK;
}

static int nt_splits(int64_t tiles_all, int64_t K) {
  if (tiles_all >= 512) return 1;                     // two workgroups per CU already: a slab pass (~7 us) costs more than it buys
  const int64_t target = 1024;
  int64_t sp = target / (tiles_all > 0 ? tiles_all : 1);
  const int64_t maxs = K / 128;                       // at least 128 of K per split
  if (sp > maxs) sp = maxs;
  if (sp > 16) sp = 16;
  return (int)(sp < 1 ? 1 : sp);
}

size_t tt_gemm_nt_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  return sizeof(float) * 16 * (size_t)M * (size_t)N + 256;     // worst case: 16 slabs
}

int tt_gemm_nt_batched(hipStream_t st, const GemmNT* it, int n) {
  GemmBatch b{};
  SlabBatch sb{};
  bool vec = true, can_split = true;
  int m = 0;
  int64_t tiles_all = 0, kmin = INT64_MAX, maxtotal = 1;
  for (int i = 0; i < n; ++i) {
    if (it[i].M == 0 || it[i].N == 0) continue;
    tiles_all += tt_cdiv(it[i].M, BM) * tt_cdiv(it[i].N, BN);
    kmin = it[i].K < kmin ? it[i].K : kmin;
    can_split = can_split && it[i].workspace && it[i].workspace_bytes >= tt_gemm_nt_workspace_bytes(it[i].M, it[i].N, it[i].K);
  }
  int zsplits = 1;
  bool any_split = false;
  for (int i = 0; i < n; ++i) {
    if (it[i].M == 0 || it[i].N == 0) continue;
    const int sp = (can_split && it[0].bf16) ? nt_splits(tiles_all, it[i].K) : 1;
    zsplits = sp > zsplits ? sp : zsplits;
    any_split = any_split || sp > 1;
  }
  (void)kmin;
  for (int i = 0; i < n; ++i) {
    if (it[i].M == 0 || it[i].N == 0) continue;
    // every problem of a split launch goes through slabs (a problem whose K is too short simply uses one slab)
    const int splits = any_split ? nt_splits(tiles_all, it[i].K) : 1;
    const int kchunk = (int)(tt_cdiv(tt_cdiv(it[i].K > 0 ? it[i].K : 1, splits), BK16) * BK16);
    if (any_split) {
      float* slabs = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(it[i].workspace) + 255) & ~uintptr_t(255));
      b.a[m] = GemmArgs{it[i].A, it[i].lda, it[i].W, it[i].ldw, slabs, it[i].N, it[i].M * it[i].N, (int)it[i].M, (int)it[i].N,
                        (int)it[i].K, kchunk, splits, nullptr, 0, it[i].alpha, nullptr, it[i].a_bf16 ? 1 : 0, 0, 0};
      sb.a[m] = SlabArgs{slabs, it[i].M * it[i].N, splits, it[i].C, it[i].ldc, (int)it[i].M, (int)it[i].N, nullptr, nullptr,
                         it[i].bias, it[i].relu ? 1 : 0, it[i].c_bf16 ? 1 : 0};
      maxtotal = it[i].M * it[i].N > maxtotal ? it[i].M * it[i].N : maxtotal;
    } else {
      b.a[m] = GemmArgs{it[i].A, it[i].lda, it[i].W, it[i].ldw, it[i].C, it[i].ldc, 0, (int)it[i].M, (int)it[i].N, (int)it[i].K,
                        kchunk, 1, it[i].bias, it[i].relu ? 1 : 0, it[i].alpha, nullptr, it[i].a_bf16 ? 1 : 0, 0, it[i].c_bf16 ? 1 : 0};
    }
    if ((it[i].a_bf16 || it[i].c_bf16) && !it[0].bf16) {
      tt_set_error("tt_gemm_nt: bf16 tensors need the bf16 compute path");
      return TT_ERR_INVALID_ARG;
    }
    vec = vec && vec_ok(it[i].A, it[i].lda, it[i].a_bf16) && vec_ok(it[i].W, it[i].ldw);
    ++m;
  }
  if (!m) return TT_OK;
  if (int rc = launch_gemm<0, 0>(st, b, m, zsplits, vec, false, it[0].bf16)) return rc;
  if (it[0].defer) {
    for (int i = 0, k = 0; i < n; ++i) {
      if (!it[i].defer) { tt_set_error("tt_gemm_nt: defer must be set on every problem of a launch"); return TT_ERR_INVALID_ARG; }
      if (it[i].M == 0 || it[i].N == 0) { *it[i].defer = NtDeferred{nullptr, 0, 0}; continue; }
      *it[i].defer = any_split ? NtDeferred{sb.a[k].slabs, sb.a[k].slab_stride, sb.a[k].splits} : NtDeferred{nullptr, 0, 0};
      ++k;
    }
    return TT_OK;
  }
  if (any_split) {
    int blocks = (int)tt_cdiv(maxtotal, THREADS);
    if (blocks > 1024) blocks = 1024;
    slab_reduce_kernel<<<dim3((unsigned)blocks, (unsigned)m), THREADS, 0, st>>>(sb);
    TT_LAUNCH_CHECK();
  }
  return TT_OK;
}

int tt_gemm_nn_batched(hipStream_t st, const GemmNN* it, int n) {
  GemmBatch b{};
  bool vec = true;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (it[i].M == 0 || it[i].N == 0) continue;
    b.a[m++] = GemmArgs{it[i].A, it[i].lda, it[i].W, it[i].ldw, it[i].C, it[i].ldc, 0, (int)it[i].M, (int)it[i].N, (int)it[i].K,
                        (int)(tt_cdiv(it[i].K > 0 ? it[i].K : 1, BK16) * BK16), 1, nullptr, 0, 1.f, nullptr, 0, 0, it[i].c_bf16 ? 1 : 0};
    if (it[i].c_bf16 && !it[0].bf16) {
      tt_set_error("tt_gemm_nn: a bf16 output needs the bf16 compute path");
      return TT_ERR_INVALID_ARG;
    }
    vec = vec && vec_ok(it[i].A, it[i].lda) && vec_ok(it[i].W, it[i].ldw);
  }
  return m ? launch_gemm<0, 1>(st, b, m, 1, vec, false, n > 0 && it[0].bf16) : TT_OK;
}

size_t tt_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t R) {
  return sizeof(float) * (size_t)tn_splits(M, N, R) * ((size_t)M * (size_t)N + (size_t)M) + 256;
}

// (TnPending: tt_gemm.h)
TnPending* tt_gemm_tn_pending_create() { return new TnPending(); }
void tt_gemm_tn_pending_destroy(TnPending* p) { delete p; }

int tt_gemm_tn_defer(tt_ctx* ctx, TnPending* p) {
  if (!p || p->n == 0) return TT_OK;
  if (!ctx->deferred) ctx->deferred = tt_gemm_tn_pending_create();
  TT_CHECK_ARG(ctx->deferred->n == 0, "deferred slab reduction: the previous one was never flushed");
  *ctx->deferred = *p;
  p->n = 0;
  p->maxtotal = 1;
  return TT_OK;
}

int tt_gemm_deferred_flush(tt_ctx* ctx, hipStream_t st) { return ctx && ctx->deferred ? tt_gemm_tn_flush(st, ctx->deferred) : TT_OK; }

int tt_gemm_tn_flush(hipStream_t st, TnPending* p) {
  if (!p || p->n == 0) return TT_OK;
  int blocks = (int)tt_cdiv(p->maxtotal, THREADS);
  if (blocks > 1024) blocks = 1024;
  slab_reduce_kernel<<<dim3((unsigned)blocks, (unsigned)p->n), THREADS, 0, st>>>(p->sb);
  p->n = 0;
  p->maxtotal = 1;
  TT_LAUNCH_CHECK();
  return TT_OK;
}

int tt_gemm_tn_batched(hipStream_t st, const GemmTN* it, int n, TnPending* pending) {
  GemmBatch b{};
  SlabBatch sb{};
  bool vec = true, colsum = false;
  int m = 0, zs = 1;
  int64_t maxtotal = 1;
  for (int i = 0; i < n; ++i) {
    const GemmTN& t = it[i];
    if (t.M == 0 || t.N == 0) continue;
    const int splits = tn_splits(t.M, t.N, t.R);
    if (!t.workspace || t.workspace_bytes < tt_gemm_tn_workspace_bytes(t.M, t.N, t.R)) {
      tt_set_error("tt_gemm_tn: workspace %zu < required %zu", t.workspace_bytes, tt_gemm_tn_workspace_bytes(t.M, t.N, t.R));
      return TT_ERR_WORKSPACE;
    }
    float* slabs = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(t.workspace) + 255) & ~uintptr_t(255));
    float* cslab = slabs + (size_t)splits * (size_t)t.M * (size_t)t.N;
    const int kchunk = (int)(tt_cdiv(tt_cdiv(t.R > 0 ? t.R : 1, splits), BK16) * BK16);
    b.a[m] = GemmArgs{t.A, t.lda, t.B, t.ldb, slabs, t.N, t.M * t.N, (int)t.M, (int)t.N, (int)t.R, kchunk, splits, nullptr, 0, 1.f,
                      t.colsum_out ? cslab : nullptr, t.a_bf16 ? 1 : 0, t.b_bf16 ? 1 : 0, 0};
    if ((t.a_bf16 || t.b_bf16) && !it[0].bf16) {
      tt_set_error("tt_gemm_tn: bf16 tensors need the bf16 compute path");
      return TT_ERR_INVALID_ARG;
    }
    sb.a[m] = SlabArgs{slabs, t.M * t.N, splits, t.C, t.ldc, (int)t.M, (int)t.N, cslab, t.colsum_out, nullptr, 0};
    vec = vec && vec_ok(t.A, t.lda, t.a_bf16) && vec_ok(t.B, t.ldb, t.b_bf16);
    colsum = colsum || t.colsum_out != nullptr;
    zs = splits > zs ? splits : zs;
    const int64_t tot = t.M * t.N + (t.colsum_out ? t.M : 0);
    maxtotal = tot > maxtotal ? tot : maxtotal;
    ++m;
  }
  if (!m) return TT_OK;
  if (int rc = launch_gemm<1, 1>(st, b, m, zs, vec, colsum, it[0].bf16)) return rc;
  if (pending) {
    if (pending->n + m > kSlabItems)
      if (int rc = tt_gemm_tn_flush(st, pending)) return rc;
    for (int i = 0; i < m; ++i) pending->sb.a[pending->n++] = sb.a[i];
    pending->maxtotal = maxtotal > pending->maxtotal ? maxtotal : pending->maxtotal;
    return TT_OK;
  }
  int blocks = (int)tt_cdiv(maxtotal, THREADS);
  if (blocks > 1024) blocks = 1024;
  slab_reduce_kernel<<<dim3((unsigned)blocks, (unsigned)m), THREADS, 0, st>>>(sb);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

static int back_g_splits(int64_t H, int64_t din, int64_t B) {
  int s = tn_splits(H, din, B);
  return s > 32 ? 32 : s;                               // the projection finish re-reads these slabs per 16-column block
}
size_t tt_gemm_back_g_workspace_bytes(int64_t H, int64_t h0, int64_t din, int64_t B) {
  return sizeof(float) * (size_t)back_g_splits(H, din, B) * (size_t)(H / BM) * (size_t)h0 * (size_t)din + 256;
}

bool tt_gemm_back_supported(const GemmBack* it, int n) {
  if (n < 1 || n > TT_MAX_SIDES) return false;
  for (int i = 0; i < n; ++i) {
    const GemmBack& g = it[i];
    if (g.B < 64 || g.B % 64 || g.H < 64 || g.H % 64 || g.H > kProjMaxH || g.kx % 64 || g.h0 % 64 || g.h0 < 64 || g.h0 >= g.kx || g.din % 64) return false;
    // every (split, 64-row block of h) writes an [h0, din] slab: beyond (H / 64) x h0 = 256 the slabs outweigh what the one launch
    // saves (H = 256, h0 = 512: 67 MB of slabs, 61 us to add them up) -- the separate launches take those shapes
    if ((g.H / BM) * g.h0 > 256) return false;
    if (g.x_bf16 != it[0].x_bf16) return false;
    if (!tt_aligned(g.dpre, 16) || !tt_aligned(g.x, 16) || g.ldx % 8 || !tt_aligned(g.dense, 16) || g.ld_dense % 4 || !tt_aligned(g.w, 16) ||
        !tt_aligned(g.dx, 16) || g.ld_dx % 4)
      return false;
    if (!g.ws_dw || g.ws_dw_bytes < tt_gemm_tn_workspace_bytes(g.H, g.kx, g.B) || !g.ws_g ||
        g.ws_g_bytes < tt_gemm_back_g_workspace_bytes(g.H, g.h0, g.din, g.B))
      return false;
  }
  return true;
}

int tt_gemm_back_batched(hipStream_t st, const GemmBack* it, int n, TnPending* pending) {
  if (!pending || !tt_gemm_back_supported(it, n)) {
    tt_set_error("tt_gemm_back: unsupported shapes / missing pending queue");
    return TT_ERR_INVALID_ARG;
  }
  BackBatch b{};
  SlabArgs sa[3 * TT_MAX_SIDES];
  int ns = 0, wg = 0;
  int64_t maxtotal = 1;
  auto kchunk_of = [](int64_t R, int splits) { return (int)(tt_cdiv(tt_cdiv(R, splits), BK16) * BK16); };
  for (int role = 0; role < 3; ++role)                   // long problems first
    for (int i = 0; i < n; ++i) {
      const GemmBack& g = it[i];
      const int s0 = tn_splits(g.H, g.kx, g.B), s1 = back_g_splits(g.H, g.din, g.B);
      float* slabs0 = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(g.ws_dw) + 255) & ~uintptr_t(255));
      float* cslab0 = slabs0 + (size_t)s0 * (size_t)g.H * (size_t)g.kx;
      float* slabs1 = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(g.ws_g) + 255) & ~uintptr_t(255));
      const int p = b.n++;
      int tm, tn_;
      if (role == 0) {
        b.g[p] = GemmArgs{g.dpre, g.H, reinterpret_cast<const float*>(g.x), g.ldx, slabs0, g.kx, (int64_t)g.H * g.kx, g.H, g.kx, (int)g.B,
                          kchunk_of(g.B, s0), s0, nullptr, 0, 1.f, cslab0, 0, g.x_bf16 ? 1 : 0, 0};
        tm = g.H / BM; tn_ = g.kx / BN;
        sa[ns++] = SlabArgs{slabs0, (int64_t)g.H * g.kx, s0, g.dw, g.kx, g.H, g.kx, cslab0, g.db, nullptr, 0};
        maxtotal = (int64_t)g.H * g.kx + g.H > maxtotal ? (int64_t)g.H * g.kx + g.H : maxtotal;
      } else if (role == 1) {
        b.g[p] = GemmArgs{g.dpre, g.H, g.dense, g.ld_dense, slabs1, g.din, (int64_t)g.h0 * g.din, g.H, g.din, (int)g.B,
                          kchunk_of(g.B, s1), s1, nullptr, 0, 1.f, nullptr, 0, 0, 0};
        b.g[p].proj_w = g.w; b.g[p].proj_ldw = g.kx; b.g[p].proj_h0 = g.h0;
        tm = g.H / BM; tn_ = g.din / BN;
        sa[ns++] = SlabArgs{slabs1, (int64_t)g.h0 * g.din, s1 * (g.H / BM), g.dwp, g.din, g.h0, g.din, nullptr, nullptr, nullptr, 0};
        SlabArgs pj{nullptr, 0, 0, nullptr, 0, g.H, 0, cslab0, g.dbp, nullptr, 0};
        pj.proj_w = g.w; pj.proj_ldw = g.kx; pj.proj_h0 = g.h0; pj.proj_colsum_splits = s0;
        sa[ns++] = pj;
        maxtotal = (int64_t)g.h0 * g.din > maxtotal ? (int64_t)g.h0 * g.din : maxtotal;
      } else {
        float* c = g.dx_bf16 ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(g.dx) + g.h0) : g.dx + g.h0;
        // (the K = 64 form can read W's bf16 shadow: the same values it would round on the way into LDS)
        const bool w16 = g.w16 != nullptr && g.H == 64 && tt_aligned(g.w16, 16);
        const float* wsrc = w16 ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(g.w16) + g.h0) : g.w + g.h0;
        b.g[p] = GemmArgs{g.dpre, g.H, wsrc, g.kx, c, g.ld_dx, 0, (int)g.B, g.kx - g.h0, g.H, (int)(tt_cdiv(g.H, BK16) * BK16), 1,
                          nullptr, 0, 1.f, nullptr, 0, w16 ? 1 : 0, g.dx_bf16 ? 1 : 0};
        tm = (int)(g.B / BM); tn_ = (g.kx - g.h0) / BN;
        if (g.H == 64) tn_ = (int)tt_cdiv(tn_, kNnCols);    // K = 64: a workgroup takes kNnCols column tiles (gemm_nn_k64_tiles)
      }
      b.role[p] = role;
      b.tiles_m[p] = tm;
      b.tiles[p] = tm * tn_;
      wg += tm * tn_ * b.g[p].splits;
      b.wg_end[p] = wg;
    }
  if (it[0].x_bf16) gemm_back_kernel<true><<<wg, THREADS, 0, st>>>(b);
  else gemm_back_kernel<false><<<wg, THREADS, 0, st>>>(b);
  TT_LAUNCH_CHECK();
  if (pending->n + ns > kSlabItems)
    if (int rc = tt_gemm_tn_flush(st, pending)) return rc;
  for (int i = 0; i < ns; ++i) pending->sb.a[pending->n++] = sa[i];
  pending->maxtotal = maxtotal > pending->maxtotal ? maxtotal : pending->maxtotal;
  return TT_OK;
}

int tt_gemm_nt(hipStream_t st, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C,
               int64_t ldc, int64_t M, int64_t N, int64_t K, bool relu, float alpha) {
  GemmNT it{A, lda, W, ldw, bias, C, ldc, M, N, K, relu, alpha};
  return tt_gemm_nt_batched(st, &it, 1);
}
