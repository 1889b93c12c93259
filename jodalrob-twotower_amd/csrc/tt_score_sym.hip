// Single-pass symmetric forward of the in-batch-negative score + softmax-CE (two_tower_train_task.py:99-179) on the square
// training problem: S = N C^T is swept ONCE; every 32x32 tile feeds both softmax directions.
//
// The two-direction kernel of tt_score_bf16.hip computes every tile (MFMA + exp2) twice, once per direction, because a
// 32x32 MFMA result holds one index on the lanes and the other in the registers: sums over the register index are
// in-lane adds, sums over the lane index are cross-lane reductions (80 DPP steps per tile).  Here a wave keeps ONE tile J
// of company rows b as its resident operand and sweeps a chunk of notice tiles I; X[b = register][a = lane]:
//   * column direction (per b, sum over a): 16 per-lane accumulators that live across the whole sweep -- the cross-lane
//     reduction happens once per sweep, not once per tile;
//   * row direction (per a, sum over b): the in-lane sum over the 16 registers is the tile's partial for 32 rows a; it goes
//     to a wave-private LDS slot (one ds_write per tile).  After the sweep the workgroup adds the slots of its 8 waves
//     (8 tiles J) in wave order and writes ONE partial per (J-group, a) to a slab [n_groups][B].
// A second launch (finish1, B/256 workgroups) adds the slab rows in a fixed order, forms the per-row terms of the loss and
// the metrics, and leaves rowsum / colsum / their reciprocals for the backward kernel; finish2 (one workgroup) adds the
// workgroups' partial sums.  Every sum has a fixed order: results are bitwise reproducible.
//
// Top-1 accuracy: per (J, a) the tile's maximum goes to a "before the positive" or "after the positive" slot (ties before
// the positive beat it, ties after do not -- torch.argmax takes the first maximum); the diagonal tile splits per element.
// sum of all scores (negative_similarity_mean): sum_ab n_a . c_b = (sum_a n_a) . (sum_b c_b) -- finish1 adds the operand
// images' columns instead of the kernel adding 67 M products.
#include "tt_score_bf16.h"
#include "tt_riders.h"

#include <stdlib.h>

namespace {

using namespace ttscore;

constexpr int kSymWaves = 8;           // tiles J per workgroup
constexpr int kFin2Fold = 32;          // partial records a workgroup of the pre-reduction (score_sym_fold_kernel) folds into one
constexpr int kSymThreads = kSymWaves * 64;

// x[lane] + x[lane ^ 32] in every lane: v_permlane32_swap exchanges the upper half of one register with the lower half of the
// other -- on two copies of x that leaves {lo, lo} and {hi, hi}.  A VALU op: no LDS round trip as __shfl_xor(x, 32) (ds_bpermute)
// (inline asm: through __builtin_amdgcn_permlane32_swap hipcc 7.2 drops the second result and adds r[0] to itself.  The
//  s_nop covers the VALU-write -> permlane-read wait states the compiler would have inserted for its own instruction)
__device__ __forceinline__ void half_swap(float x, float& lo, float& hi) {
  unsigned a = __builtin_bit_cast(unsigned, x), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  lo = __builtin_bit_cast(float, a);
  hi = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float half_sum(float x) {
  float p, q;
  half_swap(x, p, q);
  return p + q;
}
__device__ __forceinline__ float half_max(float x) {
  float p, q;
  half_swap(x, p, q);
  return fmaxf(p, q);
}

struct SymArgs {
  const void* a_rows;                  // notice image (rows image: bf16, or fp8 for the FP8 kernels), may carry the exponent scale
  const void* b_rows;                  // company image
  int R, nT, NI;                       // rows, 32-row tiles, notice tiles per workgroup
  int64_t Rp;                          // slab row stride (floats)
  float c1, c2;                        // non-unit form: exp2(acc * c1 + c2)
  float* rs; float* mb; float* ma;     // [n_groups][Rp]: exp-sum / max before / max after the positive, per (J group, a)
  float* cs;                           // [n_chunks][Rp]: exp-sum per (a chunk, b)
  float* diag_raw;                     // [R] the positives' products, as the MFMA delivers them
  int want_rank;
};

// Notice tiles are staged through LDS once per WORKGROUP (all 8 waves sweep the same tiles I): read straight from L2 by
// every wave, the operand stream was 4 KB per 32 x 32 tile -- 268 MB per launch at B = 8192, D = 64 -- and the kernel ran at
// the L2's ~11 TB/s, not at its VALU rate.  A stage = TS tiles (8 KB; 16 KB at D = 64 and D = 256) in the images' own fragment order, so
// the copy is verbatim (16 bytes per thread) and a wave's ds_read_b128 of a fragment is 1 KB contiguous: conflict-free.
// Double buffered, one barrier per stage; the next stage's global loads are in flight while this one is computed.
template <int KS, bool FP8>
struct SymStage {
  static constexpr int kTileB = FP8 ? KS * 512 : KS * 1024;                 // bytes of one 32-row tile of the rows image
  // tiles per stage (D = 64: 2 -> 4 tiles, half the barriers: sweep 22.2 -> 20.8 us).  8-KB tiles: 2 for bf16 D = 128; the fp8
  // D = 256 kernel holds two company tiles per wave and spills with two notice tiles' fragments in flight (256 VGPRs + scratch)
  static constexpr int TS = kTileB <= 4096 ? 4 : ((kTileB <= 8192 && !FP8) ? 2 : 1);
  static constexpr int kBytes = TS * kTileB;
  static constexpr int LPT = kBytes / (kSymThreads * 16);                   // 16-byte pieces per thread per stage
  static_assert(LPT >= 1 && LPT * kSymThreads * 16 == kBytes && kTileB % 1024 == 0, "stage must be whole 16-byte pieces, a wave's 64 inside one tile");
};

// JT company tiles per wave (fp8, D = 256: 2).  With one tile per wave and one notice tile per stage a wave meets a workgroup
// barrier every 32 x 32 tile and waits 35 % of its life (profiles/r03_pmc_configs4_fp8.json); with two, a stage's fragments feed two
// MFMA chains, the barriers and stage copies per score halve, and the slab of row partials has half the rows.  The two tiles'
// row partials are added (and their maxima joined) in registers before the slot is written: tiles J0, J0 + 1 lie on the same
// side of the positives of every notice tile but their own.
template <int KS, bool UNIT, bool FP8, int JT>
__global__ __launch_bounds__(kSymThreads) void score_fwd_sym_kernel(SymArgs g) {
  using ST = SymStage<KS, FP8>;
  constexpr int TS = ST::TS, LPT = ST::LPT, kTileB = ST::kTileB, K64 = FP8 ? KS / 4 : 1;
  static_assert(!FP8 || KS % 4 == 0, "fp8 operands come in K = 64 steps");
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [3][8 waves][NI * 32] slots | 2 stage buffers
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NI = g.NI, nT = g.nT, R = g.R;
  const int slots = NI * 32;
  float* s_sum = lds + (0 * kSymWaves + wave) * slots;
  float* s_mb = lds + (1 * kSymWaves + wave) * slots;
  float* s_ma = lds + (2 * kSymWaves + wave) * slots;
  char* stage = reinterpret_cast<char*>(lds + 3 * kSymWaves * slots);
  const int J0 = ((int)blockIdx.x * kSymWaves + wave) * JT;  // this wave's company tiles J0 .. J0 + JT - 1
  const int I0 = (int)blockIdx.y * NI, I1 = min(I0 + NI, nT);
  const bool active = J0 < nT;
  for (int i = lane; i < slots; i += 64) { s_sum[i] = 0.f; s_mb[i] = kNegBig; s_ma[i] = kNegBig; }
  const float c1 = g.c1, c2 = g.c2;
  auto ex = [&](float x) { return UNIT ? __builtin_amdgcn_exp2f(x) : __builtin_amdgcn_exp2f(__builtin_fmaf(x, c1, c2)); };
  float colacc[JT][16];
#pragma unroll
  for (int j = 0; j < JT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) colacc[j][r] = 0.f;
  bf16x8 bres[JT][FP8 ? 1 : KS];
  i32x8 bres8[JT][K64];
  const int n_full = R / 32;                               // tiles below n_full hold 32 valid rows
  bool jact[JT], jfull[JT];
  float dg_keep[JT];                                       // the positives of tile J (met once per wave, if J is in this chunk)
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    jact[j] = J0 + j < nT;
    jfull[j] = jact[j] && J0 + j < n_full;
    dg_keep[j] = kNegBig;
    if (FP8) load_f8frag<K64>(reinterpret_cast<const char*>(g.b_rows), jact[j] ? J0 + j : 0, c, h, bres8[j]);
    else load_bfrag<(FP8 ? 1 : KS)>(reinterpret_cast<const __bf16*>(g.b_rows), jact[j] ? J0 + j : 0, c, h, bres[j]);
  }
  // stage loader: the stage's tiles go STRAIGHT into the LDS buffer by LDS-DMA (global_load_lds_dwordx4: lane l of a wave writes
  // 16 bytes at the wave's base + 16 l), piece p = tid + 512 q of the stage at byte 16 p; tiles past the image's end are clamped
  // to its last tile (their results are never used).  Round 3: the copy used to pass through two named registers per thread and a
  // ds_write; the DMA of stage st + 1 is issued right behind the barrier that frees its buffer and has the whole stage to land.
  const int nst = (I1 - I0 + TS - 1) / TS;
  auto stage_dma = [&](int st, int buf) {
#pragma unroll
    for (int q = 0; q < LPT; ++q) {
      const int off = (q * kSymThreads + (int)threadIdx.x) * 16;           // byte offset inside the stage
      const int tl = off / kTileB;                                         // tile of the stage this piece belongs to (wave-uniform)
      const int tile = min(I0 + st * TS + tl, nT - 1);
      const char* src = reinterpret_cast<const char*>(g.a_rows) + (int64_t)tile * kTileB + (off - tl * kTileB);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(stage + buf * ST::kBytes + (q * kSymThreads + wave * 64) * 16), 16, 0, 0);
    }
  };
  stage_dma(0, 0);
  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of stage st have landed ...
    __syncthreads();                                       // ... and everybody's; the other buffer is no longer read by anyone
    if (st + 1 < nst) stage_dma(st + 1, (st + 1) & 1);
    if (active) {
      const char* sb = stage + (st & 1) * ST::kBytes;
#pragma unroll
      for (int tl = 0; tl < TS; ++tl) {
        const int I = I0 + st * TS + tl;
        // the tile's fragments in one burst of LDS reads (left to itself hipcc reads two, waits, issues two MFMAs, reads two
        // ...: at D = 256 every second MFMA then pays a full LDS round trip), the MFMA chains behind counted lgkmcnt waits
        f32x16 acc[JT];
#pragma unroll
        for (int j = 0; j < JT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if (FP8) {
          i32x8 af8[K64];
          load_f8frag<K64>(sb, tl, c, h, af8);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < K64; ++s)
#pragma unroll
            for (int j = 0; j < JT; ++j) acc[j] = mfma_f8(bres8[j][s], af8[s], acc[j]);
        } else {
          bf16x8 af[FP8 ? 1 : KS];
#pragma unroll
          for (int s = 0; s < (FP8 ? 1 : KS); ++s) af[s] = *reinterpret_cast<const bf16x8*>(sb + (((tl * KS + s) * 2 + h) * 32 + c) * 16);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < (FP8 ? 1 : KS); ++s)
#pragma unroll
            for (int j = 0; j < JT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bres[j][s], af[s], acc[j], 0, 0, 0);
        }
        const int il = (I - I0) * 32 + c;
        float rs_tot = 0.f, xb_tot = kNegBig, xa_tot = kNegBig;
#pragma unroll
        for (int j = 0; j < JT; ++j) {
          const int J = J0 + j;
          if (I < I1 && I != J && I < n_full && jfull[j]) {  // plain tile: 32 x 32 valid scores, all on one side of the positives
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) e[r] = ex(acc[j][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) colacc[j][r] = add_asm(colacc[j][r], e[r]);
            float t0 = (e[0] + e[1]) + (e[2] + e[3]), t1 = (e[4] + e[5]) + (e[6] + e[7]);
            float t2 = (e[8] + e[9]) + (e[10] + e[11]), t3 = (e[12] + e[13]) + (e[14] + e[15]);
            rs_tot += half_sum((t0 + t1) + (t2 + t3));
            if (g.want_rank) {
              float m = max3_asm(acc[j][0], acc[j][1], acc[j][2]);
#pragma unroll
              for (int r = 3; r < 15; r += 2) m = max3_asm(m, acc[j][r], acc[j][r + 1]);
              m = half_max(fmaxf(m, acc[j][15]));
              // every b of tile J lies before (J < I) / after the positive of every a of tile I
              if (J < I) xb_tot = fmaxf(xb_tot, m);
              else xa_tot = fmaxf(xa_tot, m);
            }
          } else if (I < I1 && jact[j]) {                  // the diagonal tile and the ragged last tiles: per element
            const int a = 32 * I + c;
            float rsum = 0.f, xb = kNegBig, xa = kNegBig, dg = kNegBig;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int b = 32 * J + rowmap(r, h);
              const float x = acc[j][r];
              const bool valid = b < R && a < R;
              const float e = valid ? ex(x) : 0.f;
              colacc[j][r] += e;
              rsum += e;
              xb = (valid && b < a) ? fmaxf(xb, x) : xb;
              xa = (valid && b > a) ? fmaxf(xa, x) : xa;
              dg = (b == a) ? x : dg;
            }
            rs_tot += half_sum(rsum);
            xb_tot = fmaxf(xb_tot, half_max(xb));
            xa_tot = fmaxf(xa_tot, half_max(xa));
            dg = half_max(dg);
            if (I == J) dg_keep[j] = dg;                   // taken from the MFMA result itself, so ties compare bit for bit
          }
        }
        if (I < I1 && h == 0) { s_sum[il] = rs_tot; s_mb[il] = xb_tot; s_ma[il] = xa_tot; }
      }
    }
  }
  if (active) {
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int J = J0 + j;
      if (!jact[j]) continue;
      if (J >= I0 && J < I1 && h == 0 && 32 * J + c < R) g.diag_raw[32 * J + c] = dg_keep[j];
      // column direction: one cross-lane reduction per sweep (fixed butterfly order), lanes c == 0 hold the 32 sums
#pragma unroll
      for (int o = 16; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < 16; ++r) colacc[j][r] += __shfl_xor(colacc[j][r], o);
      if (c == 0) {
        float* dst = g.cs + (int64_t)blockIdx.y * g.Rp + 32 * J;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[rowmap(r, h)] = colacc[j][r];  // (rows beyond R: slab padding, never read)
      }
    }
  }
  __syncthreads();
  // row direction: the 8 waves' slots in wave order -> one partial per (J group, a)
  for (int i = threadIdx.x; i < slots; i += kSymThreads) {
    const int a = I0 * 32 + i;                              // < Rp by construction (sym_layout)
    float s = 0.f, xb = kNegBig, xa = kNegBig;
#pragma unroll
    for (int w = 0; w < kSymWaves; ++w) {
      s += lds[(0 * kSymWaves + w) * slots + i];
      xb = fmaxf(xb, lds[(1 * kSymWaves + w) * slots + i]);
      xa = fmaxf(xa, lds[(2 * kSymWaves + w) * slots + i]);
    }
    const int64_t o = (int64_t)blockIdx.x * g.Rp + a;
    g.rs[o] = s;
    if (g.want_rank) { g.mb[o] = xb; g.ma[o] = xa; }
  }
}

// finish2 reads every workgroup's partial record in ONE workgroup: 4.6 us for 128 records (B = 8192), 50 us for 1024 (B = 65536:
// 2 MB through one CU).  From 256 records on, this launch folds them 32 to one first (fixed order: record k of the fold after
// record k - 1), and finish2 adds the folded ones.
__global__ __launch_bounds__(1024) void score_sym_fold_kernel(const float* __restrict__ part, int n_wg, int stride, float* __restrict__ part2) {
  const int j = threadIdx.x;
  if (j >= stride) return;
  const int r0 = (int)blockIdx.x * kFin2Fold, r1 = min(r0 + kFin2Fold, n_wg);
  float v[kFin2Fold];
#pragma unroll
  for (int k = 0; k < kFin2Fold; ++k) v[k] = r0 + k < r1 ? part[(int64_t)(r0 + k) * stride + j] : 0.f;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < kFin2Fold; ++k) s += v[k];
  part2[(int64_t)blockIdx.x * stride + j] = s;
}

// ---- finish1: slabs -> per-row sums, reciprocals, top-1 flags; per-workgroup partial sums of the loss terms; column sums
// of both operand images (for the sum of all scores).  64 rows per workgroup: wave q adds the slab rows g = q, q + 4, ...
// (independent loads, 8 in flight), the four partial results are combined in wave order.
constexpr int kFinRows = 64;
struct Fin1Args {
  const float* rs; const float* mb; const float* ma; const float* cs; const float* diag_raw;
  int n_groups, n_chunks, R, KS;
  int64_t Rp;
  float kexp, unscale, shift;
  int unit, want_rank;
  float* rowsum; float* colsum; float* inv_row; float* inv_col; float* diag; int32_t* row_rank;
  const void* a_rows; const void* b_rows;
  int fp8;                             // the rows images hold fp8 (tt_score_bf16.h) instead of bf16
  float* part;                         // [n_wg][4 + 2 * Dp]: l, hits, dsum, (pad), U[Dp], V[Dp]
};

__device__ __forceinline__ float slab_sum4(const float* __restrict__ slab, int64_t Rp, int n, int q, int i) {
  float s = 0.f;
  for (int g0 = q; g0 < n; g0 += 32) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = g0 + 4 * k < n ? slab[(int64_t)(g0 + 4 * k) * Rp + i] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  return s;
}
__device__ __forceinline__ float slab_max4(const float* __restrict__ slab, int64_t Rp, int n, int q, int i) {
  float s = kNegBig;
  for (int g0 = q; g0 < n; g0 += 32) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = g0 + 4 * k < n ? slab[(int64_t)(g0 + 4 * k) * Rp + i] : kNegBig;
#pragma unroll
    for (int k = 0; k < 8; ++k) s = fmaxf(s, v[k]);
  }
  return s;
}

__global__ __launch_bounds__(256) void score_sym_finish1_kernel(Fin1Args f) {
  const int t = threadIdx.x, lane = t & 63, q = t >> 6;
  const int i = blockIdx.x * kFinRows + lane;              // slab rows are padded: i < Rp always
  const int Dp = f.KS * 16, stride = 4 + 2 * Dp;
  float* part = f.part + (int64_t)blockIdx.x * stride;
  __shared__ float red[4][4][kFinRows];                    // [quantity][wave][row]
  red[0][q][lane] = slab_sum4(f.rs, f.Rp, f.n_groups, q, i);
  red[1][q][lane] = slab_sum4(f.cs, f.Rp, f.n_chunks, q, i);
  if (f.want_rank) {
    red[2][q][lane] = slab_max4(f.mb, f.Rp, f.n_groups, q, i);
    red[3][q][lane] = slab_max4(f.ma, f.Rp, f.n_groups, q, i);
  }
  // column sums of the images over this workgroup's 2 tiles.  bf16: chunk id = (k-step * 2 + half) * 32 + row, 8 values of
  // columns 16 ks + 8 half + j.  fp8: chunk id = ((k64-step * 2 + part) * 2 + half) * 32 + row, 16 values of columns
  // 64 s + 32 half + 16 part + j (as stored: the factor 64 is taken out again below).
  __shared__ float cols[2][256];
  if (!f.fp8) {
    const int chunks = f.KS * 64;                          // per tile
    for (int img = 0; img < 2; ++img) {
      const __bf16* base = reinterpret_cast<const __bf16*>(img ? f.b_rows : f.a_rows);
      for (int ch0 = 0; ch0 < chunks; ch0 += 256) {
        const int ch = ch0 + t;
        float acc8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc8[j] = 0.f;
        if (ch < chunks) {
#pragma unroll
          for (int qq = 0; qq < kFinRows / 32; ++qq) {
            const int64_t tile = (int64_t)blockIdx.x * (kFinRows / 32) + qq;
            if (tile * 32 < f.R) {                         // (rows beyond R inside a tile are zero in the image)
              const bf16x8 vv = *reinterpret_cast<const bf16x8*>(base + (tile * chunks + ch) * 8);
#pragma unroll
              for (int j = 0; j < 8; ++j) acc8[j] += (float)vv[j];
            }
          }
        }
        // the 32 threads of a (k-step, half) group hold the 32 rows: butterfly over the low 5 lane bits
#pragma unroll
        for (int o = 16; o > 0; o >>= 1)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc8[j] += __shfl_xor(acc8[j], o);
        if (ch < chunks && (t & 31) == 0) {
          const int dbase = (ch >> 5) * 8;                 // (k-step * 2 + half) * 8 = first column of the chunk
#pragma unroll
          for (int j = 0; j < 8; ++j) cols[img][dbase + j] = acc8[j];
        }
      }
    }
  } else {
    const int chunks = f.KS * 32;                          // 16-byte chunks per tile: Dp * 32 / 16
    for (int img = 0; img < 2; ++img) {
      const char* base = reinterpret_cast<const char*>(img ? f.b_rows : f.a_rows);
      for (int ch0 = 0; ch0 < chunks; ch0 += 256) {
        const int ch = ch0 + t;
        float acc16[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc16[j] = 0.f;
        if (ch < chunks) {
#pragma unroll
          for (int qq = 0; qq < kFinRows / 32; ++qq) {
            const int64_t tile = (int64_t)blockIdx.x * (kFinRows / 32) + qq;
            if (tile * 32 < f.R) {
              const i32x4 vv = *reinterpret_cast<const i32x4*>(base + (tile * chunks + ch) * 16);
#pragma unroll
              for (int w4 = 0; w4 < 4; ++w4) {
                acc16[4 * w4 + 0] += __builtin_amdgcn_cvt_f32_fp8(vv[w4], 0);
                acc16[4 * w4 + 1] += __builtin_amdgcn_cvt_f32_fp8(vv[w4], 1);
                acc16[4 * w4 + 2] += __builtin_amdgcn_cvt_f32_fp8(vv[w4], 2);
                acc16[4 * w4 + 3] += __builtin_amdgcn_cvt_f32_fp8(vv[w4], 3);
              }
            }
          }
        }
#pragma unroll
        for (int o = 16; o > 0; o >>= 1)
#pragma unroll
          for (int j = 0; j < 16; ++j) acc16[j] += __shfl_xor(acc16[j], o);
        if (ch < chunks && (t & 31) == 0) {
          const int g5 = ch >> 5;                          // (s * 2 + part) * 2 + half
          const int dbase = 64 * (g5 >> 2) + 32 * (g5 & 1) + 16 * ((g5 >> 1) & 1);
#pragma unroll
          for (int j = 0; j < 16; ++j) cols[img][dbase + j] = acc16[j] * (1.f / kFp8Up);
        }
      }
    }
  }
  __syncthreads();
  float l = 0.f, hit = 0.f, dsum = 0.f;
  if (q == 0) {
    if (i < f.R) {
      const float rsum = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
      const float csum = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
      const float draw = f.diag_raw[i], d = draw * f.unscale;
      f.inv_row[i] = 1.f / rsum;
      f.inv_col[i] = 1.f / csum;
      const float rs_shifted = f.unit ? rsum * f.kexp : rsum, cs_shifted = f.unit ? csum * f.kexp : csum;
      f.rowsum[i] = rs_shifted;
      f.colsum[i] = cs_shifted;
      f.diag[i] = d;
      l = (logf(rs_shifted) + f.shift - d) + (logf(cs_shifted) + f.shift - d);
      dsum = d;
      if (f.want_rank) {
        const float xb = fmaxf(fmaxf(red[2][0][lane], red[2][1][lane]), fmaxf(red[2][2][lane], red[2][3][lane]));
        const float xa = fmaxf(fmaxf(red[3][0][lane], red[3][1][lane]), fmaxf(red[3][2][lane], red[3][3][lane]));
        const int rk = (xb < draw && xa <= draw) ? 0 : 1;
        f.row_rank[i] = rk;
        hit = rk == 0 ? 1.f : 0.f;
      }
    } else {
      // rows past the end, up to this workgroup's 64: tt_score_bwd_bf16 reads the reciprocals a whole 32-row tile at a time
      f.inv_row[i] = 0.f; f.inv_col[i] = 0.f; f.rowsum[i] = 1.f; f.colsum[i] = 1.f;
    }
    float v[3] = {l, hit, dsum};                           // butterfly inside the one wave that holds the rows
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int j = 0; j < 3; ++j) v[j] += __shfl_xor(v[j], o);
    if (lane == 0) { part[0] = v[0]; part[1] = v[1]; part[2] = v[2]; part[3] = 0.f; }
  }
  for (int d = t; d < 2 * Dp; d += 256) part[4 + d] = cols[d / Dp][d % Dp];
}

__global__ __launch_bounds__(kRiderThreads) void score_sym_finish2_kernel(Finish2Rider fr) { finish2_body(fr); }

struct SymLayout {
  int nT, NI, n_groups, n_chunks, n_wg, Dp;
  int64_t Rp;
  size_t off_rs, off_mb, off_ma, off_cs, off_diag, off_part, off_part2, bytes;
};

inline SymLayout sym_layout(const tt_ctx* ctx, int64_t R, int D) {
  SymLayout L;
  L.Dp = padded_d(D);
  L.nT = (int)tt_cdiv(R, 32);
  L.n_groups = (int)tt_cdiv(L.nT, kSymWaves);
  // notice tiles per workgroup: enough workgroups for ~2 per CU, at least 4 tiles per sweep, at most 32 (LDS: 3 * 8 * NI * 128 B)
  int ni = (int)tt_cdiv((int64_t)L.nT * L.n_groups, 2 * (ctx ? ctx->num_cus : 256));
  ni = ni < 4 ? 4 : (ni > 16 ? 16 : ni);                  // 16 tiles: 48 KB of LDS slots per workgroup
  ni = (ni + 3) / 4 * 4;                                  // whole stages (a stage is 1, 2 or 4 tiles)
  L.NI = ni;
  L.n_chunks = (int)tt_cdiv(L.nT, ni);
  L.n_wg = (int)tt_cdiv(R, kFinRows);
  L.Rp = rup((int64_t)L.n_chunks * ni * 32, 256);          // slab rows cover every slot a workgroup writes (>= R)
  size_t o = 0;
  auto take = [&](size_t n) { size_t at = o; o += (n + 255) & ~size_t(255); return at; };
  L.off_rs = take(sizeof(float) * L.n_groups * L.Rp);
  L.off_mb = take(sizeof(float) * L.n_groups * L.Rp);
  L.off_ma = take(sizeof(float) * L.n_groups * L.Rp);
  L.off_cs = take(sizeof(float) * L.n_chunks * L.Rp);
  L.off_diag = take(sizeof(float) * L.Rp);
  L.off_part = take(sizeof(float) * L.n_wg * (4 + 2 * 256));   // (sized for the widest record: fp8 operands pad D up to 64)
  L.off_part2 = take(sizeof(float) * tt_cdiv(L.n_wg, kFin2Fold) * (4 + 2 * 256));
  L.bytes = o + 256;
  return L;
}

}  // namespace

static int sym_forward(tt_ctx* ctx, const void* N_packed, const void* C_packed, int64_t B, int32_t D, float inv_t, float shift,
                       float ab_scale, int32_t want_rank, float* rowsum, float* colsum, float* inv_row, float* inv_col, float* diag,
                       int32_t* row_rank, float* out8, float* loss_out, void* workspace, size_t workspace_bytes, tt_stream stream,
                       bool fp8, const char* who) {
  TT_CHECK_ARG(ctx && N_packed && C_packed && rowsum && colsum && inv_row && inv_col && diag && out8 && workspace, "%s: NULL argument", who);
  TT_CHECK_ARG(!want_rank || row_rank, "%s: want_rank needs row_rank", who);
  TT_CHECK_ARG(B >= 1 && B < ((int64_t)1 << 30) && D >= 1 && D <= 256, "%s: bad shape B=%lld D=%d", who, (long long)B, D);
  if (2.f * fabsf(inv_t) > 80.f) {
    tt_set_error("%s: 1/temperature = %g: fixed-shift softmax needs 2/T <= 80", who, inv_t);
    return TT_ERR_UNSUPPORTED;
  }
  tt_ctx sized = *ctx;
  sized.num_cus = 256;                                     // the layout must not depend on the device: it sizes the workspace
  SymLayout L = sym_layout(&sized, B, D);
  if (fp8) L.Dp = padded_d8(D);                            // (the part record is sized for 256 columns: sym_layout)
  if (workspace_bytes < L.bytes) {
    tt_set_error("%s: workspace %zu < required %zu", who, workspace_bytes, L.bytes);
    return TT_ERR_WORKSPACE;
  }
  char* ws = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
  const float ab = ab_scale == 0.f ? 1.f : ab_scale;
  const bool unit = ab == inv_t * kLog2e;                  // exactly: the caller got the scale from tt_score_unit_scale(inv_t)
  SymArgs g{};
  g.a_rows = fp8 ? static_cast<const void*>(view8(N_packed, B, D).rows8) : static_cast<const void*>(view(N_packed, B, D).rows);
  g.b_rows = fp8 ? static_cast<const void*>(view8(C_packed, B, D).rows8) : static_cast<const void*>(view(C_packed, B, D).rows);
  g.R = (int)B; g.nT = L.nT; g.NI = L.NI; g.Rp = L.Rp;
  g.c1 = inv_t * kLog2e / ab;
  g.c2 = -shift * kLog2e;
  g.rs = reinterpret_cast<float*>(ws + L.off_rs);
  g.mb = reinterpret_cast<float*>(ws + L.off_mb);
  g.ma = reinterpret_cast<float*>(ws + L.off_ma);
  g.cs = reinterpret_cast<float*>(ws + L.off_cs);
  g.diag_raw = reinterpret_cast<float*>(ws + L.off_diag);
  g.want_rank = want_rank ? 1 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  // fp8, D = 256: two company tiles per wave (half the workgroups along J, half the slab rows)
  const int jt = (fp8 && L.Dp == 256) ? 2 : 1;
  const int n_groups = (int)tt_cdiv(L.nT, kSymWaves * jt);
  const dim3 grid((unsigned)n_groups, (unsigned)L.n_chunks);
  const size_t slot_bytes = sizeof(float) * 3 * kSymWaves * L.NI * 32;
#define TT_SYM(KS, F8, JT_)                                                                                             \
  do {                                                                                                                  \
    const size_t lds = slot_bytes + 2 * SymStage<KS, F8>::kBytes;                                                       \
    if (unit) score_fwd_sym_kernel<KS, true, F8, JT_><<<grid, kSymThreads, lds, st>>>(g);                               \
    else score_fwd_sym_kernel<KS, false, F8, JT_><<<grid, kSymThreads, lds, st>>>(g);                                   \
  } while (0)
  if (fp8) {
    if (L.Dp == 64) TT_SYM(4, true, 1);
    else if (L.Dp == 128) TT_SYM(8, true, 1);
    else TT_SYM(16, true, 2);
  } else {
    if (L.Dp == 32) TT_SYM(2, false, 1);
    else if (L.Dp == 64) TT_SYM(4, false, 1);
    else if (L.Dp == 128) TT_SYM(8, false, 1);
    else TT_SYM(16, false, 1);
  }
#undef TT_SYM
  TT_LAUNCH_CHECK();
  Fin1Args f{};
  f.rs = g.rs; f.mb = g.mb; f.ma = g.ma; f.cs = g.cs; f.diag_raw = g.diag_raw;
  f.n_groups = n_groups; f.n_chunks = L.n_chunks; f.R = (int)B; f.KS = L.Dp / 16; f.Rp = L.Rp;
  f.kexp = exp2f(g.c2); f.unscale = inv_t / ab; f.shift = shift; f.unit = unit ? 1 : 0; f.want_rank = g.want_rank;
  f.rowsum = rowsum; f.colsum = colsum; f.inv_row = inv_row; f.inv_col = inv_col; f.diag = diag; f.row_rank = row_rank;
  f.a_rows = g.a_rows; f.b_rows = g.b_rows;
  f.fp8 = fp8 ? 1 : 0;
  f.part = reinterpret_cast<float*>(ws + L.off_part);
  score_sym_finish1_kernel<<<(unsigned)L.n_wg, 256, 0, st>>>(f);
  TT_LAUNCH_CHECK();
  Finish2Rider fr{f.part, L.n_wg, L.Dp, (float)B, f.unscale, out8, loss_out};
  if (L.n_wg >= 256) {                                   // large batches: fold the records 32 to one before the one-workgroup reduction
    float* part2 = reinterpret_cast<float*>(ws + L.off_part2);
    const int n2 = (int)tt_cdiv(L.n_wg, kFin2Fold);
    score_sym_fold_kernel<<<n2, 1024, 0, st>>>(f.part, L.n_wg, 4 + 2 * L.Dp, part2);
    TT_LAUNCH_CHECK();
    fr.part = part2;
    fr.n_wg = n2;
  }
  if (ctx->defer_riders & 2) {                           // rides beside the towers' tail_bwd (tt_riders.h): nothing on the device reads it
    if (ctx->riders->f_wg > 0)
      if (int rc = tt_riders_flush(ctx, st)) return rc;
    ctx->riders->f = fr;
    ctx->riders->f_wg = 1;
    return TT_OK;
  }
  score_sym_finish2_kernel<<<1, kRiderThreads, 0, st>>>(fr);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

extern "C" {

size_t tt_score_fwd_sym_workspace_bytes(int64_t B, int32_t D) {
  if (B < 1 || D < 1 || D > 256) return 0;
  tt_ctx fake{};
  fake.num_cus = 256;
  return sym_layout(&fake, B, D).bytes;
}

int tt_score_fwd_sym_bf16(tt_ctx* ctx, const void* N_packed, const void* C_packed, int64_t B, int32_t D, float inv_t, float shift,
                          float ab_scale, int32_t want_rank, float* rowsum, float* colsum, float* inv_row, float* inv_col,
                          float* diag, int32_t* row_rank, float* out8, float* loss_out, void* workspace, size_t workspace_bytes,
                          tt_stream stream) {
  return sym_forward(ctx, N_packed, C_packed, B, D, inv_t, shift, ab_scale, want_rank, rowsum, colsum, inv_row, inv_col, diag, row_rank,
                     out8, loss_out, workspace, workspace_bytes, stream, false, "tt_score_fwd_sym_bf16");
}

int tt_score_fwd_sym_fp8(tt_ctx* ctx, const void* N_packed, const void* C_packed, int64_t B, int32_t D, float inv_t, float shift,
                         float ab_scale, int32_t want_rank, float* rowsum, float* colsum, float* inv_row, float* inv_col,
                         float* diag, int32_t* row_rank, float* out8, float* loss_out, void* workspace, size_t workspace_bytes,
                         tt_stream stream) {
  return sym_forward(ctx, N_packed, C_packed, B, D, inv_t, shift, ab_scale, want_rank, rowsum, colsum, inv_row, inv_col, diag, row_rank,
                     out8, loss_out, workspace, workspace_bytes, stream, true, "tt_score_fwd_sym_fp8");
}

}  // extern "C"
