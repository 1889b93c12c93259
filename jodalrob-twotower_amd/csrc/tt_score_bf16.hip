// bf16-MFMA fast path of the in-batch-negative score + symmetric softmax-CE (same maths as
// tt_score.hip, operands rounded to bf16, f32 accumulation on v_mfma_f32_32x32x16_bf16).
//
// Wave-level design (no LDS in the main loops):
//   * a wave owns 32*AT rows `a` of A and sweeps the rows `b` of Bm in 32-row tiles; the NW waves of a
//     workgroup share the same A rows and split the b tiles round-robin.
//   * the tile is computed TRANSPOSED, X = Bm_tile . A_tile^T, so an accumulator register holds
//     X[b = rowmap(reg, lane>>5)][a = lane&31]: every per-`a` quantity (1/sumexp_a, the diagonal score,
//     the running exp-sum and rank count) is ONE value per lane, and the accumulator registers of a tile,
//     converted pairwise to bf16, ARE the A operand of the second MFMA (X^T . Bm) -- no lane movement.
//   * Bm is read in two images written once per step by tt_score_pack_bf16, both in MFMA fragment order so
//     that every wave-instruction reads 1 KB contiguous (a row-major image made each load touch 32 cache
//     lines and left the kernels bound by L1 tag throughput): [tile][k-step][half][row][8] for the operands
//     of the first product, and [tile][k-step][half][d][8] whose 16-B chunks are exactly the B operand of
//     the second product in the k-permutation the accumulator registers impose.
//   * partial results of the NW waves are combined through LDS in a fixed tree order.
#include "tt_score_bf16.h"

#include <stdlib.h>

namespace {

using namespace ttscore;

// ---- pack ------------------------------------------------------------------------------------------
struct PackArgs { const float* X; int64_t R, Rp; __bf16* rows; __bf16* frag; float scale; };
struct PackBatch { PackArgs a[2]; };

__global__ __launch_bounds__(256) void pack_bf16_kernel(PackBatch batch, int D, int Dp) {
  const PackArgs& pa = batch.a[blockIdx.y];
  const float* __restrict__ X = pa.X;
  const int64_t R = pa.R, Rp = pa.Rp;
  __bf16* __restrict__ rows = pa.rows;
  __bf16* __restrict__ frag = pa.frag;
  const float sc = pa.scale;
  const int64_t nchunk = Rp * Dp / 8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < 2 * nchunk; c += stride) {
    bf16x8 v;
    if (c < nchunk) {                                   // k-fragment image [t][k-step][half][row in tile][8]
      const int ci = (int)(c & 31), hh = (int)((c >> 5) & 1);
      const int64_t q = c >> 6;
      const int ks = (int)(q % (Dp / 16));
      const int64_t row = 32 * (q / (Dp / 16)) + ci;
      const int d0 = 16 * ks + 8 * hh;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)((row < R && d0 + j < D) ? X[row * D + d0 + j] * sc : 0.f);
      *reinterpret_cast<bf16x8*>(rows + c * 8) = v;
    } else {                                            // fragment-ordered image [t][s][h][d][8]
      const int64_t f = c - nchunk;
      const int d = (int)(f % Dp);
      const int64_t rest = f / Dp;
      const int h = (int)(rest & 1), s = (int)((rest >> 1) & 1);
      const int64_t t = rest >> 2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t row = 32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
        v[j] = (__bf16)((row < R && d < D) ? X[row * D + d] * sc : 0.f);
      }
      *reinterpret_cast<bf16x8*>(frag + f * 8) = v;
    }
  }
}

struct DirFwd {
  const __bf16* a_rows;
  const __bf16* b_rows;
  int64_t Ra, Rb, off;
  float* sumexp;
  float* diag;
  int32_t* rank;
  float* sumscore;
  int32_t rank_mode;   // 0 none, 1 top-1 flag (rank = 0/1), 2 full rank
  float c1;            // exponent scale for this direction's products: inv_t * log2(e) / (scale the operand images carry)
  float unscale;       // product -> s / T
  float* inv_sumexp;   // optional out: 1 / (the sum as accumulated), the factor the backward kernel multiplies by
};
struct FwdArgs {
  DirFwd d[2];
  float c2, kexp;      // exponent offset (-shift * log2 e); UNIT kernels leave it out of the terms and scale the row sums by 2^c2
};

struct DirBwd {
  const __bf16* a_rows;
  const __bf16* b_rows;
  const __bf16* b_frag;
  int64_t Ra, Rb, off;
  const float* sumexp_a;
  const float* sumexp_b;
  float* dA;
  float c1;            // as DirFwd::c1
  float out_scale;     // scale / (the B image's scale)
  const float* inv_a;  // optional: DirFwd::inv_sumexp of the A rows / of the B rows (then no reciprocals in the tile loop)
  const float* inv_b;
  const char* b_frag8;  // fp8 packing only: the B rows' fp8 fragment image (score_bwd_rows8_kernel)
};
struct BwdArgs {
  DirBwd d[2];
  float c2, kexp;
  const float* d_loss;
  int D;
};

template <int KS, int AT>
__device__ __forceinline__ void mfma1(const bf16x8 (&bf)[KS], const bf16x8 (&ares)[AT][KS], f32x16 (&acc)[AT]) {
#pragma unroll
  for (int i = 0; i < AT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int i = 0; i < AT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[s], ares[i][s], acc[i], 0, 0, 0);
}

template <int KS, int AT>
__device__ __forceinline__ void gemm1(const __bf16* __restrict__ b_rows, int64_t t, int c, int h, const bf16x8 (&ares)[AT][KS],
                                      f32x16 (&acc)[AT]) {
  bf16x8 bf[KS];
  load_bfrag<KS>(b_rows, t, c, h, bf);
  mfma1<KS, AT>(bf, ares, acc);
}

// ---- forward ---------------------------------------------------------------------------------------
// UNIT: every direction's exponent scale is exactly 1 (one operand image was packed times inv_t * log2 e): the softmax
// term is exp2(acc) with no multiply-add in front -- one VALU op less per score in kernels that are bound by VALU issue --
// and the constant factor 2^c2 of the fixed shift goes onto the finished row sums (|acc| <= log2(e) / T <= 58 for the
// supported temperatures: no overflow without the shift).  Rank / diagonal comparisons are scale-free.
template <int KS, int AT, int NW, bool UNIT>
__global__ __launch_bounds__(NW * 64) void score_fwd_bf16_kernel(FwdArgs args) {
  constexpr int ROWS = 32 * AT;
  __shared__ float part_e[NW][ROWS];
  __shared__ float part_s[NW][ROWS];
  __shared__ int part_c[NW][ROWS];
  // the direction's fields are picked with scalar selects (indexing the by-value argument with blockIdx.y made
  // every field -- and with it the whole tile loop's control flow -- live in vector registers), and row counts /
  // positions are 32-bit: the loop counter, the tile classification and their branches then run on the SALU
  const bool d1 = blockIdx.y != 0;
  DirFwd dr;
  dr.a_rows = d1 ? args.d[1].a_rows : args.d[0].a_rows;
  dr.b_rows = d1 ? args.d[1].b_rows : args.d[0].b_rows;
  dr.sumexp = d1 ? args.d[1].sumexp : args.d[0].sumexp;
  dr.diag = d1 ? args.d[1].diag : args.d[0].diag;
  dr.rank = d1 ? args.d[1].rank : args.d[0].rank;
  dr.sumscore = d1 ? args.d[1].sumscore : args.d[0].sumscore;
  dr.rank_mode = d1 ? args.d[1].rank_mode : args.d[0].rank_mode;
  const float c1 = d1 ? args.d[1].c1 : args.d[0].c1, unscale = d1 ? args.d[1].unscale : args.d[0].unscale;
  float* const inv_out = d1 ? args.d[1].inv_sumexp : args.d[0].inv_sumexp;
  const float c2 = args.c2;
  auto ex = [&](float x) { return UNIT ? __builtin_amdgcn_exp2f(x) : __builtin_amdgcn_exp2f(__builtin_fmaf(x, c1, c2)); };
  const int Ra = (int)(d1 ? args.d[1].Ra : args.d[0].Ra), Rb = (int)(d1 ? args.d[1].Rb : args.d[0].Rb);
  const int off = (int)(d1 ? args.d[1].off : args.d[0].off);
  const int a0 = (int)blockIdx.x * ROWS;
  if (a0 >= Ra) return;
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nT = (Rb + 31) / 32;
  bf16x8 ares[AT][KS];
#pragma unroll
  for (int i = 0; i < AT; ++i) load_bfrag<KS>(dr.a_rows, a0 / 32 + i, c, h, ares[i]);
  int pos[AT];
  float dg[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i) { pos[i] = a0 + 32 * i + c + off; dg[i] = kNegBig; }
  const int posmin = a0 + off, posmax = a0 + ROWS - 1 + off;
  // the diagonal scores, taken from the MFMA result itself so that ties compare bit-for-bit
  if (posmax >= 0 && posmin < Rb) {
    const int td0 = posmin > 0 ? posmin / 32 : 0;
    const int td1 = (posmax / 32) < nT - 1 ? posmax / 32 : nT - 1;
    for (int t = td0; t <= td1; ++t) {
      f32x16 acc[AT];
      gemm1<KS, AT>(dr.b_rows, t, c, h, ares, acc);
#pragma unroll
      for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * t + rowmap(r, h) == pos[i]) dg[i] = acc[i][r];
    }
#pragma unroll
    for (int i = 0; i < AT; ++i) dg[i] = fmaxf(dg[i], __shfl_xor(dg[i], 32));
  }
  // per-lane accumulators of this wave's share of the b tiles
  float se[AT], ss[AT], mb[AT], ma[AT];      // exp-sum, score-sum, max before / after the positive (top-1 mode)
  int cnt[AT];                                // full-rank mode
#pragma unroll
  for (int i = 0; i < AT; ++i) { se[i] = 0.f; ss[i] = 0.f; cnt[i] = 0; mb[i] = kNegBig; ma[i] = kNegBig; }
  const int mode = dr.rank ? dr.rank_mode : 0;                 // wave-uniform
  const bool want_ss = dr.sumscore != nullptr;
  // (a two-accumulator variant that issues tile t+NW's MFMAs before the epilogue of tile t measured SLOWER:
  //  199 VGPRs, 72-78 us vs 63-65 us -- kept single-buffered)
  // tile classes by index alone (two scalar compares per tile; deriving them from row positions cost ~30 SALU instructions
  // per tile): t < t_before: full tiles entirely before the workgroup's positives; t_after <= t < n_full: entirely after
  const int n_full = Rb / 32;
  const int t_before = min(posmin > 0 ? posmin / 32 : 0, n_full);
  const int t_after = posmax >= 0 ? posmax / 32 + 1 : 0;
  auto epilogue = [&](const f32x16 (&acc)[AT], int t) {
    const int b_lo = 32 * t;
    const bool before = t < t_before, after = t >= t_after && t < n_full;
    const bool full_tile = t < n_full;
    if (before || after) {
#pragma unroll
      for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) se[i] += ex(acc[i][r]);
      if (want_ss) {
#pragma unroll
        for (int i = 0; i < AT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) ss[i] += acc[i][r];
      }
      if (mode == 1) {            // top-1 only: running maxima, one v_max3 per two elements
        if (before) {
#pragma unroll
          for (int i = 0; i < AT; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mb[i] = max3_asm(mb[i], acc[i][r], acc[i][r + 1]);
        } else {
#pragma unroll
          for (int i = 0; i < AT; ++i)
#pragma unroll
            for (int r = 0; r < 16; r += 2) ma[i] = max3_asm(ma[i], acc[i][r], acc[i][r + 1]);
        }
      } else if (mode == 2) {     // full rank: ties before the positive count, after it they do not
        if (before) {
#pragma unroll
          for (int i = 0; i < AT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) cnt[i] += acc[i][r] >= dg[i] ? 1 : 0;
        } else {
#pragma unroll
          for (int i = 0; i < AT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) cnt[i] += acc[i][r] > dg[i] ? 1 : 0;
        }
      }
    } else if (full_tile) {       // the tile(s) holding the positives: every b is a row, only before / after is per lane.
      // Selects, no per-element branches: one wave of every workgroup meets this tile, and the general form below (exec-mask
      // branches around every element, ~600 VALU instructions against 56 for a plain tile) made that wave the straggler.
#pragma unroll
      for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) se[i] += ex(acc[i][r]);
      if (want_ss) {
#pragma unroll
        for (int i = 0; i < AT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) ss[i] += acc[i][r];
      }
#pragma unroll
      for (int i = 0; i < AT; ++i) {
        const int q = pos[i] - b_lo - 4 * h;              // b < pos  <=>  (r & 3) + 8 * (r >> 2) < q
        if (mode == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2);
            const float x = acc[i][r];
            mb[i] = fmaxf(mb[i], k < q ? x : kNegBig);
            ma[i] = fmaxf(ma[i], k > q ? x : kNegBig);
          }
        } else if (mode == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2);
            const float x = acc[i][r];
            cnt[i] += ((k < q && x >= dg[i]) || (k > q && x > dg[i])) ? 1 : 0;
          }
        }
      }
    } else {                      // the ragged last tile (Rb not a multiple of 32): per-element care
#pragma unroll
      for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int b = b_lo + rowmap(r, h);
          const float x = acc[i][r];
          const bool valid = b < Rb, bef = valid && b < pos[i], aft = valid && b > pos[i];
          se[i] += valid ? ex(x) : 0.f;
          ss[i] += valid ? x : 0.f;
          mb[i] = bef ? fmaxf(mb[i], x) : mb[i];
          ma[i] = aft ? fmaxf(ma[i], x) : ma[i];
          cnt[i] += ((bef && x >= dg[i]) || (aft && x > dg[i])) ? 1 : 0;
        }
    }
  };
  // operand prefetch PFD tiles ahead (a ring of register buffers, the loop unrolled over it so no buffer is copied): a tile
  // is ~150 ns of work for a wave, an L2 hit several times that -- one tile of lookahead left the waves waiting on loads
  constexpr int PFD = KS <= 4 ? 3 : (KS <= 8 ? 2 : 1);
  bf16x8 bq[PFD][KS];
#pragma unroll
  for (int p = 0; p < PFD; ++p)
    if (wave + p * NW < nT) load_bfrag<KS>(dr.b_rows, wave + p * NW, c, h, bq[p]);
  for (int t0 = wave; t0 < nT; t0 += PFD * NW) {
#pragma unroll
    for (int p = 0; p < PFD; ++p) {
      const int t = t0 + p * NW;
      if (t < nT) {                                                         // wave-uniform
        f32x16 acc[AT];
        mfma1<KS, AT>(bq[p], ares, acc);
        if (t + PFD * NW < nT) load_bfrag<KS>(dr.b_rows, t + PFD * NW, c, h, bq[p]);    // refill behind the MFMAs that read it
        epilogue(acc, t);
      }
    }
  }
  __shared__ float part_mb[NW][ROWS], part_ma[NW][ROWS];
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    se[i] += __shfl_xor(se[i], 32);
    ss[i] += __shfl_xor(ss[i], 32);
    cnt[i] += __shfl_xor(cnt[i], 32);
    mb[i] = fmaxf(mb[i], __shfl_xor(mb[i], 32));
    ma[i] = fmaxf(ma[i], __shfl_xor(ma[i], 32));
    if (h == 0) {
      part_e[wave][32 * i + c] = se[i]; part_s[wave][32 * i + c] = ss[i]; part_c[wave][32 * i + c] = cnt[i];
      part_mb[wave][32 * i + c] = mb[i]; part_ma[wave][32 * i + c] = ma[i];
    }
  }
  __shared__ float dg_s[ROWS];
  if (wave == 0 && h == 0) {
#pragma unroll
    for (int i = 0; i < AT; ++i) dg_s[32 * i + c] = dg[i];
  }
  __syncthreads();
  if (threadIdx.x < ROWS) {
    const int a = a0 + threadIdx.x;
    if (a < Ra) {
      float e = 0.f, sc = 0.f, xb = kNegBig, xa = kNegBig;
      int k = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        e += part_e[w][threadIdx.x]; sc += part_s[w][threadIdx.x]; k += part_c[w][threadIdx.x];
        xb = fmaxf(xb, part_mb[w][threadIdx.x]); xa = fmaxf(xa, part_ma[w][threadIdx.x]);
      }
      dr.sumexp[a] = UNIT ? e * args.kexp : e;
      if (inv_out) inv_out[a] = 1.f / e;
      if (mode == 2) dr.rank[a] = k;
      else if (mode == 1) dr.rank[a] = (xb < dg_s[threadIdx.x] && xa <= dg_s[threadIdx.x]) ? 0 : 1;
      if (dr.sumscore) dr.sumscore[a] = sc * unscale;                            // products -> sum_b s_ab / T
    }
  }
  if (wave == 0 && h == 0 && dr.diag) {
#pragma unroll
    for (int i = 0; i < AT; ++i) {
      const int a = a0 + 32 * i + c;
      if (a < Ra) dr.diag[a] = dg[i] > -1.0e38f ? dg[i] * unscale : 0.f;
    }
  }
}

// ---- backward --------------------------------------------------------------------------------------
// One tile's worth of streamed operands: the b tile's fragments for the first product, its fragment-ordered image for the
// second, and the 16 softmax reciprocals (or exp-sums) of its rows this lane half needs.
template <int KS>
struct BwdTile {
  bf16x8 b[KS];
  bf16x8 bm[2][KS / 2];
  float4 iv[4];
};

template <int KS>
__device__ __forceinline__ void bwd_tile_load(BwdTile<KS>& T, const __bf16* __restrict__ b_rows, const __bf16* __restrict__ b_frag,
                                              const float* __restrict__ ivsrc, int t, int c, int h) {
  constexpr int Dp = KS * 16, DT = KS / 2;
  load_bfrag<KS>(b_rows, t, c, h, T.b);
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int d = 0; d < DT; ++d)
      T.bm[s][d] = *reinterpret_cast<const bf16x8*>(b_frag + (((((int64_t)t * 2 + s) * 2 + h) * Dp + 32 * d + c) * 8));
#pragma unroll
  for (int q = 0; q < 4; ++q) T.iv[q] = *reinterpret_cast<const float4*>(ivsrc + 32 * t + 4 * h + 8 * q);
}

// Structure of the tile loop: TWO operand buffers, the loop unrolled over them, every load issued unconditionally (past the
// wave's last tile: that tile again) and in a fixed order.  The first version prefetched under `if (t + NW < nT)` and loaded
// the reciprocals inside the tile body: hipcc then cannot count the loads in flight and drained them all (s_waitcnt
// vmcnt(0)) twice per tile -- the prefetch never overlapped anything and the kernel ran at a third of its issue rate.
// The per-row reciprocal arrays must be readable up to a multiple of 32 rows (tt_score_bwd_dir).
template <int KS, int AT, int NW, bool UNIT>
__global__ __launch_bounds__(NW * 64) void score_bwd_bf16_kernel(BwdArgs args) {
  constexpr int Dp = KS * 16, ROWS = 32 * AT, DT = KS / 2;
  __shared__ float red[(NW / 2) * ROWS * Dp];
  const bool d1 = blockIdx.y != 0;                    // scalar selects, 32-bit positions: see the forward kernel
  DirBwd dr;
  dr.a_rows = d1 ? args.d[1].a_rows : args.d[0].a_rows;
  dr.b_rows = d1 ? args.d[1].b_rows : args.d[0].b_rows;
  dr.b_frag = d1 ? args.d[1].b_frag : args.d[0].b_frag;
  dr.sumexp_a = d1 ? args.d[1].sumexp_a : args.d[0].sumexp_a;
  dr.sumexp_b = d1 ? args.d[1].sumexp_b : args.d[0].sumexp_b;
  dr.dA = d1 ? args.d[1].dA : args.d[0].dA;
  const float c1 = d1 ? args.d[1].c1 : args.d[0].c1, out_scale = d1 ? args.d[1].out_scale : args.d[0].out_scale;
  const float* const inv_a = d1 ? args.d[1].inv_a : args.d[0].inv_a;
  const float* const inv_b = d1 ? args.d[1].inv_b : args.d[0].inv_b;
  const float c2 = args.c2, kx = UNIT ? args.kexp : 1.f;     // without the forward's reciprocals: 1 / raw sum = 2^c2 / stored sum
  const int Ra = (int)(d1 ? args.d[1].Ra : args.d[0].Ra), Rb = (int)(d1 ? args.d[1].Rb : args.d[0].Rb);
  const int off = (int)(d1 ? args.d[1].off : args.d[0].off);
  const int a0 = (int)blockIdx.x * ROWS;
  if (a0 >= Ra) return;
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nT = (Rb + 31) / 32;
  bf16x8 ares[AT][KS];
  float ia[AT];
  int pos[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    load_bfrag<KS>(dr.a_rows, a0 / 32 + i, c, h, ares[i]);
    const int a = a0 + 32 * i + c;
    ia[i] = a < Ra ? (inv_a ? inv_a[a] : __builtin_amdgcn_rcpf(dr.sumexp_a[a]) * kx) : 0.f;
    pos[i] = a + off;
  }
  const int posmin = a0 + off, posmax = a0 + ROWS - 1 + off;
  f32x16 dacc[AT][DT];
#pragma unroll
  for (int i = 0; i < AT; ++i)
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[i][d][r] = 0.f;
  const bool have_inv = inv_b != nullptr;                      // wave-uniform
  const float* const ivsrc = have_inv ? inv_b : dr.sumexp_b;
  const int tlast = nT - 1;
  auto compute = [&](const BwdTile<KS>& T, int t) {
    const int b_lo = 32 * t;
    float ib[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) { ib[4 * q] = T.iv[q].x; ib[4 * q + 1] = T.iv[q].y; ib[4 * q + 2] = T.iv[q].z; ib[4 * q + 3] = T.iv[q].w; }
    if (!have_inv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ib[r] = __builtin_amdgcn_rcpf(ib[r]) * kx;
    }
    if (b_lo + 31 >= Rb) {                                     // the ragged last tile: rows past the end weigh nothing
#pragma unroll
      for (int r = 0; r < 16; ++r) ib[r] = b_lo + rowmap(r, h) < Rb ? ib[r] : 0.f;
    }
    const bool band = !(b_lo + 31 < posmin || b_lo > posmax);
#pragma unroll
    for (int i = 0; i < AT; ++i) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T.b[s], ares[i][s], acc, 0, 0, 0);
      float w[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        w[r] = (UNIT ? __builtin_amdgcn_exp2f(acc[r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(acc[r], c1, c2))) * (ia[i] + ib[r]);
      if (b_lo + 31 >= Rb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = b_lo + rowmap(r, h) < Rb ? w[r] : 0.f;
      }
      if (band) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (b_lo + rowmap(r, h) == pos[i]) w[r] -= 2.f;
      }
      bf16x8 wf[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[s][j] = (__bf16)w[8 * s + j];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int d = 0; d < DT; ++d) dacc[i][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], T.bm[s][d], dacc[i][d], 0, 0, 0);
    }
  };
  BwdTile<KS> T0, T1;
  bwd_tile_load<KS>(T0, dr.b_rows, dr.b_frag, ivsrc, min(wave, tlast), c, h);
  __builtin_amdgcn_sched_barrier(0);                           // (issue order pinned: see the forward kernels)
  for (int t = wave; t < nT; t += 2 * NW) {
    bwd_tile_load<KS>(T1, dr.b_rows, dr.b_frag, ivsrc, min(t + NW, tlast), c, h);
    compute(T0, t);
    bwd_tile_load<KS>(T0, dr.b_rows, dr.b_frag, ivsrc, min(t + 2 * NW, tlast), c, h);
    if (t + NW < nT) compute(T1, t + NW);
  }
  // fixed-order tree over the NW waves: upper half writes, lower half adds
#pragma unroll
  for (int half = NW / 2; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
      float* slab = red + (wave - half) * ROWS * Dp;
#pragma unroll
      for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) slab[(32 * i + rowmap(r, h)) * Dp + 32 * d + c] = dacc[i][d][r];
    }
    __syncthreads();
    if (wave < half) {
      const float* slab = red + wave * ROWS * Dp;
#pragma unroll
      for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) dacc[i][d][r] += slab[(32 * i + rowmap(r, h)) * Dp + 32 * d + c];
    }
    __syncthreads();
  }
  if (wave == 0) {
    const float g = args.d_loss[0] * out_scale;
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int a = a0 + 32 * i + rowmap(r, h);
          const int dd = 32 * d + c;
          if (a < Ra && dd < args.D) dr.dA[(int64_t)a * args.D + dd] = dacc[i][d][r] * g;
        }
  }
}

// ---- backward, b-split form with ONE streamed image (D <= 64) ---------------------------------------------------------
// Ablations of the kernel above (compile-time, tools/bench_score.py, B = 8192, D = 64): full 50.0 us; without the exp2 45.9;
// without the gradient MFMAs 35.7; without ANY MFMA or exp2 32.4 -- two thirds of the time is moving the opposite side's
// operands: each workgroup streams 2 MB (the rows image for S AND the fragment-ordered image of the same values for the
// gradient product) at the ~65 GB/s a CU draws from L2.  Here only the rows image is streamed: a wave parks the tile's row
// fragments (it has them in registers for the S product) in a wave-private LDS tile [32 b][64 d] and reads the gradient
// product's operand -- 8 b-values of one column d per lane, in the accumulator's row order -- back with the transposing
// LDS read ds_read_b64_tr_b16 (a 16-lane group reads a 4-row x 16-column block, lane i gets column i).  Half the L2
// bytes for 8 KB of LDS traffic per tile.  Same values, same order of operations: bit-identical.
// Result: 45.9 -> 44.4 us in the step.  The same ablations on this kernel: full 49.1; no exp2 44.9; no gradient MFMAs 30.8;
// nothing 23.4 -- streaming is no longer the largest term; per SIMD the gradient + S MFMAs (34 GFLOP for both directions: every
// direction recomputes S) need ~20 us of the matrix pipe, the softmax VALU work ~24 us, the loads ~20 us, and a wave runs the
// three one after the other.
template <int KS, int AT, bool UNIT>
__global__ __launch_bounds__(512) void score_bwd_tr_kernel(BwdArgs args) {
  constexpr int NW = 8, Dp = KS * 16, ROWS = 32 * AT, DT = KS / 2;
  constexpr int TLD = Dp + 8;                                   // LDS row of the parked tile: 144 B at D = 64 (conflict-free b128 writes)
  __shared__ float red[(NW / 2) * ROWS * Dp];
  __shared__ __attribute__((aligned(16))) __bf16 park[NW][32 * TLD];
  using s16x4 = __attribute__((ext_vector_type(4))) short;
  using s16x8 = __attribute__((ext_vector_type(8))) short;
  const bool d1 = blockIdx.y != 0;
  DirBwd dr;
  dr.a_rows = d1 ? args.d[1].a_rows : args.d[0].a_rows;
  dr.b_rows = d1 ? args.d[1].b_rows : args.d[0].b_rows;
  dr.sumexp_a = d1 ? args.d[1].sumexp_a : args.d[0].sumexp_a;
  dr.sumexp_b = d1 ? args.d[1].sumexp_b : args.d[0].sumexp_b;
  dr.dA = d1 ? args.d[1].dA : args.d[0].dA;
  const float c1 = d1 ? args.d[1].c1 : args.d[0].c1, out_scale = d1 ? args.d[1].out_scale : args.d[0].out_scale;
  const float* const inv_a = d1 ? args.d[1].inv_a : args.d[0].inv_a;
  const float* const inv_b = d1 ? args.d[1].inv_b : args.d[0].inv_b;
  const float c2 = args.c2, kx = UNIT ? args.kexp : 1.f;
  const int Ra = (int)(d1 ? args.d[1].Ra : args.d[0].Ra), Rb = (int)(d1 ? args.d[1].Rb : args.d[0].Rb);
  const int off = (int)(d1 ? args.d[1].off : args.d[0].off);
  const int a0 = (int)blockIdx.x * ROWS;
  if (a0 >= Ra) return;
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nT = (Rb + 31) / 32;
  bf16x8 ares[AT][KS];
  float ia[AT];
  int pos[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    load_bfrag<KS>(dr.a_rows, a0 / 32 + i, c, h, ares[i]);
    const int a = a0 + 32 * i + c;
    ia[i] = a < Ra ? (inv_a ? inv_a[a] : __builtin_amdgcn_rcpf(dr.sumexp_a[a]) * kx) : 0.f;
    pos[i] = a + off;
  }
  const int posmin = a0 + off, posmax = a0 + ROWS - 1 + off;
  f32x16 dacc[AT][DT];
#pragma unroll
  for (int i = 0; i < AT; ++i)
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[i][d][r] = 0.f;
  const bool have_inv = inv_b != nullptr;
  const float* const ivsrc = have_inv ? inv_b : dr.sumexp_b;
  const int tlast = nT - 1;
  __bf16* const tile = park[wave];
  // parked-tile addresses: write = row c, columns 16 s + 8 h; transposing read = block row q of this lane's 16-lane group,
  // columns 4 p of the group's 16 (lane 4 q + p supplies the address, lane i of the group receives column i)
  __bf16* const wr_at = tile + c * TLD + 8 * h;
  const int g16 = lane & 15, cg = (lane >> 4) & 1;
  const __bf16* const tr_at = tile + (4 * h + (g16 >> 2)) * TLD + 16 * cg + 4 * (g16 & 3);
  struct Tile { bf16x8 b[KS]; float4 iv[4]; };
  auto load = [&](Tile& T, int t) {
    load_bfrag<KS>(dr.b_rows, t, c, h, T.b);
#pragma unroll
    for (int q = 0; q < 4; ++q) T.iv[q] = *reinterpret_cast<const float4*>(ivsrc + 32 * t + 4 * h + 8 * q);
  };
  auto compute = [&](const Tile& T, int t) {
    const int b_lo = 32 * t;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) *reinterpret_cast<bf16x8*>(wr_at + 16 * s2) = T.b[s2];
    float ib[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) { ib[4 * q] = T.iv[q].x; ib[4 * q + 1] = T.iv[q].y; ib[4 * q + 2] = T.iv[q].z; ib[4 * q + 3] = T.iv[q].w; }
    if (!have_inv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ib[r] = __builtin_amdgcn_rcpf(ib[r]) * kx;
    }
    if (b_lo + 31 >= Rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ib[r] = b_lo + rowmap(r, h) < Rb ? ib[r] : 0.f;
    }
    const bool band = !(b_lo + 31 < posmin || b_lo > posmax);
    // the gradient product's operand: for k-step s2 and column block d, rows {16 s2 + 4 h + 0..3, 16 s2 + 8 + 4 h + 0..3}
    bf16x8 bm[2][DT];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(tr_at + (16 * s2) * TLD + 32 * d));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(tr_at + (16 * s2 + 8) * TLD + 32 * d));
        const s16x8 v{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        bm[s2][d] = __builtin_bit_cast(bf16x8, v);
      }
#pragma unroll
    for (int i = 0; i < AT; ++i) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T.b[s2], ares[i][s2], acc, 0, 0, 0);
      float w[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        w[r] = (UNIT ? __builtin_amdgcn_exp2f(acc[r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(acc[r], c1, c2))) * (ia[i] + ib[r]);
      if (b_lo + 31 >= Rb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = b_lo + rowmap(r, h) < Rb ? w[r] : 0.f;
      }
      if (band) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (b_lo + rowmap(r, h) == pos[i]) w[r] -= 2.f;
      }
      bf16x8 wf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[s2][j] = (__bf16)w[8 * s2 + j];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int d = 0; d < DT; ++d) dacc[i][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s2], bm[s2][d], dacc[i][d], 0, 0, 0);
    }
  };
  Tile T0, T1;
  load(T0, min(wave, tlast));
  __builtin_amdgcn_sched_barrier(0);
  for (int t = wave; t < nT; t += 2 * NW) {
    load(T1, min(t + NW, tlast));
    compute(T0, t);
    load(T0, min(t + 2 * NW, tlast));
    if (t + NW < nT) compute(T1, t + NW);
  }
#pragma unroll
  for (int half = NW / 2; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
      float* slab = red + (wave - half) * ROWS * Dp;
#pragma unroll
      for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) slab[(32 * i + rowmap(r, h)) * Dp + 32 * d + c] = dacc[i][d][r];
    }
    __syncthreads();
    if (wave < half) {
      const float* slab = red + wave * ROWS * Dp;
#pragma unroll
      for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) dacc[i][d][r] += slab[(32 * i + rowmap(r, h)) * Dp + 32 * d + c];
    }
    __syncthreads();
  }
  if (wave == 0) {
    const float g = args.d_loss[0] * out_scale;
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int a = a0 + 32 * i + rowmap(r, h);
          const int dd = 32 * d + c;
          if (a < Ra && dd < args.D) dr.dA[(int64_t)a * args.D + dd] = dacc[i][d][r] * g;
        }
  }
}

// ---- backward, large-batch form ---------------------------------------------------------------------
// When there are enough rows a for every SIMD to own its own (Ra / 64 x directions >= ~1024), nothing has to be split along b:
// a workgroup is 4 waves, ONE per SIMD with the whole 512-register file (amdgpu_waves_per_eu(1, 1)), each wave owns two
// 32-row tiles of a -- A fragments (2 x KS) and the dA accumulators (2 x Dp / 32 tiles of 16 registers: 256 at D = 256) stay
// in registers for the whole sweep -- and ALL four waves stream the same b tiles, staged ONCE per workgroup through LDS
// (rows image + fragment image + the 32 reciprocals of a tile: 32 KB at D = 256, double buffered).  The b-split form above
// reads those 32 KB once per WAVE: 268 GB of L2 traffic at B = 65536, D = 256, which is what bounded it (15 TB/s at 23 %
// MFMA busy); here it is 34 GB.  No cross-wave reduction at the end: a wave's rows are its own.
// Inside a wave the two a tiles give the matrix pipe independent work while the VALU forms the softmax weights: S(0) S(1)
// | E(0) beside S(1) | dA(0) | E(1) beside dA(0) | dA(1).
// FP8: the S products take fp8 operands (rows images in the layout of tt_score_bf16.h, K = 64 per MFMA at twice the bf16 rate,
// half the A-fragment registers and half the staged bytes); the second products stay bf16.
// AT = 2, NWV = 4: one wave per SIMD with the whole register file; AT = 1, NWV = 8: two waves per SIMD (256 registers each:
// the hardware then runs one wave's softmax weights beside the other's MFMAs).
template <int KS, bool UNIT, bool FP8, int AT, int NWV>
__global__ __launch_bounds__(NWV * 64) void score_bwd_rows_kernel(BwdArgs args) {
  constexpr int NTH = NWV * 64, DT = KS / 2, Dp = KS * 16, K64 = FP8 ? KS / 4 : 1;
  constexpr int kRowsB = FP8 ? KS * 512 : KS * 1024, kFragB = KS * 1024, kIvB = 256, kStageB = kRowsB + kFragB + kIvB;
  constexpr int kPieces = (kRowsB + kFragB) / 16, kPPT = (kPieces + NTH - 1) / NTH;   // 16-byte pieces per thread per stage (last one ragged)
  static_assert(!FP8 || KS % 4 == 0, "fp8 operands come in K = 64 steps");
  // bf16 operands at D = 256: the wave's own A fragments are 64 registers beside 128 of accumulators -- they live in LDS behind
  // the two stages ([wave][k-step][lane] 16-byte pieces: 8 KB per wave) and are read back four k-steps at a time
  constexpr bool ALDS = !FP8 && KS == 16;
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const bool d1 = blockIdx.y != 0;
  DirBwd dr;
  dr.a_rows = d1 ? args.d[1].a_rows : args.d[0].a_rows;
  dr.b_rows = d1 ? args.d[1].b_rows : args.d[0].b_rows;
  dr.b_frag = d1 ? args.d[1].b_frag : args.d[0].b_frag;
  dr.sumexp_a = d1 ? args.d[1].sumexp_a : args.d[0].sumexp_a;
  dr.sumexp_b = d1 ? args.d[1].sumexp_b : args.d[0].sumexp_b;
  dr.dA = d1 ? args.d[1].dA : args.d[0].dA;
  const float c1 = d1 ? args.d[1].c1 : args.d[0].c1, out_scale = d1 ? args.d[1].out_scale : args.d[0].out_scale;
  const float* const inv_a = d1 ? args.d[1].inv_a : args.d[0].inv_a;
  const float* const inv_b = d1 ? args.d[1].inv_b : args.d[0].inv_b;
  const float c2 = args.c2, kx = UNIT ? args.kexp : 1.f;
  const int Ra = (int)(d1 ? args.d[1].Ra : args.d[0].Ra), Rb = (int)(d1 ? args.d[1].Rb : args.d[0].Rb);
  const int off = (int)(d1 ? args.d[1].off : args.d[0].off);
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nT = (Rb + 31) / 32, nTa_img = (int)(rup(Ra, 64) / 32);
  const int at0 = ((int)blockIdx.x * NWV + wave) * AT;                      // this wave's first a tile
  if ((int)blockIdx.x * NWV * AT * 32 >= Ra) return;                        // (whole workgroup)
  bf16x8 ares[AT][(FP8 || ALDS) ? 1 : KS];
  i32x8 ares8[AT][K64];
  bf16x8* const a_lds = reinterpret_cast<bf16x8*>(lds_raw + 2 * kStageB) + (size_t)wave * AT * KS * 64 + lane;   // (ALDS) [a tile][k-step][lane]
  float ia[AT];
  int pos[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    if (FP8) load_f8frag<K64>(reinterpret_cast<const char*>(dr.a_rows), min(at0 + i, nTa_img - 1), c, h, ares8[i]);
    else if (ALDS) {
      const __bf16* pa = dr.a_rows + (((int64_t)min(at0 + i, nTa_img - 1) * KS * 2 + h) * 32 + c) * 8;
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) a_lds[(i * KS + s2) * 64] = *reinterpret_cast<const bf16x8*>(pa + s2 * 512);    // (wave-private: no barrier)
    } else load_bfrag<((FP8 || ALDS) ? 1 : KS)>(dr.a_rows, min(at0 + i, nTa_img - 1), c, h, ares[i]);
    const int a = 32 * (at0 + i) + c;
    ia[i] = a < Ra ? (inv_a ? inv_a[a] : __builtin_amdgcn_rcpf(dr.sumexp_a[a]) * kx) : 0.f;
    pos[i] = a + off;
  }
  const int posmin = 32 * at0 + off, posmax = 32 * (at0 + AT) - 1 + off;
  f32x16 dacc[AT][DT];
#pragma unroll
  for (int i = 0; i < AT; ++i)
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[i][d][r] = 0.f;
  const bool have_inv = inv_b != nullptr;
  const float* const ivsrc = have_inv ? inv_b : dr.sumexp_b;
  // stage loader: the tile's [rows image | fragment image | 32 reciprocals] (contiguous per tile in global memory) go STRAIGHT into
  // the LDS stage by LDS-DMA (global_load_lds_dwordx4: lane l of a wave writes 16 bytes at the wave's base + 16 l), piece
  // p = tid + NTH q at offset 16 p.  Round 3: the copy used to pass through registers (up to eight 16-byte pieces per thread,
  // loaded a tile ahead and stored before the barrier): at D = 256 those 16-20 registers were the difference between 256
  // registers and 8-36 dwords of scratch per lane in the tile loop.  The DMA of tile t + 1 is issued right behind the barrier that
  // frees its buffer and has the whole tile to land; a wave waits for its own pieces (vmcnt(0)) before the next barrier.
  const char* const g_rows = reinterpret_cast<const char*>(dr.b_rows);
  const char* const g_frag = reinterpret_cast<const char*>(dr.b_frag);
  auto stage_dma = [&](int tn, int buf) {
    char* const base = lds_raw + buf * kStageB;
#pragma unroll
    for (int q = 0; q < kPPT; ++q) {
      const int p = tid + NTH * q;
      if ((q + 1) * NTH <= kPieces || p < kPieces) {
        const char* src = p < kRowsB / 16 ? g_rows + (int64_t)tn * kRowsB + p * 16 : g_frag + (int64_t)tn * kFragB + (p - kRowsB / 16) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(base + (wave * 64 + NTH * q) * 16), 16, 0, 0);
      }
    }
    if (wave == 0 && lane < 8)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ivsrc + 32 * tn + 4 * lane),
                                       (__attribute__((address_space(3))) void*)(base + kRowsB + kFragB), 16, 0, 0);
  };
  stage_dma(0, 0);
  for (int t = 0; t < nT; ++t) {
    const int nb = (t + 1) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of tile t have landed ...
    __syncthreads();                                       // ... and everybody's; buffer nb is no longer read by anyone
    if (t + 1 < nT) stage_dma(t + 1, nb);
    const char* rb = lds_raw + (t & 1) * kStageB;
    const char* fb = rb + kRowsB;
    const float* ivp = reinterpret_cast<const float*>(fb + kFragB);
    const int b_lo = 32 * t;
    // S tiles of both a tiles: the b fragments are read once, four at a time
    f32x16 acc[AT];
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (FP8) {
#pragma unroll
      for (int s0 = 0; s0 < K64; s0 += 2) {
        i32x8 bf8[2];
#pragma unroll
        for (int j = 0; j < 2 && s0 + j < K64; ++j) {
          const char* q = rb + ((s0 + j) * 4 + h) * 512 + c * 16;
          const i32x4 lo = *reinterpret_cast<const i32x4*>(q);
          const i32x4 hi = *reinterpret_cast<const i32x4*>(q + 1024);
          bf8[j] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int j = 0; j < 2 && s0 + j < K64; ++j)
#pragma unroll
          for (int i = 0; i < AT; ++i) acc[i] = mfma_f8(bf8[j], ares8[i][s0 + j], acc[i]);
      }
    } else {
#pragma unroll
      for (int s0 = 0; s0 < (FP8 ? 1 : KS); s0 += 4) {
        bf16x8 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(rb + (((s0 + j) * 2 + h) * 32 + c) * 16);
        bf16x8 af[AT][4];
        if (ALDS) {
#pragma unroll
          for (int i = 0; i < AT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) af[i][j] = a_lds[(i * KS + s0 + j) * 64];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < AT; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], ALDS ? af[i][j] : ares[i][(FP8 || ALDS) ? 0 : s0 + j], acc[i], 0, 0, 0);
      }
    }
    float ib[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(ivp + 4 * h + 8 * q);
      ib[4 * q] = v.x; ib[4 * q + 1] = v.y; ib[4 * q + 2] = v.z; ib[4 * q + 3] = v.w;
    }
    if (!have_inv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ib[r] = __builtin_amdgcn_rcpf(ib[r]) * kx;
    }
    const bool ragged = b_lo + 31 >= Rb;
    if (ragged) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ib[r] = b_lo + rowmap(r, h) < Rb ? ib[r] : 0.f;
    }
    const bool band = !(b_lo + 31 < posmin || b_lo > posmax);
#pragma unroll
    for (int i = 0; i < AT; ++i) {
      float w[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        w[r] = (UNIT ? __builtin_amdgcn_exp2f(acc[i][r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(acc[i][r], c1, c2))) * (ia[i] + ib[r]);
      if (ragged) {
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = b_lo + rowmap(r, h) < Rb ? w[r] : 0.f;
      }
      if (band) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (b_lo + rowmap(r, h) == pos[i]) w[r] -= 2.f;
      }
      bf16x8 wf[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[s][j] = (__bf16)w[8 * s + j];
      // second product, software-pipelined over batches of BQ fragment reads: batch k + 1 is in flight while batch k's MFMAs
      // issue -- with one wave per SIMD nobody else covers an LDS round trip.  (BQ = 2 for the bf16 D = 256 form: its two a
      // tiles' A fragments alone are 128 registers, and batches of four left 20 bytes of scratch per lane)
      constexpr int BQ = (KS == 16 && !FP8) ? 2 : 4;
      constexpr int NB1 = (DT + BQ - 1) / BQ, NBATCH = 2 * NB1;
      bf16x8 bmq[2][BQ];
      auto bm_read = [&](int k, bf16x8 (&dst)[BQ]) {
        const int s = k / NB1, d0 = BQ * (k % NB1);
#pragma unroll
        for (int j = 0; j < BQ; ++j)
          if (d0 + j < DT) dst[j] = *reinterpret_cast<const bf16x8*>(fb + (((s * 2 + h) * Dp + 32 * (d0 + j) + c) * 16));
      };
      bm_read(0, bmq[0]);
#pragma unroll
      for (int k = 0; k < NBATCH; ++k) {
        if (k + 1 < NBATCH) bm_read(k + 1, bmq[(k + 1) & 1]);
        const int s = k / NB1, d0 = BQ * (k % NB1);
#pragma unroll
        for (int j = 0; j < BQ; ++j)
          if (d0 + j < DT) dacc[i][d0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s], bmq[k & 1][j], dacc[i][d0 + j], 0, 0, 0);
      }
    }
  }
  const float g = args.d_loss[0] * out_scale;
#pragma unroll
  for (int i = 0; i < AT; ++i)
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int a = 32 * (at0 + i) + rowmap(r, h);
        const int dd = 32 * d + c;
        if (a < Ra && dd < args.D) dr.dA[(int64_t)a * args.D + dd] = dacc[i][d][r] * g;
      }
}

// ---- backward, large-batch form, fp8 operands for BOTH products ------------------------------------------
// score_bwd_rows_kernel<.., FP8> halves the matrix-pipe time of the S products only; two thirds of that kernel's MFMA cycles
// are the gradient products dA += W B (K = the b rows), still bf16.  Here they run on v_mfma_scale_f32_32x32x64_f8f6f4 too:
// K = 64 b rows per instruction, so the sweep goes over PAIRS of 32-row tiles, the softmax weights of a pair are converted
// to e4m3 and the b rows come from a third image of the packing (fp8 fragment image: tt_score_bf16.h).
// What makes 3 mantissa bits usable for weights that span 2^-20 .. 1 along a row is the instruction's block scale.  A lane
// (row a = c, half h) of the first operand holds 32 bytes = here: 16 weights of the pair's first tile, then 16 of its second;
// the hardware's blocks are [bytes 0..15 of both halves] scaled by lane (c, 0)'s scale register and [bytes 16..31 of both
// halves] scaled by lane (c, 1)'s (probe: tools/probe/fp8_block_scale.hip) -- i.e. one block = row a x the 32 rows of ONE b
// tile, and its two lanes meet through v_permlane32_swap.  Per block: m = largest weight, scale = 2^(floor(log2 m) - 7)
// (m / scale in [128, 256): the top of e4m3's range, 448), conversion with v_cvt_scalef32_pk_fp8_f32 (divides by the scale's
// power of two, round-to-nearest-even: same probe), the scale's exponent goes into the MFMA: MX-style dynamic block scaling,
// no a-priori bound on the weights.  Weights more than 2^16 below their block's largest flush to zero -- at most 32 x 2^-17
// of that largest one per block.
// The diagonal's weight (e_aa (1/rowsum + 1/colsum) - 2, the one weight of a row that is O(1) and carries the positive
// pair's pull) never goes through e4m3: the lane that meets it keeps it in f32, zeroes it in the block, and the epilogue
// adds (w_aa - 2) * b[pos_a] from the bf16 fragment image.
// Oracle model of this arithmetic: oracle_np.score_ce_bwd(..., block_fp8=True).
// Measured and not kept: running the gradient products of the second wave of every SIMD one pair late (at the top of the
// next pair's interval, fragment image triple-buffered) so that one wave's weights phase faces the other's MFMAs -- bit-identical,
// configs[4] step 7.88 against 7.77 ms: as in the bf16 kernels the two pipes' times add up whatever the waves' phases
// (profiles/NOTES.md).
template <int KS, bool UNIT, int AT, int NWV>
__global__ __launch_bounds__(NWV * 64) void score_bwd_rows8_kernel(BwdArgs args) {
  constexpr int NTH = NWV * 64, DT = KS / 2, Dp = KS * 16, K64 = KS / 4, NF = 2;
  // LDS: [rows-image tiles of a PAIR x 2 | fp8 fragment image of a pair x NF | 64 reciprocals x 2 | A fragments (ALDS)]
  constexpr int kRowsB = KS * 1024, kFragB = KS * 1024, kIvB = 256;
  constexpr int kFragOff = 2 * kRowsB, kIvOff = kFragOff + NF * kFragB, kALdsOff = kIvOff + 2 * kIvB;
  constexpr int kPieces = (kRowsB + kFragB) / 16, kPPT = kPieces / NTH, kRowQ = kRowsB / 16 / NTH;
  static_assert(KS % 4 == 0 && kPieces % NTH == 0 && (kRowsB / 16) % NTH == 0, "fp8 operands come in K = 64 steps; whole 16-byte pieces per thread");
  // D = 256, two waves per SIMD (256 registers each): the wave's own A fragments (32 registers) live in LDS behind the two
  // stages, [wave][k64-step][part][lane] 16-byte pieces (8 KB per wave), and are read back beside the b fragments
  constexpr bool ALDS = KS == 16;
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const bool d1 = blockIdx.y != 0;
  const char* const a_rows = reinterpret_cast<const char*>(d1 ? args.d[1].a_rows : args.d[0].a_rows);
  const char* const g_rows = reinterpret_cast<const char*>(d1 ? args.d[1].b_rows : args.d[0].b_rows);
  const char* const g_frag = d1 ? args.d[1].b_frag8 : args.d[0].b_frag8;
  const __bf16* const b_frag16 = d1 ? args.d[1].b_frag : args.d[0].b_frag;
  const float* const sumexp_a = d1 ? args.d[1].sumexp_a : args.d[0].sumexp_a;
  const float* const sumexp_b = d1 ? args.d[1].sumexp_b : args.d[0].sumexp_b;
  float* const dA = d1 ? args.d[1].dA : args.d[0].dA;
  const float c1 = d1 ? args.d[1].c1 : args.d[0].c1, out_scale = d1 ? args.d[1].out_scale : args.d[0].out_scale;
  const float* const inv_a = d1 ? args.d[1].inv_a : args.d[0].inv_a;
  const float* const inv_b = d1 ? args.d[1].inv_b : args.d[0].inv_b;
  const float c2 = args.c2, kx = UNIT ? args.kexp : 1.f;
  const int Ra = (int)(d1 ? args.d[1].Ra : args.d[0].Ra), Rb = (int)(d1 ? args.d[1].Rb : args.d[0].Rb);
  const int off = (int)(d1 ? args.d[1].off : args.d[0].off);
  const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nP = (Rb + 63) / 64, nTa_img = (int)(rup(Ra, 64) / 32);
  const int at0 = ((int)blockIdx.x * NWV + wave) * AT;
  if ((int)blockIdx.x * NWV * AT * 32 >= Ra) return;                        // (whole workgroup)
  i32x8 ares8[AT][ALDS ? 1 : K64];
  i32x4* const a_lds = reinterpret_cast<i32x4*>(lds_raw + kALdsOff) + (size_t)wave * AT * K64 * 128 + lane;
  float ia[AT], wd[AT];
  int pos[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    if (ALDS) {
      const char* pa = a_rows + ((int64_t)min(at0 + i, nTa_img - 1) * K64 * 4 + h) * 512 + c * 16;
#pragma unroll
      for (int s2 = 0; s2 < K64; ++s2) {                                    // (wave-private: no barrier)
        a_lds[((i * K64 + s2) * 2) * 64] = *reinterpret_cast<const i32x4*>(pa + (s2 * 4) * 512);
        a_lds[((i * K64 + s2) * 2 + 1) * 64] = *reinterpret_cast<const i32x4*>(pa + (s2 * 4 + 2) * 512);
      }
    } else load_f8frag<(ALDS ? 1 : K64)>(a_rows, min(at0 + i, nTa_img - 1), c, h, ares8[i]);
    const int a = 32 * (at0 + i) + c;
    ia[i] = a < Ra ? (inv_a ? inv_a[a] : __builtin_amdgcn_rcpf(sumexp_a[a]) * kx) : 0.f;
    pos[i] = a + off;
    wd[i] = 0.f;                                                            // stays 0 in the lanes that never meet the diagonal
  }
  const int posmin = 32 * at0 + off, posmax = 32 * (at0 + AT) - 1 + off;
  f32x16 dacc[AT][DT];
#pragma unroll
  for (int i = 0; i < AT; ++i)
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[i][d][r] = 0.f;
  const bool have_inv = inv_b != nullptr;
  const float* const ivsrc = have_inv ? inv_b : sumexp_b;
  const int iv_last = (int)rup(Rb, 32) - 4;                                  // the per-row arrays are readable up to a multiple of 32 rows
  // stage = the pair's [two rows-image tiles | fp8 fragment image | 64 reciprocals], by LDS-DMA as in score_bwd_rows_kernel
  auto stage_dma = [&](int pn, int rbuf, int fbuf) {
#pragma unroll
    for (int q = 0; q < kPPT; ++q) {
      const char* src = q < kRowQ ? g_rows + (int64_t)pn * kRowsB + (tid + NTH * q) * 16 : g_frag + (int64_t)pn * kFragB + (tid + NTH * (q - kRowQ)) * 16;
      char* dst = q < kRowQ ? lds_raw + rbuf * kRowsB + (wave * 64 + NTH * q) * 16 : lds_raw + kFragOff + fbuf * kFragB + (wave * 64 + NTH * (q - kRowQ)) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    if (wave == 0 && lane < 16)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ivsrc + min(64 * pn + 4 * lane, iv_last)),
                                       (__attribute__((address_space(3))) void*)(lds_raw + kIvOff + rbuf * kIvB), 16, 0, 0);
  };
  i32x8 wA[AT];
  int e8[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    wA[i] = i32x8{0, 0, 0, 0, 0, 0, 0, 0};
    e8[i] = 0;
  }
  // gradient products of one pair: one MFMA per 32 columns of d and a tile, K = the pair's 64 b rows; fragment d + 1 is in
  // flight while fragment d's MFMAs issue
  auto grad = [&](const char* fb) {
    i32x8 bm8[2];
    auto bm_read = [&](int d, i32x8& dst) {
      const char* q = fb + (d * 4 + h) * 512 + c * 16;
      const i32x4 lo = *reinterpret_cast<const i32x4*>(q);
      const i32x4 hi = *reinterpret_cast<const i32x4*>(q + 1024);
      dst = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    };
    bm_read(0, bm8[0]);
#pragma unroll
    for (int d = 0; d < DT; ++d) {
      if (d + 1 < DT) bm_read(d + 1, bm8[(d + 1) & 1]);
#pragma unroll
      for (int i = 0; i < AT; ++i)
        dacc[i][d] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wA[i], bm8[d & 1], dacc[i][d], 0, 0, 0, e8[i], 0, kFp8ScaleE8M0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stage_dma(0, 0, 0);
  for (int p = 0; p < nP; ++p) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p + 1 < nP) stage_dma(p + 1, (p + 1) & 1, (p + 1) & 1);
    const char* rb = lds_raw + (p & 1) * kRowsB;
    const char* fb = lds_raw + kFragOff + (p & 1) * kFragB;
    const float* ivp = reinterpret_cast<const float*>(lds_raw + kIvOff + (p & 1) * kIvB);
    const int b_lo = 64 * p;
    f32x16 acc[AT][2];
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
    // S products, software-pipelined by hand: k-step s + 1's fragments are in flight while step s's MFMAs issue (left to
    // itself hipcc hoists all 24 reads of the phase in front of the first MFMA: 96 registers, and spills accumulators)
    i32x8 bf8[2][2], af8[2][AT];
    auto s_read = [&](int s, i32x8 (&bdst)[2], i32x8 (&adst)[AT]) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const char* q = rb + t * (kRowsB / 2) + (s * 4 + h) * 512 + c * 16;
        const i32x4 lo = *reinterpret_cast<const i32x4*>(q);
        const i32x4 hi = *reinterpret_cast<const i32x4*>(q + 1024);
        bdst[t] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
      if (ALDS) {
#pragma unroll
        for (int i = 0; i < AT; ++i) {
          const i32x4 lo = a_lds[((i * K64 + s) * 2) * 64], hi = a_lds[((i * K64 + s) * 2 + 1) * 64];
          adst[i] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      }
    };
    s_read(0, bf8[0], af8[0]);
#pragma unroll
    for (int s = 0; s < K64; ++s) {
      if (s + 1 < K64) s_read(s + 1, bf8[(s + 1) & 1], af8[(s + 1) & 1]);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < AT; ++i) acc[i][t] = mfma_f8(bf8[s & 1][t], ALDS ? af8[s & 1][i] : ares8[i][ALDS ? 0 : s], acc[i][t]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // (pins the MFMAs to this block: their results are first read behind the branches below, and the machine sinker otherwise
    // moves the whole chain there, behind all of the phase's reads)
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
      for (int t = 0; t < 2; ++t) asm volatile("" : "+v"(acc[i][t]));
    const bool ragged = b_lo + 63 >= Rb;
    const bool band = !(b_lo + 63 < posmin || b_lo > posmax);
    // softmax weights of the pair, in place in the S accumulators, and the largest one this lane holds of either tile
    float wmax[AT][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float ib[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(ivp + 32 * t + 4 * h + 8 * q);
        ib[4 * q] = v.x; ib[4 * q + 1] = v.y; ib[4 * q + 2] = v.z; ib[4 * q + 3] = v.w;
      }
      if (!have_inv) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ib[r] = __builtin_amdgcn_rcpf(ib[r]) * kx;
      }
#pragma unroll
      for (int i = 0; i < AT; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[i][t][r] = (UNIT ? __builtin_amdgcn_exp2f(acc[i][t][r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(acc[i][t][r], c1, c2))) * (ia[i] + ib[r]);
        if (ragged) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][t][r] = b_lo + 32 * t + rowmap(r, h) < Rb ? acc[i][t][r] : 0.f;
        }
        if (band) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (b_lo + 32 * t + rowmap(r, h) == pos[i]) {
              wd[i] = acc[i][t][r] - 2.f;
              acc[i][t][r] = 0.f;
            }
        }
        wmax[i][t] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r += 2) wmax[i][t] = max3_asm(wmax[i][t], acc[i][t][r], acc[i][t][r + 1]);
      }
    }
#pragma unroll
    for (int i = 0; i < AT; ++i) {
      // the block's two lanes: after the swap lane (c, 0) holds both halves' maxima of tile 0, lane (c, 1) those of tile 1 --
      // the tile whose scale the MFMA takes from that lane
      const auto mm = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, wmax[i][0]), __builtin_bit_cast(unsigned, wmax[i][1]), false, false);
      const unsigned m_own = max(mm[0], mm[1]);                            // (non-negative floats order like their bit patterns)
      // scale = 2^(floor(log2 m) - 7), floored at 2^-110 (an all-zero block converts to zeros at any scale)
      const unsigned sb_own = (max(m_own, 0x0C000000u) & 0x7F800000u) - (7u << 23);
      e8[i] = (int)(sb_own >> 23);
      // tile 0's scale in every lane / tile 1's.  (Elements taken out by constant index: indexed with the unrolled loop's
      // variable, sb[t], hipcc 7.2 folds both reads to element 0 and converts both tiles with tile 0's scale.)
      const auto sb = __builtin_amdgcn_permlane32_swap(sb_own, sb_own, false, false);
      const unsigned sb0 = sb[0], sb1 = sb[1];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float scale = __builtin_bit_cast(float, t ? sb1 : sb0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // (the conversion writes one 16-bit half of its destination and keeps the other.  Through the builtin the first of
          //  a word's two conversions needs a defined `old` value -- a v_mov per word, 8 of the pair's ~144 vector instructions;
          //  as inline asm its destination is write-only and the second conversion fills the other half)
          int u;
          asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "=v"(u) : "v"(acc[i][t][4 * q]), "v"(acc[i][t][4 * q + 1]), "v"(scale));
          asm("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3 op_sel:[0,0,0,1]" : "+v"(u) : "v"(acc[i][t][4 * q + 2]), "v"(acc[i][t][4 * q + 3]), "v"(scale));
          wA[i][4 * t + q] = u;
        }
      }
    }
    grad(fb);
  }
  // epilogue: + (w_aa - 2) b[pos_a] with the bf16 image's row (a lane holds the diagonal weight of row a = its column index c,
  // the accumulators are laid out by row: one cross-lane read per register)
  const float g = args.d_loss[0] * out_scale;
  const int nTb_img = (int)(rup(Rb, 64) / 32);
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    const float wd_row = wd[i] + __shfl_xor(wd[i], 32);                    // one of the two halves met it (or neither: 0)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int arow = rowmap(r, h);
      const float wv = __shfl(wd_row, arow);
      const int pb = 32 * (at0 + i) + arow + off;
      const int a = 32 * (at0 + i) + arow;
      const bool diag_ok = pb >= 0 && pb < 32 * nTb_img;
      const int pbc = diag_ok ? pb : 0;
      const int t = pbc >> 5, rr = pbc & 31, s = rr >> 4, r16 = rr & 15, hh = (r16 >> 2) & 1, j = ((r16 >> 3) << 2) | (r16 & 3);
      const __bf16* prow = b_frag16 + ((((int64_t)t * 2 + s) * 2 + hh) * Dp) * 8 + j;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const int dd = 32 * d + c;
        if (a < Ra && dd < args.D) {
          const float bv = diag_ok ? (float)prow[dd * 8] : 0.f;
          dA[(int64_t)a * args.D + dd] = __builtin_fmaf(wv, bv, dacc[i][d][r]) * g;
        }
      }
    }
  }
}

// ---- fp8 pack: [fp8 rows image | bf16 fragment image | fp8 fragment image] (tt_score_bf16.h) ----------------
// saturating: past e4m3's largest finite value the conversion would give NaN (64 * scale * x reaches 448 once |x| / T > 4.85,
// e.g. a row with one dominant coordinate at T = 0.2)
__device__ __forceinline__ float fp8_clamp(float v) { return __builtin_fminf(__builtin_fmaxf(v, -448.f), 448.f); }

__global__ __launch_bounds__(256) void pack_fp8_kernel(PackBatch batch, int D, int Dp) {
  const PackArgs& pa = batch.a[blockIdx.y];
  const float* __restrict__ X = pa.X;
  const int64_t R = pa.R, Rp = pa.Rp;
  char* __restrict__ rows8 = reinterpret_cast<char*>(pa.rows);
  __bf16* __restrict__ frag = pa.frag;
  const float sc = pa.scale;
  const int64_t n8 = Rp * Dp / 16, nfr = Rp * Dp / 8;      // 16-byte chunks of the rows image / of the bf16 fragment image
  char* __restrict__ frag8 = reinterpret_cast<char*>(frag) + Rp * Dp * 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int cpt = Dp * 2;                                   // fp8 chunks per 32-row tile
  for (int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ci < 2 * n8 + nfr; ci += stride) {
    if (ci >= n8 + nfr) {                                   // fp8 fragment image [P][d][part][h][c][16]
      const int64_t f = ci - n8 - nfr;
      const int64_t P = f / (Dp * 4);
      const int w = (int)(f - P * (Dp * 4)), c = w & 31, h = (w >> 5) & 1, part = (w >> 6) & 1, d = w >> 7;
      const int col = 32 * d + c;
      i32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t row = 64 * P + 32 * part + rowmap(4 * q + j, h);
          v[j] = (row < R && col < D) ? fp8_clamp(X[row * D + col] * sc * kFp8Up) : 0.f;
        }
        int u = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        u = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], u, true);
        o[q] = u;
      }
      *reinterpret_cast<i32x4*>(frag8 + f * 16) = o;
    } else if (ci < n8) {
      const int64_t t = ci / cpt;
      const int w = (int)(ci - t * cpt), row_in = w & 31, g5 = w >> 5;
      const int d0 = 64 * (g5 >> 2) + 32 * (g5 & 1) + 16 * ((g5 >> 1) & 1);
      const int64_t row = 32 * t + row_in;
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = (row < R && d0 + j < D) ? fp8_clamp(X[row * D + d0 + j] * sc * kFp8Up) : 0.f;
      i32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int u = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * q], v[4 * q + 1], 0, false);
        u = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * q + 2], v[4 * q + 3], u, true);
        o[q] = u;
      }
      *reinterpret_cast<i32x4*>(rows8 + ci * 16) = o;
    } else {                                               // fragment-ordered bf16 image [t][s][h][d][8], as pack_bf16_kernel
      const int64_t f = ci - n8;
      const int d = (int)(f % Dp);
      const int64_t rest = f / Dp;
      const int h = (int)(rest & 1), s = (int)((rest >> 1) & 1);
      const int64_t t = rest >> 2;
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t row = 32 * t + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
        v[j] = (__bf16)((row < R && d < D) ? X[row * D + d] * sc : 0.f);
      }
      *reinterpret_cast<bf16x8*>(frag + f * 8) = v;
    }
  }
}


// The same three images from ONE read of X: a workgroup stages 64 rows (a tile pair) in LDS -- coalesced, 16 bytes per lane -- and
// every thread then assembles 16-byte chunks of the images from there.  pack_fp8_kernel reads X once per image, the rows image
// with 64 different rows per wave-instruction (536 MB of traffic for 134 MB of input at B = 65536, D = 256: 179 us).  Same
// arithmetic per element: bit-identical images (tests compare the packed buffers with torch's conversion).
constexpr int kPackTileRows = 64, kPackPad = 4;
__global__ __launch_bounds__(256) void pack_fp8_tile_kernel(PackBatch batch, int D, int Dp) {
  extern __shared__ __attribute__((aligned(16))) float xt[];          // [64][Dp + 4]
  const PackArgs& pa = batch.a[blockIdx.y];
  const float* __restrict__ X = pa.X;
  const int64_t R = pa.R, Rp = pa.Rp;
  const int64_t P = blockIdx.x;
  if (P * kPackTileRows >= Rp) return;
  char* __restrict__ rows8 = reinterpret_cast<char*>(pa.rows);
  __bf16* __restrict__ frag = pa.frag;
  char* __restrict__ frag8 = reinterpret_cast<char*>(frag) + Rp * Dp * 2;
  const float sc = pa.scale;
  const int ld = Dp + kPackPad, tid = threadIdx.x;
  // stage: rows 64 P .. 64 P + 63, columns [0, Dp); zero outside [0, R) x [0, D)
  if (D == Dp && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
    const int per_row = Dp / 4;
    for (int e = tid; e < kPackTileRows * per_row; e += 256) {
      const int r = e / per_row, c4 = e - r * per_row;
      const int64_t row = P * kPackTileRows + r;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < R) v = reinterpret_cast<const float4*>(X + row * D)[c4];
      *reinterpret_cast<float4*>(xt + r * ld + 4 * c4) = v;
    }
  } else {
    for (int e = tid; e < kPackTileRows * Dp; e += 256) {
      const int r = e / Dp, c = e - r * Dp;
      const int64_t row = P * kPackTileRows + r;
      xt[r * ld + c] = (row < R && c < D) ? X[row * D + c] : 0.f;
    }
  }
  __syncthreads();
  // fp8 rows image: tiles 2 P, 2 P + 1; chunk w of a tile = (row w & 31, group w >> 5)
  const int cpt = Dp * 2;
  for (int e = tid; e < 2 * cpt; e += 256) {
    const int t = e / cpt, w = e - t * cpt, row_in = w & 31, g5 = w >> 5;
    const int d0 = 64 * (g5 >> 2) + 32 * (g5 & 1) + 16 * ((g5 >> 1) & 1);
    const float* src = xt + (32 * t + row_in) * ld + d0;
    i32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
      int u = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_clamp(v.x * sc * kFp8Up), fp8_clamp(v.y * sc * kFp8Up), 0, false);
      u = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_clamp(v.z * sc * kFp8Up), fp8_clamp(v.w * sc * kFp8Up), u, true);
      o[q] = u;
    }
    *reinterpret_cast<i32x4*>(rows8 + ((2 * P + t) * cpt + w) * 16) = o;
  }
  // bf16 fragment image [t][s][h][d][8]
  for (int e = tid; e < 8 * Dp; e += 256) {
    const int d = e % Dp, rest = e / Dp, h = rest & 1, s2 = (rest >> 1) & 1, t = rest >> 2;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)(xt[(32 * t + 16 * s2 + 8 * (j >> 2) + 4 * h + (j & 3)) * ld + d] * sc);
    *reinterpret_cast<bf16x8*>(frag + ((((2 * P + t) * 2 + s2) * 2 + h) * Dp + d) * 8) = v;
  }
  // fp8 fragment image [P][d][part][h][c][16]
  for (int e = tid; e < 4 * Dp; e += 256) {
    const int c = e & 31, h = (e >> 5) & 1, part = (e >> 6) & 1, d = e >> 7;
    const int col = 32 * d + c;
    i32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fp8_clamp(xt[(32 * part + rowmap(4 * q + j, h)) * ld + col] * sc * kFp8Up);
      int u = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
      u = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], u, true);
      o[q] = u;
    }
    *reinterpret_cast<i32x4*>(frag8 + (P * (Dp * 4) + e) * 16) = o;
  }
}

}  // namespace

extern "C" {

size_t tt_score_pack_bytes(int64_t R, int32_t D) {
  if (R < 0 || D < 1 || D > 256) return 0;
  return (size_t)(4 * rup(R > 0 ? R : 1, 64) * padded_d(D));
}

float tt_score_unit_scale(float inv_t) { return inv_t * kLog2e; }

int tt_score_pack2_bf16(tt_ctx* ctx, const float* X0, int64_t R0, void* packed0, const float* X1, int64_t R1, void* packed1,
                        int32_t D, float scale0, float scale1, tt_stream stream) {
  TT_CHECK_ARG(ctx && X0 && packed0, "tt_score_pack_bf16: NULL argument");
  TT_CHECK_ARG(R0 >= 1 && D >= 1 && (X1 == nullptr || (packed1 && R1 >= 1)), "tt_score_pack_bf16: bad shape");
  if (D > 256) {
    tt_set_error("tt_score_pack_bf16: D=%d > 256 not supported", D);
    return TT_ERR_UNSUPPORTED;
  }
  TT_CHECK_ARG(tt_aligned(packed0, 16) && tt_aligned(packed1, 16), "tt_score_pack_bf16: packed buffers must be 16-byte aligned");
  const int Dp = padded_d(D);
  PackBatch b{};
  const int n = X1 ? 2 : 1;
  int64_t maxchunks = 1;
  for (int i = 0; i < n; ++i) {
    const int64_t R = i ? R1 : R0, Rp = rup(R, 64);     // a workgroup reads up to 64 consecutive rows of its operand
    __bf16* base = reinterpret_cast<__bf16*>(i ? packed1 : packed0);
    const float sc = i ? scale1 : scale0;
    b.a[i] = PackArgs{i ? X1 : X0, R, Rp, base, base + Rp * Dp, sc == 0.f ? 1.f : sc};
    const int64_t chunks = 2 * Rp * Dp / 8;
    maxchunks = chunks > maxchunks ? chunks : maxchunks;
  }
  int64_t grid = tt_cdiv(maxchunks, 256);
  const int64_t cap = (int64_t)ctx->num_cus * 4;
  if (grid > cap) grid = cap;
  pack_bf16_kernel<<<dim3((unsigned)grid, (unsigned)n), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(b, D, Dp);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

int tt_score_pack_bf16(tt_ctx* ctx, const float* X, int64_t R, int32_t D, float scale, void* packed, tt_stream stream) {
  return tt_score_pack2_bf16(ctx, X, R, packed, nullptr, 0, nullptr, D, scale, 1.f, stream);
}

int tt_score_fwd_bf16(tt_ctx* ctx, const tt_score_fwd_dir* dirs, int32_t n_dirs, int32_t D, float inv_t, float shift,
                      tt_stream stream) {
  TT_CHECK_ARG(ctx && dirs && (n_dirs == 1 || n_dirs == 2), "tt_score_fwd_bf16: need 1 or 2 directions");
  TT_CHECK_ARG(D >= 1 && D <= 256, "tt_score_fwd_bf16: D=%d not in [1,256]", D);
  if (2.f * fabsf(inv_t) > 80.f) {
    tt_set_error("tt_score_fwd_bf16: 1/temperature = %g: fixed-shift softmax needs 2/T <= 80", inv_t);
    return TT_ERR_UNSUPPORTED;
  }
  FwdArgs a{};
  int64_t maxRa = 0;
  bool unit = true;
  for (int i = 0; i < 2; ++i) {
    const tt_score_fwd_dir& d = dirs[i < n_dirs ? i : 0];
    TT_CHECK_ARG(d.A_packed && d.B_packed && d.sumexp && d.Ra >= 1 && d.Rb >= 1, "tt_score_fwd_bf16: bad direction %d", i);
    const float ab = d.ab_scale == 0.f ? 1.f : d.ab_scale;
    a.d[i] = DirFwd{view(d.A_packed, d.Ra, D).rows, view(d.B_packed, d.Rb, D).rows, d.Ra, d.Rb, d.diag_offset, d.sumexp, d.diag, d.rank, d.sumscore,
                      d.rank ? (d.rank_mode == 1 ? 1 : 2) : 0, inv_t * kLog2e / ab, inv_t / ab, d.inv_sumexp};
    unit = unit && ab == inv_t * kLog2e;                // exactly: the caller got the scale from tt_score_unit_scale(inv_t)
    maxRa = d.Ra > maxRa ? d.Ra : maxRa;
  }
  a.c2 = -shift * kLog2e;
  a.kexp = exp2f(a.c2);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int Dp = padded_d(D);
#define TT_FWD(KS, AT, NW)                                                                                     \
  do {                                                                                                         \
    const dim3 grid((unsigned)tt_cdiv(maxRa, 32 * AT), (unsigned)n_dirs);                                      \
    if (unit) score_fwd_bf16_kernel<KS, AT, NW, true><<<grid, NW * 64, 0, st>>>(a);                            \
    else score_fwd_bf16_kernel<KS, AT, NW, false><<<grid, NW * 64, 0, st>>>(a);                                \
  } while (0)
  // D <= 64: 32 rows per wave (AT = 1), 8 waves, 2 workgroups per CU measured best (43.6 us; AT = 2: 48.0, 4 waves: 53.4,
  // 16 waves: 48.7 at B = 8192)
  if (Dp == 32) TT_FWD(2, 2, 8);
  else if (Dp == 64) TT_FWD(4, 1, 8);
  else if (Dp == 128) TT_FWD(8, 2, 8);
  else TT_FWD(16, 1, 8);
#undef TT_FWD
  TT_LAUNCH_CHECK();
  return TT_OK;
}

int tt_score_bwd_bf16(tt_ctx* ctx, const tt_score_bwd_dir* dirs, int32_t n_dirs, int32_t D, float inv_t, float shift,
                      const float* d_loss, float scale, tt_stream stream) {
  TT_CHECK_ARG(ctx && dirs && d_loss && (n_dirs == 1 || n_dirs == 2), "tt_score_bwd_bf16: need 1 or 2 directions");
  TT_CHECK_ARG(D >= 1 && D <= 256, "tt_score_bwd_bf16: D=%d not in [1,256]", D);
  BwdArgs a{};
  int64_t maxRa = 0;
  bool unit = true;
  for (int i = 0; i < 2; ++i) {
    const tt_score_bwd_dir& d = dirs[i < n_dirs ? i : 0];
    TT_CHECK_ARG(d.A_packed && d.B_packed && d.sumexp_a && d.sumexp_b && d.dA && d.Ra >= 1 && d.Rb >= 1,
                 "tt_score_bwd_bf16: bad direction %d", i);
    TT_CHECK_ARG(tt_aligned(d.sumexp_b, 16), "tt_score_bwd_bf16: sumexp_b must be 16-byte aligned");
    const PackedView vb = view(d.B_packed, d.Rb, D);
    const float ab = d.ab_scale == 0.f ? 1.f : d.ab_scale, bs = d.b_scale == 0.f ? 1.f : d.b_scale;
    a.d[i] = DirBwd{view(d.A_packed, d.Ra, D).rows, vb.rows, vb.frag, d.Ra, d.Rb, d.diag_offset, d.sumexp_a, d.sumexp_b, d.dA,
                      inv_t * kLog2e / ab, scale / bs, d.inv_a, d.inv_b};
    TT_CHECK_ARG(d.inv_b == nullptr || tt_aligned(d.inv_b, 16), "tt_score_bwd_bf16: inv_b must be 16-byte aligned");
    unit = unit && ab == inv_t * kLog2e;
    maxRa = d.Ra > maxRa ? d.Ra : maxRa;
  }
  a.c2 = -shift * kLog2e;
  a.kexp = exp2f(a.c2);
  a.d_loss = d_loss;
  a.D = D;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int Dp = padded_d(D);
#define TT_BWD(KS, AT, NW)                                                                                     \
  do {                                                                                                         \
    const dim3 grid((unsigned)tt_cdiv(maxRa, 32 * AT), (unsigned)n_dirs);                                      \
    if (unit) score_bwd_bf16_kernel<KS, AT, NW, true><<<grid, NW * 64, 0, st>>>(a);                            \
    else score_bwd_bf16_kernel<KS, AT, NW, false><<<grid, NW * 64, 0, st>>>(a);                                \
  } while (0)
  // enough rows for every SIMD to own 64 of them: the workgroup-staged form (no split along b, operands shared through LDS)
  if (maxRa >= ctx->score_bwd_rows_min && Dp >= 64) {
#define TT_BWD_ROWS(KS, AT_, NWV_)                                                                             \
  do {                                                                                                         \
    const dim3 grid((unsigned)tt_cdiv(maxRa, 32 * AT_ * NWV_), (unsigned)n_dirs);                              \
    const size_t lds = 2 * (size_t)(KS * 2048 + 256) + (KS == 16 ? (size_t)NWV_ * AT_ * KS * 1024 : 0);        \
    if (unit) {                                                                                                \
      TT_LDS_ONCE(lds, &score_bwd_rows_kernel<KS, true, false, AT_, NWV_>);                                    \
      score_bwd_rows_kernel<KS, true, false, AT_, NWV_><<<grid, NWV_ * 64, lds, st>>>(a);                      \
    } else {                                                                                                   \
      TT_LDS_ONCE(lds, &score_bwd_rows_kernel<KS, false, false, AT_, NWV_>);                                   \
      score_bwd_rows_kernel<KS, false, false, AT_, NWV_><<<grid, NWV_ * 64, lds, st>>>(a);                     \
    }                                                                                                          \
  } while (0)
    // D = 256: one a tile per wave with its A fragments in LDS (16 KB per wave: four waves beside the two 33-KB stages) -- with
    // two a tiles per wave the A fragments (128 registers) and the accumulators (256) filled the whole file and hipcc shuttled
    // hundreds of values between AGPRs, VGPRs and scratch (20-116 bytes of scratch per lane in the tile loop)
    if (Dp == 64) TT_BWD_ROWS(4, 2, 4);
    else if (Dp == 128) TT_BWD_ROWS(8, 2, 4);
    else TT_BWD_ROWS(16, 1, 4);
#undef TT_BWD_ROWS
    TT_LAUNCH_CHECK();
    return TT_OK;
  }
  if (Dp == 32) TT_BWD(2, 2, 8);
  else if (Dp == 64) {                                    // one streamed image, transposing LDS reads
    const dim3 grid((unsigned)tt_cdiv(maxRa, 64), (unsigned)n_dirs);
    if (unit) score_bwd_tr_kernel<4, 2, true><<<grid, 512, 0, st>>>(a);
    else score_bwd_tr_kernel<4, 2, false><<<grid, 512, 0, st>>>(a);
  } else if (Dp == 128) {                                 // the one-image form, one a tile per workgroup
    const dim3 grid((unsigned)tt_cdiv(maxRa, 32), (unsigned)n_dirs);
    if (unit) score_bwd_tr_kernel<8, 1, true><<<grid, 512, 0, st>>>(a);
    else score_bwd_tr_kernel<8, 1, false><<<grid, 512, 0, st>>>(a);
  }
  else TT_BWD(16, 1, 4);
#undef TT_BWD
  TT_LAUNCH_CHECK();
  return TT_OK;
}

size_t tt_score_pack_fp8_bytes(int64_t R, int32_t D) {
  if (R < 0 || D < 1 || D > 256) return 0;
  return (size_t)(4 * rup(R > 0 ? R : 1, 64) * padded_d8(D));
}

int tt_score_pack2_fp8(tt_ctx* ctx, const float* X0, int64_t R0, void* packed0, const float* X1, int64_t R1, void* packed1,
                       int32_t D, float scale0, float scale1, tt_stream stream) {
  TT_CHECK_ARG(ctx && X0 && packed0, "tt_score_pack2_fp8: NULL argument");
  TT_CHECK_ARG(R0 >= 1 && D >= 1 && D <= 256 && (X1 == nullptr || (packed1 && R1 >= 1)), "tt_score_pack2_fp8: bad shape");
  TT_CHECK_ARG(tt_aligned(packed0, 16) && tt_aligned(packed1, 16), "tt_score_pack2_fp8: packed buffers must be 16-byte aligned");
  const int Dp = padded_d8(D);
  PackBatch b{};
  const int n = X1 ? 2 : 1;
  int64_t maxchunks = 1;
  for (int i = 0; i < n; ++i) {
    const int64_t R = i ? R1 : R0, Rp = rup(R, 64);
    char* base = reinterpret_cast<char*>(i ? packed1 : packed0);
    const float sc = i ? scale1 : scale0;
    b.a[i] = PackArgs{i ? X1 : X0, R, Rp, reinterpret_cast<__bf16*>(base), reinterpret_cast<__bf16*>(base + Rp * Dp), sc == 0.f ? 1.f : sc};
    const int64_t chunks = 2 * (Rp * Dp / 16) + Rp * Dp / 8;
    maxchunks = chunks > maxchunks ? chunks : maxchunks;
  }
  int64_t maxRp = 0;
  for (int i = 0; i < n; ++i) maxRp = b.a[i].Rp > maxRp ? b.a[i].Rp : maxRp;
  if (maxRp >= 4096) {                                   // enough tile pairs to fill the chip: one read of X through LDS
    const size_t lds = (size_t)kPackTileRows * (Dp + kPackPad) * sizeof(float);
    TT_LDS_ONCE(lds, &pack_fp8_tile_kernel);
    pack_fp8_tile_kernel<<<dim3((unsigned)(maxRp / kPackTileRows), (unsigned)n), 256, lds, reinterpret_cast<hipStream_t>(stream)>>>(b, D, Dp);
    TT_LAUNCH_CHECK();
    return TT_OK;
  }
  int64_t grid = tt_cdiv(maxchunks, 256);
  const int64_t cap = (int64_t)ctx->num_cus * 4;
  if (grid > cap) grid = cap;
  pack_fp8_kernel<<<dim3((unsigned)grid, (unsigned)n), 256, 0, reinterpret_cast<hipStream_t>(stream)>>>(b, D, Dp);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

int tt_score_bwd_fp8(tt_ctx* ctx, const tt_score_bwd_dir* dirs, int32_t n_dirs, int32_t D, float inv_t, float shift,
                     const float* d_loss, float scale, tt_stream stream) {
  TT_CHECK_ARG(ctx && dirs && d_loss && (n_dirs == 1 || n_dirs == 2), "tt_score_bwd_fp8: need 1 or 2 directions");
  TT_CHECK_ARG(D >= 1 && D <= 256, "tt_score_bwd_fp8: D=%d not in [1,256]", D);
  BwdArgs a{};
  int64_t maxRa = 0;
  bool unit = true;
  for (int i = 0; i < 2; ++i) {
    const tt_score_bwd_dir& d = dirs[i < n_dirs ? i : 0];
    TT_CHECK_ARG(d.A_packed && d.B_packed && d.sumexp_a && d.sumexp_b && d.dA && d.Ra >= 1 && d.Rb >= 1, "tt_score_bwd_fp8: bad direction %d", i);
    TT_CHECK_ARG(tt_aligned(d.sumexp_b, 16) && (d.inv_b == nullptr || tt_aligned(d.inv_b, 16)), "tt_score_bwd_fp8: per-row arrays must be 16-byte aligned");
    const PackedView8 va = view8(d.A_packed, d.Ra, D), vb = view8(d.B_packed, d.Rb, D);
    const float ab = d.ab_scale == 0.f ? 1.f : d.ab_scale, bs = d.b_scale == 0.f ? 1.f : d.b_scale;
    a.d[i] = DirBwd{reinterpret_cast<const __bf16*>(va.rows8), reinterpret_cast<const __bf16*>(vb.rows8), vb.frag, d.Ra, d.Rb, d.diag_offset,
                    d.sumexp_a, d.sumexp_b, d.dA, inv_t * kLog2e / ab, scale / bs, d.inv_a, d.inv_b, vb.frag8};
    unit = unit && ab == inv_t * kLog2e;
    maxRa = d.Ra > maxRa ? d.Ra : maxRa;
  }
  a.c2 = -shift * kLog2e;
  a.kexp = exp2f(a.c2);
  a.d_loss = d_loss;
  a.D = D;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int Dp = padded_d8(D);
  // D = 256: two waves per SIMD, one a tile each (the 128 accumulator registers of a 32 x 256 block leave room for nothing
  // more); narrower: one wave per SIMD with two a tiles.
  if (ctx->fp8_grad) {                                     // TT_OPT_FP8_GRAD (default): e4m3 gradient products, block-scaled weights
#define TT_BWD88(KS, AT_, NWV_)                                                                                \
  do {                                                                                                         \
    const dim3 grid((unsigned)tt_cdiv(maxRa, 32 * AT_ * NWV_), (unsigned)n_dirs);                              \
    const size_t lds = (size_t)4 * KS * 1024 + 512 + (KS == 16 ? (size_t)NWV_ * AT_ * KS * 512 : 0);           \
    if (unit) {                                                                                                \
      TT_LDS_ONCE(lds, &score_bwd_rows8_kernel<KS, true, AT_, NWV_>);                                          \
      score_bwd_rows8_kernel<KS, true, AT_, NWV_><<<grid, NWV_ * 64, lds, st>>>(a);                            \
    } else {                                                                                                   \
      TT_LDS_ONCE(lds, &score_bwd_rows8_kernel<KS, false, AT_, NWV_>);                                         \
      score_bwd_rows8_kernel<KS, false, AT_, NWV_><<<grid, NWV_ * 64, lds, st>>>(a);                           \
    }                                                                                                          \
  } while (0)
    if (Dp == 64) TT_BWD88(4, 2, 4);
    else if (Dp == 128) TT_BWD88(8, 2, 4);
    else TT_BWD88(16, 1, 8);
#undef TT_BWD88
    TT_LAUNCH_CHECK();
    return TT_OK;
  }
#define TT_BWD8(KS, AT_, NWV_)                                                                                 \
  do {                                                                                                         \
    const dim3 grid((unsigned)tt_cdiv(maxRa, 32 * AT_ * NWV_), (unsigned)n_dirs);                              \
    const size_t lds = 2 * (size_t)(KS * 512 + KS * 1024 + 256);                                               \
    if (unit) {                                                                                                \
      TT_LDS_ONCE(lds, &score_bwd_rows_kernel<KS, true, true, AT_, NWV_>);                                     \
      score_bwd_rows_kernel<KS, true, true, AT_, NWV_><<<grid, NWV_ * 64, lds, st>>>(a);                       \
    } else {                                                                                                   \
      TT_LDS_ONCE(lds, &score_bwd_rows_kernel<KS, false, true, AT_, NWV_>);                                    \
      score_bwd_rows_kernel<KS, false, true, AT_, NWV_><<<grid, NWV_ * 64, lds, st>>>(a);                      \
    }                                                                                                          \
  } while (0)
  if (Dp == 64) TT_BWD8(4, 2, 4);
  else if (Dp == 128) TT_BWD8(8, 2, 4);
  else TT_BWD8(16, 1, 8);
#undef TT_BWD8
  TT_LAUNCH_CHECK();
  return TT_OK;
}

}  // extern "C"
