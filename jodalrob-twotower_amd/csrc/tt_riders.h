// "Riders": two small roles of the training step that depend on nothing the launches around them produce and are worth one launch
// each in the step's dependent chain (~5 us + a boundary) -- the compaction of the keyed duplicate-row plan (needs every sort
// workgroup's head count; first consumer: tt_embed_grad_bwd) and the last reduction of the symmetric score forward (loss + metrics:
// read by the host only).  With TT_OPT_DEFER_RIDERS set on the context, tt_dedup_plan_keyed* / tt_score_fwd_sym_* leave them queued
// here instead of launching them, and the towers' fused tail kernels (1024-thread workgroups, like both roles) run them in one
// extra grid row: the compaction beside tail_fwd, the loss reduction beside tail_bwd.  tt_flush_deferred, tt_embed_grad_bwd and the
// tt_adam_* entries launch whatever is still queued (towers whose shapes do not take the fused tail).  Bodies are shared between the
// stand-alone kernels and the host kernels, so results are bit-identical (test).
#pragma once
#include "tt_common.h"

constexpr int kRiderThreads = 1024;
// the long-row list of the segmented gradient reduction (rows with more than kPlanLongSeg slots are summed chunk by chunk), built
// by the plan's compaction instead of by the reduction itself: the reduction's row and chunk passes then run as ONE launch
constexpr int kPlanLongSeg = 64;
struct PlanLong {
  int32_t* counters;     // [0] chunks, [1] long rows (zeroed by keyed_sort_kernel, filled by the compaction)
  int32_t* long_row; int32_t* long_base; int32_t* chunk_lo; int32_t* chunk_hi;
};
struct CompactRider {
  const int32_t* uniq_stage; const int32_t* seg_stage; const int32_t* ucount; const int32_t* ubase; const int32_t* uend;
  PlanLong pl;
  int n_keys; int64_t M;
  int32_t* unique_rows; int32_t* seg_offsets; int32_t* n_unique;
};
struct Finish2Rider {
  const float* part; int n_wg, Dp; float fb, unscale; float* out; float* loss_out;
};
struct tt_riders {
  CompactRider c; int c_wg;        // c_wg > 0: queued, needs that many workgroups
  Finish2Rider f; int f_wg;
};

#ifdef __HIPCC__
// workgroup ki of the keyed plan's compaction (kRiderThreads threads)
__device__ __forceinline__ void compact_body(const CompactRider& cr, int ki) {
  const int32_t* __restrict__ uniq_stage = cr.uniq_stage;
  const int32_t* __restrict__ seg_stage = cr.seg_stage;
  const int32_t* __restrict__ ucount = cr.ucount;
  const int32_t* __restrict__ ubase = cr.ubase;
  const int32_t* __restrict__ uend = cr.uend;
  const PlanLong pl = cr.pl;
  const int n_keys = cr.n_keys;
  const int64_t M = cr.M;
  int32_t* __restrict__ unique_rows = cr.unique_rows;
  int32_t* __restrict__ seg_offsets = cr.seg_offsets;
  int32_t* __restrict__ n_unique = cr.n_unique;
  constexpr int kKeyedThreads = kRiderThreads;

  // head counts of all (key, share) pairs -- a few dozen to a few hundred -- added up in parallel (integer sums: any order)
  __shared__ int sred[2][kKeyedThreads / 64];
  int before = 0, all = 0;
  for (int q = threadIdx.x; q < n_keys; q += kKeyedThreads) {
    const int v = ucount[q];
    all += v;
    if (q < ki) before += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    before += __shfl_xor(before, o);
    all += __shfl_xor(all, o);
  }
  if ((threadIdx.x & 63) == 0) { sred[0][threadIdx.x >> 6] = before; sred[1][threadIdx.x >> 6] = all; }
  __syncthreads();
  before = 0; all = 0;
#pragma unroll
  for (int w = 0; w < kKeyedThreads / 64; ++w) { before += sred[0][w]; all += sred[1][w]; }
  const int U = ucount[ki];
  const int64_t gbase = ubase[ki];
  for (int u = threadIdx.x; u < U; u += kKeyedThreads) {
    unique_rows[before + u] = uniq_stage[gbase + u];
    const int32_t s0 = seg_stage[gbase + u];
    seg_offsets[before + u] = s0;
    if (pl.counters) {                                   // long rows -> chunk list (what seg_reduce_kernel registers otherwise)
      const int32_t s1 = u + 1 < U ? seg_stage[gbase + u + 1] : uend[ki];
      if (s1 - s0 > kPlanLongSeg) {
        const int32_t nch = (s1 - s0 + kPlanLongSeg - 1) / kPlanLongSeg;
        const int32_t cb = atomicAdd(&pl.counters[0], nch);
        const int32_t li = atomicAdd(&pl.counters[1], 1);
        pl.long_row[li] = before + u;
        pl.long_base[li] = cb;
        for (int32_t c = 0; c < nch; ++c) {
          pl.chunk_lo[cb + c] = s0 + c * kPlanLongSeg;
          pl.chunk_hi[cb + c] = min(s1, s0 + (c + 1) * kPlanLongSeg);
        }
      }
    }
  }
  if (ki == 0 && threadIdx.x == 0) {
    n_unique[0] = all;
    seg_offsets[all] = (int32_t)M;
  }
}

// the one workgroup (kRiderThreads threads) that adds the symmetric forward's partial records in a fixed order (thread (j, q):
// records q, q + 4, ... of entry j, the four partial sums in order); loss and metrics (out8 as tt_score_loss_finish)
__device__ __forceinline__ void finish2_body(const Finish2Rider& fr) {
  const float* __restrict__ part = fr.part;
  const int n_wg = fr.n_wg, Dp = fr.Dp;
  const float fb = fr.fb, unscale = fr.unscale;
  float* __restrict__ out = fr.out;
  float* __restrict__ loss_out = fr.loss_out;

  __shared__ float red[4][4 + 2 * 256];
  __shared__ float prod[256];
  const int t = threadIdx.x, q = t >> 8, j0 = t & 255, stride = 4 + 2 * Dp;
  for (int j = j0; j < stride; j += 256) {
    float s = 0.f;
    for (int w0 = q; w0 < n_wg; w0 += 32) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = w0 + 4 * k < n_wg ? part[(int64_t)(w0 + 4 * k) * stride + j] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    red[q][j] = s;
  }
  __syncthreads();
  if (t < 256) {
    float p = 0.f;
    if (t < Dp) {
      const float u = (red[0][4 + t] + red[1][4 + t]) + (red[2][4 + t] + red[3][4 + t]);
      const float v = (red[0][4 + Dp + t] + red[1][4 + Dp + t]) + (red[2][4 + Dp + t] + red[3][4 + Dp + t]);
      p = u * v;
    }
    prod[t] = p;
  }
  __syncthreads();
  if (t < 64) {
    float p = (prod[t] + prod[t + 64]) + (prod[t + 128] + prod[t + 192]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o);
    if (t == 0) {
      const float tot = p * unscale;                       // sum of all s_ab / T = (sum_a n_a) . (sum_b c_b) / T
      const float l = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
      const float hit = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
      const float dsum = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
      const float pos = dsum / fb;
      const float neg = (tot - dsum) / (fb * fb - fb);     // mean over the off-diagonal (nan for B == 1, as torch)
      out[0] = 0.5f * l / fb;
      out[1] = hit / fb;
      out[2] = pos;
      out[3] = neg;
      out[4] = pos - neg;
      out[5] = 0.f;                                        // column-direction top-1 rate: first-call diagnostic only (two-direction kernel)
      out[6] = tot;
      out[7] = 0.f;
      if (loss_out) loss_out[0] = out[0];
    }
  }
}
#endif

// launches whatever the context still holds (stand-alone kernels); defined in tt_ctx.hip
int tt_riders_flush(tt_ctx* ctx, hipStream_t st);
