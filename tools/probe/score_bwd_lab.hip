// Laboratory for the b-split score backward at B = 8192, D = 64 (the dominant kernel of the configs[1] step): the kernel
// under development with compile-time ablation flags, timed by HIP events over both directions, beside the shipping entry
// point of libtwotower_hip.so.  NOT product code: a kernel moves into csrc/tt_score_bf16.hip once it wins here.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ijodalrob-twotower_amd/csrc -o tools/probe/score_bwd_lab \
//           tools/probe/score_bwd_lab.hip -Ljodalrob-twotower_amd -ltwotower_hip -Wl,-rpath,'$ORIGIN/../../jodalrob-twotower_amd'
#include "tt_score_bf16.h"

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <type_traits>
#include <vector>

using namespace ttscore;

struct DirBwd {
  const __bf16* a_rows;
  const __bf16* b_rows;
  int Ra, Rb, off;
  const float* inv_a;
  const float* inv_b;
  float* dA;
  float out_scale;
};
struct LabArgs { DirBwd d[2]; const float* d_loss; int D; unsigned long long* stamps; };

enum : int { F_NOEXP = 1, F_NOGRAD = 2, F_NOS = 4, F_NOLOAD = 8, F_NOPARK = 16, F_NOBAND = 32, F_ASM = 64, F_SPLITBAND = 128, F_SCHED = 256,
             F_PRIO = 512, F_INBAND = 1024 };

__device__ __forceinline__ float mul_asm(float a, float b) {
  float r;
  asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// RING register sets of fragments in flight; IVLDS: reciprocals of the b rows staged in LDS (else 4 global loads per tile)
template <int FLAGS, int RING, bool IVLDS>
__global__ __launch_bounds__(512) void lab_bwd(LabArgs args) {
  constexpr int KS = 4, AT = 2, NW = 8, Dp = 64, ROWS = 64, DT = 2, TLD = Dp + 8;
  constexpr int kParkB = NW * 32 * TLD * 2;
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  float* const red = reinterpret_cast<float*>(lds3);
  __bf16* const park0 = reinterpret_cast<__bf16*>(lds3);
  float* const ivl = reinterpret_cast<float*>(lds3 + kParkB);
  using s16x4 = __attribute__((ext_vector_type(4))) short;
  using s16x8 = __attribute__((ext_vector_type(8))) short;
  const DirBwd& dr = args.d[blockIdx.y != 0];
  const int Ra = dr.Ra, Rb = dr.Rb, off = dr.off;
  const int a0 = (int)blockIdx.x * ROWS;
  if (a0 >= Ra) return;
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nT = (Rb + 31) / 32, tlast = nT - 1;
  // (named register sets: an array of sets indexed in an unrolled loop ended up in scratch memory)
  bf16x8 F0[KS], F1[KS], F2[KS], F3[KS];
  load_bfrag<KS>(dr.b_rows, min(wave, tlast), c, h, F0);
  if (RING > 1) load_bfrag<KS>(dr.b_rows, min(wave + NW, tlast), c, h, F1);
  if (RING > 2) load_bfrag<KS>(dr.b_rows, min(wave + 2 * NW, tlast), c, h, F2);
  if (RING > 3) load_bfrag<KS>(dr.b_rows, min(wave + 3 * NW, tlast), c, h, F3);
  if (IVLDS)
    for (int b = threadIdx.x; b < 32 * nT; b += 512) ivl[b] = b < Rb ? dr.inv_b[b] : 0.f;
  bf16x8 ares[AT][KS];
  float ia[AT];
  int pos[AT];
#pragma unroll
  for (int i = 0; i < AT; ++i) {
    load_bfrag<KS>(dr.a_rows, a0 / 32 + i, c, h, ares[i]);
    const int a = a0 + 32 * i + c;
    ia[i] = a < Ra ? dr.inv_a[a] : 0.f;
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
  __bf16* const tile = park0 + wave * 32 * TLD;
  __bf16* const wr_at = tile + c * TLD + 8 * h;
  const int g16 = lane & 15, cg = (lane >> 4) & 1;
  const __bf16* const tr_at = tile + (4 * h + (g16 >> 2)) * TLD + 16 * cg + 4 * (g16 & 3);
  __syncthreads();
  const unsigned long long sc0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
  if ((FLAGS & F_PRIO) && wave >= 4) __builtin_amdgcn_s_setprio(1);

  auto body = [&](bf16x8 (&Fq)[KS], int t, int t_next, auto band_tag) {
    constexpr bool BAND = decltype(band_tag)::value;
    const int b_lo = 32 * t;
    const bool inband = (FLAGS & F_INBAND) && !(b_lo + 31 < posmin || b_lo > posmax);
    if (!(FLAGS & F_NOPARK)) {
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) *reinterpret_cast<bf16x8*>(wr_at + 16 * s2) = Fq[s2];
    }
    f32x16 acc[AT];
#pragma unroll
    for (int i = 0; i < AT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = (FLAGS & F_NOS) ? (float)Fq[r & 3][r & 7] : 0.f;
    if (!(FLAGS & F_NOS)) {
#pragma unroll
      for (int i = 0; i < AT; ++i)
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fq[s2], ares[i][s2], acc[i], 0, 0, 0);
    }
    bf16x8 keep[KS];
    if (FLAGS & F_NOPARK) {
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) keep[s2] = Fq[s2];
    }
    if (!(FLAGS & F_NOLOAD)) load_bfrag<KS>(dr.b_rows, t_next, c, h, Fq);
    float ib[16];
    if (IVLDS) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(ivl + b_lo + 4 * h + 8 * q);
        ib[4 * q] = v.x; ib[4 * q + 1] = v.y; ib[4 * q + 2] = v.z; ib[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(dr.inv_b + b_lo + 4 * h + 8 * q);
        ib[4 * q] = v.x; ib[4 * q + 1] = v.y; ib[4 * q + 2] = v.z; ib[4 * q + 3] = v.w;
      }
    }
    bf16x8 bm[2][DT];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        if (FLAGS & F_NOPARK) {
          bm[s2][d] = keep[2 * s2 + d];
        } else {
          const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tr_at + (16 * s2) * TLD + 32 * d));
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tr_at + (16 * s2 + 8) * TLD + 32 * d));
          const s16x8 v{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
          bm[s2][d] = __builtin_bit_cast(bf16x8, v);
        }
      }
#pragma unroll
    for (int i = 0; i < AT; ++i) {
      float w[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = (FLAGS & F_NOEXP) ? acc[i][r] : __builtin_amdgcn_exp2f(acc[i][r]);
        w[r] = (FLAGS & F_ASM) ? mul_asm(e, add_asm(ia[i], ib[r])) : e * (ia[i] + ib[r]);
      }
      if (BAND || inband) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (b_lo + rowmap(r, h) == pos[i]) w[r] -= 2.f;
      }
      bf16x8 wf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[s2][j] = (__bf16)w[8 * s2 + j];
      if (!(FLAGS & F_NOGRAD)) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int d = 0; d < DT; ++d) dacc[i][d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s2], bm[s2][d], dacc[i][d], 0, 0, 0);
      } else {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int j = 0; j < 8; ++j) dacc[i][s2][j] += (float)wf[s2][j] + (float)bm[s2][0][j];
      }
    }
    if (FLAGS & F_SCHED) {
      // one MFMA, then five vector instructions, sixteen times (the tile has 16 MFMAs and ~100 vector instructions)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
    }
  };
  auto compute = [&](bf16x8 (&Fq)[KS], int t, int t_next) {
    const int b_lo = 32 * t;
    const bool band = !(b_lo + 31 < posmin || b_lo > posmax);
    if (FLAGS & (F_NOBAND | F_INBAND)) body(Fq, t, t_next, std::false_type{});
    else if (FLAGS & F_SPLITBAND) { if (band) body(Fq, t, t_next, std::true_type{}); else body(Fq, t, t_next, std::false_type{}); }
    else body(Fq, t, t_next, std::true_type{});            // (per-element compare on every tile: reference point only)
  };
  int t = wave;
  for (; t + (RING - 1) * NW < nT; t += RING * NW) {
    compute(F0, t, min(t + RING * NW, tlast));
    if (RING > 1) compute(F1, t + NW, min(t + (RING + 1) * NW, tlast));
    if (RING > 2) compute(F2, t + 2 * NW, min(t + (RING + 2) * NW, tlast));
    if (RING > 3) compute(F3, t + 3 * NW, min(t + (RING + 3) * NW, tlast));
  }
  if (RING > 1 && t < nT) compute(F0, t, tlast);
  if (RING > 2 && t + NW < nT) compute(F1, t + NW, tlast);
  if (RING > 3 && t + 2 * NW < nT) compute(F2, t + 2 * NW, tlast);
  if (FLAGS & F_PRIO) __builtin_amdgcn_s_setprio(0);
  {
    const unsigned long long sc1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && args.stamps) {
      unsigned long long* q = args.stamps + 4 * ((blockIdx.y * gridDim.x + blockIdx.x) * NW + wave);
      q[0] = sc0; q[1] = sc1; q[2] = sr0; q[3] = sr1;
    }
  }
  __syncthreads();
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
    const float g = args.d_loss[0] * dr.out_scale;
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


// ---- software-pipelined form: S(u + 1), the softmax weights of unit u and the gradient product of unit u - 1 in ONE straight-line
// block per half step (unit = one a tile of one b tile), so that 8 MFMAs and ~100 vector instructions without mutual dependencies
// stand side by side; band tiles (the -2 on the diagonal) are swept first, in the plain form.
enum : int { P_SCHED = 1, P_ASM = 2, P_PRIO = 4 };
template <int PF>
__global__ __launch_bounds__(512) void lab_pipe(LabArgs args) {
  constexpr int KS = 4, NW = 8, Dp = 64, ROWS = 64, TLD = Dp + 8;
  constexpr int kParkB = NW * 32 * TLD * 2;
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  float* const red = reinterpret_cast<float*>(lds3);
  __bf16* const park0 = reinterpret_cast<__bf16*>(lds3);
  constexpr int kAresB = 2 * KS * 64 * 16;                                        // the workgroup's two a tiles, fragment order: 8 KB
  bf16x8* const aresl = reinterpret_cast<bf16x8*>(lds3 + kParkB);                 // [a tile][k-step][lane]
  float* const ivl = reinterpret_cast<float*>(lds3 + kParkB + kAresB);
  using s16x4 = __attribute__((ext_vector_type(4))) short;
  using s16x8 = __attribute__((ext_vector_type(8))) short;
  const DirBwd& dr = args.d[blockIdx.y != 0];
  const int Ra = dr.Ra, Rb = dr.Rb, off = dr.off;
  const int a0 = (int)blockIdx.x * ROWS;
  if (a0 >= Ra) return;
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nT = (Rb + 31) / 32, tlast = nT - 1;
  for (int b = threadIdx.x; b < 32 * nT; b += 512) ivl[b] = b < Rb ? dr.inv_b[b] : 0.f;
  {                                                          // the a tiles' fragments: one 16-byte piece per thread (2 x 4 x 64 pieces)
    const int i = threadIdx.x >> 8, s2 = (threadIdx.x >> 6) & 3;
    aresl[threadIdx.x] = *reinterpret_cast<const bf16x8*>(dr.a_rows + (((int64_t)(a0 / 32 + i) * KS * 2 + h) * 32 + c) * 8 + s2 * 512);
  }
  const bf16x8* const ares0 = aresl + lane, * const ares1 = aresl + KS * 64 + lane;       // k-step s2 at [64 * s2]
  const int aa0 = a0 + c, aa1 = a0 + 32 + c;
  const float ia0 = aa0 < Ra ? dr.inv_a[aa0] : 0.f, ia1 = aa1 < Ra ? dr.inv_a[aa1] : 0.f;
  const int pos0 = aa0 + off, pos1 = aa1 + off;
  const int posmin = a0 + off, posmax = a0 + ROWS - 1 + off;
  const int tb_lo = max(posmin, 0) / 32, tb_hi = min(posmax / 32, tlast);       // band tiles: at most three
  f32x16 d00, d01, d10, d11;                                                     // dacc[a tile][column block]
#pragma unroll
  for (int r = 0; r < 16; ++r) { d00[r] = 0.f; d01[r] = 0.f; d10[r] = 0.f; d11[r] = 0.f; }
  __bf16* const tile = park0 + wave * 32 * TLD;
  __bf16* const wr_at = tile + c * TLD + 8 * h;
  const int g16 = lane & 15, cg = (lane >> 4) & 1;
  const __bf16* const tr_at = tile + (4 * h + (g16 >> 2)) * TLD + 16 * cg + 4 * (g16 & 3);
  __syncthreads();
  auto park = [&](const bf16x8 (&Fq)[KS]) {
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) *reinterpret_cast<bf16x8*>(wr_at + 16 * s2) = Fq[s2];
  };
  auto trread = [&](bf16x8 (&bm)[4]) {                      // bm[2 * s2 + d]
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tr_at + (16 * s2) * TLD + 32 * d));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tr_at + (16 * s2 + 8) * TLD + 32 * d));
        const s16x8 v{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        bm[2 * s2 + d] = __builtin_bit_cast(bf16x8, v);
      }
  };
  auto readib = [&](float (&ib)[16], int t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(ivl + 32 * t + 4 * h + 8 * q);
      ib[4 * q] = v.x; ib[4 * q + 1] = v.y; ib[4 * q + 2] = v.z; ib[4 * q + 3] = v.w;
    }
  };
  auto smm = [&](const bf16x8 (&Fq)[KS], const bf16x8* ar) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fq[s2], ar[64 * s2], acc, 0, 0, 0);
    return acc;
  };
  auto weights = [&](const f32x16& acc, float ia, const float (&ib)[16], bf16x8 (&wf)[2], int b_lo, int pos, bool band) {
    float w[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __builtin_amdgcn_exp2f(acc[r]);
      w[r] = (PF & P_ASM) ? mul_asm(e, add_asm(ia, ib[r])) : e * (ia + ib[r]);
    }
    if (band) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (b_lo + rowmap(r, h) == pos) w[r] -= 2.f;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) wf[s2][j] = (__bf16)w[8 * s2 + j];
  };
  auto grad = [&](f32x16& dlo, f32x16& dhi, const bf16x8 (&wf)[2], const bf16x8 (&bm)[4]) {
    dlo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], bm[0], dlo, 0, 0, 0);
    dhi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], bm[1], dhi, 0, 0, 0);
    dlo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], bm[2], dlo, 0, 0, 0);
    dhi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], bm[3], dhi, 0, 0, 0);
  };
  const unsigned long long sc0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
  // ---- band tiles of this wave, plain form
  for (int t = tb_lo; t <= tb_hi; ++t) {
    if ((t & (NW - 1)) != wave) continue;
    bf16x8 Fb[KS], bm[4], wf[2];
    float ib[16];
    load_bfrag<KS>(dr.b_rows, t, c, h, Fb);
    park(Fb);
    readib(ib, t);
    trread(bm);
    const f32x16 s0 = smm(Fb, ares0), s1 = smm(Fb, ares1);
    weights(s0, ia0, ib, wf, 32 * t, pos0, true);
    grad(d00, d01, wf, bm);
    weights(s1, ia1, ib, wf, 32 * t, pos1, true);
    grad(d10, d11, wf, bm);
  }
  // ---- the other tiles, pipelined.  Tile sequence of this wave: wave, wave + 8, ... without the band tiles
  auto next_tile = [&](int t) {
    t += NW;
    if (t >= tb_lo && t <= tb_hi) t += NW;
    return t;
  };
  int t0 = wave;
  if (t0 >= tb_lo && t0 <= tb_hi) t0 += NW;
  int n_left = 0;
  for (int t = t0; t < nT; t = next_tile(t)) ++n_left;
  if (n_left > 0) {
    if (PF & P_PRIO) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
    bf16x8 FA[KS], FB[KS];
    int tA = t0, tB = next_tile(tA), tC = next_tile(tB);
    load_bfrag<KS>(dr.b_rows, min(tA, tlast), c, h, FA);
    load_bfrag<KS>(dr.b_rows, min(tB, tlast), c, h, FB);
    park(FA);
    f32x16 acc0 = smm(FA, ares0), acc1;
    bf16x8 wf0[2], wf1[2], bmp[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) { wf1[0][j] = (__bf16)0.f; wf1[1][j] = (__bf16)0.f; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) bmp[q][j] = (__bf16)0.f;
    // one tile per call: Fc holds tile tc (parked, its S for a tile 0 in acc0), Fn tile tn (in flight); Fc is refilled with tile tn2
    auto halfstep = [&](bf16x8 (&Fc)[KS], bf16x8 (&Fn)[KS], int tc, int tn2) {
      float ib[16];
      readib(ib, tc);
      acc1 = smm(Fc, ares1);
      load_bfrag<KS>(dr.b_rows, min(tn2, tlast), c, h, Fc);
      weights(acc0, ia0, ib, wf0, 0, 0, false);
      grad(d10, d11, wf1, bmp);                             // a tile 1 of the previous tile
      trread(bmp);                                          // this tile's operand (stays for the next call's product too)
      park(Fn);                                             // (behind the reads in the LDS queue)
      acc0 = smm(Fn, ares0);
      weights(acc1, ia1, ib, wf1, 0, 0, false);
      grad(d00, d01, wf0, bmp);
      if (PF & P_SCHED) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);
        }
      }
    };
    while (true) {
      halfstep(FA, FB, tA, tC);
      if (--n_left == 0) break;
      tA = tB; tB = tC; tC = next_tile(tC);                 // (roles: FB now current)
      halfstep(FB, FA, tA, tC);
      if (--n_left == 0) break;
      tA = tB; tB = tC; tC = next_tile(tC);
    }
    grad(d10, d11, wf1, bmp);
    if (PF & P_PRIO) __builtin_amdgcn_s_setprio(0);
  }
  {
    const unsigned long long sc1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && args.stamps) {
      unsigned long long* q = args.stamps + 4 * ((blockIdx.y * gridDim.x + blockIdx.x) * NW + wave);
      q[0] = sc0; q[1] = sc1; q[2] = sr0; q[3] = sr1;
    }
  }
  __syncthreads();
  f32x16* dd[4] = {&d00, &d01, &d10, &d11};
#pragma unroll
  for (int half = NW / 2; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
      float* slab = red + (wave - half) * ROWS * Dp;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(32 * (q >> 1) + rowmap(r, h)) * Dp + 32 * (q & 1) + c] = (*dd[q])[r];
    }
    __syncthreads();
    if (wave < half) {
      const float* slab = red + wave * ROWS * Dp;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) (*dd[q])[r] += slab[(32 * (q >> 1) + rowmap(r, h)) * Dp + 32 * (q & 1) + c];
    }
    __syncthreads();
  }
  if (wave == 0) {
    const float g = args.d_loss[0] * dr.out_scale;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int a = a0 + 32 * (q >> 1) + rowmap(r, h);
        const int dcol = 32 * (q & 1) + c;
        if (a < Ra && dcol < args.D) dr.dA[(int64_t)a * args.D + dcol] = (*dd[q])[r] * g;
      }
  }
}


// ---- hand-placed form: the software pipeline of lab_pipe with the instruction ORDER fixed in the source: sixteen slots per tile, each
// one MFMA followed by seven vector instructions and the slot's share of LDS / global traffic, a sched_barrier(0) behind every slot
// (hipcc keeps the order and still allocates registers and inserts the waits).  Per tile and wave:
//   slots 0-3   S(k, a tile 1)          | exp2 x 16 of a tile 0, first sums
//   slots 4-7   dA(k - 1, a tile 1)     | rest of a tile 0's weights -> wf0;            reload of the consumed fragments
//   slots 8-11  S(k + 1, a tile 0)      | exp2 x 16 of a tile 1 ...;                    transposing reads of tile k, park of tile k + 1
//   slots 12-15 dA(k, a tile 0)         | rest of a tile 1's weights -> wf1
#define SB() __builtin_amdgcn_sched_barrier(0)
using bf16x2v = __attribute__((ext_vector_type(2))) __bf16;
using f32x2v = __attribute__((ext_vector_type(2))) float;
struct WF { bf16x2v p[8]; };                               // 16 weights of one a tile as bf16 pairs: p[0..3] = k-step 0, p[4..7] = k-step 1
__device__ __forceinline__ bf16x8 wf_half(const WF& w, int s2) {
  struct Q { bf16x2v a, b, c, d; } q{w.p[4 * s2], w.p[4 * s2 + 1], w.p[4 * s2 + 2], w.p[4 * s2 + 3]};
  return __builtin_bit_cast(bf16x8, q);
}

template <int PF, int NW>
__global__ __launch_bounds__(NW * 64) void lab_hand(LabArgs args) {
  constexpr int KS = 4, Dp = 64, ROWS = 64, TLD = Dp + 8;
  constexpr int kParkB = NW * 32 * TLD * 2;
  extern __shared__ __attribute__((aligned(16))) char lds3[];
  float* const red = reinterpret_cast<float*>(lds3);
  __bf16* const park0 = reinterpret_cast<__bf16*>(lds3);
  float* const ivl = reinterpret_cast<float*>(lds3 + kParkB);
  using s16x4 = __attribute__((ext_vector_type(4))) short;
  using s16x8 = __attribute__((ext_vector_type(8))) short;
  const DirBwd& dr = args.d[blockIdx.y != 0];
  const int Ra = dr.Ra, Rb = dr.Rb, off = dr.off;
  const int a0 = (int)blockIdx.x * ROWS;
  if (a0 >= Ra) return;
  const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nT = (Rb + 31) / 32, tlast = nT - 1;
  for (int b = threadIdx.x; b < 32 * nT; b += NW * 64) ivl[b] = b < Rb ? dr.inv_b[b] : 0.f;
  bf16x8 ares0[KS], ares1[KS];
  load_bfrag<KS>(dr.a_rows, a0 / 32, c, h, ares0);
  load_bfrag<KS>(dr.a_rows, a0 / 32 + 1, c, h, ares1);
  const int aa0 = a0 + c, aa1 = a0 + 32 + c;
  const float ia0 = aa0 < Ra ? dr.inv_a[aa0] : 0.f, ia1 = aa1 < Ra ? dr.inv_a[aa1] : 0.f;
  const int pos0 = aa0 + off, pos1 = aa1 + off;
  const int posmin = a0 + off, posmax = a0 + ROWS - 1 + off;
  const int tb_lo = max(posmin, 0) / 32, tb_hi = min(posmax / 32, tlast);
  f32x16 d00, d01, d10, d11;
#pragma unroll
  for (int r = 0; r < 16; ++r) { d00[r] = 0.f; d01[r] = 0.f; d10[r] = 0.f; d11[r] = 0.f; }
  __bf16* const tile = park0 + wave * 32 * TLD;
  __bf16* const wr_at = tile + c * TLD + 8 * h;
  const int g16 = lane & 15, cg = (lane >> 4) & 1;
  const __bf16* const tr_at = tile + (4 * h + (g16 >> 2)) * TLD + 16 * cg + 4 * (g16 & 3);
  __syncthreads();
  const unsigned long long sc0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
  auto park = [&](const bf16x8 (&Fq)[KS]) {
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) *reinterpret_cast<bf16x8*>(wr_at + 16 * s2) = Fq[s2];
  };
  auto tr1 = [&](int s2, int d) {
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tr_at + (16 * s2) * TLD + 32 * d));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tr_at + (16 * s2 + 8) * TLD + 32 * d));
    const s16x8 v{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto ib4 = [&](int t, int q) { return *reinterpret_cast<const float4*>(ivl + 32 * t + 4 * h + 8 * q); };
  // ---- band tiles of this wave, plain form
  for (int t = tb_lo; t <= tb_hi; ++t) {
    if ((t % NW) != wave) continue;
    bf16x8 Fb[KS];
    load_bfrag<KS>(dr.b_rows, t, c, h, Fb);
    park(Fb);
    float ib[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float4 v = ib4(t, q); ib[4 * q] = v.x; ib[4 * q + 1] = v.y; ib[4 * q + 2] = v.z; ib[4 * q + 3] = v.w; }
    bf16x8 bm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bm[q] = tr1(q >> 1, q & 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fb[s2], i ? ares1[s2] : ares0[s2], acc, 0, 0, 0);
      float w[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        w[r] = __builtin_amdgcn_exp2f(acc[r]) * ((i ? ia1 : ia0) + ib[r]);
        if (32 * t + rowmap(r, h) == (i ? pos1 : pos0)) w[r] -= 2.f;
      }
      bf16x8 wf[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[s2][j] = (__bf16)w[8 * s2 + j];
      f32x16& dlo = i ? d10 : d00;
      f32x16& dhi = i ? d11 : d01;
      dlo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], bm[0], dlo, 0, 0, 0);
      dhi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], bm[1], dhi, 0, 0, 0);
      dlo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], bm[2], dlo, 0, 0, 0);
      dhi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], bm[3], dhi, 0, 0, 0);
    }
  }
  auto next_tile = [&](int t) {
    t += NW;
    if (t >= tb_lo && t <= tb_hi) t += NW;
    return t;
  };
  int t0 = wave;
  if (t0 >= tb_lo && t0 <= tb_hi) t0 += NW;
  int n_left = 0;
  for (int t = t0; t < nT; t = next_tile(t)) ++n_left;
  if (n_left > 0) {
    bf16x8 FA[KS], FB[KS];
    int tA = t0, tB = next_tile(tA), tC = next_tile(tB);
    load_bfrag<KS>(dr.b_rows, min(tA, tlast), c, h, FA);
    load_bfrag<KS>(dr.b_rows, min(tB, tlast), c, h, FB);
    park(FA);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[s2], ares0[s2], acc0, 0, 0, 0);
    WF wf0, wf1;
    bf16x8 bm[4];
#pragma unroll
    for (int q = 0; q < 8; ++q) { wf1.p[q][0] = (__bf16)0.f; wf1.p[q][1] = (__bf16)0.f; }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) bm[q][j] = (__bf16)0.f;
    auto halfstep = [&](bf16x8 (&Fc)[KS], bf16x8 (&Fn)[KS], int tc, int tn2) {
      float ib[16], e[16];
      const __bf16* src = dr.b_rows + (((int64_t)min(tn2, tlast) * KS * 2 + h) * 32 + c) * 8;
      // vector instruction j (0 .. 55) of one a tile's weights: 16 exp2, then per pair (r, r + 1): two sums, two products, one convert
#define TT_W_OP(j, ACC, IA, WFX)                                                                              \
      do {                                                                                                    \
        if (PF & 1) break;                                                                                    \
        if ((j) < 16) e[(j)] = (PF & 4) ? ACC[(j)] * 0.5f : __builtin_amdgcn_exp2f(ACC[(j)]);                                              \
        else {                                                                                                \
          constexpr int pr = ((j) - 16) / 5, st = ((j) - 16) % 5;                                             \
          if (st == 0) ib[2 * pr] = IA + ib[2 * pr];                                                          \
          else if (st == 1) ib[2 * pr + 1] = IA + ib[2 * pr + 1];                                             \
          else if (st == 2) e[2 * pr] = e[2 * pr] * ib[2 * pr];                                               \
          else if (st == 3) e[2 * pr + 1] = e[2 * pr + 1] * ib[2 * pr + 1];                                   \
          else WFX.p[pr] = __builtin_convertvector(f32x2v{e[2 * pr], e[2 * pr + 1]}, bf16x2v);                \
        }                                                                                                     \
      } while (0)
#define TT_W7(s, ACC, IA, WFX) TT_W_OP(7 * (s), ACC, IA, WFX); TT_W_OP(7 * (s) + 1, ACC, IA, WFX); TT_W_OP(7 * (s) + 2, ACC, IA, WFX); \
      TT_W_OP(7 * (s) + 3, ACC, IA, WFX); TT_W_OP(7 * (s) + 4, ACC, IA, WFX); TT_W_OP(7 * (s) + 5, ACC, IA, WFX); TT_W_OP(7 * (s) + 6, ACC, IA, WFX)
      // reciprocals of tile tc (first needed by vector instruction 16: slot 2)
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float4 v = ib4(tc, q); ib[4 * q] = v.x; ib[4 * q + 1] = v.y; ib[4 * q + 2] = v.z; ib[4 * q + 3] = v.w; }
      float ib1[16];                                         // a tile 1 adds its own reciprocal to the same 16 values: keep the originals
#pragma unroll
      for (int r = 0; r < 16; ++r) ib1[r] = ib[r];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
      // slots 0-3: S(k, 1)
      if (!(PF & 2)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fc[0], ares1[0], acc1, 0, 0, 0); TT_W7(0, acc0, ia0, wf0); SB();
      if (!(PF & 2)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fc[1], ares1[1], acc1, 0, 0, 0); TT_W7(1, acc0, ia0, wf0); SB();
      if (!(PF & 2)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fc[2], ares1[2], acc1, 0, 0, 0); TT_W7(2, acc0, ia0, wf0); SB();
      if (!(PF & 2)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fc[3], ares1[3], acc1, 0, 0, 0); TT_W7(3, acc0, ia0, wf0); SB();
      // slots 4-7: dA(k - 1, 1); the consumed fragment registers are refilled
      if (!(PF & 2)) d10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf1, 0), bm[0], d10, 0, 0, 0); TT_W7(4, acc0, ia0, wf0);
      Fc[0] = *reinterpret_cast<const bf16x8*>(src); SB();
      if (!(PF & 2)) d11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf1, 0), bm[1], d11, 0, 0, 0); TT_W7(5, acc0, ia0, wf0);
      Fc[1] = *reinterpret_cast<const bf16x8*>(src + 512); SB();
      if (!(PF & 2)) d10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf1, 1), bm[2], d10, 0, 0, 0); TT_W7(6, acc0, ia0, wf0);
      Fc[2] = *reinterpret_cast<const bf16x8*>(src + 1024); SB();
      if (!(PF & 2)) d11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf1, 1), bm[3], d11, 0, 0, 0); TT_W7(7, acc0, ia0, wf0);
      Fc[3] = *reinterpret_cast<const bf16x8*>(src + 1536); SB();
      // slots 8-11: S(k + 1, 0); tile k's gradient operand is read back, then tile k + 1 is parked (LDS queue keeps the order)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; ib[r] = ib1[r]; }
      if (!(PF & 2)) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fn[0], ares0[0], acc0, 0, 0, 0); TT_W7(0, acc1, ia1, wf1);
      bm[0] = tr1(0, 0); bm[1] = tr1(0, 1); SB();
      if (!(PF & 2)) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fn[1], ares0[1], acc0, 0, 0, 0); TT_W7(1, acc1, ia1, wf1);
      bm[2] = tr1(1, 0); bm[3] = tr1(1, 1); SB();
      if (!(PF & 2)) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fn[2], ares0[2], acc0, 0, 0, 0); TT_W7(2, acc1, ia1, wf1);
      *reinterpret_cast<bf16x8*>(wr_at) = Fn[0]; *reinterpret_cast<bf16x8*>(wr_at + 16) = Fn[1]; SB();
      if (!(PF & 2)) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fn[3], ares0[3], acc0, 0, 0, 0); TT_W7(3, acc1, ia1, wf1);
      *reinterpret_cast<bf16x8*>(wr_at + 32) = Fn[2]; *reinterpret_cast<bf16x8*>(wr_at + 48) = Fn[3]; SB();
      // slots 12-15: dA(k, 0)
      if (!(PF & 2)) d00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf0, 0), bm[0], d00, 0, 0, 0); TT_W7(4, acc1, ia1, wf1); SB();
      if (!(PF & 2)) d01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf0, 0), bm[1], d01, 0, 0, 0); TT_W7(5, acc1, ia1, wf1); SB();
      if (!(PF & 2)) d00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf0, 1), bm[2], d00, 0, 0, 0); TT_W7(6, acc1, ia1, wf1); SB();
      if (!(PF & 2)) d01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf0, 1), bm[3], d01, 0, 0, 0); TT_W7(7, acc1, ia1, wf1); SB();
#undef TT_W7
#undef TT_W_OP
    };
    while (true) {
      halfstep(FA, FB, tA, tC);
      if (--n_left == 0) break;
      tA = tB; tB = tC; tC = next_tile(tC);
      halfstep(FB, FA, tA, tC);
      if (--n_left == 0) break;
      tA = tB; tB = tC; tC = next_tile(tC);
    }
    d10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf1, 0), bm[0], d10, 0, 0, 0);
    d11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf1, 0), bm[1], d11, 0, 0, 0);
    d10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf1, 1), bm[2], d10, 0, 0, 0);
    d11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf_half(wf1, 1), bm[3], d11, 0, 0, 0);
  }
  {
    const unsigned long long sc1 = __builtin_amdgcn_s_memtime(), sr1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && args.stamps) {
      unsigned long long* q = args.stamps + 4 * ((blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave);
      q[0] = sc0; q[1] = sc1; q[2] = sr0; q[3] = sr1;
    }
  }
  __syncthreads();
  f32x16* dd[4] = {&d00, &d01, &d10, &d11};
#pragma unroll
  for (int half = NW / 2; half >= 1; half >>= 1) {
    if (wave >= half && wave < 2 * half) {
      float* slab = red + (wave - half) * ROWS * Dp;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[(32 * (q >> 1) + rowmap(r, h)) * Dp + 32 * (q & 1) + c] = (*dd[q])[r];
    }
    __syncthreads();
    if (wave < half) {
      const float* slab = red + wave * ROWS * Dp;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) (*dd[q])[r] += slab[(32 * (q >> 1) + rowmap(r, h)) * Dp + 32 * (q & 1) + c];
    }
    __syncthreads();
  }
  if (wave == 0) {
    const float g = args.d_loss[0] * dr.out_scale;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int a = a0 + 32 * (q >> 1) + rowmap(r, h);
        const int dcol = 32 * (q & 1) + c;
        if (a < Ra && dcol < args.D) dr.dA[(int64_t)a * args.D + dcol] = (*dd[q])[r] * g;
      }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static LabArgs g_args;
static float* g_ref[2];

static void report_stamps() {
  const int nw = 2 * 128 * 8;
  std::vector<unsigned long long> h(4 * nw);
  CK(hipMemcpy(h.data(), g_args.stamps, h.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, ns = 0, cmax = 0;
  unsigned long long r_lo = ~0ull, r_hi = 0;
  int cnt = 0;
  for (int w = 0; w < nw; ++w) {
    if (h[4 * w + 1] == 0) continue;
    ++cnt;
    const double c = (double)(h[4 * w + 1] - h[4 * w]), n = 10.0 * (double)(h[4 * w + 3] - h[4 * w + 2]);
    cyc += c; ns += n; cmax = c > cmax ? c : cmax;
    r_lo = h[4 * w + 2] < r_lo ? h[4 * w + 2] : r_lo; r_hi = h[4 * w + 3] > r_hi ? h[4 * w + 3] : r_hi;
  }
  printf("      sweep per wave: mean %.0f cycles = %.2f us (max %.0f cycles), in-kernel clock %.2f GHz; first start .. last end %.2f us\n", cyc / cnt,
         ns / cnt * 1e-3, cmax, cyc / ns, 10.0 * (double)(r_hi - r_lo) * 1e-3);
}

template <int FLAGS, int RING, bool IVLDS>
static void run(const char* name, bool check) {
  const int B = 8192;
  const size_t lds = 8 * 32 * 72 * 2 + (IVLDS ? (size_t)B * 4 : 0);
  const size_t need = lds > 65536 ? lds : 65536;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_bwd<FLAGS, RING, IVLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 grid(B / 64, 2);
  for (int i = 0; i < 3; ++i) lab_bwd<FLAGS, RING, IVLDS><<<grid, 512, need>>>(g_args);
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0.f;
  const int reps = 20;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0));
    lab_bwd<FLAGS, RING, IVLDS><<<grid, 512, need>>>(g_args);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best; sum += ms;
  }
  double err = -1.0;
  if (check) {
    std::vector<float> got((size_t)B * 64), ref((size_t)B * 64);
    err = 0.0;
    for (int d = 0; d < 2; ++d) {
      CK(hipMemcpy(got.data(), g_args.d[d].dA, got.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(ref.data(), g_ref[d], ref.size() * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < got.size(); ++i) err += got[i] != ref[i];
    }
  }
  printf("%-64s flags %4d ring %d ivlds %d : %7.2f us min  %7.2f us mean  (%s)\n", name, FLAGS, RING, (int)IVLDS, best * 1e3, sum / reps * 1e3,
         check ? (err == 0.0 ? "bit-identical to the library" : "DIFFERS from the library") : "ablation: not checked");
  report_stamps();
}

template <int PF, bool HAND = false, int NWV = 8>
static void run_pipe(const char* name) {
  const int B = 8192;
  const size_t need = NWV == 8 ? 8 * 32 * 72 * 2 + 8192 + (size_t)B * 4 : 98304;       // (4 waves: 96 KB so that a CU takes one workgroup)
  void (*kern)(LabArgs) = HAND ? &lab_hand<PF, NWV> : &lab_pipe<PF>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 grid(B / 64, 2);
  CK(hipMemset(g_args.stamps, 0, 4 * 2 * 128 * 8 * 8));
  for (int i = 0; i < 3; ++i) kern<<<grid, NWV * 64, need>>>(g_args);
  CK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0.f;
  const int reps = 20;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0));
    kern<<<grid, NWV * 64, need>>>(g_args);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best; sum += ms;
  }
  std::vector<float> got((size_t)B * 64), ref((size_t)B * 64);
  double num = 0, den = 0, ndiff = 0;
  for (int d = 0; d < 2; ++d) {
    CK(hipMemcpy(got.data(), g_args.d[d].dA, got.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ref.data(), g_ref[d], ref.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < got.size(); ++i) { num += ((double)got[i] - ref[i]) * ((double)got[i] - ref[i]); den += (double)ref[i] * ref[i]; ndiff += got[i] != ref[i]; }
  }
  printf("%-64s %s flags %d : %7.2f us min  %7.2f us mean  (rel. L2 difference to the library %.2e, %.0f elements differ)\n", name, HAND ? "hand" : "pipe", PF, best * 1e3,
         sum / reps * 1e3, sqrt(num / den), ndiff);
  report_stamps();
}

int main() {
  const int B = 8192, D = 64;
  tt_ctx* ctx = nullptr;
  if (tt_ctx_create(0, &ctx) != 0) { printf("ctx: %s\n", tt_last_error_string()); return 1; }
  std::vector<float> hn((size_t)B * D), hc((size_t)B * D), hinv(B);
  srand(7);
  auto unit_rows = [&](std::vector<float>& x) {
    for (int r = 0; r < B; ++r) {
      double n2 = 0;
      for (int d = 0; d < D; ++d) { float v = (float)rand() / RAND_MAX - 0.5f; x[(size_t)r * D + d] = v; n2 += (double)v * v; }
      for (int d = 0; d < D; ++d) x[(size_t)r * D + d] /= (float)sqrt(n2);
    }
  };
  unit_rows(hn); unit_rows(hc);
  for (int r = 0; r < B; ++r) hinv[r] = 1.0f / (B * (0.9f + 0.2f * rand() / RAND_MAX));
  float *dn, *dc, *inv_r, *inv_c, *sum_r, *sum_c, *dN, *dC, *one;
  void *pn, *pc;
  const size_t pbytes = tt_score_pack_bytes(B, D);
  CK(hipMalloc(&dn, hn.size() * 4)); CK(hipMalloc(&dc, hc.size() * 4)); CK(hipMalloc(&inv_r, B * 4)); CK(hipMalloc(&inv_c, B * 4));
  CK(hipMalloc(&sum_r, B * 4)); CK(hipMalloc(&sum_c, B * 4));
  CK(hipMalloc(&dN, hn.size() * 4)); CK(hipMalloc(&dC, hn.size() * 4)); CK(hipMalloc(&g_ref[0], hn.size() * 4)); CK(hipMalloc(&g_ref[1], hn.size() * 4));
  CK(hipMalloc(&one, 4)); CK(hipMalloc(&pn, pbytes)); CK(hipMalloc(&pc, pbytes));
  CK(hipMemcpy(dn, hn.data(), hn.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(inv_r, hinv.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(inv_c, hinv.data(), B * 4, hipMemcpyHostToDevice));
  std::vector<float> hs(B, 1.0f);
  CK(hipMemcpy(sum_r, hs.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sum_c, hs.data(), B * 4, hipMemcpyHostToDevice));
  const float h1 = 1.0f;
  CK(hipMemcpy(one, &h1, 4, hipMemcpyHostToDevice));
  const float sn = tt_score_unit_scale(1.0f);
  if (tt_score_pack2_bf16(ctx, dn, B, pn, dc, B, pc, D, sn, 1.0f, nullptr) != 0) { printf("pack: %s\n", tt_last_error_string()); return 1; }
  // the library's own backward: reference bits + its time (whatever form the library picks, and the one-tile-ahead form)
  tt_score_bwd_dir dirs[2] = {};
  dirs[0].A_packed = pn; dirs[0].B_packed = pc; dirs[0].Ra = B; dirs[0].Rb = B; dirs[0].sumexp_a = sum_r; dirs[0].sumexp_b = sum_c; dirs[0].dA = g_ref[0];
  dirs[0].ab_scale = sn; dirs[0].b_scale = 1.0f; dirs[0].inv_a = inv_r; dirs[0].inv_b = inv_c;
  dirs[1].A_packed = pc; dirs[1].B_packed = pn; dirs[1].Ra = B; dirs[1].Rb = B; dirs[1].sumexp_a = sum_c; dirs[1].sumexp_b = sum_r; dirs[1].dA = g_ref[1];
  dirs[1].ab_scale = sn; dirs[1].b_scale = sn; dirs[1].inv_a = inv_c; dirs[1].inv_b = inv_r;
  const float scale = 1.0f / (2.0f * B);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  {
    float best = 1e9f;
    for (int i = 0; i < 23; ++i) {
      CK(hipEventRecord(e0));
      if (tt_score_bwd_bf16(ctx, dirs, 2, D, 1.0f, 1.0f, one, scale, nullptr) != 0) { printf("bwd: %s\n", tt_last_error_string()); return 1; }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 3) best = ms < best ? ms : best;
    }
    printf("library tt_score_bwd_bf16: %7.2f us min (HIP events around one launch: ~5 us of launch overhead inside)\n", best * 1e3);
  }
  const size_t half = (size_t)B * 64;                     // rows image: B x 64 bf16
  (void)half;
  g_args.d[0] = DirBwd{(const __bf16*)pn, (const __bf16*)pc, B, B, 0, inv_r, inv_c, dN, scale / 1.0f};
  g_args.d[1] = DirBwd{(const __bf16*)pc, (const __bf16*)pn, B, B, 0, inv_c, inv_r, dC, scale / sn};
  g_args.d_loss = one; g_args.D = D;
  CK(hipMalloc(&g_args.stamps, 4 * 2 * 128 * 8 * 8));
  CK(hipMemset(g_args.stamps, 0, 4 * 2 * 128 * 8 * 8));
#include "score_bwd_lab_runs.inc"
  return 0;
}
