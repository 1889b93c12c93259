// Tower MLP (BaseTower.forward after the lookup, and its backward): Linear layers on the f32 MFMA
// GEMM tiles of tt_gemm.hip, BatchNorm1d statistics as deterministic two-stage column reductions,
// ReLU / BN / dropout / L2-normalise as fused elementwise and row-wise kernels.
#include "tt_gemm.h"

namespace {

constexpr int kThreads = 256;
constexpr float kBnEps = 1e-5f, kBnMomentum = 0.1f, kNormEps = 1e-12f;
constexpr int kMaxChunks = 64;   // row chunks of the two-stage column reductions

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
  const float d = b.mean - a.mean;
  o.mean = a.mean + d * (b.n / o.n);
  o.m2 = a.m2 + b.m2 + d * d * (a.n * b.n / o.n);
  return o;
}

// grid (colblocks of 64, nchunks); thread = (column, row-lane of 4)
__global__ __launch_bounds__(kThreads) void bn_stats_partial_kernel(const float* __restrict__ pre, int B, int H, int rows_per_chunk,
                                                                   float* __restrict__ partial /*[nchunks][3][H]*/) {
  __shared__ Wf sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(B, r0 + rows_per_chunk);
  Wf w{0.f, 0.f, 0.f};
  if (c < H) {
    for (int r = r0 + rl; r < r1; r += 4) {
      const float a = fmaxf(pre[(int64_t)r * H + c], 0.f);
      w.n += 1.f;
      const float d = a - w.mean;
      w.mean += d / w.n;
      w.m2 += d * (a - w.mean);
    }
  }
  sh[rl][threadIdx.x & 63] = w;
  __syncthreads();
  if (rl == 0 && c < H) {
    Wf o = sh[0][threadIdx.x];
    o = wf_combine(o, sh[1][threadIdx.x]);
    o = wf_combine(o, sh[2][threadIdx.x]);
    o = wf_combine(o, sh[3][threadIdx.x]);
    float* p = partial + (int64_t)blockIdx.y * 3 * H;
    p[c] = o.n; p[H + c] = o.mean; p[2 * H + c] = o.m2;
  }
}

// finish: workgroup = 64 columns x 4 chunk-lanes; lane j combines chunks j, j+4, ... then the four
// partial results are combined in lane order (fixed order => reproducible)
__global__ __launch_bounds__(kThreads) void bn_stats_finish_kernel(const float* __restrict__ partial, int nchunks, int H,
                                                                  float* __restrict__ mean, float* __restrict__ rstd,
                                                                  float* __restrict__ running_mean, float* __restrict__ running_var) {
  __shared__ Wf sh[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), jl = threadIdx.x >> 6;
  Wf o{0.f, 0.f, 0.f};
  if (c < H) {
    Wf v[kMaxChunks / 4];
#pragma unroll
    for (int i = 0; i < kMaxChunks / 4; ++i) {            // all loads first (latency-bound otherwise)
      const int k = jl + 4 * i;
      const float* p = partial + (int64_t)k * 3 * H;
      v[i] = k < nchunks ? Wf{p[c], p[H + c], p[2 * H + c]} : Wf{0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < kMaxChunks / 4; ++i) o = wf_combine(o, v[i]);
  }
  sh[jl][threadIdx.x & 63] = o;
  __syncthreads();
  if (jl != 0 || c >= H) return;
  o = wf_combine(wf_combine(sh[0][threadIdx.x], sh[1][threadIdx.x]), wf_combine(sh[2][threadIdx.x], sh[3][threadIdx.x]));
  const float var = o.n > 0.f ? o.m2 / o.n : 0.f;
  mean[c] = o.mean;
  rstd[c] = 1.f / sqrtf(var + kBnEps);
  if (running_mean) {   // nn.BatchNorm1d: momentum 0.1, unbiased variance in the running estimate
    running_mean[c] = (1.f - kBnMomentum) * running_mean[c] + kBnMomentum * o.mean;
    running_var[c] = (1.f - kBnMomentum) * running_var[c] + kBnMomentum * (o.n > 1.f ? o.m2 / (o.n - 1.f) : var);
  }
}

__global__ void bn_eval_prepare_kernel(const float* __restrict__ rm, const float* __restrict__ rv, int H, float* __restrict__ mean,
                                       float* __restrict__ rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < H) {
    mean[c] = rm[c];
    rstd[c] = 1.f / sqrtf(rv[c] + kBnEps);
  }
}

__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const float* __restrict__ pre, int64_t total, int H,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ g, const float* __restrict__ b, bool drop, float p,
                                                           uint64_t seed0, const uint64_t* __restrict__ seed_dev, uint64_t salt,
                                                           float* __restrict__ act) {
  const uint64_t seed = drop ? seed_of(seed0, seed_dev) : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % H);
    const float a = fmaxf(pre[i], 0.f);
    const float y = (a - mean[c]) * rstd[c] * g[c] + b[c];
    act[i] = y * dropout_scale(drop, p, seed, salt + (uint64_t)i);
  }
}

// ---- deterministic two-stage column sums -----------------------------------------------------
// OP 0: v0 = x[r, c]                                   (bias gradients)
// OP 1: da = d_act * dropscale ; xhat from pre ; v0 = da, v1 = da * xhat   (BatchNorm backward sums)
struct ColArgs {
  const float* x; int64_t ldx;
  const float* pre; const float* mean; const float* rstd;
  bool drop; float p; uint64_t seed, salt;
  const uint64_t* seed_dev;
};

template <int OP>
__global__ __launch_bounds__(kThreads) void colsum_partial_kernel(ColArgs a, int B, int H, int rows_per_chunk,
                                                                 float* __restrict__ partial /*[nchunks][2][H]*/) {
  __shared__ float sh[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(B, r0 + rows_per_chunk);
  float s0 = 0.f, s1 = 0.f;
  const uint64_t seed = (OP == 1 && a.drop) ? seed_of(a.seed, a.seed_dev) : 0;
  if (c < H) {
#pragma unroll 8
    for (int r = r0 + rl; r < r1; r += 4) {
      if (OP == 0) {
        s0 += a.x[(int64_t)r * a.ldx + c];
      } else {
        const int64_t i = (int64_t)r * H + c;
        const float da = a.x[(int64_t)r * a.ldx + c] * dropout_scale(a.drop, a.p, seed, a.salt + (uint64_t)i);
        const float xh = (fmaxf(a.pre[i], 0.f) - a.mean[c]) * a.rstd[c];
        s0 += da;
        s1 += da * xh;
      }
    }
  }
  sh[0][rl][threadIdx.x & 63] = s0;
  sh[1][rl][threadIdx.x & 63] = s1;
  __syncthreads();
  if (rl == 0 && c < H) {
    float* p = partial + (int64_t)blockIdx.y * 2 * H;
    p[c] = ((sh[0][0][threadIdx.x] + sh[0][1][threadIdx.x]) + sh[0][2][threadIdx.x]) + sh[0][3][threadIdx.x];
    if (OP == 1) p[H + c] = ((sh[1][0][threadIdx.x] + sh[1][1][threadIdx.x]) + sh[1][2][threadIdx.x]) + sh[1][3][threadIdx.x];
  }
}

__global__ __launch_bounds__(kThreads) void colsum_finish_kernel(const float* __restrict__ partial, int nchunks, int H, int nv,
                                                                float* __restrict__ out0, float* __restrict__ out1) {
  __shared__ float sh[2][4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), jl = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f;
  if (c < H) {
    float v0[kMaxChunks / 4], v1[kMaxChunks / 4];
#pragma unroll
    for (int i = 0; i < kMaxChunks / 4; ++i) {
      const int k = jl + 4 * i;
      const float* p = partial + (int64_t)k * 2 * H;
      v0[i] = k < nchunks ? p[c] : 0.f;
      v1[i] = (k < nchunks && nv > 1) ? p[H + c] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < kMaxChunks / 4; ++i) { s0 += v0[i]; s1 += v1[i]; }
  }
  sh[0][jl][threadIdx.x & 63] = s0;
  sh[1][jl][threadIdx.x & 63] = s1;
  __syncthreads();
  if (jl != 0 || c >= H) return;
  out0[c] = (sh[0][0][threadIdx.x] + sh[0][1][threadIdx.x]) + (sh[0][2][threadIdx.x] + sh[0][3][threadIdx.x]);
  if (nv > 1) out1[c] = (sh[1][0][threadIdx.x] + sh[1][1][threadIdx.x]) + (sh[1][2][threadIdx.x] + sh[1][3][threadIdx.x]);
}

// BN backward apply, in place on the gradient buffer:  d_act -> d_pre
//   train: d_a = g rstd (da - S1/B - xhat S2/B)   eval: d_a = da g rstd ;   d_pre = d_a [pre > 0]
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(float* __restrict__ d, const float* __restrict__ pre, int64_t total, int H,
                                                               float invB, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ g, const float* __restrict__ S1,
                                                               const float* __restrict__ S2, bool train, bool drop, float p, uint64_t seed0,
                                                               const uint64_t* __restrict__ seed_dev, uint64_t salt) {
  const uint64_t seed = drop ? seed_of(seed0, seed_dev) : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % H);
    const float pr = pre[i];
    const float da = d[i] * dropout_scale(drop, p, seed, salt + (uint64_t)i);
    float dv;
    if (train) {
      const float xh = (fmaxf(pr, 0.f) - mean[c]) * rstd[c];
      dv = g[c] * rstd[c] * (da - S1[c] * invB - xh * (S2[c] * invB));
    } else {
      dv = da * g[c] * rstd[c];
    }
    d[i] = pr > 0.f ? dv : 0.f;
  }
}

// ---- row-wise L2 normalise: one wave per row ---------------------------------------------------
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}

__global__ __launch_bounds__(kThreads) void l2norm_fwd_kernel(const float* __restrict__ y, int B, int D, float* __restrict__ emb) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) {
    const float v = y[(int64_t)row * D + c];
    ss += v * v;
  }
  const float den = fmaxf(sqrtf(wave_sum(ss)), kNormEps);
  for (int c = lane; c < D; c += 64) emb[(int64_t)row * D + c] = y[(int64_t)row * D + c] / den;
}

__global__ __launch_bounds__(kThreads) void l2norm_bwd_kernel(const float* __restrict__ y, const float* __restrict__ emb,
                                                             const float* __restrict__ d_emb, int B, int D, float* __restrict__ d_y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B) return;
  float ss = 0.f, dot = 0.f;
  for (int c = lane; c < D; c += 64) {
    const int64_t i = (int64_t)row * D + c;
    ss += y[i] * y[i];
    dot += emb[i] * d_emb[i];
  }
  const float nrm = sqrtf(wave_sum(ss));
  dot = wave_sum(dot);
  const float den = fmaxf(nrm, kNormEps);
  for (int c = lane; c < D; c += 64) {
    const int64_t i = (int64_t)row * D + c;
    d_y[i] = nrm > kNormEps ? (d_emb[i] - emb[i] * dot) / den : d_emb[i] / den;
  }
}

// ---- host side ---------------------------------------------------------------------------------
inline int ew_grid(const tt_ctx* ctx, int64_t n) {
  const int64_t cap = (int64_t)ctx->num_cus * 8;
  int64_t b = tt_cdiv(n, kThreads);
  return (int)(b < 1 ? 1 : (b < cap ? b : cap));
}

inline int chunks_for(int64_t B, int H) {
  int64_t n = 256 / tt_cdiv(H, 64);
  if (n > kMaxChunks) n = kMaxChunks;
  const int64_t mx = tt_cdiv(B, 64);
  if (n > mx) n = mx;
  return (int)(n < 1 ? 1 : n);
}

struct WsLayout {
  char* gemm;
  size_t gemm_bytes;
  float* col;
  size_t col_bytes;
  size_t total;
};

inline int in_width(const tt_tower_params* p, int i) { return i == 0 ? p->h0 + p->kcat_e : p->hidden[i - 1]; }
inline int last_width(const tt_tower_params* p) { return p->n_hidden == 0 ? p->h0 + p->kcat_e : p->hidden[p->n_hidden - 1]; }

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
  WsLayout w;
  w.gemm_bytes = (g + 255) & ~size_t(255);
  w.col_bytes = sizeof(float) * 3 * (size_t)hmax * (size_t)(tt_cdiv(B > 0 ? B : 1, 64) < 512 ? tt_cdiv(B > 0 ? B : 1, 64) : 512) + 256;
  w.gemm = base;
  w.col = reinterpret_cast<float*>(base ? base + w.gemm_bytes : nullptr);
  w.total = w.gemm_bytes + w.col_bytes;
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

int colsum(const tt_ctx* ctx, hipStream_t st, int op, const ColArgs& a, int64_t B, int H, float* partial, float* out0, float* out1) {
  const int nchunks = chunks_for(B, H);
  const int rpc = (int)tt_cdiv(B, nchunks);
  dim3 grid((unsigned)tt_cdiv(H, 64), (unsigned)nchunks);
  if (op == 0) colsum_partial_kernel<0><<<grid, kThreads, 0, st>>>(a, (int)B, H, rpc, partial);
  else colsum_partial_kernel<1><<<grid, kThreads, 0, st>>>(a, (int)B, H, rpc, partial);
  TT_LAUNCH_CHECK();
  colsum_finish_kernel<<<(unsigned)tt_cdiv(H, 64), kThreads, 0, st>>>(partial, nchunks, H, op == 0 ? 1 : 2, out0, out1);
  TT_LAUNCH_CHECK();
  (void)ctx;
  return TT_OK;
}

}  // namespace

extern "C" {

size_t tt_tower_workspace_bytes(const tt_tower_params* p, int64_t B) {
  if (!p || p->n_hidden < 0 || p->n_hidden > TT_MAX_HIDDEN) return 0;
  return ws_layout(p, B, nullptr).total;
}

int tt_tower_mlp_fwd(tt_ctx* ctx, const tt_tower_params* p, const tt_tower_acts* a, int64_t B, int32_t train, float dropout_p,
                     uint64_t seed, const uint64_t* seed_dev, void* workspace, size_t workspace_bytes, tt_stream stream) {
  TT_CHECK_ARG(ctx && a, "tt_tower_mlp_fwd: NULL argument");
  if (int rc = check_params(p, "tt_tower_mlp_fwd")) return rc;
  TT_CHECK_ARG(B >= 0 && B < ((int64_t)1 << 24), "tt_tower_mlp_fwd: B=%lld out of range", (long long)B);
  TT_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "tt_tower_mlp_fwd: dropout_p=%f not in [0,1)", dropout_p);
  if (B == 0) return TT_OK;
  TT_CHECK_ARG(a->dense && a->x && a->y && a->emb, "tt_tower_mlp_fwd: NULL activation buffers");
  if (!workspace || workspace_bytes < tt_tower_workspace_bytes(p, B)) {
    tt_set_error("tt_tower_mlp_fwd: workspace %zu < required %zu", workspace_bytes, tt_tower_workspace_bytes(p, B));
    return TT_ERR_WORKSPACE;
  }
  const WsLayout ws = ws_layout(p, B, reinterpret_cast<char*>(workspace));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int wx = p->h0 + p->kcat_e;
  // x[:, 0:h0] = dense W_proj^T + b_proj        (base_tower.py:133)
  if (int rc = tt_gemm_nt(st, a->dense, p->din, p->w_proj, p->din, p->b_proj, a->x, wx, B, p->h0, p->din, false)) return rc;
  const float* in = a->x;
  int in_w = wx;
  const bool drop = train && dropout_p > 0.f;
  for (int i = 0; i < p->n_hidden; ++i) {
    const int H = p->hidden[i];
    TT_CHECK_ARG(a->pre[i] && a->act[i] && a->mean[i] && a->rstd[i], "tt_tower_mlp_fwd: NULL buffers of block %d", i);
    if (int rc = tt_gemm_nt(st, in, in_w, p->w[i], in_w, p->b[i], a->pre[i], H, B, H, in_w, false)) return rc;
    if (train) {
      const int nchunks = chunks_for(B, H);
      const int rpc = (int)tt_cdiv(B, nchunks);
      bn_stats_partial_kernel<<<dim3((unsigned)tt_cdiv(H, 64), (unsigned)nchunks), kThreads, 0, st>>>(a->pre[i], (int)B, H, rpc, ws.col);
      TT_LAUNCH_CHECK();
      bn_stats_finish_kernel<<<(unsigned)tt_cdiv(H, 64), kThreads, 0, st>>>(ws.col, nchunks, H, a->mean[i], a->rstd[i], p->bn_rm[i], p->bn_rv[i]);
      TT_LAUNCH_CHECK();
    } else {
      bn_eval_prepare_kernel<<<(unsigned)tt_cdiv(H, 256), 256, 0, st>>>(p->bn_rm[i], p->bn_rv[i], H, a->mean[i], a->rstd[i]);
      TT_LAUNCH_CHECK();
    }
    bn_apply_kernel<<<ew_grid(ctx, B * H), kThreads, 0, st>>>(a->pre[i], B * H, H, a->mean[i], a->rstd[i], p->bn_w[i], p->bn_b[i], drop,
                                                               dropout_p, seed, seed_dev, (uint64_t)(i + 1) << 40, a->act[i]);
    TT_LAUNCH_CHECK();
    in = a->act[i];
    in_w = H;
  }
  if (int rc = tt_gemm_nt(st, in, in_w, p->w_out, in_w, p->b_out, a->y, p->d_out, B, p->d_out, in_w, false)) return rc;
  l2norm_fwd_kernel<<<(unsigned)tt_cdiv(B, 4), kThreads, 0, st>>>(a->y, (int)B, p->d_out, a->emb);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

int tt_tower_mlp_bwd(tt_ctx* ctx, const tt_tower_params* p, const tt_tower_acts* a, const float* d_emb, const tt_tower_grads* g,
                     int64_t B, int32_t train, float dropout_p, uint64_t seed, const uint64_t* seed_dev, void* workspace,
                     size_t workspace_bytes, tt_stream stream) {
  TT_CHECK_ARG(ctx && a && g && d_emb, "tt_tower_mlp_bwd: NULL argument");
  if (int rc = check_params(p, "tt_tower_mlp_bwd")) return rc;
  TT_CHECK_ARG(B >= 1 && B < ((int64_t)1 << 24), "tt_tower_mlp_bwd: B=%lld out of range", (long long)B);
  TT_CHECK_ARG(g->w_proj && g->b_proj && g->w_out && g->b_out && g->d_x && g->d_y, "tt_tower_mlp_bwd: NULL gradient buffers");
  if (!workspace || workspace_bytes < tt_tower_workspace_bytes(p, B)) {
    tt_set_error("tt_tower_mlp_bwd: workspace %zu < required %zu", workspace_bytes, tt_tower_workspace_bytes(p, B));
    return TT_ERR_WORKSPACE;
  }
  const WsLayout ws = ws_layout(p, B, reinterpret_cast<char*>(workspace));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int wx = p->h0 + p->kcat_e;
  const int nh = p->n_hidden;
  const bool drop = train && dropout_p > 0.f;
  l2norm_bwd_kernel<<<(unsigned)tt_cdiv(B, 4), kThreads, 0, st>>>(a->y, a->emb, d_emb, (int)B, p->d_out, g->d_y);
  TT_LAUNCH_CHECK();
  const float* in_last = nh == 0 ? a->x : a->act[nh - 1];
  const int lw = last_width(p);
  if (int rc = tt_gemm_tn(st, g->d_y, p->d_out, in_last, lw, g->w_out, lw, p->d_out, lw, B, ws.gemm, ws.gemm_bytes, g->b_out)) return rc;
  float* dcur = nh == 0 ? g->d_x : g->scratch[nh - 1];
  TT_CHECK_ARG(dcur, "tt_tower_mlp_bwd: NULL scratch buffer");
  if (int rc = tt_gemm_nn(st, g->d_y, p->d_out, p->w_out, lw, dcur, lw, B, lw, p->d_out)) return rc;
  for (int i = nh - 1; i >= 0; --i) {
    const int H = p->hidden[i];
    const int iw = in_width(p, i);
    TT_CHECK_ARG(g->w[i] && g->b[i] && g->bn_w[i] && g->bn_b[i], "tt_tower_mlp_bwd: NULL gradient buffers of block %d", i);
    const uint64_t salt = (uint64_t)(i + 1) << 40;
    ColArgs cb{};
    cb.x = dcur; cb.ldx = H; cb.pre = a->pre[i]; cb.mean = a->mean[i]; cb.rstd = a->rstd[i];
    cb.drop = drop; cb.p = dropout_p; cb.seed = seed; cb.salt = salt; cb.seed_dev = seed_dev;
    // S1 = sum da -> bn bias grad ; S2 = sum da*xhat -> bn weight grad
    if (int rc = colsum(ctx, st, 1, cb, B, H, ws.col, g->bn_b[i], g->bn_w[i])) return rc;
    bn_bwd_apply_kernel<<<ew_grid(ctx, B * H), kThreads, 0, st>>>(dcur, a->pre[i], B * H, H, 1.f / (float)B, a->mean[i], a->rstd[i],
                                                                   p->bn_w[i], g->bn_b[i], g->bn_w[i], train != 0, drop, dropout_p, seed, seed_dev, salt);
    TT_LAUNCH_CHECK();
    const float* in_i = i == 0 ? a->x : a->act[i - 1];
    if (int rc = tt_gemm_tn(st, dcur, H, in_i, iw, g->w[i], iw, H, iw, B, ws.gemm, ws.gemm_bytes, g->b[i])) return rc;
    float* dnext = i == 0 ? g->d_x : g->scratch[i - 1];
    TT_CHECK_ARG(dnext, "tt_tower_mlp_bwd: NULL scratch buffer");
    if (int rc = tt_gemm_nn(st, dcur, H, p->w[i], iw, dnext, iw, B, iw, H)) return rc;
    dcur = dnext;
  }
  // dense projection: d_x[:, 0:h0]
  if (int rc = tt_gemm_tn(st, g->d_x, wx, a->dense, p->din, g->w_proj, p->din, p->h0, p->din, B, ws.gemm, ws.gemm_bytes, g->b_proj)) return rc;
  return TT_OK;
}

int tt_linear_fwd(tt_ctx* ctx, const float* X, int64_t ldx, const float* W, const float* bias, float* Y, int64_t ldy, int64_t M,
                  int32_t N, int32_t K, int32_t relu, tt_stream stream) {
  TT_CHECK_ARG(ctx && (M == 0 || (X && W && Y)), "tt_linear_fwd: NULL argument");
  TT_CHECK_ARG(M >= 0 && N >= 1 && K >= 1 && ldx >= K && ldy >= N, "tt_linear_fwd: bad shape");
  TT_CHECK_ARG(M < ((int64_t)1 << 31), "tt_linear_fwd: M too large");
  return tt_gemm_nt(reinterpret_cast<hipStream_t>(stream), X, ldx, W, K, bias, Y, ldy, M, N, K, relu != 0);
}

}  // extern "C"
