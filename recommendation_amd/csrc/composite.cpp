// Whole-network entry points of the interaction models: the launch sequences of the DCN-v2 cross network and of the
// DIN local activation unit behind ONE call each way, for hosts that drive the library from C/C++ (no torch, no
// Python).  Nothing here computes: every step is one of the library's own launches, in the order and with the
// operands the Python hosts (dcn.py, din.py) use, so the two routes leave the same bits.
#include <stdint.h>
#include <stddef.h>

#include "common.h"

namespace {

size_t up256(size_t x) { return (x + 255) & ~(size_t) 255; }

// K-slices of a weight-gradient contraction [n_out][k_in] over m rows: enough workgroups to fill the chip, every
// slice a whole number of 64-row tiles and all slices in use.
int wgrad_splits(int64_t n_out, int64_t k_in, int64_t m) {
  const int64_t tiles = mi_cdiv(n_out, 128) * mi_cdiv(k_in, 128);
  const int64_t t64 = m / 64;
  int64_t s = 512 / (tiles > 0 ? tiles : 1);
  if (s > t64) s = t64;
  if (s < 1) s = 1;
  while (s > 1 && mi_cdiv(t64, mi_cdiv(t64, s)) != s) --s;
  return (int) s;
}

mi_gemm_epilogue epi(int mode, int flags, void* out, int64_t ld_out, const float* bias = nullptr,
                     const void* in0 = nullptr, int64_t ld_in0 = 0, const void* in1 = nullptr, int64_t ld_in1 = 0,
                     void* out1 = nullptr, int64_t ld_out1 = 0, int64_t bias_ld = 0, int bias_rows = 0) {
  mi_gemm_epilogue e;
  e.mode = mode; e.flags = flags; e.out = out; e.ld_out = ld_out; e.bias = bias; e.in0 = in0; e.ld_in0 = ld_in0;
  e.in1 = in1; e.ld_in1 = ld_in1; e.out1 = out1; e.ld_out1 = ld_out1; e.bias_ld = bias_ld; e.bias_rows = bias_rows;
  return e;
}

struct Adam {
  float lr_t;
  const float* lr_dev;
  float b1, b2, eps;
  bool apply;  // false: gradients only (into grad_out), parameters and their bf16 copies untouched
};

#define MI_TRY(call)               \
  do {                             \
    const int rc_ = (call);        \
    if (rc_ != MI_OK) return rc_;  \
  } while (0)

// dW = dY^T X as split-K slabs, the fused reduction + Adam + bf16 re-cast, and the transposed copy.
int weight_step(const mi_adam_param& p, int64_t n_out, int64_t k_in, const void* dY, int64_t ld_dy, const void* X,
                int64_t ld_x, int64_t m, float* slab, const Adam& a, void* s) {
  const int splits = wgrad_splits(n_out, k_in, m);
  const mi_gemm_epilogue e = epi(MI_EPI_SLAB, 0, slab, k_in);
  MI_TRY(mi_gemm_bf16(1, 1, n_out, k_in, m, dY, ld_dy, X, ld_x, splits, &e, s));
  if (!a.apply)
    return mi_slab_adam(slab, splits, n_out * k_in, n_out * k_in, nullptr, nullptr, nullptr, 0.f, nullptr, 0.f, 0.f, 0.f,
                        nullptr, p.grad_out, s);
  MI_TRY(mi_slab_adam(slab, splits, n_out * k_in, n_out * k_in, p.w, p.m, p.v, a.lr_t, a.lr_dev, a.b1, a.b2, a.eps,
                      p.w_bf16, p.grad_out, s));
  return mi_transpose_bf16(p.w_bf16, k_in, (int) n_out, (int) k_in, p.w_bf16_t, n_out, s);
}

// db = column sums of dY in `slices` partials, then the same fused reduction + Adam.
int bias_step(const mi_adam_param& p, int n, const void* dY, int64_t ld_dy, int64_t m, int slices, float* slab,
              const Adam& a, void* s) {
  MI_TRY(mi_colsum_bf16(dY, ld_dy, m, n, slices, slab, n, s));
  if (!a.apply)
    return mi_slab_adam(slab, slices, n, n, nullptr, nullptr, nullptr, 0.f, nullptr, 0.f, 0.f, 0.f, nullptr, p.grad_out, s);
  return mi_slab_adam(slab, slices, n, n, p.w, p.m, p.v, a.lr_t, a.lr_dev, a.b1, a.b2, a.eps, nullptr, p.grad_out, s);
}

bool param_ok(const mi_adam_param& p, bool matrix) { return p.w && p.m && p.v && (!matrix || (p.w_bf16 && p.w_bf16_t)); }

// ---- cross network ----------------------------------------------------------------------------
struct CrossWs {
  char* x[MI_CROSS_MAX_LAYERS + 1];  // x[0] unused (the caller's x0)
  char* u[MI_CROSS_MAX_LAYERS];
  char* dx[MI_CROSS_MAX_LAYERS + 1];
  char* du[2];
  float* slab;
  size_t bytes;
  int vec_slices;
};

int cross_slices(int64_t B) {
  const int64_t v = B / 64;
  return (int) (v < 1 ? 1 : (v > 256 ? 256 : v));
}

CrossWs cross_carve(void* ws, int64_t B, int width, int L) {
  CrossWs c = {};
  char* p = (char*) ws;
  const size_t act = up256((size_t) B * width * 2);
  for (int l = 1; l <= L; ++l) { c.x[l] = p; p += act; }
  for (int l = 0; l < L; ++l) { c.u[l] = p; p += act; }
  for (int l = 0; l <= L; ++l) { c.dx[l] = p; p += act; }
  for (int i = 0; i < 2; ++i) { c.du[i] = p; p += act; }
  c.vec_slices = cross_slices(B);
  const size_t a = (size_t) wgrad_splits(width, width, B) * width * width, b = (size_t) c.vec_slices * width;
  c.slab = (float*) p;
  p += up256((a > b ? a : b) * 4);
  c.bytes = (size_t) (p - (char*) ws);
  return c;
}

int cross_check(const char* who, const mi_cross_net* net, const void* x0, int64_t ld, int64_t B, const void* ws,
                size_t ws_bytes) {
  MI_REQUIRE(net && x0 && ws, "%s: null pointer", who);
  MI_REQUIRE(net->n_layers >= 1 && net->n_layers <= MI_CROSS_MAX_LAYERS, "%s: 1..%d layers", who, MI_CROSS_MAX_LAYERS);
  MI_REQUIRE(net->width >= 64 && net->width % 64 == 0 && ld >= net->width, "%s: width must be a multiple of 64 <= ld", who);
  MI_REQUIRE(B >= 64 && B % 64 == 0, "%s: batch must be a multiple of 64 (the contraction tile depth)", who);
  MI_REQUIRE(((uintptr_t) ws & 255) == 0, "%s: workspace must be 256-byte aligned", who);
  for (int l = 0; l < net->n_layers; ++l)
    MI_REQUIRE(param_ok(net->W[l], true) && param_ok(net->b[l], false), "%s: layer %d has a null parameter pointer", who, l);
  const size_t need = mi_cross_workspace_bytes(B, net->width, net->n_layers);
  if (ws_bytes < need) {
    mi_set_error("%s: workspace %zu < %zu", who, ws_bytes, need);
    return MI_ERR_WORKSPACE;
  }
  return MI_OK;
}

// ---- DIN activation unit ------------------------------------------------------------------------
struct DinWs {
  char *A, *dA, *H1, *dZ1, *H2, *dH2;
  float *a, *da, *rowbias, *S, *dq, *qpart, *gWf, *gq, *part, *part_sums, *slab;
  int part_slices;
  size_t bytes;
  int vec_slices, a_cols;
};

DinWs din_carve(void* ws, int64_t B, const mi_din_lau* n) {
  DinWs d = {};
  char* p = (char*) ws;
  const int64_t M = B * n->L;
  const int D = n->dim, H = n->h1;
  d.a_cols = (n->factored ? 2 : 4) * D;
  auto take = [&](size_t bytes) { char* r = p; p += up256(bytes); return r; };
  d.A = take((size_t) M * d.a_cols * 2);
  d.dA = take((size_t) M * d.a_cols * 2);
  d.H1 = take((size_t) M * H * 2);
  d.dZ1 = take((size_t) M * H * 2);
  d.H2 = take((size_t) M * n->h2 * 2);
  d.dH2 = take((size_t) M * n->h2 * 2);
  d.a = (float*) take((size_t) M * 4);
  d.da = (float*) take((size_t) M * 4);
  if (n->factored) {
    d.rowbias = (float*) take((size_t) B * H * 4);
    d.S = (float*) take((size_t) B * H * 4);
    d.dq = (float*) take((size_t) B * D * 4);
    d.qpart = (float*) take((size_t) mi_din_qterm_part_floats(B, H, D) * 4);
    d.gWf = (float*) take((size_t) H * 2 * D * 4);
    d.gq = (float*) take((size_t) (D * H + H) * 4);
    // per-sample sums of the fused backward, and their reduction to at most 256 runs of samples
    int ps = (int) (B < 256 ? B : 256);
    while (ps > 1 && mi_cdiv(B, mi_cdiv(B, ps)) != ps) --ps;
    d.part_slices = ps;
    d.part = (float*) take((size_t) B * mi_din_lau_bwd_part_floats() * 4);
    d.part_sums = (float*) take((size_t) ps * mi_din_lau_bwd_part_floats() * 4);
  }
  const int64_t v = M / 64;
  d.vec_slices = (int) (v < 1 ? 1 : (v > 1024 ? 1024 : v));
  size_t s1 = (size_t) wgrad_splits(H, d.a_cols, M) * H * d.a_cols, s2 = (size_t) wgrad_splits(n->h2, H, M) * n->h2 * H,
         s3 = (size_t) d.vec_slices * (H + 4), s4 = (size_t) d.vec_slices * (n->h2 + 4);
  size_t sl = s1 > s2 ? s1 : s2;
  sl = sl > s3 ? sl : s3;
  sl = sl > s4 ? sl : s4;
  d.slab = (float*) take(sl * 4);
  d.bytes = (size_t) (p - (char*) ws);
  return d;
}

int din_check(const char* who, const mi_din_lau* n, int64_t B, const void* ws, size_t ws_bytes) {
  MI_REQUIRE(n && ws, "%s: null pointer", who);
  MI_REQUIRE(n->dim == 64, "%s: only dim 64 is built", who);
  MI_REQUIRE(n->h1 >= 64 && n->h1 % 64 == 0 && n->h2 >= 64 && n->h2 % 64 == 0, "%s: hidden widths must be multiples of 64", who);
  MI_REQUIRE(n->L >= 1 && B >= 1 && (B * n->L) % 64 == 0, "%s: batch * history length must be a multiple of 64", who);
  MI_REQUIRE(!n->factored || n->h1 <= 128, "%s: the factored first layer is built for a first hidden width up to 128", who);
  MI_REQUIRE(n->W1.w && n->W1.m && n->W1.v && param_ok(n->b1, false) && param_ok(n->W2, true) && param_ok(n->b2, false) &&
                 param_ok(n->w3, false),
             "%s: null parameter pointer", who);
  if (n->factored) MI_REQUIRE(n->W1f && n->W1fT && n->Wq && n->WqT, "%s: the factored form needs W1f, W1fT, Wq, WqT", who);
  else MI_REQUIRE(n->W1.w_bf16 && n->W1.w_bf16_t, "%s: the full-row form needs the bf16 copies of W1", who);
  MI_REQUIRE(((uintptr_t) ws & 255) == 0, "%s: workspace must be 256-byte aligned", who);
  const size_t need = mi_din_lau_workspace_bytes(B, n);
  if (ws_bytes < need) {
    mi_set_error("%s: workspace %zu < %zu", who, ws_bytes, need);
    return MI_ERR_WORKSPACE;
  }
  return MI_OK;
}

}  // namespace

extern "C" int mi_gemm_wgrad_splits(int64_t n_out, int64_t k_in, int64_t m) { return wgrad_splits(n_out, k_in, m); }

// (layouts are worked out from a stand-in base address, never dereferenced)
static char* const kLayoutBase = reinterpret_cast<char*>((uintptr_t) 4096);

extern "C" size_t mi_cross_workspace_bytes(int64_t B, int width, int n_layers) {
  if (B < 1 || width < 1 || n_layers < 1 || n_layers > MI_CROSS_MAX_LAYERS) return 0;
  return cross_carve(kLayoutBase, B, width, n_layers).bytes;
}

extern "C" int mi_cross_workspace_part(int64_t B, int width, int n_layers, int what, int index, size_t* offset,
                                       size_t* bytes) {
  MI_REQUIRE(offset && bytes && B >= 1 && width >= 1 && n_layers >= 1 && n_layers <= MI_CROSS_MAX_LAYERS,
             "mi_cross_workspace_part: bad arguments");
  const CrossWs c = cross_carve(kLayoutBase, B, width, n_layers);
  const char* p = nullptr;
  size_t n = (size_t) B * width * 2;
  switch (what) {
    case MI_CROSS_PART_X: if (index >= 1 && index <= n_layers) p = c.x[index]; break;
    case MI_CROSS_PART_U: if (index >= 0 && index < n_layers) p = c.u[index]; break;
    case MI_CROSS_PART_DX: if (index >= 0 && index <= n_layers) p = c.dx[index]; break;
    case MI_CROSS_PART_DU: if (index >= 0 && index < 2) p = c.du[index]; break;
    case MI_CROSS_PART_SLAB:
      if (index == 0) {
        p = (const char*) c.slab;
        n = c.bytes - (size_t) (p - kLayoutBase);
      }
      break;
    default: break;
  }
  if (!p) {
    mi_set_error("mi_cross_workspace_part: no part %d[%d]", what, index);
    return MI_ERR_UNSUPPORTED;
  }
  *offset = (size_t) (p - kLayoutBase);
  *bytes = n;
  return MI_OK;
}

extern "C" int mi_cross_fwd(const mi_cross_net* net, const void* x0, int64_t ld, int64_t B, void* workspace,
                            size_t workspace_bytes, const void** x_last, void* stream) {
  MI_TRY(cross_check("mi_cross_fwd", net, x0, ld, B, workspace, workspace_bytes));
  const int L = net->n_layers, W = net->width;
  const CrossWs c = cross_carve(workspace, B, W, L);
  const void* xl = x0;
  int64_t ldl = ld;
  for (int l = 0; l < L; ++l) {
    // u_l = x_l W_l^T + b_l (kept for the backward), x_{l+1} = x0 * u_l + x_l
    const mi_gemm_epilogue e = epi(MI_EPI_CROSS_FWD, 0, c.x[l + 1], W, net->b[l].w, x0, ld, xl, ldl, c.u[l], W);
    MI_TRY(mi_gemm_bf16(0, 0, B, W, W, xl, ldl, net->W[l].w_bf16, W, 1, &e, stream));
    xl = c.x[l + 1];
    ldl = W;
  }
  if (x_last) *x_last = xl;
  return MI_OK;
}

extern "C" int mi_cross_bwd(const mi_cross_net* net, const void* x0, int64_t ld, int64_t B, void* workspace,
                            size_t workspace_bytes, int du_last_ready, int apply_adam, float lr_t,
                            const float* lr_dev, float beta1, float beta2, float eps, float* dx0, int64_t ld_dx0,
                            void* stream) {
  MI_TRY(cross_check("mi_cross_bwd", net, x0, ld, B, workspace, workspace_bytes));
  MI_REQUIRE(dx0 && ld_dx0 >= net->width, "mi_cross_bwd: dx0 needs ld_dx0 >= width");
  const int L = net->n_layers, W = net->width;
  if (!apply_adam)
    for (int l = 0; l < L; ++l)
      MI_REQUIRE(net->W[l].grad_out && net->b[l].grad_out, "mi_cross_bwd: gradients only, but layer %d has no grad_out", l);
  const CrossWs c = cross_carve(workspace, B, W, L);
  const Adam a{lr_t, lr_dev, beta1, beta2, eps, apply_adam != 0};
  // du_{L-1} = dx_L * x0, unless the producer of dx_L stored it already (an MI_EPI_CROSS_BWD epilogue with out1)
  if (!du_last_ready) MI_TRY(mi_mul_bf16(c.dx[L], W, x0, ld, B, W, c.du[L & 1], W, stream));
  for (int l = L - 1; l >= 0; --l) {
    const void* du = c.du[(l + 1) & 1];
    const void* xl = l > 0 ? (const void*) c.x[l] : x0;
    const int64_t ldl = l > 0 ? W : ld;
    // dx_l = du_l W_l + dx_{l+1} with the weights the forward saw; for l > 0 also du_{l-1} = dx_l * x0
    const mi_gemm_epilogue e = l > 0 ? epi(MI_EPI_CROSS_BWD, 0, c.dx[l], W, nullptr, x0, ld, c.dx[l + 1], W, c.du[l & 1], W)
                                     : epi(MI_EPI_CROSS_BWD, 0, c.dx[0], W, nullptr, nullptr, 0, c.dx[1], W);
    MI_TRY(mi_gemm_bf16(0, 0, B, W, W, du, W, net->W[l].w_bf16_t, W, 1, &e, stream));
    MI_TRY(weight_step(net->W[l], W, W, du, W, xl, ldl, B, c.slab, a, stream));
    MI_TRY(bias_step(net->b[l], W, du, W, B, c.vec_slices, c.slab, a, stream));
  }
  const void* dxs[MI_CROSS_MAX_LAYERS + 1];
  const void* us[MI_CROSS_MAX_LAYERS];
  for (int l = 0; l <= L; ++l) dxs[l] = c.dx[l];
  for (int l = 0; l < L; ++l) us[l] = c.u[l];
  return mi_cross_input_grad(dxs, us, L, B, W, W, dx0, ld_dx0, stream);
}

extern "C" size_t mi_din_lau_workspace_bytes(int64_t B, const mi_din_lau* net) {
  if (!net || B < 1 || net->L < 1 || net->dim < 1 || net->h1 < 1 || net->h2 < 1) return 0;
  return din_carve(kLayoutBase, B, net).bytes;
}

extern "C" int mi_din_lau_workspace_part(int64_t B, const mi_din_lau* net, int what, size_t* offset, size_t* bytes) {
  MI_REQUIRE(net && offset && bytes && B >= 1, "mi_din_lau_workspace_part: bad arguments");
  const DinWs d = din_carve(kLayoutBase, B, net);
  const int64_t M = B * net->L;
  const char* base = kLayoutBase;
  const char* p;
  size_t n;
  switch (what) {
    case MI_DIN_PART_A: p = d.A; n = (size_t) M * d.a_cols * 2; break;
    case MI_DIN_PART_DA: p = d.dA; n = (size_t) M * d.a_cols * 2; break;
    case MI_DIN_PART_H1: p = d.H1; n = (size_t) M * net->h1 * 2; break;
    case MI_DIN_PART_DZ1: p = d.dZ1; n = (size_t) M * net->h1 * 2; break;
    case MI_DIN_PART_H2: p = d.H2; n = (size_t) M * net->h2 * 2; break;
    case MI_DIN_PART_DH2: p = d.dH2; n = (size_t) M * net->h2 * 2; break;
    case MI_DIN_PART_WEIGHTS: p = (const char*) d.a; n = (size_t) M * 4; break;
    default:
      mi_set_error("mi_din_lau_workspace_part: no part %d", what);
      return MI_ERR_UNSUPPORTED;
  }
  *offset = (size_t) (p - base);
  *bytes = n;
  return MI_OK;
}

extern "C" int mi_din_lau_fwd(const mi_din_lau* net, const float* E, int64_t rows, const void* hist, const void* qid,
                              int idx_bits, const int32_t* len, int64_t B, void* workspace, size_t workspace_bytes,
                              float* feat, int64_t ld_feat, int32_t* ids_all, int32_t* err, void* stream) {
  MI_TRY(din_check("mi_din_lau_fwd", net, B, workspace, workspace_bytes));
  MI_REQUIRE(E && hist && qid && len && feat && ids_all && ld_feat >= 2 * net->dim, "mi_din_lau_fwd: null pointer / ld_feat < 2*dim");
  const DinWs d = din_carve(workspace, B, net);
  const int D = net->dim, H = net->h1, H2 = net->h2, L = net->L;
  const int64_t M = B * L;
  float* q = feat + D;  // feat = [pooled | q]
  if (mi_din_lau_fwd_fused_ok(D, H, H2, L, net->factored))
    // one launch, one workgroup per sample: the same values as the sequence below, bit for bit
    return mi_din_lau_fwd_fused(E, rows, hist, qid, idx_bits, len, B, L, net->W1f, net->Wq, net->b1.w, net->W2.w_bf16,
                                net->b2.w, net->w3.w, d.A, d.H1, d.H2, d.a, ids_all, feat, ld_feat, d.rowbias, err, stream);
  MI_TRY(mi_din_lau_build(E, rows, D, hist, qid, idx_bits, len, B, L, d.A, ids_all, q, ld_feat, d.a_cols, err, stream));
  if (net->factored) {
    // the q term of the first layer, once per sample (fp32), enters as a bias row per L positions
    MI_TRY(mi_din_qterm_fwd(q, ld_feat, net->Wq, net->b1.w, B, H, D, d.rowbias, stream));
    const mi_gemm_epilogue e = epi(MI_EPI_BIAS_ACT, MI_EPI_RELU, d.H1, H, d.rowbias, nullptr, 0, nullptr, 0, nullptr, 0, H, L);
    MI_TRY(mi_gemm_bf16(0, 0, M, H, d.a_cols, d.A, d.a_cols, net->W1f, d.a_cols, 1, &e, stream));
  } else {
    const mi_gemm_epilogue e = epi(MI_EPI_BIAS_ACT, MI_EPI_RELU, d.H1, H, net->b1.w);
    MI_TRY(mi_gemm_bf16(0, 0, M, H, d.a_cols, d.A, d.a_cols, net->W1.w_bf16, d.a_cols, 1, &e, stream));
  }
  {
    const mi_gemm_epilogue e = epi(MI_EPI_BIAS_ACT, MI_EPI_RELU, d.H2, H2, net->b2.w);
    MI_TRY(mi_gemm_bf16(0, 0, M, H2, H, d.H1, H, net->W2.w_bf16, H, 1, &e, stream));
  }
  MI_TRY(mi_head_fwd(d.H2, H2, H2, net->w3.w, net->w3.w + H2, nullptr, 0.f, M, d.a, nullptr, nullptr, stream));
  return mi_din_lau_pool(d.A, d.a_cols, d.a, ids_all, B, L, D, feat, ld_feat, stream);
}

extern "C" int mi_din_lau_bwd(const mi_din_lau* net, int64_t B, void* workspace, size_t workspace_bytes,
                              const float* feat, int64_t ld_feat, const float* dfeat, int64_t ld_dfeat,
                              const int32_t* ids_all, float lr_t, const float* lr_dev, float beta1, float beta2,
                              float eps, float* dE, void* stream) {
  MI_TRY(din_check("mi_din_lau_bwd", net, B, workspace, workspace_bytes));
  MI_REQUIRE(feat && dfeat && ids_all && dE && ld_feat >= 2 * net->dim && ld_dfeat >= 2 * net->dim,
             "mi_din_lau_bwd: null pointer / leading dimension < 2*dim");
  const DinWs d = din_carve(workspace, B, net);
  const int D = net->dim, H = net->h1, H2 = net->h2, L = net->L;
  const int64_t M = B * L;
  const Adam a{lr_t, lr_dev, beta1, beta2, eps, true};
  const float* q = feat + D;
  const float* dq_above = dfeat + D;  // what the layers above send to q directly
  if (mi_din_lau_fwd_fused_ok(D, H, H2, L, net->factored)) {
    // The data path in one launch per sample (dA and the weights' gradient da never reach HBM), then the weight
    // steps from what it left: dZ2, dZ1, the per-sample sums S, and the head's / second bias' gradients as
    // `part_slices` partial rows [w3 (H2 + 4) | b2 (H2)].
    const int P = mi_din_lau_bwd_part_floats();
    MI_TRY(mi_din_lau_bwd_fused(B, L, d.A, d.H1, d.H2, d.a, ids_all, feat, ld_feat, dfeat, ld_dfeat, net->w3.w,
                                net->W2.w_bf16_t, net->W1fT, d.dH2, d.dZ1, dE, d.S, d.part, d.part_slices, d.part_sums,
                                stream));
    MI_TRY(mi_slab_adam(d.part_sums, d.part_slices, H2 + 4, P, net->w3.w, net->w3.m, net->w3.v, a.lr_t, a.lr_dev, a.b1, a.b2,
                        a.eps, nullptr, net->w3.grad_out, stream));
    MI_TRY(weight_step(net->W2, H2, H, d.dH2, H2, d.H1, H, M, d.slab, a, stream));
    MI_TRY(mi_slab_adam(d.part_sums + H2 + 4, d.part_slices, H2, P, net->b2.w, net->b2.m, net->b2.v, a.lr_t, a.lr_dev, a.b1,
                        a.b2, a.eps, nullptr, net->b2.grad_out, stream));
    // d q (the candidate's row of dE, left as dq_above + sum_j k_j * gp) += S Wq; partials of q^T S (-> d Wa), sums of S (-> d b1)
    float* dq_rows = dE + (int64_t) L * D;
    MI_TRY(mi_din_qterm_bwd(q, ld_feat, d.S, net->Wq, B, H, D, dq_rows, (int64_t) (L + 1) * D, dq_rows, (int64_t) (L + 1) * D,
                            d.qpart, stream));
    const int64_t stride = (int64_t) D * H + H;
    const int chunks = (int) (mi_din_qterm_part_floats(B, H, D) / stride);
    MI_TRY(mi_slab_adam(d.qpart, chunks, stride, stride, nullptr, nullptr, nullptr, 0.f, nullptr, 0.f, 0.f, 0.f, nullptr, d.gq, stream));
    MI_TRY(mi_slab_adam(d.gq + (size_t) D * H, 1, H, H, net->b1.w, net->b1.m, net->b1.v, a.lr_t, a.lr_dev, a.b1, a.b2, a.eps,
                        nullptr, net->b1.grad_out, stream));
    const int s = wgrad_splits(H, d.a_cols, M);
    {
      const mi_gemm_epilogue e = epi(MI_EPI_SLAB, 0, d.slab, d.a_cols);
      MI_TRY(mi_gemm_bf16(1, 1, H, d.a_cols, M, d.dZ1, H, d.A, d.a_cols, s, &e, stream));
    }
    MI_TRY(mi_slab_adam(d.slab, s, (int64_t) H * d.a_cols, (int64_t) H * d.a_cols, nullptr, nullptr, nullptr, 0.f, nullptr, 0.f,
                        0.f, 0.f, nullptr, d.gWf, stream));
    return mi_din_w1_step(d.gWf, 1, (int64_t) H * d.a_cols, d.gq, 1, stride, H, D, net->W1.w, net->W1.m, net->W1.v, a.lr_t,
                          a.lr_dev, a.b1, a.b2, a.eps, net->W1f, net->W1fT, net->Wq, net->WqT, net->W1.grad_out, stream);
  }
  // the unit, output side first; data gradients with the OLD weights, then the weight steps
  MI_TRY(mi_din_lau_pool_bwd(d.A, d.a_cols, dfeat, ld_dfeat, ids_all, B, L, D, d.da, stream));
  MI_HIP(hipMemsetAsync(d.slab, 0, (size_t) d.vec_slices * (H2 + 4) * 4, (hipStream_t) stream));
  MI_TRY(mi_head_bwd(d.da, net->w3.w, d.H2, H2, H2, M, d.dH2, H2, d.vec_slices, d.slab, H2 + 4, stream));
  MI_TRY(mi_slab_adam(d.slab, d.vec_slices, H2 + 4, H2 + 4, net->w3.w, net->w3.m, net->w3.v, a.lr_t, a.lr_dev, a.b1, a.b2,
                      a.eps, nullptr, net->w3.grad_out, stream));
  {
    const mi_gemm_epilogue e = epi(MI_EPI_MASK, 0, d.dZ1, H, nullptr, d.H1, H);
    MI_TRY(mi_gemm_bf16(0, 0, M, H, H2, d.dH2, H2, net->W2.w_bf16_t, H2, 1, &e, stream));
  }
  MI_TRY(weight_step(net->W2, H2, H, d.dH2, H2, d.H1, H, M, d.slab, a, stream));
  MI_TRY(bias_step(net->b2, H2, d.dH2, H2, M, d.vec_slices, d.slab, a, stream));
  if (!net->factored) {
    const mi_gemm_epilogue e = epi(MI_EPI_MASK, 0, d.dA, d.a_cols);
    MI_TRY(mi_gemm_bf16(0, 0, M, d.a_cols, H, d.dZ1, H, net->W1.w_bf16_t, H, 1, &e, stream));
    MI_TRY(weight_step(net->W1, H, d.a_cols, d.dZ1, H, d.A, d.a_cols, M, d.slab, a, stream));
    MI_TRY(bias_step(net->b1, H, d.dZ1, H, M, d.vec_slices, d.slab, a, stream));
    return mi_din_lau_scatter(d.A, d.dA, d.a, dfeat, ld_dfeat, dq_above, ld_dfeat, ids_all, B, L, D, dE, stream);
  }
  // factored first layer: S[b] = sum of dZ1 over the sample's positions (column sums, one slice per sample)
  MI_TRY(mi_colsum_bf16(d.dZ1, H, M, H, (int) B, d.S, H, stream));
  // d q += S Wq; per 32 samples a partial of q^T S (-> d Wa) and of the column sums of S (-> d b1)
  MI_TRY(mi_din_qterm_bwd(q, ld_feat, d.S, net->Wq, B, H, D, dq_above, ld_dfeat, d.dq, D, d.qpart, stream));
  const int64_t stride = (int64_t) D * H + H;
  const int chunks = (int) (mi_din_qterm_part_floats(B, H, D) / stride);
  MI_TRY(mi_slab_adam(d.qpart, chunks, stride, stride, nullptr, nullptr, nullptr, 0.f, nullptr, 0.f, 0.f, 0.f, nullptr, d.gq, stream));
  MI_TRY(mi_slab_adam(d.gq + (size_t) D * H, 1, H, H, net->b1.w, net->b1.m, net->b1.v, a.lr_t, a.lr_dev, a.b1, a.b2, a.eps,
                      nullptr, net->b1.grad_out, stream));
  {
    const mi_gemm_epilogue e = epi(MI_EPI_MASK, 0, d.dA, d.a_cols);
    MI_TRY(mi_gemm_bf16(0, 0, M, d.a_cols, H, d.dZ1, H, net->W1fT, H, 1, &e, stream));
  }
  const int s = wgrad_splits(H, d.a_cols, M);
  {
    const mi_gemm_epilogue e = epi(MI_EPI_SLAB, 0, d.slab, d.a_cols);
    MI_TRY(mi_gemm_bf16(1, 1, H, d.a_cols, M, d.dZ1, H, d.A, d.a_cols, s, &e, stream));
  }
  MI_TRY(mi_slab_adam(d.slab, s, (int64_t) H * d.a_cols, (int64_t) H * d.a_cols, nullptr, nullptr, nullptr, 0.f, nullptr, 0.f,
                      0.f, 0.f, nullptr, d.gWf, stream));
  MI_TRY(mi_din_w1_step(d.gWf, 1, (int64_t) H * d.a_cols, d.gq, 1, stride, H, D, net->W1.w, net->W1.m, net->W1.v, a.lr_t,
                        a.lr_dev, a.b1, a.b2, a.eps, net->W1f, net->W1fT, net->Wq, net->WqT, net->W1.grad_out, stream));
  return mi_din_lau_scatter_fact(d.A, d.dA, d.a, dfeat, ld_dfeat, d.dq, D, q, ld_feat, ids_all, B, L, D, dE, stream);
}
