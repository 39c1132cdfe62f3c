// Residual-stage step driver (host code): enqueues every kernel of one X3D residual stage, forward or backward,
// from ONE C call (reference model/x3d.py:331-412 ResStage / ResBlock / BottleneckBlock, driven by
// `self.x3d.blocks[i](x)` at reference model/trainer.py:126-139).
//
// Why it exists: a B=32 BCD step is ~1000 kernel launches; issued one by one through Python + ctypes they cost
// 19-22 ms of host time per step (round-1 measurement), a floor the kernels had almost reached.  Here the
// per-block launch sequence runs in C++ (sub-microsecond argument marshalling), the activations of a whole stage
// live in ONE workspace carved by a deterministic plan (no allocator calls between kernels), the f64 statistics
// accumulators of the stage are zeroed by one memset, and the leaf gradients (weight gradients) go to an internal
// side stream forked / joined with events.
#include "../../include/change3d_hip.h"
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "common.h"
#include "launch_hints.h"
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

bool c3d_detail_pw_gemm_wg_supported(int Kp, int Np, int wg_mode);   // pw_gemm_wg.hip: the fused kernel's own LDS plan
bool c3d_detail_pw_gemm_masksum_supported(int Kp, int Np);           // pw_gemm_wg.hip: C3D_WG_MASKSUM

namespace {

inline int cpad(int c) { return (c + 7) / 8 * 8; }
inline size_t al(size_t v) { return (v + 255) / 256 * 256; }
inline size_t es(int dtype) { return dtype == C3D_DT_F32 ? 4 : 2; }
constexpr int S = C3D_STAT_STRIPES;
enum { SC_NONE = 0, SC_IDENTITY = 1, SC_BN = 2, SC_RAW = 3 };

#define RC(call)              \
  do {                        \
    const int rc_ = (call);   \
    if (rc_ != 0) return rc_; \
  } while (0)
#define HIPRC(call)                            \
  do {                                         \
    const hipError_t e_ = (call);              \
    if (e_ != hipSuccess) return (int)e_;      \
  } while (0)

// ------------------------------------------------------------------------------------------ workspace plan
struct BlkGeom {
  int H, W, Ho, Wo;
  int64_t M, Mo;   // rows in / out
  int Cin, Ci, Co, Cinp, Cip, Cop, s, Cr;
  bool se, sc_conv, sc_bn;
};

struct BlkFwd {   // byte offsets into ws_fwd
  size_t a, b, c, sc, y;                                        // activations (y: SIZE_MAX for the last block)
  size_t ss_a, mr_a, ss_b, mr_b, gate, hid, ss_c, mr_c, ss_1, mr_1;   // f32 vectors
  size_t sums_a, nc_b, sums_c, sums_1;                          // f64 accumulators
  size_t tick;                                                  // u32 [4] last-workgroup tickets (a, c, shortcut)
  // pointwise weights as LDS images (c3d_pw_pack_weights): forward orientation and transposed (data gradient);
  // SIZE_MAX where the narrow GEMM kernel does not take the shape
  size_t img_a, img_at, img_c, img_ct, img_s, img_st;
};

// Ring depth of the backward temporaries: block i shares its slot with block i+R, so the side stream (weight gradients)
// may run up to R-1 blocks behind the data-gradient chain before the main stream has to wait for it.
#ifndef C3D_BWD_RING_DEFAULT
#define C3D_BWD_RING_DEFAULT 3
#endif
constexpr int BWD_RING_MAX = 8;
int bwd_ring() {
  static const int r = [] {
    const char* s = c3d_env("C3D_BWD_RING");
    // measured on MI355X (B=32 bf16): 2, 3, 4 slots -> 34.04 / 34.10 / 34.32 ms per step before the weight gradients were
    // forked ahead of their data gradients; 32.62 / 32.45 ms for 2 / 3 slots after (three interleaved repeats each)
    const int v = s ? atoi(s) : C3D_BWD_RING_DEFAULT;
    return v < 2 ? 2 : (v > BWD_RING_MAX ? BWD_RING_MAX : v);
  }();
  return r;
}

struct BlkBwd {   // byte offsets into ws_bwd (ring slot for the big tensors)
  size_t g, t1, t2, dxs, dx;
  size_t coef_c, coef_1, coef_a, cA, cC, cB;                   // f32 vectors
  size_t dsums_c, dsums_1, nc3, dsums_a;                        // f64 accumulators
  size_t tick;                                                  // u32 [4] last-workgroup tickets (c (+shortcut), a)
};

struct Plan {
  std::vector<BlkGeom> g;
  std::vector<BlkFwd> f;
  std::vector<BlkBwd> b;
  size_t fwd_acc_off = 0, fwd_acc_bytes = 0, fwd_total = 0;
  size_t bwd_acc_off = 0, bwd_acc_bytes = 0, bwd_total = 0, wgrad_ws = 0, wgrad_ws2 = 0, wgrad_ws_fused = 0, wgrad_ws_fused_slot = 0;
  size_t y_bytes = 0, dx_bytes = 0;
};

struct Carver {
  size_t off = 0;
  size_t take(size_t bytes) { const size_t o = off; off += al(bytes); return o; }
};

// pointwise weight gradient inside the data-gradient launch (c3d_pw_args.wg_mode; csrc/pw_gemm_impl.h): bf16 layers whose
// accumulator image fits in LDS beside the tiles -- the res2 / res3 shapes (K, N <= 112 padded)
inline bool fuse_wgrad(const c3d_stage_desc* d, int Kp, int Np, int wg_mode) {
  return !(d->flags & C3D_STAGE_SEPARATE_WGRAD) && d->dtype == C3D_DT_BF16 && Kp <= 112 && Np <= 112 && c3d_knob("C3D_PW_WG", 1) &&
         c3d_detail_pw_gemm_wg_supported(Kp, Np, wg_mode);
}

int g_mask_in_dgrad = 3;   // c3d_set_option(C3D_OPT_MASK_IN_DGRAD, ...): bit 0 = the ReLU mask of the previous block's output in the conv_a data
                           // gradient, bit 1 = that block's BatchNorm_c-backward sums there too (no c3d_block_out_bwd launch at all)
int g_fold_se = 1;         // c3d_set_option(C3D_OPT_FOLD_SE, ...): SE gate computed by conv_c's workgroups (forward)
int g_fuse_wgrad = 3;      // c3d_set_option(C3D_OPT_FUSE_WGRAD, ...): bit 0 conv_a, bit 1 conv_c
int g_wgrad_chain = 1;     // c3d_set_option(C3D_OPT_PW_WGRAD_V2, value): bit 1 clear = chained reduction of the separate weight gradients

int make_plan(const c3d_stage_desc* d, Plan& P) {
  if (!d || d->n_blocks <= 0 || !d->blocks || d->B <= 0 || d->T <= 0 || d->H <= 0 || d->W <= 0) return C3D_E_BADARG;
  if (d->dtype != C3D_DT_F32 && d->dtype != C3D_DT_BF16) return C3D_E_BADARG;
  const size_t e = es(d->dtype);
  const int n = d->n_blocks;
  P.g.resize(n); P.f.resize(n); P.b.resize(n);
  int H = d->H, W = d->W;
  for (int i = 0; i < n; ++i) {
    const c3d_block_desc& k = d->blocks[i];
    if (k.cin <= 0 || k.cinner <= 0 || k.cout <= 0 || (k.stride != 1 && k.stride != 2)) return C3D_E_BADARG;
    if (i > 0 && k.cin != d->blocks[i - 1].cout) return C3D_E_BADARG;
    if (!k.has_sc_conv && (k.cin != k.cout || k.stride != 1)) return C3D_E_BADARG;
    if (k.has_sc_bn && !k.has_sc_conv) return C3D_E_BADARG;
    BlkGeom& G = P.g[i];
    G.H = H; G.W = W; G.s = k.stride;
    G.Ho = (H - 1) / k.stride + 1; G.Wo = (W - 1) / k.stride + 1;
    G.M = (int64_t)d->B * d->T * H * W; G.Mo = (int64_t)d->B * d->T * G.Ho * G.Wo;
    G.Cin = k.cin; G.Ci = k.cinner; G.Co = k.cout;
    G.Cinp = cpad(k.cin); G.Cip = cpad(k.cinner); G.Cop = cpad(k.cout);
    G.se = k.se_width > 0; G.Cr = k.se_width; G.sc_conv = k.has_sc_conv != 0; G.sc_bn = k.has_sc_bn != 0;
    // The narrow pointwise kernels (channel counts up to 224) address rows with 32-bit byte offsets into bounds-checked buffer
    // resources (csrc/pw_gemm.hip): a tensor of 2 GiB or more would be refused by the first launch that meets it, in the MIDDLE
    // of a stage pass.  Refuse the stage here instead -- c3d_stage_ws_bytes is the caller's first contact with a geometry.
    // (bf16, 256 x 256, T = 3: B <= 96 per GPU; f32: half of that.  The wide (res5) kernels have no such limit.)
    {
      const int cmax = std::max(std::max(G.Cinp, G.Cip), G.Cop);
      if (cmax <= 224 && (int64_t)std::max(G.M, G.Mo) * cmax * (int64_t)e >= ((int64_t)1 << 31)) return C3D_E_UNSUPPORTED;
    }
    H = G.Ho; W = G.Wo;
  }
  // ---- forward workspace: activations, then f32 vectors, then ONE contiguous f64 accumulator region
  Carver cf;
  for (int i = 0; i < n; ++i) {
    const BlkGeom& G = P.g[i];
    BlkFwd& F = P.f[i];
    F.a = cf.take((size_t)G.M * G.Cip * e);
    F.b = cf.take((size_t)G.Mo * G.Cip * e);
    F.c = cf.take((size_t)G.Mo * G.Cop * e);
    F.sc = G.sc_conv ? cf.take((size_t)G.Mo * G.Cop * e) : SIZE_MAX;
    F.y = i + 1 < n ? cf.take((size_t)G.Mo * G.Cop * e) : SIZE_MAX;
  }
  for (int i = 0; i < n; ++i) {
    const BlkGeom& G = P.g[i];
    BlkFwd& F = P.f[i];
    F.ss_a = cf.take(2 * G.Cip * 4); F.mr_a = cf.take(2 * G.Cip * 4);
    F.ss_b = cf.take(2 * G.Cip * 4); F.mr_b = cf.take(2 * G.Cip * 4);
    F.gate = G.se ? cf.take((size_t)d->B * G.Cip * 4) : SIZE_MAX;
    F.hid = G.se ? cf.take((size_t)d->B * G.Cr * 4) : SIZE_MAX;
    F.ss_c = cf.take(2 * G.Cop * 4); F.mr_c = cf.take(2 * G.Cop * 4);
    F.ss_1 = G.sc_bn ? cf.take(2 * G.Cop * 4) : SIZE_MAX;
    F.mr_1 = G.sc_bn ? cf.take(2 * G.Cop * 4) : SIZE_MAX;
  }
  P.fwd_acc_off = cf.off;
  for (int i = 0; i < n; ++i) {
    const BlkGeom& G = P.g[i];
    BlkFwd& F = P.f[i];
    F.sums_a = cf.take((size_t)S * 2 * G.Ci * 8);
    F.nc_b = cf.take((size_t)d->B * G.Cip * 2 * 8);
    F.sums_c = cf.take((size_t)S * 2 * G.Co * 8);
    F.sums_1 = G.sc_bn ? cf.take((size_t)S * 2 * G.Co * 8) : SIZE_MAX;
    F.tick = cf.take(16);
  }
  P.fwd_acc_bytes = cf.off - P.fwd_acc_off;
  for (int i = 0; i < n; ++i) {
    const BlkGeom& G = P.g[i];
    BlkFwd& F = P.f[i];
    auto img = [&](int Np, int Kp) -> size_t {
      const int64_t b = c3d_pw_weight_image_bytes(Np, Kp, d->dtype);
      return b > 0 ? cf.take((size_t)b) : SIZE_MAX;
    };
    F.img_a = img(G.Cip, G.Cinp); F.img_at = img(G.Cinp, G.Cip);
    F.img_c = img(G.Cop, G.Cip); F.img_ct = img(G.Cip, G.Cop);
    F.img_s = G.sc_conv ? img(G.Cop, G.Cinp) : SIZE_MAX;
    F.img_st = G.sc_conv ? img(G.Cinp, G.Cop) : SIZE_MAX;
  }
  P.fwd_total = cf.off;
  // ---- backward workspace: bwd_ring() ring slots of big temporaries (the side stream may lag the data-gradient chain
  //      by ring-1 blocks), per-block f32 coefficient vectors, one f64 accumulator region, the split-K scratch of
  //      the pointwise weight gradient
  size_t mx_g = 0, mx_t1 = 0, mx_t2 = 0, mx_dxs = 0, mx_dx = 0;
  int64_t wsf = 0, wsf_fused = 0;
  for (int i = 0; i < n; ++i) {
    const BlkGeom& G = P.g[i];
    if (fuse_wgrad(d, G.Cop, G.Cip, C3D_WG_SWISH)) wsf_fused = std::max(wsf_fused, c3d_pw_gemm_wg_ws_floats(G.Co, G.Ci));
    if (fuse_wgrad(d, G.Cip, G.Cinp, C3D_WG_ROWS)) wsf_fused = std::max(wsf_fused, c3d_pw_gemm_wg_ws_floats(G.Ci, G.Cin));
    // (the cooperative conv_a data + weight gradient, csrc/pw_cdgrad.hip: reserved whatever C3D_OPT_PW_CDG says right now)
    if (d->dtype == C3D_DT_BF16 && c3d_detail_pw_cdg_a_supported(G.Cip, G.Cinp, G.M)) wsf_fused = std::max(wsf_fused, c3d_pw_gemm_wg_ws_floats(G.Ci, G.Cin));
    if (d->dtype == C3D_DT_BF16 && c3d_detail_pw_cdg_c_supported(G.Cop, G.Cip, G.Mo, (int64_t)d->T * G.Ho * G.Wo))
      wsf_fused = std::max(wsf_fused, c3d_pw_gemm_wg_ws_floats(G.Co, G.Ci));
    mx_g = std::max(mx_g, (size_t)G.Mo * G.Cop * e);
    mx_t1 = std::max(mx_t1, (size_t)G.Mo * G.Cip * e);
    mx_t2 = std::max(mx_t2, (size_t)G.M * G.Cip * e);
    if (G.sc_conv) mx_dxs = std::max(mx_dxs, (size_t)G.Mo * G.Cinp * e);
    if (i > 0) mx_dx = std::max(mx_dx, (size_t)G.M * G.Cinp * e);
    wsf = std::max(wsf, c3d_pw_wgrad_ws_floats(G.Co, G.Ci));
    wsf = std::max(wsf, c3d_pw_wgrad_ws_floats(G.Ci, G.Cin));
    wsf = std::max(wsf, c3d_pw_wgrad_ws_floats(G.Co, G.Cin));
  }
  Carver cb;
  const int R = bwd_ring();
  size_t ring[BWD_RING_MAX][4];
  for (int r = 0; r < R; ++r) {
    ring[r][0] = cb.take(mx_g); ring[r][1] = cb.take(mx_t1); ring[r][2] = cb.take(mx_t2);
    ring[r][3] = mx_dxs ? cb.take(mx_dxs) : SIZE_MAX;
  }
  // dx of block i is dy of block i - 1 -- and, when the conv_a data gradient masked it (c3d_pw_args.wg_mask_out /
  // C3D_WG_MASKSUM), that block's g as well, which its SIDE-stream weight gradients read: one slot more than the ring, so that
  // block i - R - 1 overwrites it after the side marks of blocks >= i - 1 are joined (c3d_stage_bwd's lag rule)
  size_t ring_dx[BWD_RING_MAX + 1];
  for (int r = 0; r < R + 1; ++r) ring_dx[r] = mx_dx ? cb.take(mx_dx) : SIZE_MAX;
  P.wgrad_ws = cb.take((size_t)wsf * 4);
  P.wgrad_ws2 = cb.take((size_t)wsf * 4);   // chained weight-gradient launches alternate between the two (c3d_pw_wgrad_args.chain)
  // slot 0: the wave-private kernel's fused variant (kernel, then its reducer, on the main stream); slots 1..2n: one per
  // cooperative data + weight gradient launch of a backward pass -- their partials are reduced behind ONE fork at the end of
  // the pass (c3d_stage_bwd), not launch by launch
  P.wgrad_ws_fused = wsf_fused ? cb.take((size_t)wsf_fused * 4 * (1 + 2 * (size_t)n)) : SIZE_MAX;
  P.wgrad_ws_fused_slot = (size_t)wsf_fused * 4;
  for (int i = 0; i < n; ++i) {
    const BlkGeom& G = P.g[i];
    BlkBwd& Bk = P.b[i];
    const int r = i % R;
    Bk.g = ring[r][0]; Bk.t1 = ring[r][1]; Bk.t2 = ring[r][2]; Bk.dxs = ring[r][3];
    Bk.dx = i > 0 ? ring_dx[i % (R + 1)] : SIZE_MAX;
    Bk.coef_c = cb.take(3 * G.Cop * 4);
    Bk.coef_1 = G.sc_bn ? cb.take(3 * G.Cop * 4) : SIZE_MAX;
    Bk.coef_a = cb.take(3 * G.Cip * 4);
    Bk.cA = cb.take(G.Cip * 4); Bk.cC = cb.take(G.Cip * 4); Bk.cB = cb.take((size_t)d->B * G.Cip * 4);
  }
  P.bwd_acc_off = cb.off;
  for (int i = 0; i < n; ++i) {
    const BlkGeom& G = P.g[i];
    BlkBwd& Bk = P.b[i];
    Bk.dsums_c = cb.take(2 * G.Co * 8);
    Bk.dsums_1 = G.sc_bn ? cb.take(2 * G.Co * 8) : SIZE_MAX;
    Bk.nc3 = cb.take((size_t)d->B * G.Cip * 3 * 8);
    Bk.dsums_a = cb.take(2 * G.Ci * 8);
    Bk.tick = cb.take(16);
  }
  P.bwd_acc_bytes = cb.off - P.bwd_acc_off;
  P.bwd_total = cb.off;
  P.y_bytes = (size_t)P.g[n - 1].Mo * P.g[n - 1].Cop * e;
  P.dx_bytes = (size_t)P.g[0].M * P.g[0].Cinp * e;
  return 0;
}

// ------------------------------------------------------------------------------------------ per-launch profile
// c3d_prof_begin / c3d_prof_end (include/change3d_hip.h): every kernel the driver enqueues is bracketed by a HIP event
// pair on the stream it is launched on and billed its algorithmic bytes, so bench.py's per-kernel table and `roofline`
// block are taken through THIS launch sequence (round 2 took them through a second, Python, copy of it).
struct ProfRec { char name[64]; hipEvent_t e0, e1; double bytes; };
std::vector<ProfRec> g_prof;
int g_prof_flags = -1;   // < 0: off; bit 0: weight gradients inline on the main stream; bit 1: names carry shape / mode
char g_prof_tag[32] = "";   // detail mode: geometry of the block being enqueued, appended to names without a shape of their own

template <typename F>
int prof_call(const char* name, double bytes, hipStream_t s, F&& fn) {
  if (g_prof_flags < 0) return fn();
  ProfRec r;
  if ((g_prof_flags & 2) && g_prof_tag[0] && !std::strchr(name, '['))
    std::snprintf(r.name, sizeof(r.name), "%s[%s]", name, g_prof_tag);
  else
    std::snprintf(r.name, sizeof(r.name), "%s", name);
  r.bytes = bytes;
  HIPRC(hipEventCreate(&r.e0));
  HIPRC(hipEventCreate(&r.e1));
  HIPRC(hipEventRecord(r.e0, s));
  const int rc = fn();
  HIPRC(hipEventRecord(r.e1, s));
  g_prof.push_back(r);
  return rc;
}
inline bool prof_detail() { return g_prof_flags >= 0 && (g_prof_flags & 2); }

// ------------------------------------------------------------------------------------------ side stream
struct SideCtx {
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> pool;
  size_t next = 0;
  std::deque<std::pair<uint64_t, hipEvent_t>> marks;   // (sequence, done event) of side work not yet joined
  uint64_t seq = 0;
  hipEvent_t ev() {
    if (pool.size() < 256) {
      hipEvent_t e;
      // fork / done marks between two streams of ONE device: no timing, and no system-scope fence -- the default event makes
      // the recording queue write its caches back for the host and for peer devices at every mark (7-10 us of main-queue
      // bubble per fork in the round-5 trace, two forks per residual block); what leaves the device (the gradient all-reduce,
      // the host reading the loss) is ordered by the caller's own events / synchronisation behind c3d_side_join
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) return nullptr;
      pool.push_back(e);
      return e;
    }
    next = (next + 1) % pool.size();
    return pool[next];
  }
};

std::mutex g_mu;
SideCtx g_side[16];

int g_side_on = 1;        // c3d_set_option(C3D_OPT_SIDE_STREAM, ...)
bool side_enabled() { return g_side_on != 0 && !(g_prof_flags >= 0 && (g_prof_flags & 1)); }

SideCtx* side_ctx() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  SideCtx& c = g_side[dev];
  if (!c.side) {
    // C3D_SIDE_PRIO=1: lowest stream priority for the weight-gradient stream (A/B knob)
    static const bool low = c3d_env("C3D_SIDE_PRIO") && atoi(c3d_env("C3D_SIDE_PRIO")) == 1;
    int lo = 0, hi = 0;
    if (low && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess) {
      if (hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, lo) != hipSuccess) return nullptr;
    } else if (hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) != hipSuccess) {
      return nullptr;
    }
  }
  return &c;
}

// Run `fn(stream)` on the side stream after everything issued so far on `main` (inline when disabled).
template <typename F>
int side_run(hipStream_t main, F&& fn) {
  if (!side_enabled()) return fn(main);
  std::lock_guard<std::mutex> lk(g_mu);
  SideCtx* c = side_ctx();
  if (!c) return fn(main);
  hipEvent_t fork = c->ev();
  if (!fork) return fn(main);
  HIPRC(hipEventRecord(fork, main));
  HIPRC(hipStreamWaitEvent(c->side, fork, 0));
  c3d_side_launch = 1;           // launch hint (launch_hints.h): this kernel runs beside the data-gradient chain
  const int rc_fn = fn(c->side);
  c3d_side_launch = 0;
  RC(rc_fn);
  hipEvent_t done = c->ev();
  if (!done) return (int)hipErrorOutOfMemory;
  HIPRC(hipEventRecord(done, c->side));
  c->marks.emplace_back(++c->seq, done);
  while (c->marks.size() > 64) c->marks.pop_front();   // older work is ordered before the newer marks on the side stream
  return 0;
}

uint64_t side_mark() {
  std::lock_guard<std::mutex> lk(g_mu);
  SideCtx* c = side_ctx();
  return c ? c->seq : 0;
}

// `main` waits for the side work issued up to sequence `upto` (everything if upto == UINT64_MAX).
int side_join(hipStream_t main, uint64_t upto) {
  std::lock_guard<std::mutex> lk(g_mu);
  SideCtx* c = side_ctx();
  if (!c) return 0;
  hipEvent_t last = nullptr;
  while (!c->marks.empty() && c->marks.front().first <= upto) {
    last = c->marks.front().second;
    c->marks.pop_front();
  }
  if (last) HIPRC(hipStreamWaitEvent(main, last, 0));
  return 0;
}

// ------------------------------------------------------------------------------------------ launch helpers
struct PwCall {
  c3d_pw_args a;
  PwCall(const void* x, const float* w, void* y, int64_t M, int K, int N, int w_sn, int w_sk, int dtype) {
    std::memset(&a, 0, sizeof(a));
    a.x = x; a.w = w; a.y = y; a.M = M; a.K = K; a.Kp = cpad(K); a.N = N; a.Np = cpad(N);
    a.w_sn = w_sn; a.w_sk = w_sk; a.dtype = dtype;
    a.pro_mode = C3D_PRO_NONE; a.epi_mode = C3D_EPI_STORE; a.row_mode = C3D_ROWS_DENSE;
  }
};

struct WgCall {
  c3d_pw_wgrad_args a;
  WgCall(const void* p, const void* q, float* dw, float* ws, int64_t M, int K, int N, int dw_sn, int dw_sk, int dtype) {
    std::memset(&a, 0, sizeof(a));
    a.p = p; a.q = q; a.dw = dw; a.ws = ws; a.M = M; a.K = K; a.Kp = cpad(K); a.N = N; a.Np = cpad(N);
    a.dw_sn = dw_sn; a.dw_sk = dw_sk; a.dtype = dtype; a.q_mode = C3D_PRO_NONE; a.row_mode = C3D_ROWS_DENSE;
  }
};

// c3d_stage_desc.flags (include/change3d_hip.h): the unfused launch sequences, kept callable so that the fused ones can
// be tested bit for bit against them (tests/test_model_gpu.py) -- the default (flags = 0) is the measured-best sequence
inline bool fin_consumer(const c3d_stage_desc* d) { return !(d->flags & C3D_STAGE_SEPARATE_FINALIZE); }
inline bool fuse_residual(const c3d_stage_desc* d) { return !(d->flags & C3D_STAGE_SEPARATE_RESIDUAL); }
inline bool use_pw_img(const c3d_stage_desc* d) { return !(d->flags & C3D_STAGE_NO_WEIGHT_IMAGES); }

inline c3d_bn_fin fin_consume(const double* sums, const c3d_bn_ptrs& bn, double count, float momentum, float eps,
                              float* ss, float* mr) {
  c3d_bn_fin f;
  std::memset(&f, 0, sizeof(f));
  f.gamma = bn.gamma; f.beta = bn.beta; f.running_mean = bn.running_mean; f.running_var = bn.running_var;
  f.nbt = bn.num_batches_tracked; f.ss = ss; f.mr = mr; f.count = count; f.momentum = momentum; f.eps = eps;
  f.training = 1; f.sums = sums;
  return f;
}

// backward consumer side: the AFFINE2 prologues of the data-gradient GEMM and of the weight-gradient kernel rebuild
// A|B|C from the single-stripe sums; `accumulate` (the data-gradient GEMM only) adds dgamma / dbeta
inline c3d_bn_fin fin_coef_consume(const double* dsums, const c3d_bn_ptrs& bn, double count, const float* mr,
                                   bool accumulate) {
  c3d_bn_fin f;
  std::memset(&f, 0, sizeof(f));
  f.sums = dsums; f.gamma = bn.gamma; f.mr = const_cast<float*>(mr); f.count = count;
  if (accumulate) { f.running_mean = bn.dgamma; f.running_var = bn.dbeta; }
  return f;
}

inline char* at(void* base, size_t off) { return off == SIZE_MAX ? nullptr : reinterpret_cast<char*>(base) + off; }
template <typename T> inline T* atT(void* base, size_t off) { return reinterpret_cast<T*>(at(base, off)); }


// Zero fill of the accumulator regions with an ordinary kernel launch: hipMemsetAsync goes through the runtime's blit path,
// and the kernel trace shows a ~32 us hole in the main queue in front of every one of them (7 per BCD step = 0.22 ms;
// tools/trace_step.sh).  The regions are carved in 256-byte units, 16-byte aligned.
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* __restrict__ p, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0, 0, 0, 0);
}
int zero_fill(void* p, size_t bytes, hipStream_t st) {
  if (!bytes) return 0;
  if (((uintptr_t)p & 15) || (bytes & 15)) { HIPRC(hipMemsetAsync(p, 0, bytes, st)); return 0; }
  const size_t n16 = bytes >> 4;
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  zero_fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, st>>>(reinterpret_cast<uint4*>(p), n16);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// ------------------------------------------------------------------------------------------ eval: folded BatchNorm
// Eval-mode BatchNorm is the per-channel affine map y = x*scale + shift with scale = gamma / sqrt(running_var + eps),
// shift = beta - running_mean*scale (reference scripts/train_BCD.py:92-154 runs the model under model.eval()).
// The scale is folded into the rows of the producing convolution's weight matrix ONCE (c3d_stage_fold_bn); what is
// left of every BatchNorm is a bias that the consumer adds on operand load.  The eval forward then needs no
// statistics, no finalize launches (4-5 kernels per block instead of 7) and no saved activations (ring workspace).
__global__ void fold_bn_kernel(const float* __restrict__ w, float* __restrict__ wf, int N, int K, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var,
                               float eps, float* __restrict__ ss, int Cp) {
  const int n = blockIdx.x;
  if (n >= Cp) return;
  float sc = 0.f, sh = 0.f;
  if (n < N) {
    // same arithmetic as bn_finalize_kernel's eval branch: rstd in f64, rounded once
    const float rstd = (float)(1.0 / sqrt((double)var[n] + (double)eps));
    sc = gamma[n] * rstd;
    sh = beta[n] - mean[n] * sc;
    for (int k = threadIdx.x; k < K; k += blockDim.x) wf[(size_t)n * K + k] = w[(size_t)n * K + k] * sc;
  }
  if (threadIdx.x == 0) { ss[n] = n < N ? 1.f : 0.f; ss[Cp + n] = sh; }
}

struct BlkFold { size_t w_a, w_b, w_c, w_sc, ss_a, ss_b, ss_c, ss_1; };
struct BlkEval { size_t a, b, c, sc, y, gate, hid, nc_b; };
struct FoldPlan {
  std::vector<BlkFold> f;
  std::vector<BlkEval> e;
  size_t fold_total = 0, ws_total = 0, acc_off = 0, acc_bytes = 0;
};

int make_fold_plan(const c3d_stage_desc* d, const Plan& P, FoldPlan& Q) {
  const int n = d->n_blocks;
  const size_t e = es(d->dtype);
  Q.f.resize(n); Q.e.resize(n);
  Carver cf;
  for (int i = 0; i < n; ++i) {
    const BlkGeom& G = P.g[i];
    BlkFold& F = Q.f[i];
    F.w_a = cf.take((size_t)G.Ci * G.Cin * 4); F.w_b = cf.take((size_t)G.Ci * 27 * 4);
    F.w_c = cf.take((size_t)G.Co * G.Ci * 4);
    F.w_sc = G.sc_conv ? cf.take((size_t)G.Co * G.Cin * 4) : SIZE_MAX;
    F.ss_a = cf.take(2 * G.Cip * 4); F.ss_b = cf.take(2 * G.Cip * 4); F.ss_c = cf.take(2 * G.Cop * 4);
    F.ss_1 = G.sc_bn ? cf.take(2 * G.Cop * 4) : SIZE_MAX;
  }
  Q.fold_total = cf.off;
  size_t mx_a = 0, mx_b = 0, mx_c = 0, mx_y = 0, mx_gate = 0, mx_hid = 0, mx_nc = 0;
  for (int i = 0; i < n; ++i) {
    const BlkGeom& G = P.g[i];
    mx_a = std::max(mx_a, (size_t)G.M * G.Cip * e); mx_b = std::max(mx_b, (size_t)G.Mo * G.Cip * e);
    mx_c = std::max(mx_c, (size_t)G.Mo * G.Cop * e); mx_y = std::max(mx_y, (size_t)G.Mo * G.Cop * e);
    mx_gate = std::max(mx_gate, (size_t)d->B * G.Cip * 4); mx_hid = std::max(mx_hid, (size_t)d->B * std::max(G.Cr, 1) * 4);
    mx_nc = std::max(mx_nc, (size_t)d->B * G.Cip * 2 * 8);
  }
  Carver cw;
  const size_t a = cw.take(mx_a), b = cw.take(mx_b), c = cw.take(mx_c), sc = cw.take(mx_c);
  const size_t y0 = cw.take(mx_y), y1 = cw.take(mx_y), gate = cw.take(mx_gate), hid = cw.take(mx_hid);
  Q.acc_off = cw.off;
  for (int i = 0; i < n; ++i) {
    BlkEval& E = Q.e[i];
    E.a = a; E.b = b; E.c = c; E.sc = sc; E.y = (i & 1) ? y1 : y0; E.gate = gate; E.hid = hid;
    E.nc_b = P.g[i].se ? cw.take((size_t)d->B * P.g[i].Cip * 2 * 8) : SIZE_MAX;   // only SE blocks need the per-sample means
  }
  Q.acc_bytes = cw.off - Q.acc_off;
  Q.ws_total = cw.off;
  return 0;
}

}  // namespace

// =================================================================================================== C ABI
extern "C" int c3d_stage_ws_bytes(const c3d_stage_desc* d, int64_t* ws_fwd_bytes, int64_t* ws_bwd_bytes, int64_t* y_bytes,
                                  int64_t* dx_bytes) {
  Plan P;
  RC(make_plan(d, P));
  if (ws_fwd_bytes) *ws_fwd_bytes = (int64_t)P.fwd_total;
  if (ws_bwd_bytes) *ws_bwd_bytes = (int64_t)P.bwd_total;
  if (y_bytes) *y_bytes = (int64_t)P.y_bytes;
  if (dx_bytes) *dx_bytes = (int64_t)P.dx_bytes;
  return 0;
}

extern "C" int c3d_stage_saved(const c3d_stage_desc* d, int32_t blk, const char* name, int64_t* offset, int64_t* bytes) {
  Plan P;
  RC(make_plan(d, P));
  if (blk < 0 || blk >= d->n_blocks || !name || !offset || !bytes) return C3D_E_BADARG;
  const BlkGeom& G = P.g[blk];
  const BlkFwd& F = P.f[blk];
  const size_t e = es(d->dtype);
  size_t off = SIZE_MAX, n = 0;
  if (!strcmp(name, "a")) { off = F.a; n = (size_t)G.M * G.Cip * e; }
  else if (!strcmp(name, "b")) { off = F.b; n = (size_t)G.Mo * G.Cip * e; }
  else if (!strcmp(name, "c")) { off = F.c; n = (size_t)G.Mo * G.Cop * e; }
  else if (!strcmp(name, "sc")) { off = F.sc; n = (size_t)G.Mo * G.Cop * e; }
  else if (!strcmp(name, "mr_a")) { off = F.mr_a; n = 2 * G.Cip * 4; }
  else if (!strcmp(name, "mr_b")) { off = F.mr_b; n = 2 * G.Cip * 4; }
  else if (!strcmp(name, "mr_c")) { off = F.mr_c; n = 2 * G.Cop * 4; }
  else if (!strcmp(name, "mr_sc")) { off = F.mr_1; n = 2 * G.Cop * 4; }
  else if (!strcmp(name, "ss_a")) { off = F.ss_a; n = 2 * G.Cip * 4; }
  else if (!strcmp(name, "ss_b")) { off = F.ss_b; n = 2 * G.Cip * 4; }
  else if (!strcmp(name, "ss_c")) { off = F.ss_c; n = 2 * G.Cop * 4; }
  else if (!strcmp(name, "ss_sc")) { off = F.ss_1; n = 2 * G.Cop * 4; }
  else if (!strcmp(name, "gate")) { off = F.gate; n = (size_t)d->B * G.Cip * 4; }
  if (off == SIZE_MAX) return C3D_E_BADARG;
  *offset = (int64_t)off; *bytes = (int64_t)n;
  return 0;
}

extern "C" int c3d_side_join(void* stream) { return side_join(reinterpret_cast<hipStream_t>(stream), UINT64_MAX); }

// ---- profiled launches -------------------------------------------------------------------------------------
namespace {

int pw_launch(const c3d_pw_args& a, hipStream_t st) {
  const double bytes = (double)a.M * ((double)a.Kp * (a.x2 ? 2 : 1) + (double)a.Np * (a.e1 ? 2 : 1) + (a.pro_out ? a.Kp : 0) +
                                     ((a.wg_mode == C3D_WG_ROWS || a.wg_mode == C3D_WG_MASKSUM) ? a.Np : 0) + (a.add_sums ? a.Np : 0)) * (double)es(a.dtype);
  char nm[64];
  if (prof_detail())
    std::snprintf(nm, sizeof(nm), "c3d_pw_gemm[M=%lld K=%d N=%d pro=%d epi=%d rows=%d%s]", (long long)a.M, a.K, a.N, a.pro_mode,
                  a.epi_mode, a.row_mode, a.wg_mode == C3D_WG_MASKSUM ? " +bob" : a.wg_mode ? (a.add_sums ? " +dW +bob" : " +dW") : "");
  else
    std::snprintf(nm, sizeof(nm), "c3d_pw_gemm");
  return prof_call(nm, bytes, st, [&] { return c3d_pw_gemm(&a, st); });
}

int wg_launch(const c3d_pw_wgrad_args& a, hipStream_t st) {
  const double bytes = (double)a.M * ((double)a.Np * (a.p2 ? 2 : 1) + (double)a.Kp) * (double)es(a.dtype);
  char nm[64];
  if (prof_detail())
    std::snprintf(nm, sizeof(nm), "c3d_pw_wgrad[M=%lld K=%d N=%d q=%d rows=%d]", (long long)a.M, a.K, a.N, a.q_mode, a.row_mode);
  else
    std::snprintf(nm, sizeof(nm), "c3d_pw_wgrad");
  return prof_call(nm, bytes, st, [&] { return c3d_pw_wgrad(&a, st); });
}

}  // namespace

extern "C" int c3d_stage_fwd(const c3d_stage_desc* d, const void* x, void* ws, void* y_out, void* stream) {
  Plan P;
  RC(make_plan(d, P));
  if (!x || !ws || !y_out) return C3D_E_BADARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int dt = d->dtype, tr = d->training ? 1 : 0, B = d->B, T = d->T;
  const double e = (double)es(dt);
  RC(zero_fill(at(ws, P.fwd_acc_off), P.fwd_acc_bytes, st));
  // Weight images of the whole stage in one launch per 64 images: the f32 master weights change once per optimizer
  // step, the four (six with a shortcut convolution) GEMMs of a block read them in ~256 workgroups each.  The backward
  // pass of this forward reads the transposed images from the same workspace.
  const bool wimg = use_pw_img(d);
  auto imgp = [&](size_t off) -> const void* { return wimg && off != SIZE_MAX ? at(ws, off) : nullptr; };
  if (wimg) {
    std::vector<c3d_pw_pack_desc> pk;
    auto add = [&](const float* w, size_t off, int N, int K, int sn, int sk) {
      if (off != SIZE_MAX) pk.push_back(c3d_pw_pack_desc{w, at(ws, off), N, cpad(N), K, cpad(K), sn, sk});
    };
    for (int i = 0; i < d->n_blocks; ++i) {
      const c3d_block_desc& k = d->blocks[i];
      const BlkGeom& G = P.g[i];
      const BlkFwd& F = P.f[i];
      add(k.w_a, F.img_a, G.Ci, G.Cin, G.Cin, 1);
      add(k.w_c, F.img_c, G.Co, G.Ci, G.Ci, 1);
      if (G.sc_conv) add(k.w_sc, F.img_s, G.Co, G.Cin, G.Cin, 1);
      add(k.w_a, F.img_at, G.Cin, G.Ci, 1, G.Cin);
      add(k.w_c, F.img_ct, G.Ci, G.Co, 1, G.Ci);
      if (G.sc_conv) add(k.w_sc, F.img_st, G.Cin, G.Co, 1, G.Cin);
    }
    RC(prof_call("c3d_pw_pack_weights", 0.0, st, [&] { return c3d_pw_pack_weights(pk.data(), (int32_t)pk.size(), dt, st); }));
  }
  const int epi = tr ? C3D_EPI_STATS : C3D_EPI_STORE;
  const void* cur = x;
  // Residual add of block i fused into conv_a of block i+1 (c3d_pw_args.pro_out): pending operands of block i
  struct Pending { const void* c; const void* sc; c3d_bn_fin fin; void* y; bool on; } pend;
  std::memset(&pend, 0, sizeof(pend));
  for (int i = 0; i < d->n_blocks; ++i) {
    const c3d_block_desc& k = d->blocks[i];
    const BlkGeom& G = P.g[i];
    if (prof_detail()) std::snprintf(g_prof_tag, sizeof(g_prof_tag), "H=%d Ci=%d s=%d se=%d", G.H, G.Ci, G.s, (int)G.se);
    const BlkFwd& F = P.f[i];
    void* a = at(ws, F.a); void* b = at(ws, F.b); void* c = at(ws, F.c); void* sc = at(ws, F.sc);
    void* y = F.y == SIZE_MAX ? y_out : at(ws, F.y);
    float* ss_a = atT<float>(ws, F.ss_a); float* mr_a = atT<float>(ws, F.mr_a);
    float* ss_b = atT<float>(ws, F.ss_b); float* mr_b = atT<float>(ws, F.mr_b);
    float* ss_c = atT<float>(ws, F.ss_c); float* mr_c = atT<float>(ws, F.mr_c);
    float* ss_1 = atT<float>(ws, F.ss_1); float* mr_1 = atT<float>(ws, F.mr_1);
    float* gate = atT<float>(ws, F.gate); float* hid = atT<float>(ws, F.hid);
    double* sums_a = atT<double>(ws, F.sums_a); double* nc_b = atT<double>(ws, F.nc_b);
    double* sums_c = atT<double>(ws, F.sums_c); double* sums_1 = atT<double>(ws, F.sums_1);
    const int64_t rps = (int64_t)T * G.Ho * G.Wo;
    // conv_a (1x1x1) + BN_a statistics
    if (pend.on) {   // y(i-1) = relu(bn_c(c) + shortcut) computed on load, written out, and fed to the GEMM
      PwCall p(pend.c, k.w_a, a, G.M, G.Cin, G.Ci, G.Cin, 1, dt);
      p.a.x2 = pend.sc; p.a.pro_mode = C3D_PRO_AFFINE2; p.a.fin = pend.fin; p.a.pro_p = pend.fin.ss; p.a.pro_out = pend.y;
      p.a.epi_mode = epi; p.a.stats = sums_a; p.a.w_img = imgp(F.img_a);
      RC(pw_launch(p.a, st));
      pend.on = false;
    } else {
      PwCall p(cur, k.w_a, a, G.M, G.Cin, G.Ci, G.Cin, 1, dt);
      p.a.epi_mode = epi; p.a.stats = sums_a; p.a.w_img = imgp(F.img_a);
      RC(pw_launch(p.a, st));
    }
    // conv_b (depthwise 3x3x3, BN_a + ReLU on load) + per-sample statistics; BN_b + SE.  BN_a is finalised by the
    // depthwise kernel's own prologue (or by a separate launch: eval mode, C3D_FIN_CONSUMER=0)
    const bool cons = tr && fin_consumer(d);
    const double dw_bytes = ((double)G.M + (double)G.Mo) * G.Cip * e;
    if (cons) {
      const c3d_bn_fin fa = fin_consume(sums_a, k.bn_a, (double)G.M, d->momentum, d->eps, ss_a, mr_a);
      RC(prof_call("c3d_dw333_fwd", dw_bytes, st, [&] {
        return c3d_dw333_fwd_fin(a, &fa, k.w_b, b, nc_b, B, T, G.H, G.W, G.Ci, G.Cip, G.s, dt, st); }));
    } else {
      RC(prof_call("c3d_bn_finalize", 0.0, st, [&] {
        return c3d_bn_finalize(sums_a, S, (double)G.M, k.bn_a.gamma, k.bn_a.beta, k.bn_a.running_mean, k.bn_a.running_var,
                               tr ? k.bn_a.num_batches_tracked : nullptr, d->momentum, d->eps, G.Ci, G.Cip, tr, ss_a, mr_a, st); }));
      RC(prof_call("c3d_dw333_fwd", dw_bytes, st, [&] {
        return c3d_dw333_fwd(a, ss_a, k.w_b, b, nc_b, B, T, G.H, G.W, G.Ci, G.Cip, G.s, dt, st); }));
    }
    // BatchNorm_b (+ the SqueezeExcitation gate).  Blocks WITHOUT SE (every odd block) need only the batch statistics:
    // conv_c's prologue rebuilds scale / shift from the per-sample sums itself (csrc/bn_fin.h bn_consume_nc; narrow
    // kernel) -- one single-workgroup launch less on the forward critical path per such block
    // (blocks WITH SE since round 4: every conv_c workgroup also computes the gate of its samples, c3d_pw_args.se_w1)
    const bool fold_b = cons && G.Cip <= 224 && G.Cop <= 224 && (!G.se || (g_fold_se && G.Cr <= 32));
    if (!fold_b)
    RC(prof_call("c3d_bn_se_finalize", 0.0, st, [&] {
      return c3d_bn_se_finalize(nc_b, B, (double)rps, k.bn_b.gamma, k.bn_b.beta, k.bn_b.running_mean, k.bn_b.running_var,
                                tr ? k.bn_b.num_batches_tracked : nullptr, d->momentum, d->eps, G.Ci, G.Cip, tr,
                                G.se ? k.se_w1 : nullptr, k.se_b1, k.se_w2, k.se_b2, G.Cr, ss_b, mr_b, gate, hid, st); }));
    // conv_c (BN_b * SE gate, Swish on load) + BN_c statistics
    {
      PwCall p(b, k.w_c, c, G.Mo, G.Ci, G.Co, G.Ci, 1, dt);
      p.a.pro_mode = C3D_PRO_BN_SE_SWISH; p.a.pro_p = ss_b; p.a.pro_gate = gate; p.a.rows_per_sample = rps;
      if (fold_b) {
        p.a.fin = fin_consume(nc_b, k.bn_b, (double)rps * B, d->momentum, d->eps, ss_b, mr_b);
        p.a.fin.batch = B;
        if (G.se) { p.a.se_w1 = k.se_w1; p.a.se_b1 = k.se_b1; p.a.se_w2 = k.se_w2; p.a.se_b2 = k.se_b2; p.a.se_hid = hid; p.a.se_cr = G.Cr; }
      }
      p.a.epi_mode = epi; p.a.stats = sums_c; p.a.w_img = imgp(F.img_c);
      RC(pw_launch(p.a, st));
    }
    auto finalize = [&](const double* sums, const c3d_bn_ptrs& bn, int C, int Cp, float* ss, float* mr) {
      return prof_call("c3d_bn_finalize", 0.0, st, [&] {
        return c3d_bn_finalize(sums, S, (double)G.Mo, bn.gamma, bn.beta, bn.running_mean, bn.running_var,
                               tr ? bn.num_batches_tracked : nullptr, d->momentum, d->eps, C, Cp, tr, ss, mr, st); });
    };
    if (!cons) RC(finalize(sums_c, k.bn_c, G.Co, G.Cop, ss_c, mr_c));
    // shortcut
    int mode = SC_IDENTITY;
    const void* scp = cur;
    if (G.sc_conv) {
      PwCall p(cur, k.w_sc, sc, G.Mo, G.Cin, G.Co, G.Cin, 1, dt);
      p.a.row_mode = G.s == 2 ? C3D_ROWS_STRIDE2 : C3D_ROWS_DENSE; p.a.H = G.H; p.a.W = G.W;
      p.a.epi_mode = G.sc_bn ? epi : C3D_EPI_STORE; p.a.stats = sums_1; p.a.w_img = imgp(F.img_s);
      RC(pw_launch(p.a, st));
      if (G.sc_bn) {
        if (!cons) RC(finalize(sums_1, k.bn_sc, G.Co, G.Cop, ss_1, mr_1));
        mode = SC_BN;
      } else {
        mode = SC_RAW;
      }
      scp = sc;
    }
    // the next block's conv_a can take over this block's residual add when that block reads dense rows of y
    // (stride 1 inside a stage), its kernels are the narrow bf16 ones, and the shortcut carries no BatchNorm
    const bool fuse_next = cons && fuse_residual(d) && i + 1 < d->n_blocks && mode != SC_BN && dt == C3D_DT_BF16 &&
                           G.Cop <= 224 && P.g[i + 1].Cip <= 224 && d->blocks[i + 1].stride == 1 && !d->blocks[i + 1].has_sc_conv;
    const double bo_bytes = (double)G.Mo * G.Cop * 3 * e;
    if (fuse_next) {
      pend.c = c; pend.sc = scp; pend.y = y; pend.on = true;
      pend.fin = fin_consume(sums_c, k.bn_c, (double)G.Mo, d->momentum, d->eps, ss_c, mr_c);
    } else if (cons) {
      const c3d_bn_fin fc = fin_consume(sums_c, k.bn_c, (double)G.Mo, d->momentum, d->eps, ss_c, mr_c);
      c3d_bn_fin f1;
      if (mode == SC_BN) f1 = fin_consume(sums_1, k.bn_sc, (double)G.Mo, d->momentum, d->eps, ss_1, mr_1);
      RC(prof_call("c3d_block_out_fwd", bo_bytes, st, [&] {
        return c3d_block_out_fwd_fin(c, &fc, scp, mode == SC_BN ? &f1 : nullptr, mode, y, G.Mo, G.Co, G.Cop, dt, st); }));
    } else {
      RC(prof_call("c3d_block_out_fwd", bo_bytes, st, [&] {
        return c3d_block_out_fwd(c, ss_c, scp, ss_1, mode, y, G.Mo, G.Cop, dt, st); }));
    }
    cur = y;
  }
  return 0;
}

extern "C" int c3d_stage_bwd(const c3d_stage_desc* d, const void* x, const void* y_out, const void* dy, void* ws,
                             void* wb, void* dx_out, void* stream) {
  Plan P;
  RC(make_plan(d, P));
  if (!x || !y_out || !dy || !ws || !wb || !dx_out) return C3D_E_BADARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int dt = d->dtype, B = d->B, T = d->T;
  const double e = (double)es(dt);
  RC(zero_fill(at(wb, P.bwd_acc_off), P.bwd_acc_bytes, st));
  float* const wgws_ab[2] = {atT<float>(wb, P.wgrad_ws), atT<float>(wb, P.wgrad_ws2)};
  int wg_n = 0;   // weight-gradient launches of this call: launch k leaves its partials in workspace k & 1, launch k + 1 reduces them
  float* wgws_fused = atT<float>(wb, P.wgrad_ws_fused);
  int cdg_n = 0;   // cooperative data + weight gradient launches of this call (workspace slot 1 + cdg_n)
  struct RedJob { const float* ws; float* dw; int K, N, parts, sk, sn; };
  std::vector<RedJob> red_jobs;   // their partials: reduced on the side stream behind one fork at the end of the pass
  c3d_detail_pw_wgrad_v2_drop();   // (nothing may be pending from a call that returned early)
  const bool wimg = use_pw_img(d);   // transposed weight images written by this step's c3d_stage_fwd (training mode)
  auto imgp = [&](size_t off) -> const void* { return wimg && off != SIZE_MAX ? at(ws, off) : nullptr; };
  const void* cur_dy = dy;
  bool premasked = false;   // cur_dy is already dy * (y > 0): the conv_a data gradient of the block above stored it that way
  bool sums_done = false;   // ...and accumulated this block's BatchNorm_c-backward sums too: no c3d_block_out_bwd for it
  std::deque<uint64_t> lag;   // side-stream marks of the blocks whose ring slots are still in flight
  for (int i = d->n_blocks - 1; i >= 0; --i) {
    const c3d_block_desc& k = d->blocks[i];
    const BlkGeom& G = P.g[i];
    if (prof_detail()) std::snprintf(g_prof_tag, sizeof(g_prof_tag), "H=%d Ci=%d s=%d se=%d", G.H, G.Ci, G.s, (int)G.se);
    const BlkFwd& F = P.f[i];
    const BlkBwd& Bk = P.b[i];
    const void* xin = i == 0 ? x : (P.f[i - 1].y == SIZE_MAX ? y_out : at(ws, P.f[i - 1].y));
    const void* a = at(ws, F.a); const void* b = at(ws, F.b); const void* c = at(ws, F.c); const void* sc = at(ws, F.sc);
    const void* y = F.y == SIZE_MAX ? y_out : at(ws, F.y);
    const float* ss_a = atT<float>(ws, F.ss_a); const float* mr_a = atT<float>(ws, F.mr_a);
    const float* ss_b = atT<float>(ws, F.ss_b); const float* mr_b = atT<float>(ws, F.mr_b);
    const float* mr_c = atT<float>(ws, F.mr_c); const float* mr_1 = atT<float>(ws, F.mr_1);
    const float* gate = atT<float>(ws, F.gate); const float* hid = atT<float>(ws, F.hid);
    const double* nc_b = atT<double>(ws, F.nc_b);
    void* g = at(wb, Bk.g); void* t1 = at(wb, Bk.t1); void* t2 = at(wb, Bk.t2); void* dxs = at(wb, Bk.dxs);
    void* dx = i == 0 ? dx_out : at(wb, Bk.dx);
    float* coef_c = atT<float>(wb, Bk.coef_c); float* coef_1 = atT<float>(wb, Bk.coef_1);
    float* coef_a = atT<float>(wb, Bk.coef_a);
    float* cA = atT<float>(wb, Bk.cA); float* cC = atT<float>(wb, Bk.cC); float* cB = atT<float>(wb, Bk.cB);
    double* dsums_c = atT<double>(wb, Bk.dsums_c); double* dsums_1 = atT<double>(wb, Bk.dsums_1);
    double* nc3 = atT<double>(wb, Bk.nc3); double* dsums_a = atT<double>(wb, Bk.dsums_a);
    const int64_t rps = (int64_t)T * G.Ho * G.Wo;
    const bool scbn = G.sc_bn;
    // BatchNorm-backward coefficients rebuilt by their consumers (bf16 kernels, narrow and wide; csrc/bn_fin.h) instead of
    // c3d_bn_bwd_coef launches
    const bool consb = fin_consumer(d) && dt == C3D_DT_BF16;
    auto coef = [&](const double* dsums, double count, const c3d_bn_ptrs& bn, const float* mr, int C, int Cp, float* out) {
      return prof_call("c3d_bn_bwd_coef", 0.0, st, [&] {
        return c3d_bn_bwd_coef(dsums, 1, count, bn.gamma, mr, C, Cp, out, bn.dgamma, bn.dbeta, st); });
    };
    // ---- y = relu(bn_c(c) + shortcut)
    if (premasked && sums_done) {   // c3d_block_out_bwd of this block ran inside the conv_a data gradient of the block above
      g = const_cast<void*>(cur_dy);
    } else if (premasked) {   // the mask was applied where dy was produced (c3d_pw_args.wg_mask_out): statistics only, g IS dy
      g = const_cast<void*>(cur_dy);
      RC(prof_call("c3d_block_out_bwd", (double)G.Mo * G.Cop * (scbn ? 3 : 2) * e, st, [&] {
        return c3d_block_out_bwd(cur_dy, nullptr, c, scbn ? sc : nullptr, nullptr, mr_c, scbn ? mr_1 : nullptr, dsums_c,
                                 scbn ? dsums_1 : nullptr, G.Mo, G.Co, G.Cop, dt, st); }));
    } else
    RC(prof_call("c3d_block_out_bwd", (double)G.Mo * G.Cop * (scbn ? 5 : 4) * e, st, [&] {
      return c3d_block_out_bwd(cur_dy, y, c, scbn ? sc : nullptr, g, mr_c, scbn ? mr_1 : nullptr, dsums_c,
                               scbn ? dsums_1 : nullptr, G.Mo, G.Co, G.Cop, dt, st); }));
    if (!consb) RC(coef(dsums_c, (double)G.Mo, k.bn_c, mr_c, G.Co, G.Cop, coef_c));
    // ---- conv_c data gradient, Swish / SE backward in the epilogue; weight gradient on the side stream (it needs
    //      coef_c, not the data gradient: it is forked BEFORE the data-gradient launch)
    const bool coop_wc = (c3d_option_pw_cdg & 2) && dt == C3D_DT_BF16 && !(d->flags & C3D_STAGE_SEPARATE_WGRAD) && imgp(F.img_ct) != nullptr &&
                         c3d_detail_pw_cdg_c_supported(G.Cop, G.Cip, G.Mo, rps);
    const bool fuse_wc = coop_wc || ((g_fuse_wgrad & 2) && fuse_wgrad(d, G.Cop, G.Cip, C3D_WG_SWISH) && G.Cop <= 48);
    if (!fuse_wc)
    RC(side_run(st, [&](hipStream_t s2) {
      WgCall w(g, b, k.dw_c, wgws_ab[wg_n++ & 1], G.Mo, G.Ci, G.Co, G.Ci, 1, dt);
      w.a.chain = g_wgrad_chain;
      w.a.p2 = c; w.a.p_coef = coef_c; w.a.q_mode = C3D_PRO_BN_SE_SWISH; w.a.q_ss = ss_b; w.a.q_gate = gate;
      w.a.rows_per_sample = rps;
      if (consb) w.a.p_fin = fin_coef_consume(dsums_c, k.bn_c, (double)G.Mo, mr_c, false);
      return wg_launch(w.a, s2);
    }));
    {
      PwCall p(g, k.w_c, t1, G.Mo, G.Co, G.Ci, 1, G.Ci, dt);
      if (fuse_wc) { p.a.wg_mode = C3D_WG_SWISH; p.a.wg_dw = k.dw_c; p.a.wg_ws = wgws_fused; }
      p.a.x2 = c; p.a.pro_mode = C3D_PRO_AFFINE2; p.a.pro_p = coef_c;
      if (consb) p.a.fin = fin_coef_consume(dsums_c, k.bn_c, (double)G.Mo, mr_c, true);
      p.a.epi_mode = C3D_EPI_SWISH_SE_BWD; p.a.e1 = b; p.a.epi_p = ss_b; p.a.epi_gate = gate; p.a.epi_q = mr_b;
      p.a.stats = nc3; p.a.rows_per_sample = rps; p.a.w_img = imgp(F.img_ct);
      if (coop_wc && side_enabled()) {   // (the partials' reducer: deferred, see conv_a below)
        float* const wsk = wgws_fused + (size_t)(1 + cdg_n) * (P.wgrad_ws_fused_slot / 4);
        p.a.wg_ws = wsk;
        c3d_cdg_defer_reduce = 1; c3d_cdg_parts = 0;
        const int rcl = pw_launch(p.a, st);
        c3d_cdg_defer_reduce = 0;
        RC(rcl);
        if (c3d_cdg_parts > 0) { ++cdg_n; red_jobs.push_back({wsk, p.a.wg_dw, p.a.K, p.a.N, c3d_cdg_parts, p.a.w_sk, p.a.w_sn}); }
      } else {
        RC(pw_launch(p.a, st));
      }
    }
    // BatchNorm_b / SE backward coefficients.  Blocks without SE (stride 1 always): the fused depthwise backward kernel
    // rebuilds A | B | C from the per-sample sums in its prologue -- no coefficient launch on the critical path
    const bool fold_b = fin_consumer(d) && !G.se && G.s == 1;
    if (!fold_b)
    RC(prof_call("c3d_se_bn_bwd_coef", 0.0, st, [&] {
      return c3d_se_bn_bwd_coef(nc3, nc_b, B, (double)rps, k.bn_b.gamma, mr_b, ss_b, G.Ci, G.Cip, G.se ? k.se_w1 : nullptr,
                                k.se_w2, gate, hid, G.Cr, cA, cC, cB, k.bn_b.dgamma, k.bn_b.dbeta, k.dse_w1, k.dse_b1,
                                k.dse_w2, k.dse_b2, st); }));
    // ---- depthwise conv_b: data gradient and weight gradient in ONE pass over t1, b, a (csrc/dw_bwd_fused.hip; stride 1
    //      and the stride-2 first block of a stage, any extents)
    if (fold_b) {
      c3d_bn_fin fb;
      std::memset(&fb, 0, sizeof(fb));
      fb.sums = nc3; fb.batch = B; fb.gamma = k.bn_b.gamma; fb.mr = const_cast<float*>(mr_b); fb.count = (double)rps * B;
      fb.running_mean = k.bn_b.dgamma; fb.running_var = k.bn_b.dbeta;
      RC(prof_call("c3d_dw333_bwd_fused", ((double)G.Mo * 2 + (double)G.M * 2) * G.Cip * e, st, [&] {
        return c3d_dw333_bwd_fused_fin(t1, b, &fb, k.w_b, a, ss_a, mr_a, t2, dsums_a, k.dw_b, B, T, G.H, G.W, G.Ci, G.Cip, 1, dt, st); }));
    } else {
      RC(prof_call("c3d_dw333_bwd_fused", ((double)G.Mo * 2 + (double)G.M * 2) * G.Cip * e, st, [&] {
        return c3d_dw333_bwd_fused(t1, b, cA, cB, cC, k.w_b, a, ss_a, mr_a, t2, dsums_a, k.dw_b, B, T, G.H, G.W, G.Ci, G.Cip, G.s, dt, st); }));
    }
    if (!consb) RC(coef(dsums_a, (double)G.M, k.bn_a, mr_a, G.Ci, G.Cip, coef_a));
    // ---- shortcut branch
    const void* res = g;
    int res_mode = 0;
    if (G.sc_conv) {
      const int rm = G.s == 2 ? C3D_ROWS_STRIDE2 : C3D_ROWS_DENSE;
      PwCall p(g, k.w_sc, dxs, G.Mo, G.Co, G.Cin, 1, G.Cin, dt);
      p.a.w_img = imgp(F.img_st);
      if (scbn) {
        if (!consb) RC(coef(dsums_1, (double)G.Mo, k.bn_sc, mr_1, G.Co, G.Cop, coef_1));
        p.a.x2 = sc; p.a.pro_mode = C3D_PRO_AFFINE2; p.a.pro_p = coef_1;
        if (consb) p.a.fin = fin_coef_consume(dsums_1, k.bn_sc, (double)G.Mo, mr_1, true);
      }
      RC(pw_launch(p.a, st));
      RC(side_run(st, [&](hipStream_t s2) {
        WgCall w(g, xin, k.dw_sc, wgws_ab[wg_n++ & 1], G.Mo, G.Cin, G.Co, G.Cin, 1, dt);
        w.a.chain = g_wgrad_chain;
        if (scbn) {
          w.a.p2 = sc; w.a.p_coef = coef_1;
          if (consb) w.a.p_fin = fin_coef_consume(dsums_1, k.bn_sc, (double)G.Mo, mr_1, false);
        }
        w.a.row_mode = rm; w.a.H = G.H; w.a.W = G.W;
        return wg_launch(w.a, s2);
      }));
      res = dxs;
      res_mode = G.s == 2 ? 1 : 0;
    }
    // ---- conv_a data gradient (+ shortcut gradient in the epilogue) and weight gradient (forked first: it needs the
    //      coefficients, not the data gradient)
    // ... or fused into the data-gradient launch: the wave-private kernel's variant (K, N <= 112) or the cooperative kernel
    // (csrc/pw_cdgrad.hip: any of the three stage widths, dense shortcut gradient, packed weight image)
    const bool coop_wa = (c3d_option_pw_cdg & 1) && dt == C3D_DT_BF16 && !(d->flags & C3D_STAGE_SEPARATE_WGRAD) && res_mode == 0 &&
                         imgp(F.img_at) != nullptr && c3d_detail_pw_cdg_a_supported(G.Cip, G.Cinp, G.M);
    const bool fuse_wa = coop_wa || ((g_fuse_wgrad & 1) && fuse_wgrad(d, G.Cip, G.Cinp, C3D_WG_ROWS));
    bool mask_next = false, sums_next = false;
    if (!fuse_wa)
    RC(side_run(st, [&](hipStream_t s2) {
      WgCall w(t2, xin, k.dw_a, wgws_ab[wg_n++ & 1], G.M, G.Cin, G.Ci, G.Cin, 1, dt);
      w.a.chain = g_wgrad_chain;
      w.a.p2 = a; w.a.p_coef = coef_a;
      if (consb) w.a.p_fin = fin_coef_consume(dsums_a, k.bn_a, (double)G.M, mr_a, false);
      return wg_launch(w.a, s2);
    }));
    {
      PwCall p(t2, k.w_a, dx, G.M, G.Ci, G.Cin, 1, G.Cin, dt);
      if (fuse_wa) { p.a.wg_mode = C3D_WG_ROWS; p.a.wg_x3 = xin; p.a.wg_dw = k.dw_a; p.a.wg_ws = wgws_fused; }
      // xin is the previous block's output y: its ReLU mask goes onto dx here -- dx IS that block's g then (the dx slots
      // outlive that block's side-stream weight gradients: make_plan) -- and, unless its shortcut has a BatchNorm of its own,
      // its BatchNorm_c-backward sums are taken in the same epilogue (add_sums): no c3d_block_out_bwd launch for it.  Without
      // the fused weight gradient the same epilogue is the C3D_WG_MASKSUM kernel (res4: the 7-tile bucket).
      const bool masksum = !fuse_wa && dt == C3D_DT_BF16 && c3d_detail_pw_gemm_masksum_supported(G.Cip, G.Cinp);
      mask_next = i > 0 && (g_mask_in_dgrad & 1) && res_mode == 0 && (fuse_wa || (masksum && (g_mask_in_dgrad & 2)));
      sums_next = mask_next && (g_mask_in_dgrad & 2) && !P.g[i - 1].sc_bn;
      if (mask_next && !fuse_wa && !sums_next) mask_next = false;   // (C3D_WG_MASKSUM always sums)
      p.a.wg_mask_out = mask_next ? 1 : 0;
      if (mask_next && !fuse_wa) { p.a.wg_mode = C3D_WG_MASKSUM; p.a.wg_x3 = xin; }
      if (sums_next) {
        p.a.add_c = at(ws, P.f[i - 1].c); p.a.add_mr = atT<float>(ws, P.f[i - 1].mr_c);
        p.a.add_sums = atT<double>(wb, P.b[i - 1].dsums_c);
      }
      p.a.x2 = a; p.a.pro_mode = C3D_PRO_AFFINE2; p.a.pro_p = coef_a;
      if (consb) p.a.fin = fin_coef_consume(dsums_a, k.bn_a, (double)G.M, mr_a, true);
      p.a.epi_mode = C3D_EPI_ADD; p.a.e1 = res; p.a.res_mode = res_mode; p.a.H = G.H; p.a.W = G.W;
      p.a.w_img = imgp(F.img_at);
      if (coop_wa && side_enabled()) {
        // The cooperative kernel leaves its weight-gradient partials in a buffer of its own; ALL reducers of the pass are
        // launched behind one fork at its end.  (Per launch -- on the side stream, six rotating buffers -- every fork was a
        // barrier packet on the main queue: 11 us in front of every conv_c launch with the side queue otherwise idle,
        // profiles/r06_trace_gaps.txt; on the main stream each reducer is 5 us of the data-gradient chain.)
        float* const wsk = wgws_fused + (size_t)(1 + cdg_n) * (P.wgrad_ws_fused_slot / 4);
        p.a.wg_ws = wsk;
        c3d_cdg_defer_reduce = 1; c3d_cdg_parts = 0;
        const int rcl = pw_launch(p.a, st);
        c3d_cdg_defer_reduce = 0;
        RC(rcl);
        if (c3d_cdg_parts > 0) { ++cdg_n; red_jobs.push_back({wsk, p.a.wg_dw, p.a.K, p.a.N, c3d_cdg_parts, p.a.w_sk, p.a.w_sn}); }
      } else {
        RC(pw_launch(p.a, st));
      }
    }
    // the side stream may lag by ring-1 blocks: block i-1 reuses the ring slot of block i-1+ring
    lag.push_back(side_mark());
    if ((int)lag.size() >= bwd_ring()) { RC(side_join(st, lag.front())); lag.pop_front(); }
    cur_dy = dx;
    premasked = mask_next;
    sums_done = sums_next;
  }
  // the last chained weight gradient's partials (its own reducer launch, on the stream it ran on)
  if (wg_n || !red_jobs.empty())
    RC(side_run(st, [&](hipStream_t s2) {
      for (const RedJob& j : red_jobs) RC(c3d_detail_pw_wgrad_reduce(j.ws, j.dw, j.K, j.N, j.parts, j.sk, j.sn, s2));
      return wg_n ? c3d_pw_wgrad_flush(s2) : 0;
    }));
  return 0;
}

// ---- profile / runtime switches --------------------------------------------------------------------------------
int c3d_option_stem_mfma = 2, c3d_option_convt_mfma = 1;   // read by stem.hip / decoder.hip (launch_hints.h)
int c3d_option_dw_ring = 13;                               // read by dw_bwd_fused.hip / dw_conv.hip
int c3d_option_pw_wgrad_v2 = 1;                            // read by pw_wgrad.hip
int c3d_option_pw_cfwd = 3;                                // read by pw_gemm.hip
int c3d_option_pw_cdg = 3;                                 // read by pw_gemm.hip and c3d_stage_bwd
int c3d_option_dw_fwd_hv = 5;                              // read by dw_conv.hip

extern "C" int c3d_set_option(int32_t option, int32_t value) {
  switch (option) {
    case C3D_OPT_SIDE_STREAM: g_side_on = value ? 1 : 0; return 0;
    case C3D_OPT_STEM_MFMA: c3d_option_stem_mfma = value < 0 ? 0 : (value > 2 ? 2 : value); return 0;
    case C3D_OPT_CONVT_MFMA: c3d_option_convt_mfma = value ? 1 : 0; return 0;
    case C3D_OPT_FUSE_WGRAD: g_fuse_wgrad = value & 3; return 0;
    case C3D_OPT_FOLD_SE: g_fold_se = value ? 1 : 0; return 0;
    case C3D_OPT_MASK_IN_DGRAD: g_mask_in_dgrad = value & 3; return 0;
    case C3D_OPT_DW_RING: c3d_option_dw_ring = value & 15; return 0;
    case C3D_OPT_DW_FWD_HV: c3d_option_dw_fwd_hv = value & 7; return 0;
    case C3D_OPT_PW_CFWD: c3d_option_pw_cfwd = value & 3; return 0;
    case C3D_OPT_PW_CDG: c3d_option_pw_cdg = value & 3; return 0;
    case C3D_OPT_PW_WGRAD_V2: c3d_option_pw_wgrad_v2 = value & 1; g_wgrad_chain = (value & 2) ? 0 : 1; return 0;
    default: return C3D_E_BADARG;
  }
}

extern "C" int c3d_prof_begin(int32_t flags) {
  for (ProfRec& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_prof.clear();
  g_prof_flags = flags & 3;
  g_prof_tag[0] = 0;
  return 0;
}

extern "C" int c3d_prof_end(c3d_prof_row* rows, int32_t cap, int32_t* n_rows) {
  g_prof_flags = -1;
  if (!n_rows || (cap > 0 && !rows)) return C3D_E_BADARG;
  HIPRC(hipDeviceSynchronize());
  int n = 0;
  for (ProfRec& r : g_prof) {
    float ms = 0.f;
    HIPRC(hipEventElapsedTime(&ms, r.e0, r.e1));
    (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    int j = 0;
    while (j < n && std::strncmp(rows[j].name, r.name, sizeof(rows[j].name)) != 0) ++j;
    if (j == n) {
      if (n >= cap) continue;   // table full: the row is dropped (callers pass cap >= 256)
      std::memset(&rows[n], 0, sizeof(rows[n]));
      std::snprintf(rows[n].name, sizeof(rows[n].name), "%s", r.name);
      ++n;
    }
    rows[j].launches += 1; rows[j].ms_total += ms; rows[j].bytes_total += r.bytes;
  }
  g_prof.clear();
  *n_rows = n;
  return 0;
}

extern "C" int c3d_stage_fold_bytes(const c3d_stage_desc* d, int64_t* fold_bytes, int64_t* ws_eval_bytes) {
  Plan P;
  RC(make_plan(d, P));
  FoldPlan Q;
  RC(make_fold_plan(d, P, Q));
  if (fold_bytes) *fold_bytes = (int64_t)Q.fold_total;
  if (ws_eval_bytes) *ws_eval_bytes = (int64_t)Q.ws_total;
  return 0;
}

extern "C" int c3d_stage_fold_bn(const c3d_stage_desc* d, void* fold, void* stream) {
  Plan P;
  RC(make_plan(d, P));
  FoldPlan Q;
  RC(make_fold_plan(d, P, Q));
  if (!fold) return C3D_E_BADARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int i = 0; i < d->n_blocks; ++i) {
    const c3d_block_desc& k = d->blocks[i];
    const BlkGeom& G = P.g[i];
    const BlkFold& F = Q.f[i];
    struct Job { const float* w; size_t wf; int N, K; const c3d_bn_ptrs* bn; size_t ss; int Cp; bool on; };
    const Job jobs[4] = {{k.w_a, F.w_a, G.Ci, G.Cin, &k.bn_a, F.ss_a, G.Cip, true},
                         {k.w_b, F.w_b, G.Ci, 27, &k.bn_b, F.ss_b, G.Cip, true},
                         {k.w_c, F.w_c, G.Co, G.Ci, &k.bn_c, F.ss_c, G.Cop, true},
                         {k.w_sc, F.w_sc, G.Co, G.Cin, &k.bn_sc, F.ss_1, G.Cop, G.sc_bn}};
    for (const Job& j : jobs) {
      if (!j.on) continue;
      if (!j.w || !j.bn->gamma || !j.bn->beta || !j.bn->running_mean || !j.bn->running_var) return C3D_E_BADARG;
      fold_bn_kernel<<<dim3(j.Cp), dim3(64), 0, st>>>(j.w, atT<float>(fold, j.wf), j.N, j.K, j.bn->gamma, j.bn->beta,
                                                     j.bn->running_mean, j.bn->running_var, d->eps, atT<float>(fold, j.ss), j.Cp);
    }
    if (G.sc_conv && !G.sc_bn)   // shortcut convolution without BatchNorm (stage 1 block 0): plain copy of the weights
      HIPRC(hipMemcpyAsync(at(fold, F.w_sc), k.w_sc, (size_t)G.Co * G.Cin * 4, hipMemcpyDeviceToDevice, st));
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int c3d_stage_fwd_folded(const c3d_stage_desc* d, const void* fold_c, const void* x, void* ws, void* y_out,
                                    void* stream) {
  Plan P;
  RC(make_plan(d, P));
  FoldPlan Q;
  RC(make_fold_plan(d, P, Q));
  if (!fold_c || !x || !ws || !y_out) return C3D_E_BADARG;
  void* fold = const_cast<void*>(fold_c);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int dt = d->dtype, B = d->B, T = d->T;
  RC(zero_fill(at(ws, Q.acc_off), Q.acc_bytes, st));
  const void* cur = x;
  for (int i = 0; i < d->n_blocks; ++i) {
    const c3d_block_desc& k = d->blocks[i];
    const BlkGeom& G = P.g[i];
    const BlkFold& F = Q.f[i];
    const BlkEval& E = Q.e[i];
    void* a = at(ws, E.a); void* b = at(ws, E.b); void* c = at(ws, E.c); void* sc = at(ws, E.sc);
    void* y = i + 1 == d->n_blocks ? y_out : at(ws, E.y);
    float* ss_a = atT<float>(fold, F.ss_a); float* ss_b = atT<float>(fold, F.ss_b); float* ss_c = atT<float>(fold, F.ss_c);
    float* ss_1 = atT<float>(fold, F.ss_1);
    float* gate = G.se ? atT<float>(ws, E.gate) : nullptr;
    double* nc_b = atT<double>(ws, E.nc_b);
    const int64_t rps = (int64_t)T * G.Ho * G.Wo;
    {
      PwCall p(cur, atT<float>(fold, F.w_a), a, G.M, G.Cin, G.Ci, G.Cin, 1, dt);
      RC(c3d_pw_gemm(&p.a, st));
    }
    RC(c3d_dw333_fwd(a, ss_a, atT<float>(fold, F.w_b), b, nc_b, B, T, G.H, G.W, G.Ci, G.Cip, G.s, dt, st));
    if (G.se)   // SE gate from the per-sample means of the (already scaled) conv_b output: training = 2 -> ss is given
      RC(c3d_bn_se_finalize(nc_b, B, (double)rps, k.bn_b.gamma, k.bn_b.beta, k.bn_b.running_mean, k.bn_b.running_var, nullptr,
                            d->momentum, d->eps, G.Ci, G.Cip, 2, k.se_w1, k.se_b1, k.se_w2, k.se_b2, G.Cr, ss_b, nullptr,
                            gate, atT<float>(ws, E.hid), st));
    {
      PwCall p(b, atT<float>(fold, F.w_c), c, G.Mo, G.Ci, G.Co, G.Ci, 1, dt);
      p.a.pro_mode = C3D_PRO_BN_SE_SWISH; p.a.pro_p = ss_b; p.a.pro_gate = gate; p.a.rows_per_sample = rps;
      RC(c3d_pw_gemm(&p.a, st));
    }
    int mode = SC_IDENTITY;
    const void* scp = cur;
    if (G.sc_conv) {
      PwCall p(cur, atT<float>(fold, F.w_sc), sc, G.Mo, G.Cin, G.Co, G.Cin, 1, dt);
      p.a.row_mode = G.s == 2 ? C3D_ROWS_STRIDE2 : C3D_ROWS_DENSE; p.a.H = G.H; p.a.W = G.W;
      RC(c3d_pw_gemm(&p.a, st));
      mode = G.sc_bn ? SC_BN : SC_RAW;
      scp = sc;
    }
    RC(c3d_block_out_fwd(c, ss_c, scp, ss_1, mode, y, G.Mo, G.Cop, dt, st));
    cur = y;
  }
  return 0;
}
