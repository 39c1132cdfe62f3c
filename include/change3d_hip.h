/* change3d_hip.h — C ABI of libchange3d_hip.so (MI355X / gfx950 kernels for the Change3D
 * BCD hot path).  Plain pointers and sizes only; every pointer is a DEVICE pointer unless
 * noted; `stream` is a hipStream_t passed as void*.  Every function returns 0 on success
 * or a hipError_t / negative C3D_E* code.
 *
 * The reference (zhuduowang/Change3D) has no FFI: its boundary for this path is the Python
 * module surface of model/x3d.py, model/change_decoder.py, model/trainer.py and
 * model/utils.py.  Each entry point below therefore names the reference construct whose
 * device work it replaces (file:line into /root/reference); the Python mirror in
 * change3d_amd/model/ (Python) binds them with ctypes.
 *
 * Tensor layout: activations are channels-last [B][T][H][W][Cp], Cp = round_up(C,8), pad
 * channels zero; `dtype` selects the storage type of activations (C3D_F32 | C3D_BF16).
 * Parameters, BN statistics, gradients of parameters and optimizer state are always f32
 * (reductions accumulate in f64).
 */
#ifndef CHANGE3D_HIP_H
#define CHANGE3D_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C3D_DT_F32 0
#define C3D_DT_BF16 1

#define C3D_E_BADARG (-1)
#define C3D_E_UNSUPPORTED (-2)

/* library / device identification (host-only; safe to call without a GPU) */
int c3d_abi_version(void);
const char* c3d_build_info(void);
/* number of compute units of the current device (needs a GPU) */
int c3d_device_cus(void);

/* ------------------------------------------------------------------------------------
 * Pointwise (1x1x1 / 1x1) convolution family as a row GEMM on MFMA:
 *     Y[m, n] = epilogue( sum_k prologue(X)[m, k] * Wt[n, k] )
 * Replaces: nn.Conv3d k=1 conv_a / conv_c / branch1_conv (reference model/x3d.py:173-175,
 * 214-216, 302-308) forward and data-gradient, nn.Conv2d k=1 of the decoder
 * (reference model/change_decoder.py:30-45), with the neighbouring BatchNorm3d-apply,
 * SE gate, Swish and BN-backward arithmetic fused on operand load / result store.
 * ------------------------------------------------------------------------------------ */
#define C3D_PRO_NONE 0
#define C3D_PRO_BN_SE_SWISH 1 /* v = x*scale+shift; q = gate*v; out = q*sigmoid(q)            */
#define C3D_PRO_AFFINE2 2     /* out = A*x + B + C*x2   (BatchNorm backward applied on load)  */

#define C3D_EPI_STORE 0
#define C3D_EPI_STATS 1        /* store + per-channel sum / sum-of-squares into one of
                                  C3D_STAT_STRIPES striped f64 accumulator sets               */
#define C3D_STAT_STRIPES 16
#define C3D_EPI_SWISH_SE_BWD 2 /* t1 = result*swish'(gate*bn(e1))*gate; per-(sample,channel)
                                  sums (d gate, t1, t1*e1hat) -> stats [B][Np][3]            */
#define C3D_EPI_ADD 3          /* result + residual e1 (dense, or scattered from half res)    */

#define C3D_ROWS_DENSE 0  /* row m at x + m*Kp                                               */
#define C3D_ROWS_FRAME 1  /* row m at x + (m / rpg)*gstride + (m % rpg)*Kp (one frame of NDHWC) */
#define C3D_ROWS_STRIDE2 2 /* row m=(bt,ho,wo) gathers input pixel (bt,2ho,2wo) of [BT][H][W] */
#define C3D_ROWS_S2SHIFT 3 /* (wgrad only) as STRIDE2 but pixel (2ho+dy, 2wo+dx), zero outside */

/* In-kernel BatchNorm finalisation by the last workgroup of a statistics producer (csrc/bn_fin.h): when `ticket`
 * is non-NULL the producing kernel itself turns the completed sums into scale/shift (+ running statistics), or into
 * the BatchNorm-backward coefficients, instead of a separate c3d_bn_finalize / c3d_bn_bwd_coef launch.
 *   forward  (c3d_pw_gemm C3D_EPI_STATS)          : ss = scale|shift [2][Cp], mr = mean|rstd [2][Cp] (may be NULL)
 *   backward (c3d_block_out_bwd): ss = coefficients A|B|C [3][Cp], mr = saved mean|rstd (input),
 *              running_mean / running_var carry dgamma / dbeta (accumulated), beta / nbt unused
 * `ticket` is a zeroed uint32 (re-zeroed by the caller before every launch that uses it).                        */
typedef struct c3d_bn_fin {
  uint32_t* ticket;
  const float* gamma; const float* beta;
  float* running_mean; float* running_var;
  int64_t* nbt;
  float* ss; float* mr;
  double count;
  float momentum, eps;
  int32_t training, batch;   /* batch > 0: `sums` is the PER-SAMPLE layout of c3d_dw333_fwd / the Swish-SE-backward epilogue   */
  /* consumer side (c3d_dw333_fwd_fin, c3d_block_out_fwd_fin): the completed f64 sums [C3D_STAT_STRIPES][2][C] of an
   * EARLIER launch; every workgroup of the consuming kernel rebuilds scale/shift of the channels it reads, one
   * workgroup also writes ss / mr / the running statistics.  `ticket` is unused (NULL) in this mode.                */
  const double* sums;
  /* batch > 0 (BatchNorm_b of blocks WITHOUT SqueezeExcitation; the SE blocks keep c3d_bn_se_finalize / c3d_se_bn_bwd_coef):
   *   forward  (c3d_pw_gemm C3D_PRO_BN_SE_SWISH): sums = nc f64 [batch][Cp][2] -> scale|shift rebuilt by every workgroup in
   *             the 4-lane order of c3d_bn_se_finalize (bit-identical), workgroup 0 writes ss / mr / running statistics;
   *   backward (c3d_dw333_bwd_fused): sums = nc3 f64 [batch][Cp][3] -> A | B | C of c3d_se_bn_bwd_coef (no-SE branch),
   *             running_mean / running_var carry dgamma / dbeta (accumulated by one workgroup per channel chunk).      */
} c3d_bn_fin;

typedef struct c3d_pw_args {
  const void* x;         /* A-side tensor, storage dtype, rows of Kp elements                */
  const void* x2;        /* second A-side tensor for C3D_PRO_AFFINE2 (same addressing)       */
  void* y;               /* output [M][Np], storage dtype                                    */
  const void* e1;        /* epilogue tensor ([M][Np] dense; or half-res residual)            */
  const float* w;        /* weights, element (n,k) at w[n*w_sn + k*w_sk]                     */
  const float* pro_p;    /* BN_SE_SWISH: scale[Kp],shift[Kp]; AFFINE2: A[Kp],B[Kp],C[Kp]     */
  const float* pro_gate; /* BN_SE_SWISH: gate[B][Kp] or NULL (=1)                            */
  const float* epi_p;    /* SWISH_SE_BWD: scale[Np],shift[Np] of the BN applied to e1        */
  const float* epi_gate; /* SWISH_SE_BWD: gate[B][Np] or NULL (=1)                           */
  const float* epi_q;    /* SWISH_SE_BWD: mean[Np],rstd[Np] of that BN (centred sum t1*bhat)  */
  double* stats;         /* STATS: [C3D_STAT_STRIPES][2][N]; SWISH_SE_BWD: [B][Np][3]         */
  int64_t M;             /* output rows                                                      */
  int64_t gstride;       /* C3D_ROWS_FRAME: elements between consecutive row groups          */
  int64_t rows_per_sample; /* output rows per batch sample (gate / per-sample sums)          */
  int32_t K, Kp, N, Np;
  int32_t w_sn, w_sk;
  int32_t row_mode, rpg;
  int32_t H, W;          /* STRIDE2: input H,W.  EPI_ADD res_mode 1: output H,W              */
  int32_t pro_mode, epi_mode;
  int32_t res_mode;      /* EPI_ADD: 0 dense [M][Np]; 1 e1 is [BT][H/2][W/2][Np], added where h,w even */
  int32_t dtype;
  c3d_bn_fin fin;        /* C3D_EPI_STATS: fin.ticket != NULL -> the last workgroup finalises the BatchNorm.      */
                         /* C3D_PRO_AFFINE2: fin.sums != NULL -> A|B|C are rebuilt from the completed single-   */
                         /* stripe sums f64 [2][K] (fin.gamma, fin.mr = mean|rstd, fin.count; workgroup 0 adds    */
                         /* dgamma / dbeta into fin.running_mean / fin.running_var and writes fin.ss if non-NULL) */
                         /* instead of read from pro_p (narrow kernel only: K, N <= 224, no bias).                */
  const float* bias;     /* optional bias[N] added before the epilogue (linear layers of the caption decoder)  */
  void* pro_out;         /* C3D_PRO_AFFINE2, dense rows, narrow kernel: relu(A*x + B + C*x2) is what enters the GEMM  */
                         /* AND is written here [M][Kp] (the residual add of the previous block fused into conv_a:    */
                         /* x = c, x2 = shortcut, fin.training = 1 + fin.sums = BatchNorm_c's 16-stripe sums, C = 1)  */
  const void* w_img;     /* optional: image of `w` made by c3d_pw_pack_weights for the same (N, Np, K, Kp, w_sn, w_sk,  */
                         /* dtype).  The narrow kernel then copies it into LDS (one 16-byte DMA per lane) instead of     */
                         /* converting and scattering the f32 weights in every workgroup; same results bit for bit.     */
                         /* `w` must still be given (the block-tiled kernel reads it).                                  */
  /* ---- weight gradient fused into the DATA-gradient launch (round 4; reference model/x3d.py:173-175,214-216: autograd's
   * convolution_backward produces both gradients of a 1x1x1 convolution from one read of its operands).  For a data-gradient
   * call (x = dY-side tensor with the C3D_PRO_AFFINE2 prologue, w addressed transposed) the kernel ALSO accumulates
   *     dW[k][n] += sum_m P(m, k) * Q(m, n),   P = prologue(x, x2) (the rows already staged for the GEMM),
   *     wg_mode C3D_WG_SWISH : Q = swish(bn(e1) * gate), the forward operand the C3D_EPI_SWISH_SE_BWD epilogue rebuilds anyway
   *     wg_mode C3D_WG_ROWS  : Q = rows of wg_x3 ([M][Np] dense, storage dtype: the layer's forward input)
   * into dw[n*w_sn + k*w_sk] (the strides of `w`): per-workgroup partial sums go to wg_ws (f32, at least
   * c3d_pw_gemm_wg_ws_floats(K, N) elements) and a fixed-order reducer launched behind the kernel on the same stream adds them
   * to wg_dw.  Narrow bf16 kernel, dense rows, Kp <= 112 and Np <= 112 only (C3D_E_UNSUPPORTED otherwise: callers keep
   * c3d_pw_wgrad for those).  wg_mode 0 (the zero-initialised default): no weight gradient.                            */
  const void* wg_x3;
  float* wg_dw;
  float* wg_ws;
  int32_t wg_mode;
  int32_t wg_mask_out;   /* C3D_WG_ROWS with C3D_EPI_ADD: 1 = the stored output is dx * (wg_x3 > 0) -- the ReLU mask of the PREVIOUS   */
                         /* block's output (wg_x3 is that output), i.e. the g = dy * (y > 0) of c3d_block_out_bwd, which then */
                         /* only has to produce the BatchNorm-backward sums (its y = g = NULL form)                          */
  /* ---- SqueezeExcitation gate computed by the CONSUMER (round 4): with C3D_PRO_BN_SE_SWISH, fin.sums (per-sample sums of
   * c3d_dw333_fwd, fin.batch > 0) and se_w1 != NULL, every workgroup of the narrow kernel rebuilds BatchNorm_b's scale / shift
   * AND the SE gate of the samples its rows belong to (z = bn(mean_n), FC1 + ReLU, FC2 + sigmoid: the arithmetic and summation
   * order of c3d_bn_se_finalize, bit-identical) in its prologue -- no c3d_bn_se_finalize launch between conv_b and conv_c.  The
   * workgroup that holds a sample's first row writes gate[n][Kp] (-> pro_gate, which must be given: the backward pass reads it)
   * and se_hid[n][Cr].  When a workgroup would span more than 4 samples (tiny inputs) the entry point launches
   * c3d_bn_se_finalize itself and runs the unfused form.  Reference: fvcore SqueezeExcitation at model/x3d.py:194-202.      */
  const float* se_w1; const float* se_b1; const float* se_w2; const float* se_b2;
  float* se_hid;
  int32_t se_cr;
  int32_t se_reserved;
  /* ---- c3d_block_out_bwd of the PREVIOUS block fused into this data-gradient launch (round 5; reference model/x3d.py:
   * 229-236: y = relu(bn_c(c) + shortcut), whose backward is one masked pass over dy in autograd).  With C3D_EPI_ADD, dense
   * rows, bf16 and wg_x3 = that block's output y:
   *     wg_mode C3D_WG_ROWS + wg_mask_out, or wg_mode C3D_WG_MASKSUM (no weight gradient; any Kp <= 224, Np <= 112):
   *     the stored output is g = (result + e1) * (wg_x3 > 0), and with add_sums != NULL (required by C3D_WG_MASKSUM) the
   *     kernel also accumulates that block's BatchNorm_c-backward sums  add_sums f64 [2][N] += (sum g, sum g * chat),
   *     chat = (add_c - mean) * rstd, add_c = its conv_c output [M][Np], add_mr = mean[Np] | rstd[Np] -- the sums of
   *     c3d_block_out_bwd (g as stored, i.e. rounded to the storage type), which is then not launched at all.            */
  const void* add_c;
  const float* add_mr;
  double* add_sums;
} c3d_pw_args;
#define C3D_WG_NONE 0
#define C3D_WG_SWISH 1
#define C3D_WG_ROWS 2
#define C3D_WG_MASKSUM 3
int64_t c3d_pw_gemm_wg_ws_floats(int32_t K, int32_t N);

/* K, N <= 224 run the wave-private-tile kernel (pw_gemm_impl.h); wider layers (X3D res5: 432 inner channels; the
 * caption decoder's 192 -> 576 / vocabulary projections) or a non-NULL bias run the block-tiled kernel of
 * pw_wide.hip with the same prologues / epilogues (row modes DENSE and STRIDE2; no in-kernel finalisation).
 * The wave-private-tile kernels address rows through bounds-checked buffer resources with 32-bit byte offsets: a call
 * whose largest operand (M * max(Kp, Np) elements) reaches 2 GiB returns C3D_E_UNSUPPORTED. */
int c3d_pw_gemm(const c3d_pw_args* args, void* stream);

/* Weight images for c3d_pw_args.w_img: the narrow kernel's LDS operand layout (f32: [NT*16][Kpad+pad]; bf16: 8-element
 * k-chunk major [Kpad/8 + 1][NT*16][8], the bank-conflict-free form for ds_read_b128 A fragments; zero padded; opaque to callers) written once per weight VERSION instead of rebuilt by each of the ~256 workgroups of each launch
 * (the f32 master weights change once per optimizer step; a block's four GEMMs per step read two weights, each in two
 * orientations).  c3d_pw_weight_image_bytes returns 0 for shapes the narrow kernel does not take (Kp or Np > 224).
 * One launch packs up to C3D_PW_PACK_MAX images (descriptors travel as kernel arguments). */
typedef struct c3d_pw_pack_desc {
  const float* w;        /* element (n,k) at w[n*w_sn + k*w_sk] */
  void* img;             /* c3d_pw_weight_image_bytes(Np, Kp, dtype) bytes, 16-byte aligned */
  int32_t N, Np, K, Kp, w_sn, w_sk;
} c3d_pw_pack_desc;
#define C3D_PW_PACK_MAX 64
int64_t c3d_pw_weight_image_bytes(int32_t Np, int32_t Kp, int32_t dtype);
int c3d_pw_pack_weights(const c3d_pw_pack_desc* descs, int32_t n, int32_t dtype, void* stream);

/* Weight gradient of the same family:
 *     dW[n, k] += sum_m P(m, n) * Q(m, k),  P = prologue_p(p, p2),  Q = prologue_q(q)
 * (replaces the weight half of aten::convolution_backward for the k=1 convs above).
 * `ws` is an f32 workspace of at least c3d_pw_wgrad_ws_floats(N,K) elements.             */
typedef struct c3d_pw_wgrad_args {
  const void* p;  const void* p2;   /* dY-side operand ([M][Np] dense) + AFFINE2 companion   */
  const void* q;                    /* X-side operand, rows of Kp elements (row_mode applies)*/
  float* dw;                        /* element (n,k) accumulated at dw[n*dw_sn + k*dw_sk]    */
  float* ws;
  const float* p_coef;              /* AFFINE2 A,B,C [Np] each, or NULL (plain p)            */
  const float* q_ss;                /* BN_SE_SWISH scale,shift [Kp] each, or NULL            */
  const float* q_gate;              /* gate[B][Kp] or NULL                                   */
  int64_t M;
  int64_t gstride;
  int64_t rows_per_sample;
  int32_t K, Kp, N, Np;
  int32_t dw_sn, dw_sk;
  int32_t row_mode, rpg, H, W;      /* addressing of q (p is always dense)                   */
  int32_t dy, dx;                   /* C3D_ROWS_S2SHIFT: row m=(b,i,j) reads pixel (2i+dy, 2j+dx) */
  int32_t q_mode;                   /* C3D_PRO_NONE | C3D_PRO_BN_SE_SWISH                    */
  int32_t dtype;
  int32_t taps;                     /* C3D_ROWS_S2SHIFT: 16 = ONE launch covers the 4x4 taps of a ConvTranspose2d   */
  int32_t dw_tap_stride;            /* k4s2p1 weight gradient, (dy,dx) = (tap/4-1, tap%4-1), tap t accumulates at   */
                                    /* dw + t*dw_tap_stride (reference model/change_decoder.py:30-45); 0/1 = dy,dx */
  c3d_bn_fin p_fin;                 /* p_fin.sums != NULL: the AFFINE2 coefficients are rebuilt from the completed   */
                                    /* single-stripe sums (gamma, mr, count as in c3d_bn_bwd_coef) instead of p_coef  */
  int32_t chain;                    /* 1: the per-workgroup partials of this launch may stay PENDING in `ws` -- the next */
  int32_t reserved_;                /* chained launch on the same stream adds them into dw in its prologue (same fixed  */
                                    /* order as the reducer launch it replaces: bit-identical dw), c3d_pw_wgrad_flush    */
                                    /* completes the last one.  The caller alternates between two `ws` buffers and does  */
                                    /* not touch a pending `ws` / read `dw` before the flush.  0 (default): dw is        */
                                    /* complete, in stream order, when c3d_pw_wgrad returns                              */
} c3d_pw_wgrad_args;

int64_t c3d_pw_wgrad_ws_floats(int32_t N, int32_t K);
int c3d_pw_wgrad(const c3d_pw_wgrad_args* args, void* stream);
/* Adds the pending partials of this thread's last chained c3d_pw_wgrad launch (if any) into its dw, on `stream` (the stream of
 * that launch).  The stage driver calls it at the end of c3d_stage_bwd; a no-op when nothing is pending.                      */
int c3d_pw_wgrad_flush(void* stream);

/* ------------------------------------------------------------------------------------
 * Train-mode BatchNorm3d split (reference model/x3d.py:97,179,207,220,298): statistics are
 * accumulated by producer epilogues, finalised here, applied by consumer prologues.
 *   sums : f64 [stripes][2][C] (sum, sumsq), summed over the stripes;   ss : f32 scale[Cp], shift[Cp];   mr : f32 mean[Cp], rstd[Cp]
 * training=0 builds scale/shift from the running statistics (eval mode); c3d_bn_se_finalize also accepts
 * training=2: scale/shift are GIVEN in ss (BatchNorm already folded into the weights), only the SE gate is computed.
 * ------------------------------------------------------------------------------------ */
int c3d_bn_finalize(const double* sums, int32_t stripes, double count, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float momentum, float eps, int32_t C, int32_t Cp, int32_t training, float* ss,
                    float* mr, void* stream);
/* BN_b + SqueezeExcitation (fvcore SqueezeExcitation called at reference model/x3d.py:194-202):
 * nc is f64 [B][Cp][2] per-(sample,channel) sum / sumsq from c3d_dw333_fwd; gate is f32 [B][Cp];
 * hid is f32 [B][Cr] (saved ReLU output of the first FC).  w1 == NULL -> block without SE.  */
int c3d_bn_se_finalize(const double* nc, int32_t B, double cnt_per_sample, const float* gamma,
                       const float* beta, float* running_mean, float* running_var,
                       int64_t* num_batches_tracked, float momentum, float eps, int32_t C, int32_t Cp,
                       int32_t training, const float* w1, const float* b1, const float* w2,
                       const float* b2, int32_t Cr, float* ss, float* mr, float* gate, float* hid,
                       void* stream);
/* BatchNorm backward as an affine map dx = A*g + B + C*x: dsums f64 [stripes][2][C] = (sum g, sum g*xhat),
 * xhat = (x-mean)*rstd accumulated centred by the producer;
 * coef f32 A[Cp],B[Cp],C[Cp]; dgamma/dbeta are accumulated (+=).                            */
int c3d_bn_bwd_coef(const double* dsums, int32_t stripes, double count, const float* gamma, const float* mr, int32_t C,
                    int32_t Cp, float* coef, float* dgamma, float* dbeta, void* stream);
/* SE backward + BN_b backward: nc3 f64 [B][Cp][3] from C3D_EPI_SWISH_SE_BWD, ncf = forward nc.
 * db = coefA[c]*t1 + coefB[n][c] + coefC[c]*b.  FC / BN parameter gradients accumulated (+=). */
int c3d_se_bn_bwd_coef(const double* nc3, const double* ncf, int32_t B, double cnt_per_sample,
                       const float* gamma, const float* mr, const float* ss, int32_t C, int32_t Cp,
                       const float* w1, const float* w2, const float* gate, const float* hid, int32_t Cr,
                       float* coefA, float* coefC, float* coefB, float* dgamma, float* dbeta,
                       float* dw1, float* db1, float* dw2, float* db2, void* stream);

/* ------------------------------------------------------------------------------------
 * Depthwise 3x3x3 Conv3d, stride (1,s,s) (conv_b, reference model/x3d.py:184-193).
 *   fwd      : x = a (raw conv_a output) with BN_a+ReLU applied on load (ss = scale/shift);
 *              y = b raw; nc_sums f64 [B][Cp][2] accumulated (+=).
 *   bwd_data : db = coefA*t1 + coefB[n] + coefC*b on load -> t2 = dconv*(bn_a(a)>0);
 *              dsums f64 [2][C] += (sum t2, sum t2*ahat), ahat = (a-mean_a)*rstd_a (mr_a).
 *   wgrad    : dw f32 [C][27] += sum db * relu(bn_a(a)).
 * ------------------------------------------------------------------------------------ */
int c3d_dw333_fwd(const void* x, const float* ss, const float* w, void* y, double* nc_sums, int32_t B,
                  int32_t T, int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t stride, int32_t dtype,
                  void* stream);
/* Same, with the BatchNorm finalisation of the INPUT statistics folded in (no c3d_bn_finalize launch between the
 * producer of `fin->sums` and this kernel): bit-identical scale/shift, running statistics and saved vectors.  Shapes
 * without a folded kernel (stride 2) run c3d_bn_finalize + c3d_dw333_fwd internally.                              */
int c3d_dw333_fwd_fin(const void* x, const c3d_bn_fin* fin, const float* w, void* y, double* nc_sums, int32_t B,
                      int32_t T, int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t stride, int32_t dtype,
                      void* stream);
/* bwd_data AND wgrad from one staged tile of db (csrc/dw_bwd_fused.hip) -- what autograd's convolution_backward returns
 * for conv_b (reference model/x3d.py:184-193) in one pass over t1, b and a: t2 = d conv * (bn_a(a) > 0), dsums f64 [2][C] =
 * (sum t2, sum t2 * ahat), dw += (f32 atomics).  stride 1 or 2, any extents.                      */
int c3d_dw333_bwd_fused(const void* t1, const void* b, const float* coefA, const float* coefB,
                        const float* coefC, const float* w, const void* a, const float* ss_a,
                        const float* mr_a, void* t2, double* dsums, float* dw, int32_t B, int32_t T,
                        int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t stride, int32_t dtype, void* stream);
/* Same, with the BatchNorm_b backward coefficients of a block WITHOUT SqueezeExcitation rebuilt in the kernel's prologue
 * from the per-sample sums nc3 (fin_b->sums, fin_b->batch = B, fin_b->gamma, fin_b->mr = mean|rstd of BatchNorm_b,
 * fin_b->count = B*T*Ho*Wo; dgamma / dbeta += through fin_b->running_mean / running_var): no c3d_se_bn_bwd_coef launch
 * between the conv_c data gradient and this kernel.                                                                  */
int c3d_dw333_bwd_fused_fin(const void* t1, const void* b, const c3d_bn_fin* fin_b, const float* w, const void* a,
                            const float* ss_a, const float* mr_a, void* t2, double* dsums, float* dw, int32_t B,
                            int32_t T, int32_t H, int32_t W, int32_t C, int32_t Cp, int32_t stride, int32_t dtype,
                            void* stream);

/* ------------------------------------------------------------------------------------
 * Res-block output y = relu(bn_c(c) + shortcut) (reference model/x3d.py:326-327; also the
 * stem's BN+ReLU with sc_mode 0) and its backward g = dy*(y>0) with BN-backward sums.
 * sc_mode: 0 none, 1 identity, 2 shortcut*scale1+shift1 (branch1_norm), 3 raw shortcut.
 * ------------------------------------------------------------------------------------ */
int c3d_block_out_fwd(const void* c, const float* ss_c, const void* shortcut, const float* ss_1,
                      int32_t sc_mode, void* y, int64_t M, int32_t Cp, int32_t dtype, void* stream);
/* Same with BN_c (and, for sc_mode BN, the shortcut BatchNorm) finalised from their sums by the kernel itself;
 * C = real channel count.  fin_1 is NULL unless sc_mode is BN.                                                     */
int c3d_block_out_fwd_fin(const void* c, const c3d_bn_fin* fin_c, const void* shortcut, const c3d_bn_fin* fin_1,
                          int32_t sc_mode, void* y, int64_t M, int32_t C, int32_t Cp, int32_t dtype, void* stream);
/* y == NULL and g == NULL: `dy` already IS dy * (y > 0) (masked by its producer, c3d_pw_args.wg_mask_out): only the sums.
 * mr_c / mr_1 (mean | rstd rows, Cp floats each) must be 16-byte aligned (C3D_E_BADARG otherwise). */
int c3d_block_out_bwd(const void* dy, const void* y, const void* c, const void* s_bn, void* g,
                      const float* mr_c, const float* mr_1, double* dsums_c, double* dsums_1, int64_t M,
                      int32_t C, int32_t Cp, int32_t dtype, void* stream);
/* as c3d_block_out_bwd; with fin_c->ticket != NULL the last workgroup also writes the backward coefficients of
 * BatchNorm_c (fin_c) and, when s_bn is given, of the shortcut BatchNorm (fin_1; shares fin_c's ticket). */
int c3d_block_out_bwd_fin(const void* dy, const void* y, const void* c, const void* s_bn, void* g,
                          const float* mr_c, const float* mr_1, double* dsums_c, double* dsums_1, int64_t M,
                          int32_t C, int32_t Cp, int32_t dtype, const c3d_bn_fin* fin_c, const c3d_bn_fin* fin_1,
                          void* stream);

/* Encoder.enhance pieces (reference model/trainer.py:71-108); HW = H*W pixels per frame.    */
int c3d_frame_absdiff(const void* y, void* d, int32_t B, int32_t T, int64_t HW, int32_t Cp, int32_t t_pre,
                      int32_t t_post, int32_t dtype, void* stream);
int c3d_enhance_apply(const void* y, const void* e, void* out, int32_t B, int32_t T, int64_t HW,
                      int32_t Cp, int32_t t_mid, int32_t dtype, void* stream);
int c3d_enhance_bwd_mask(const void* dout, const void* e, void* de, int32_t B, int32_t T, int64_t HW,
                         int32_t Cp, int32_t t_mid, int32_t dtype, void* stream);
int c3d_enhance_bwd_apply(const void* dout, const void* y, const void* dd, void* dy, int32_t B, int32_t T,
                          int64_t HW, int32_t Cp, int32_t t_pre, int32_t t_post, int32_t dtype,
                          void* stream);
int c3d_frame_scatter(const void* src, void* dst, int32_t B, int32_t T, int64_t HW, int32_t Cp,
                      int32_t t_dst, int32_t accumulate, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Stem (reference model/x3d.py:70-106): x is the logical NCDHW f32 clip [B][3][T][H][W];
 * u is the raw channels-last output [B][T][H][W][24]; sums f64 [2][24] accumulated.
 * Backward: dv = conv_xy^T(du) with du = A*g0+B+C*u on load (coef f32 [3][24]); then
 * d w_t and the batch-summed input gradient of frames t_first..t_first+n_frames-1
 * (dP f32 [3][n_frames][H][W], the learnable perception frames, model/trainer.py:51-54;
 * with per_sample=1 dP is instead a full NCDHW gradient [B][3][T][H][W], written not summed).
 * ------------------------------------------------------------------------------------ */
int c3d_stem_fwd(const float* x, const float* w_t, const float* w_xy, void* u, double* sums, int32_t B,
                 int32_t T, int32_t H, int32_t W, int32_t dtype, void* stream);
int c3d_stem_bwd_dv(const float* x, const float* w_t, const float* w_xy, const void* g0, const void* u,
                    const float* coef, void* dv, float* dw_xy, int32_t B, int32_t T, int32_t H, int32_t W,
                    int32_t dtype, void* stream);
int c3d_stem_bwd_wx(const float* x, const float* w_t, const void* dv, float* dw_t, float* dP, int32_t B,
                    int32_t T, int32_t H, int32_t W, int32_t t_first, int32_t n_frames, int32_t per_sample,
                    int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * ChangeDecoder (reference model/change_decoder.py:30-55, 68-81).
 *   convT4s2_fwd : out[B][2h][2w][C] = bias + skip + ConvTranspose2d(k4,s2,p1)(in[B][h][w][C]);
 *                  skip is one frame of an NDHWC tensor (element batch stride skip_bstride).
 *   head3x3      : Conv2d 3x3 (24 -> NC, no bias) [+ sigmoid]; out / dout / prob are f32
 *                  NCHW [B][NC][H][W]; dw f32 [NC][24][3][3] accumulated.
 * ------------------------------------------------------------------------------------ */
int c3d_convT4s2_fwd(const void* in, const float* w, const float* bias, const void* skip,
                     int64_t skip_bstride, void* out, int32_t B, int32_t h, int32_t wd, int32_t C,
                     int32_t dtype, void* stream);
int c3d_convT4s2_bwd_data(const void* dout, const float* w, void* din, int32_t B, int32_t h, int32_t wd,
                          int32_t C, int32_t dtype, void* stream);
/* weight gradient of the transposed convolution on MFMA (bf16 storage, C = 24 | 48): dw[ci][co][ky][kx] +=
 * sum t[b,i,j,ci] * dout[b,2i-1+ky,2j-1+kx,co]; t is the layer input [B][h][wd][C], dout its output gradient
 * [B][2h][2wd][C]; ws: f32 scratch of c3d_convT4s2_wgrad_ws_floats elements.  (f32 storage uses c3d_pw_wgrad with
 * C3D_ROWS_S2SHIFT / taps = 16; this entry returns C3D_E_UNSUPPORTED for it.)                               */
int64_t c3d_convT4s2_wgrad_ws_floats(int32_t B, int32_t h, int32_t wd, int32_t C);
int c3d_convT4s2_wgrad(const void* t, const void* dout, float* dw, float* ws, int32_t B, int32_t h, int32_t wd,
                       int32_t C, int32_t dtype, void* stream);
int c3d_col_sum(const void* x, float* out, int64_t M, int32_t C, int32_t Cp, int32_t dtype, void* stream);
int c3d_head3x3_fwd(const void* x, const float* w, float* out, int32_t B, int32_t H, int32_t W, int32_t C,
                    int32_t NC, int32_t has_sigmoid, int32_t dtype, void* stream);
/* ws: optional f32 scratch of c3d_head3x3_bwd_ws_floats elements for the bf16 matrix-core kernel (per-workgroup partials of dw +
 * fixed-order reducer); NULL: every workgroup adds its partial with global atomics (~1000 of them queue on the same 216 NC
 * addresses: 3-4 x slower on the full-size heads).                                                                          */
int64_t c3d_head3x3_bwd_ws_floats(int32_t B, int32_t H, int32_t W, int32_t NC);
int c3d_head3x3_bwd(const float* dout, const float* prob, const void* x, const float* w, void* dx,
                    float* dw, float* ws, int32_t B, int32_t H, int32_t W, int32_t C, int32_t NC, int32_t has_sigmoid,
                    int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Train-step shell: BCEDiceLoss (reference model/utils.py:154-169), Adam
 * (reference scripts/train_BCD.py:284-290), thresholded confusion matrix
 * (reference scripts/train_BCD.py:204-208, utils/metric_tool.py:111-128).
 * sums4 f64 [4] = (sum bce terms, sum p*t, sum p, sum t).  hparams_dev (optional, f32 [3] =
 * lr, bias_correction1, sqrt(bias_correction2)) overrides the by-value scalars so a captured
 * HIP graph can be replayed with a new learning rate.
 * ------------------------------------------------------------------------------------ */
int c3d_bce_dice_fwd(const float* prob, const float* target, int64_t n, double* sums4, float* loss,
                     void* stream);
int c3d_bce_dice_bwd(const float* prob, const float* target, const double* sums4, const float* dloss,
                     int64_t n, float* dprob, void* stream);
int c3d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                  const float* hparams_dev, float lr, float bias_correction1, float bias_correction2_sqrt,
                  float beta1, float beta2, float eps, float weight_decay, void* stream);
int c3d_confusion2(const float* prob, const float* target, int64_t n, unsigned long long* cm4, void* stream);
/* SCD validation (reference scripts/train_SCD.py:104-178, model/utils.py:313-378 fast_hist / get_hist / SCDD_eval_all):
 * hist[a*n + b] += #{i : a[i] == a, b[i] == b} for 0 <= a < n (a = prediction, b = label; int64 device arrays), n <= 16;
 * hist has n*n + 1 u64 slots, the last one counts pairs whose b is outside [0, n) (numpy would raise there).        */
int c3d_hist2d(const int64_t* a, const int64_t* b, int64_t n_elems, int32_t n, unsigned long long* hist, void* stream);

/* ---------------------------------------------------------------------------------------
 * SCD losses.  Logits are f32 NCHW views: element (b, c, p) at
 * x[b*bstride + c*cstride + p], p < HW, NC <= 8 classes; labels are int64 [B][HW].
 *   c3d_ce2d_*   : CrossEntropyLoss2d(ignore_index) = nll_loss(log_softmax(x, 1), target, 'mean' over the
 *                  non-ignored pixels) (reference model/utils.py:171-178; used with ignore_index=0 at
 *                  scripts/train_SCD.py:226).  sums2 = (sum nll, count), zeroed by the call.
 *   c3d_cossim_* : ChangeSimilarity = cosine_embedding_loss(softmax(x1), softmax(x2), +1 where
 *                  label_change == 0 / -1 elsewhere, margin 0, 'mean') (reference model/utils.py:180-203).
 * Backward writes dense [B][NC][HW] gradients scaled by dloss[0] (NULL = 1).
 * ------------------------------------------------------------------------------------ */
int c3d_ce2d_fwd(const float* logits, const int64_t* target, int64_t B, int32_t NC, int64_t HW, int64_t bstride,
                 int64_t cstride, int64_t ignore_index, double* sums2, float* loss, void* stream);
int c3d_ce2d_bwd(const float* logits, const int64_t* target, const double* sums2, const float* dloss, int64_t B,
                 int32_t NC, int64_t HW, int64_t bstride, int64_t cstride, int64_t ignore_index, float* dlogits,
                 void* stream);
int c3d_cossim_fwd(const float* x1, const float* x2, const int64_t* label_change, int64_t B, int32_t NC, int64_t HW,
                   int64_t bstride1, int64_t cstride1, int64_t bstride2, int64_t cstride2, double* sums1, float* loss,
                   void* stream);
int c3d_cossim_bwd(const float* x1, const float* x2, const int64_t* label_change, const float* dloss, int64_t B,
                   int32_t NC, int64_t HW, int64_t bstride1, int64_t cstride1, int64_t bstride2, int64_t cstride2,
                   float* dx1, float* dx2, void* stream);

/* ------------------------------------------------------------------------------------
 * On-device input pipeline of the BCD step (reference data/transforms.py:100-154: random_flip,
 * random_exchange, normalize, to_tensor as composed at scripts/train_BCD.py:262-270) over a raw uint8 batch
 * resident in HBM.  image6: u8 [B][H][W][6] (pre RGB | post RGB); label: u8 [B][H][W] or NULL;
 * flags: u8 [B][3] = (cv2.flip(.,0), cv2.flip(.,1), exchange pre/post) per sample, or NULL (no augmentation =
 * the validation transform); mean6 / std6: f32 [6] DEVICE vectors.  Outputs: pre, post f32 [B][3][H][W]
 * = ((u8/255) - mean)/std, bit-identical to the numpy arithmetic; label_out f32 [B][1][H][W] = ceil(u8/255).
 * ------------------------------------------------------------------------------------ */
/* The (B, 3, T = K+2, H, W) f32 clip the encoder feeds to the stem (reference model/trainer.py:155-162:
 * `torch.cat([x.unsqueeze(2), perception_frames.expand(B, ...), y.unsqueeze(2)], dim=2)`): frame 0 = pre [B][3][H][W],
 * frames 1..K = the learnable perception frames [3][K][H][W] shared by the batch, frame K+1 = post.  H*W % 4 == 0.  */
int c3d_build_clip(const float* pre, const float* post, const float* frames, float* clip, int32_t B, int32_t K,
                   int32_t H, int32_t W, void* stream);
int c3d_bcd_preprocess(const uint8_t* image6, const uint8_t* label, const uint8_t* flags, const float* mean6,
                       const float* std6, float* pre, float* post, float* label_out, int32_t B, int32_t H,
                       int32_t W, void* stream);
/* SCD labels (reference data/transforms.py:300-326, 341-357; scripts/train_SCD.py:207-213): label3 u8 [B][H][W][3] =
 * (pre classes, post classes, change), same flags as the image pass -> int64 [B][3][H][W]; an exchange swaps the two
 * class maps.  The SCD image goes through c3d_bcd_preprocess (same 6-channel arithmetic, label = NULL).             */
int c3d_scd_label_preprocess(const uint8_t* label3, const uint8_t* flags, int64_t* out, int32_t B, int32_t H, int32_t W,
                             void* stream);
/* Change-captioning pairs (reference data/dataset.py:411-424, scripts/train_CC.py:466-469): img u8 [B][2][3][H][W] ->
 * pre, post f32 [B][3][H][W] = lut[channel][u8] (the host builds the 3 x 256 table with the reference's own arithmetic:
 * Normalize(FloatTensor(u8 / 255.))); swap u8 [B] or NULL exchanges the pair.  H*W % 4 == 0.                        */
int c3d_cc_preprocess(const uint8_t* img, const uint8_t* swap, const float* lut, float* pre, float* post, int32_t B,
                      int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------------------------
 * Residual-stage step driver: ONE call enqueues every kernel of `blocks[i](x)` for a whole X3D residual
 * stage (reference model/x3d.py:331-412 = ResStage of ResBlocks, driven by `self.x3d.blocks[i](x)` at
 * reference model/trainer.py:126-139), forward or backward, from C++ -- the ~25 launches per block are no
 * longer issued one by one through Python/ctypes (19-22 ms of host time per B=32 step in round 1).
 *
 * All pointers are device pointers; parameter / gradient / BN-buffer pointers are the tensors of the
 * reference's own module tree (state-dict keys in the comments).  Gradients are ACCUMULATED (+=).
 * The forward workspace `ws_fwd` (c3d_stage_ws_bytes) holds every activation the backward pass re-reads
 * (a, b, c, shortcut, block outputs), the BatchNorm scale/shift and mean/rstd vectors, SE gates and the
 * f64 statistics accumulators; the caller keeps it alive until c3d_stage_bwd has run.  `ws_bwd` holds the
 * backward temporaries.  Weight gradients run on an internal side stream that forks from / joins `stream`
 * with events (C3D_WGRAD_SIDE=0 disables it); c3d_side_join makes `stream` wait for everything issued
 * there so far (call it before reading parameter gradients and before freeing ws_fwd / ws_bwd).
 * Size limit: every activation tensor of a stage whose channel counts are <= 224 must stay under 2 GiB (32-bit byte offsets in
 * the narrow pointwise kernels); c3d_stage_ws_bytes / c3d_stage_fwd / c3d_stage_bwd return C3D_E_UNSUPPORTED for a larger
 * geometry BEFORE touching the device (bf16 at 256 x 256, T = 3: B <= 96 per GPU; f32: B <= 48).
 * ------------------------------------------------------------------------------------ */
typedef struct c3d_bn_ptrs {
  const float* gamma; const float* beta;       /* .weight / .bias                                       */
  float* running_mean; float* running_var;     /* .running_mean / .running_var (updated in training)    */
  int64_t* num_batches_tracked;                /* .num_batches_tracked (incremented in training)        */
  float* dgamma; float* dbeta;                 /* gradients (may be NULL in c3d_stage_fwd)              */
} c3d_bn_ptrs;

typedef struct c3d_block_desc {
  int32_t cin, cinner, cout, stride;           /* dim_in, dim_inner, dim_out, spatial stride (1|2)      */
  int32_t se_width;                            /* 0 = no SqueezeExcitation in this block                */
  int32_t has_sc_conv, has_sc_bn;              /* branch1_conv / branch1_norm present                   */
  int32_t reserved;
  const float* w_a; const float* w_b; const float* w_c; const float* w_sc;   /* branch2.conv_{a,b,c}.weight, branch1_conv.weight */
  float* dw_a; float* dw_b; float* dw_c; float* dw_sc;
  c3d_bn_ptrs bn_a, bn_b, bn_c, bn_sc;         /* branch2.norm_a, norm_b.0, norm_c, branch1_norm        */
  const float* se_w1; const float* se_b1; const float* se_w2; const float* se_b2;   /* norm_b.1.block.{0,2}.{weight,bias} */
  float* dse_w1; float* dse_b1; float* dse_w2; float* dse_b2;
} c3d_block_desc;

/* c3d_stage_desc.flags: the unfused launch sequences stay callable so that the fused ones are testable against them
 * bit for bit (tests/test_model_gpu.py::test_folded_batchnorm_launches_are_bit_identical_end_to_end).                */
enum {
  C3D_STAGE_SEPARATE_FINALIZE = 1,   /* c3d_bn_finalize / c3d_bn_bwd_coef launches instead of the consumers' prologues     */
  C3D_STAGE_NO_WEIGHT_IMAGES = 2,    /* every GEMM workgroup converts the f32 weights itself (no c3d_pw_pack_weights)      */
  C3D_STAGE_SEPARATE_RESIDUAL = 4,   /* c3d_block_out_fwd launches instead of the next block's conv_a prologue             */
  C3D_STAGE_SEPARATE_WGRAD = 8       /* every pointwise weight gradient through c3d_pw_wgrad on the side stream (round 3's   */
                                     /* sequence) instead of fused into the data-gradient launch where the shape allows it  */
};

typedef struct c3d_stage_desc {
  int32_t n_blocks;
  int32_t B, T, H, W;                          /* stage INPUT extent; block 0 applies the stride        */
  int32_t dtype;                               /* activation storage type                               */
  int32_t training;                            /* 1: batch statistics (+ running-stat update), 0: eval  */
  int32_t flags;                               /* C3D_STAGE_* (0 = the fused, measured-best launch sequence) */
  float momentum, eps;                         /* BatchNorm3d momentum / eps (0.1 / 1e-5)               */
  const c3d_block_desc* blocks;                /* HOST array of n_blocks descriptors                    */
} c3d_stage_desc;

/* bytes of the two workspaces and of the stage output y [B][T][Ho][Wo][cpad(cout)] / input gradient dx */
int c3d_stage_ws_bytes(const c3d_stage_desc* d, int64_t* ws_fwd_bytes, int64_t* ws_bwd_bytes, int64_t* y_bytes,
                       int64_t* dx_bytes);
/* x: [B][T][H][W][cpad(cin)] channels-last, storage dtype.  y: stage output (also re-read by backward).      */
int c3d_stage_fwd(const c3d_stage_desc* d, const void* x, void* ws_fwd, void* y, void* stream);
/* dy: gradient of y (same layout); dx: gradient of x (written).  x / y / ws_fwd as given to c3d_stage_fwd.
 * The weight gradients that are not fused into their data-gradient launch (c3d_pw_wgrad) are launched on the library's side stream, each forked ahead
 * of the data-gradient kernel that reads the same operands, with their grids capped (7/8 of the CUs for the pointwise
 * kernel: csrc/launch_hints.h, csrc/pw_wgrad.hip) so that the data-gradient chain finds free CUs; ws_bwd holds a ring of
 * three blocks' temporaries (one slot more for dx, which is also the next block's g: c3d_pw_args.add_sums), so the side
 * stream may lag the chain by two blocks.                                                                        */
int c3d_stage_bwd(const c3d_stage_desc* d, const void* x, const void* y, const void* dy, void* ws_fwd, void* ws_bwd,
                  void* dx, void* stream);
/* c3d_side_join(stream): `stream` waits (on the device) for everything the library has issued on its side stream so far.
 * CONTRACT: the fork / done events between the two streams are created with hipEventDisableSystemFence -- they order the two
 * queues of ONE device and carry no system-scope release.  After c3d_side_join the weight gradients are visible to later work
 * on `stream` on this device; a consumer OUTSIDE the device -- a peer GPU (RCCL), the host reading p.grad -- must be ordered
 * behind `stream` by the caller's own system-scope operation: a default-flag event / hipStreamSynchronize / a stream-ordered
 * D2H copy or collective enqueued on `stream` (each of these releases at system scope when it completes).  torch's
 * stream.synchronize(), tensor.cpu() and torch.distributed collectives issued on `stream` all qualify
 * (tests/test_dp_gpu.py::test_side_join_then_host_readback_and_allreduce).                                                    */
int c3d_side_join(void* stream);
/* Run-time options of the library (process-wide; not thread-safe; all default to 1).  These are the ONLY run-time
 * switches: the product library never reads the environment (tuning knobs exist in the -DC3D_TUNING build only).
 *   C3D_OPT_SIDE_STREAM : 0 = weight gradients inline on the caller's stream (needed when the step is captured in a HIP
 *                         graph, and for one-kernel-at-a-time traces)
 *   C3D_OPT_STEM_MFMA   : 0 = scalar-FMA stem kernels (csrc/stem.hip) instead of the matrix-core ones (bit-identical
 *                         u / dv / dx: tests/test_model_gpu.py::test_stem_mfma_kernels_equal_the_scalar_kernels);
 *                         1 = f32 matrix-core kernels everywhere; 2 (default) = c3d_stem_bwd_wx of bf16 storage multiplies on
 *                         the bf16 matrix cores (x and w_t rounded to bf16 for the products, f32 sums)
 *   C3D_OPT_CONVT_MFMA  : 0 = scalar ConvTranspose2d kernels on the bf16 path too
 *   C3D_OPT_FUSE_WGRAD  : which pointwise weight gradients the stage driver fuses into their data-gradient launch where the
 *                         shape allows it (c3d_pw_args.wg_mode): bit 0 = conv_a, bit 1 = conv_c; default = measured best
 *   C3D_OPT_FOLD_SE     : 0 = c3d_bn_se_finalize launches for the blocks with SqueezeExcitation (default 1: conv_c's workgroups
 *                         compute the gate of their samples, c3d_pw_args.se_w1)
 *   C3D_OPT_MASK_IN_DGRAD : a BITFIELD, default 3 (the value is masked with 3).  bit 0: conv_a's data-gradient launch of the block
 *                         ABOVE stores dx * (y > 0) -- c3d_block_out_bwd of this block only sums; bit 1 (needs bit 0): the same
 *                         epilogue also takes this block's BatchNorm_c-backward sums (c3d_pw_args.add_sums / C3D_WG_MASKSUM) and
 *                         the c3d_block_out_bwd launch is gone (35 of 41 per BCD step).  0 = c3d_block_out_bwd does everything;
 *                         1 = the round-4 mask-only path (a caller that passes 1 meaning "on" gets THAT, not the default)
 *   C3D_OPT_DW_RING     : c3d_dw333_bwd_fused (bf16, stride 1) fed by an LDS-DMA ring (global_load_lds_dwordx4, two tiles ahead
 *                         for T <= 3, one for T = 5) instead of register prefetch: bit 0 = on, bit 2 = the requests are issued
 *                         one per tap step instead of in a burst, bit 3 = also on maps under 64 x 64; results are bit-identical
 *                         to the register-prefetch kernel; default 13 (round 6, same-call A/B: 20.13 -> 20.05 ms with bit 3)
 *   C3D_OPT_PW_WGRAD_V2 : 0 = c3d_pw_wgrad of bf16 dense rows on the first kernel (operand-split staging, 32-row tiles;
 *                         csrc/pw_wgrad.hip) instead of the flat-staged, transposing-read one (csrc/pw_wgrad_v2.hip, default 1);
 *                         same operand arithmetic, products summed in another order (f32 rounding apart); bit 1 SET = the stage
 *                         driver's separate weight gradients each launch their own reducer (default: chained,
 *                         c3d_pw_wgrad_args.chain -- bit-identical gradients either way)
 *   C3D_OPT_DW_FWD_HV   : c3d_dw333_fwd, stride 1, three frames, on half-vector lanes (a lane = 4 channels x 4 output rows, tap
 *                         walk column -> frame -> input row: half the LDS reads per FMA; csrc/dw_conv.hip): bit 0 = bf16
 *                         storage, bit 1 = f32 storage; the taps are summed in another order than the 8-channel lanes (f32
 *                         rounding apart); bit 2 = the STRIDE-2 forward (bf16) on eight waves per tile instead of four -- two
 *                         waves per channel vector, one half vector each; its 131 KB tile allows one workgroup per CU -- same
 *                         tap order per channel: bit-identical outputs.  Default 5
 *   C3D_OPT_PW_CFWD     : bit field, which forward GEMMs of the training path run on the workgroup-cooperative kernel
 *                         (csrc/pw_cfwd.hip) instead of the wave-private-tile one.  bit 0: conv_c (C3D_PRO_BN_SE_SWISH +
 *                         C3D_EPI_STATS, BatchNorm_b / SE gate rebuilt from the per-sample sums); bit 1: conv_a with the
 *                         previous block's residual add in its prologue (C3D_PRO_AFFINE2 + pro_out + C3D_EPI_STATS).
 *                         Outputs (and pro_out) bit-identical, BatchNorm statistics to f32 rounding.  Default 3
 *   C3D_OPT_PW_CDG      : bit field.  bit 0: the conv_a data gradient with its weight gradient fused (C3D_PRO_AFFINE2 +
 *                         C3D_EPI_ADD + C3D_WG_ROWS, dense shortcut gradient), bit 1: the conv_c one (C3D_PRO_AFFINE2 +
 *                         C3D_EPI_SWISH_SE_BWD + C3D_WG_SWISH, rows_per_sample a multiple of the tile's rows) run on the
 *                         workgroup-cooperative kernels (csrc/pw_cdgrad.hip) where they apply -- all three stage widths of
 *                         X3D-L, the 216-wide layers included, which the wave-private kernel's fused variant (accumulator
 *                         image in LDS) left to separate c3d_pw_wgrad launches; c3d_stage_bwd then asks for the fused form
 *                         there too.  Data gradients bit-identical, sums / dW to f32 rounding.  Default 3                  */
enum { C3D_OPT_SIDE_STREAM = 0, C3D_OPT_STEM_MFMA = 1, C3D_OPT_CONVT_MFMA = 2, C3D_OPT_FUSE_WGRAD = 3, C3D_OPT_FOLD_SE = 4,
       C3D_OPT_MASK_IN_DGRAD = 5, C3D_OPT_DW_RING = 6, C3D_OPT_PW_WGRAD_V2 = 7, C3D_OPT_DW_FWD_HV = 8, C3D_OPT_PW_CFWD = 9,
       C3D_OPT_PW_CDG = 10 };
int c3d_set_option(int32_t option, int32_t value);
/* Per-launch profile of the stage driver: between c3d_prof_begin and c3d_prof_end every kernel c3d_stage_fwd /
 * c3d_stage_bwd enqueue is bracketed by a HIP event pair on its launch stream and billed its algorithmic bytes
 * (the tensors it must read / write once).  flags bit 0: weight gradients inline on the main stream (an event pair
 * then brackets exactly one kernel), bit 1: pointwise rows are keyed by shape and mode.  c3d_prof_end synchronises
 * the device and returns one row per kernel entry (aggregated).  Not thread-safe; meant for bench.py / tools.      */
typedef struct c3d_prof_row {
  char name[64];
  int32_t launches, reserved;
  float ms_total, reserved2;
  double bytes_total;
} c3d_prof_row;
int c3d_prof_begin(int32_t flags);
int c3d_prof_end(c3d_prof_row* rows, int32_t cap, int32_t* n_rows);
/* Eval / inference (reference scripts/train_BCD.py:92-154 `val()` under model.eval() + torch.no_grad()): BatchNorm
 * folded into the convolution weights.  c3d_stage_fold_bn writes, once per set of weights, W' = W * gamma/sqrt(var+eps)
 * (rows of conv_a / conv_b / conv_c / branch1_conv) and the remaining per-channel biases into `fold`
 * (c3d_stage_fold_bytes); c3d_stage_fwd_folded then runs the stage with 4-5 launches per block (no statistics, no
 * finalize kernels, no saved activations: `ws` is a small ring of c3d_stage_fold_bytes' ws_eval_bytes).          */
int c3d_stage_fold_bytes(const c3d_stage_desc* d, int64_t* fold_bytes, int64_t* ws_eval_bytes);
int c3d_stage_fold_bn(const c3d_stage_desc* d, void* fold, void* stream);
int c3d_stage_fwd_folded(const c3d_stage_desc* d, const void* fold, const void* x, void* ws, void* y, void* stream);
/* byte offsets (into ws_fwd) and sizes of what block `blk` stored: name is one of "a","b","c","sc","mr_a","mr_b",
 * "mr_c","mr_sc","ss_a","ss_b","ss_c","ss_sc","gate" -- test / debug access (returns C3D_E_BADARG if absent). */
int c3d_stage_saved(const c3d_stage_desc* d, int32_t blk, const char* name, int64_t* offset, int64_t* bytes);

/* ------------------------------------------------------------------------------------
 * Caption decoder of the change-captioning path (reference model/caption_decoder.py:526-613 CaptionDecoder, :316-423
 * Mesh_TransformerDecoderLayer, :272-314 PositionalEncoding; loss: scripts/train_CC.py:124-132).  Activations are
 * sequence-first rows (row = l*B + b, nn.MultiheadAttention's default layout) of Dp = round_up(D, 8) elements in the
 * storage dtype; the linear layers run on c3d_pw_gemm / c3d_pw_wgrad (bias through c3d_pw_args.bias, bias gradient
 * through c3d_col_sum).  Dropout masks are counter-based: (seed, element index) -> keep/drop, regenerated in backward.
 * ------------------------------------------------------------------------------------ */
/* out[l*B+b] = dropout_p(emb[tokens[b][l]] + pe[l]); tokens int64 [B][L]; emb f32 [V][D]; pe f32 [>=L][D]        */
int c3d_cap_embed_fwd(const int64_t* tokens, const float* emb, const float* pe, void* out, int32_t B, int32_t L,
                      int32_t D, int32_t V, float p, uint64_t seed, int32_t dtype, void* stream);
/* demb[tokens[b][l]] += dropout-mask * dout[l*B+b]   (f32 atomics)                                                */
int c3d_cap_embed_bwd(const int64_t* tokens, const void* dout, float* demb, int32_t B, int32_t L, int32_t D, int32_t V,
                      float p, uint64_t seed, int32_t dtype, void* stream);
/* y = x * mask(seed) / (1-p): the same call is the forward and (on the gradient) the backward                     */
int c3d_cap_dropout(const void* x, void* y, int64_t rows, int32_t D, float p, uint64_t seed, int32_t dtype, void* stream);
/* y = LayerNorm(x + a) (a may be NULL), mr f32 [rows][2] = saved (mean, rstd); backward: dx (= gradient of x and of a),
 * dgamma / dbeta accumulated (+=)                                                                                 */
int c3d_cap_layernorm_fwd(const void* x, const void* a, const float* gamma, const float* beta, void* y, float* mr,
                          int64_t rows, int32_t D, float eps, int32_t dtype, void* stream);
int c3d_cap_layernorm_bwd(const void* x, const void* a, const void* dy, const float* gamma, const float* mr, void* dx,
                          float* dgamma, float* dbeta, int64_t rows, int32_t D, int32_t dtype, void* stream);
/* multi-head attention, one workgroup per (sample, head): q/k/v/o rows (l*B+b) with leading dimensions ld* (elements),
 * head h at column h*hd; P f32 [H*B][Lq][Lk] = softmax(scale*q.k^T (+causal mask)) BEFORE dropout (saved for backward);
 * o = dropout_p(P) v.                                                                                             */
int c3d_cap_attn_fwd(const void* q, const void* k, const void* v, int32_t ldq, int32_t ldk, int32_t ldv, void* o, int32_t ldo,
                     float* P, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, float scale, int32_t causal,
                     float p, uint64_t seed, int32_t dtype, void* stream);
int c3d_cap_attn_bwd(const void* q, const void* k, const void* v, int32_t ldq, int32_t ldk, int32_t ldv, const void* dout,
                     int32_t ldo, const float* P, void* dq, void* dk, void* dv, int32_t lddq, int32_t lddk, int32_t lddv,
                     int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, float scale, float p, uint64_t seed,
                     int32_t dtype, void* stream);
/* CrossEntropyLoss(ignore_index) over the decoded steps (pack_padded_sequence of scripts/train_CC.py:124-132 without the
 * gather): logits rows (l*B+b) of round_up(V,8) columns; target = caps[b][l+1] (caps int64 [B][L], sorted by length);
 * a step counts iff l < declen[b] (int64 [B]) and target != ignore_index; loss = mean; lse f32 [L*B];
 * acc2 f64 [3] = (sum of the negative log-likelihoods, counted steps, top-1 hits = caption_accuracy(scores, targets, 1)
 * of reference model/utils.py:493-507 before its percentage scaling).  A counted step whose target lies outside [0, V)
 * makes the loss NaN (torch.nn.CrossEntropyLoss trips a device assert there): never silently ignored.              */
int c3d_cap_ce_fwd(const void* logits, const int64_t* caps, const int64_t* declen, double* acc2, float* lse, float* loss,
                   int32_t B, int32_t L, int32_t V, int64_t ignore_index, int32_t dtype, void* stream);
int c3d_cap_ce_bwd(const void* logits, const int64_t* caps, const int64_t* declen, const double* acc2, const float* lse,
                   const float* dloss, void* dlogits, int32_t B, int32_t L, int32_t V, int64_t ignore_index, int32_t dtype,
                   void* stream);
/* clip_gradient (reference model/utils.py:481-491): g = clamp(g, -limit, limit) over a flat f32 gradient buffer     */
int c3d_clamp_(float* g, int64_t n, float limit, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CHANGE3D_HIP_H */
